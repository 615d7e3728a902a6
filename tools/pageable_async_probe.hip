// Does hipMemcpyAsync(H2D) from PAGEABLE host memory read the host buffer after the call has returned?
//
// Root-cause probe for the one-in-ten wrong answer of the whole-cycle predictor in round 1
// (tests/test_predictor.py::test_src_main_test_branch_vs_oracle[True-8]): rnh_cine_gather used to upload its sample
// descriptors with hipMemcpyAsync from the caller's ctypes array, which Python frees - and the next batch's gather
// re-fills - as soon as the call returns.  This program does the same thing in isolation: while the stream is kept
// busy by a long kernel it enqueues copy k of a small pageable buffer holding the value k, immediately overwrites the
// host buffer with k + 1, and afterwards counts how many device copies hold a value other than the one the host
// buffer had AT THE TIME OF THE CALL.  Any count > 0 means the runtime deferred the read.
//
//   hipcc --offload-arch=gfx950 -O2 tools/pageable_async_probe.hip -o /tmp/pageable_probe && /tmp/pageable_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void spin_kernel(long long cycles, int *sink) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {
    }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}

int main() {
    const int rounds = 200, words = 640;          // 2560 B: 32 descriptors of 80 B
    hipStream_t st;
    CK(hipStreamCreate(&st));
    int *dev, *sink;
    CK(hipMalloc(&dev, (size_t)rounds * words * sizeof(int)));
    CK(hipMalloc(&sink, sizeof(int)));
    int *host = (int *)malloc(words * sizeof(int));                       // pageable
    int *back = (int *)malloc((size_t)rounds * words * sizeof(int));
    for (int busy = 0; busy < 2; ++busy) {
        CK(hipMemset(dev, 0xff, (size_t)rounds * words * sizeof(int)));
        CK(hipDeviceSynchronize());
        if (busy) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, 200000000LL, sink);    // ~0.1 s in front of the copies
        for (int k = 0; k < rounds; ++k) {
            for (int i = 0; i < words; ++i) host[i] = k;
            CK(hipMemcpyAsync(dev + (size_t)k * words, host, words * sizeof(int), hipMemcpyHostToDevice, st));
            for (int i = 0; i < words; ++i) host[i] = k + 1;               // what a freed-and-reused array looks like
        }
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(back, dev, (size_t)rounds * words * sizeof(int), hipMemcpyDeviceToHost));
        int stale = 0;
        for (int k = 0; k < rounds; ++k) {
            int bad = 0;
            for (int i = 0; i < words; ++i) bad |= back[(size_t)k * words + i] != k;
            stale += bad;
        }
        printf("stream %s: %d of %d pageable async copies delivered bytes written to the host buffer AFTER hipMemcpyAsync returned\n",
               busy ? "busy (0.1 s kernel queued first)" : "idle", stale, rounds);
    }
    return 0;
}
