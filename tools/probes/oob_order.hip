// Do buffer loads with out-of-range lanes keep their place in the return order on gfx950?
//
// Every wave issues load A (in range, from a buffer nothing has touched: an HBM round trip), then load B into another register, then
// s_waitcnt vmcnt(1) - "everything but the youngest load has landed" if loads return in order - and copies A's register out.  A's register
// holds a sentinel beforehand.  If B can overtake A, vmcnt(1) is satisfied while A is still in flight and the sentinel comes out.
//   mode 0: B in range (control)   1: all 64 lanes of B out of range   2: lanes 1..63 out of range   3: lanes 32..63 out of range
//   4: lanes 16..63 out of range   5: B in range but exec-masked down to lane 0   6: only lane 63 out of range
// hipcc --offload-arch=gfx950 -O2 -o oob_order oob_order.hip && ./oob_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 desc(const void *p, int nbytes) {
    const unsigned long long u = (unsigned long long)p;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    d[1] = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
    d[2] = nbytes;
    d[3] = 0x00020000;
    return d;
}

__global__ void probe(const float *big, const float *small_, float *outA, float *outB, int mode, long stride) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const i32x4 da = desc(big + wave * stride, 64 * 16);          // this wave's private kilobyte of the cold buffer
    const i32x4 db = desc(small_, 64 * 16);
    int voa = lane * 16, vob = lane * 16;
    const int oob = 0x7fffff00;
    if (mode == 1) vob = oob;
    if (mode == 2 && lane >= 1) vob = oob;
    if (mode == 3 && lane >= 32) vob = oob;
    if (mode == 4 && lane >= 16) vob = oob;
    if (mode == 6 && lane == 63) vob = oob;
    float a = -777.f, b = -555.f;
    if (mode == 5) {
        asm volatile("s_nop 4\n\tbuffer_load_dword %0, %2, %3, 0 offen\n\t"
                     "s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 1\n\t"
                     "buffer_load_dword %1, %4, %5, 0 offen\n\t"
                     "s_mov_b64 exec, s[20:21]\n\t"
                     "s_waitcnt vmcnt(1)"
                     : "+v"(a), "+v"(b) : "v"(voa), "s"(da), "v"(vob), "s"(db) : "memory", "s20", "s21");
    } else {
        asm volatile("s_nop 4\n\tbuffer_load_dword %0, %2, %3, 0 offen\n\t"
                     "buffer_load_dword %1, %4, %5, 0 offen\n\t"
                     "s_waitcnt vmcnt(1)"
                     : "+v"(a), "+v"(b) : "v"(voa), "s"(da), "v"(vob), "s"(db) : "memory");
    }
    const float seen = a;                                          // what A's register holds behind the counted wait
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b));
    outA[wave * 64 + lane] = seen;
    outB[wave * 64 + lane] = b;
}

int main() {
    const int nwaves = 256 * 8 * 4, stride = 4096;                 // floats between the waves' kilobytes: different pages / channels
    float *big, *small_, *outA, *outB;
    hipMalloc(&big, (size_t)nwaves * stride * 4);
    hipMalloc(&small_, 4096);
    hipMalloc(&outA, (size_t)nwaves * 64 * 4);
    hipMalloc(&outB, (size_t)nwaves * 64 * 4);
    std::vector<float> h((size_t)nwaves * 64);
    for (int mode = 0; mode <= 6; ++mode) {
        long stale = 0, total = 0;
        for (int rep = 0; rep < 20; ++rep) {
            hipMemset(big, 0, (size_t)nwaves * stride * 4);        // value 0.0 everywhere; also evicts the lines from the caches
            hipMemset(small_, 0, 4096);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(probe, dim3(nwaves / 4), dim3(256), 0, 0, big, small_, outA, outB, mode, (long)stride);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), outA, h.size() * 4, hipMemcpyDeviceToHost);
            for (float v : h) { stale += (v == -777.f); ++total; }
        }
        printf("mode %d: %ld of %ld lanes read A's register before its load had landed (%.4f %%)\n", mode, stale, total, 100.0 * stale / total);
    }
    return 0;
}
