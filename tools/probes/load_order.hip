// Do younger vector-memory loads overtake older ones on gfx950 when the older one is SCATTERED and cold and the younger one contiguous and hot?
// (The counted waits of csrc/conv_igemm.hip's direct variant - s_waitcnt vmcnt(L/2) - failed under cross-stream load, DESIGN.md 4d (e);
// tools/probes/oob_order.hip found out-of-range loads in order.)  Every wave: load A = buffer_load_dwordx4 with a per-lane stride (each lane its
// own cache line of a buffer nothing has touched), load B = buffer_load_dwordx4 of one hot kilobyte, s_waitcnt vmcnt(1), copy A's registers out.
// A's registers hold a sentinel beforehand: a sentinel in the output = B was counted as "the one load still in flight" while A had not landed.
//   mode 0: A contiguous (control)   1: lane stride 256 B   2: lane stride 4 KiB   3: as 2, and B is issued 8 times (vmcnt(8))
// A second stream runs a copy kernel meanwhile (argument "busy").   hipcc --offload-arch=gfx950 -O2 -o load_order load_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 desc(const void *p) {
    const unsigned long long u = (unsigned long long)p;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    d[1] = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
    d[2] = 0x7fffffff;
    d[3] = 0x00020000;
    return d;
}

__global__ void probe(const float *big, const float *hot, float *out, int mode, long wave_stride, int lane_stride) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const i32x4 da = desc(big + wave * wave_stride), db = desc(hot);
    const int voa = lane * lane_stride, vob = lane * 16;
    f32x4 a = {-777.f, -777.f, -777.f, -777.f}, b[8];
    for (int i = 0; i < 8; ++i) b[i] = f32x4{-5.f, -5.f, -5.f, -5.f};
    if (mode == 3) {
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %9, %10, 0 offen\n\t"
                     "buffer_load_dwordx4 %1, %11, %12, 0 offen\n\tbuffer_load_dwordx4 %2, %11, %12, 0 offen offset:1024\n\t"
                     "buffer_load_dwordx4 %3, %11, %12, 0 offen offset:2048\n\tbuffer_load_dwordx4 %4, %11, %12, 0 offen offset:3072\n\t"
                     "buffer_load_dwordx4 %5, %11, %12, 0 offen\n\tbuffer_load_dwordx4 %6, %11, %12, 0 offen offset:1024\n\t"
                     "buffer_load_dwordx4 %7, %11, %12, 0 offen offset:2048\n\tbuffer_load_dwordx4 %8, %11, %12, 0 offen offset:3072\n\t"
                     "s_waitcnt vmcnt(8)"
                     : "+v"(a), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
                     : "v"(voa), "s"(da), "v"(vob), "s"(db) : "memory");
    } else {
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %3, 0 offen\n\t"
                     "buffer_load_dwordx4 %1, %4, %5, 0 offen\n\t"
                     "s_waitcnt vmcnt(1)"
                     : "+v"(a), "+v"(b[0]) : "v"(voa), "s"(da), "v"(vob), "s"(db) : "memory");
    }
    const f32x4 seen = a;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += b[i].x;
    out[wave * 64 + lane] = (seen.x == -777.f || seen.y == -777.f || seen.z == -777.f || seen.w == -777.f) ? 1.f : (s == 12345.f ? 2.f : 0.f);
}

__global__ void churn(float4 *dst, const float4 *src, long n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main(int argc, char **argv) {
    const bool busy = argc > 1 && !strcmp(argv[1], "busy");
    const int nwaves = 256 * 8 * 2;
    const long wave_stride = 64 * 1024 + 256;                      // floats between two waves' regions (64 lanes x 4 KiB fit)
    float *big, *hot, *out;
    float4 *c0, *c1;
    const long cn = 64L << 20;                                     // 1 GiB each way for the other stream
    hipMalloc(&big, (size_t)nwaves * wave_stride * 4);
    hipMalloc(&hot, 8192);
    hipMalloc(&out, (size_t)nwaves * 64 * 4);
    hipMalloc(&c0, cn * 16);
    hipMalloc(&c1, cn * 16);
    hipMemset(c1, 0, cn * 16);
    hipStream_t s2;
    hipStreamCreate(&s2);
    std::vector<float> h((size_t)nwaves * 64);
    const int strides[4] = {16, 256, 4096, 4096};
    for (int mode = 0; mode < 4; ++mode) {
        long stale = 0, total = 0;
        for (int rep = 0; rep < 30; ++rep) {
            hipMemset(big, 0, (size_t)nwaves * wave_stride * 4);
            hipMemset(hot, 0, 8192);
            hipDeviceSynchronize();
            if (busy) hipLaunchKernelGGL(churn, dim3(2048), dim3(256), 0, s2, c0, c1, cn, 2);
            hipLaunchKernelGGL(probe, dim3(nwaves / 4), dim3(256), 0, 0, big, hot, out, mode, wave_stride, strides[mode]);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
            for (float v : h) { stale += (v == 1.f); ++total; }
        }
        printf("%s mode %d (lane stride %d B): %ld of %ld lanes read A before it had landed (%.5f %%)\n", busy ? "busy" : "idle", mode, strides[mode], stale, total,
               100.0 * stale / total);
    }
    return 0;
}
