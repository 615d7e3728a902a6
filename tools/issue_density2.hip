// Follow-up of issue_density.hip: v_mfma_f32_16x16x4_f32 (32 cycles) streams with the vector work of the Winograd
// design (per 32 MFMAs: 16 8-byte loads, 32 packed adds), one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NL, int NV>
__global__ void __launch_bounds__(256, 2) k16(const float *tab, float *out, int iters) {
    f32x4 acc[32];
    for (int j = 0; j < 32; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x2 *t2 = reinterpret_cast<const f32x2 *>(tab) + threadIdx.x;
    f32x2 cur[16], nxt[16];
    for (int j = 0; j < 16; ++j) cur[j] = t2[j * 256];
    int off = 0;
    for (int i = 0; i < iters; ++i) {
        off = (off + 4096) & 65535;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < NL) nxt[j] = t2[off + j * 256];
            f32x2 a = cur[j], b = cur[(j + 3) & 15];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (v & 1) b = b + cur[(j + v + 1) & 15];
                else a = a - cur[(j + v + 1) & 15];
            }
            acc[2 * j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[2 * j], 0, 0, 0);
            acc[2 * j + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[2 * j + 1], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < NL) cur[j] = nxt[j];
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 32; ++j) s += acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NL, int NV>
static void run(const float *tab, float *out, int bpc) {
    const int iters = 8000, blocks = 256 * bpc;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k16<NL, NV>), dim3(blocks), dim3(256), 0, 0, tab, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        double flop = 2.0 * 16 * 16 * 4 * 32.0 * iters * 4.0 * blocks;
        if (rep) printf("waves/SIMD=%d loads=%d pk_adds=%d per 32 MFMA(16x16x4): %.3f ms, %.1f TFLOP/s\n", bpc, NL, 2 * NV, ms, flop / ms / 1e9);
    }
}

int main() {
    float *tab, *out;
    (void)hipMalloc(&tab, (65536 + 8192) * 2 * sizeof(float) + 4096);
    (void)hipMemset(tab, 0, (65536 + 8192) * 2 * sizeof(float) + 4096);
    (void)hipMalloc(&out, 512 * 256 * sizeof(float));
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<0, 0>(tab, out, bpc);
        run<16, 0>(tab, out, bpc);
        run<16, 1>(tab, out, bpc);
        run<16, 2>(tab, out, bpc);
        run<16, 4>(tab, out, bpc);
    }
    return 0;
}
