// How many vector-memory + VALU instructions can a wave interleave with its own v_mfma_f32_32x32x2_f32 stream?
// Per iteration: NM MFMAs, NL 8-byte global loads (L1/L2 resident table), NV v_pk_add_f32.  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int LPM, int VPM>      // loads per MFMA pair, pk_adds per MFMA pair
__global__ void __launch_bounds__(256) k(const float *tab, float *out, int iters) {
    f32x16 acc[16];
    for (int j = 0; j < 16; ++j)
        for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
    const f32x2 *t2 = reinterpret_cast<const f32x2 *>(tab) + threadIdx.x;
    f32x2 cur[16], nxt[16];
    for (int j = 0; j < 16; ++j) cur[j] = t2[j * 256];
    int off = 0;
    for (int i = 0; i < iters; ++i) {
        off = (off + 4096) & 65535;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < 8 * LPM) nxt[(2 * j) % 16] = t2[off + (2 * j) * 256];
            if (j < 8 * LPM) nxt[(2 * j + 1) % 16] = t2[off + (2 * j + 1) * 256];
            f32x2 a = cur[j];
#pragma unroll
            for (int v = 0; v < VPM; ++v) a = a + cur[(j + v + 1) & 15];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, cur[(j + 5) & 15].x, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, cur[(j + 5) & 15].y, acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) cur[j] = nxt[j];
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j)
        for (int v = 0; v < 16; ++v) s += acc[j][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int LPM, int VPM>
static void run(const float *tab, float *out) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<LPM, VPM>), dim3(blocks), dim3(256), 0, 0, tab, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flop = 2.0 * 32 * 32 * 2 * 32.0 * iters * 4.0 * blocks;
        if (rep) printf("loads/MFMA=%.1f pk_add/MFMA=%.1f: %.3f ms, %.1f TFLOP/s\n", LPM * 0.5, VPM * 0.5, ms, flop / ms / 1e9);
    }
}

int main() {
    float *tab, *out;
    hipMalloc(&tab, (65536 + 8192) * 2 * sizeof(float) + 4096);
    hipMemset(tab, 0, (65536 + 8192) * 2 * sizeof(float) + 4096);
    hipMalloc(&out, 256 * 256 * sizeof(float));
    run<0, 0>(tab, out);
    run<1, 0>(tab, out);
    run<2, 0>(tab, out);
    run<2, 1>(tab, out);
    run<2, 2>(tab, out);
    run<2, 4>(tab, out);
    return 0;
}
