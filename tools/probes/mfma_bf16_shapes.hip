// Probe (round 4): sustained rate of the two bf16 MFMA shapes on random operands with nothing else going on - MI355X_MICROARCH.md
// ("DVFS give-back", item 7) reports ~1.15x the FLOP/s for v_mfma_f32_16x16x32_bf16 over 32x32x16 at equal cycles per FLOP, because the
// chip holds a higher clock under it.  Operands in registers, one or two waves per SIMD, long enough for the governor to settle.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_bf16_shapes.hip -o mfma_bf16_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 rnd(unsigned s) {
    bf16x8 v;
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u;
        v[i] = (__bf16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f);
    }
    return v;
}

template <int SHAPE>   // 0: 32x32x16 (8 accumulators of 16), 1: 16x16x32 (32 accumulators of 4): the same 128 accumulator registers, the same FLOPs per pass
__global__ void __launch_bounds__(256) loop(float *out, int iters, unsigned long long *clk) {
    const bf16x8 a0 = rnd(threadIdx.x * 7 + 1), a1 = rnd(threadIdx.x * 13 + 5), b0 = rnd(threadIdx.x * 3 + 9), b1 = rnd(threadIdx.x * 11 + 2);
    float s = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if (SHAPE == 0) {
        f32x16 acc[8];
        for (int j = 0; j < 8; ++j)
            for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((j & 1) ? a1 : a0, (j & 2) ? b1 : b0, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((j & 2) ? a1 : a0, (j & 1) ? b1 : b0, acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 8; ++j)
            for (int v = 0; v < 16; ++v) s += acc[j][v];
    } else {
        f32x4 acc[32];
        for (int j = 0; j < 32; ++j)
            for (int v = 0; v < 4; ++v) acc[j][v] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 32; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((j & 1) ? a1 : a0, (j & 2) ? b1 : b0, acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 32; ++j)
            for (int v = 0; v < 4; ++v) s += acc[j][v];
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 100 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    float *out;
    unsigned long long *clk, h[2];
    hipMalloc(&out, 512 * 256 * sizeof(float));
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wps = 1; wps <= 2; ++wps)
        for (int shape = 0; shape < 2; ++shape)
            for (int rep = 0; rep < 3; ++rep) {
                const int iters = 60000 / wps;
                hipEventRecord(e0);
                if (shape == 0) hipLaunchKernelGGL(loop<0>, dim3(256 * wps), dim3(256), 0, 0, out, iters, clk);
                else hipLaunchKernelGGL(loop<1>, dim3(256 * wps), dim3(256), 0, 0, out, iters, clk);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
                // per pass of the loop body: 16 x 32x32x16 = 32 x 16x16x32 = 262 144 MAC per wave
                const double flop = 2.0 * 262144.0 * iters * 4.0 * 256 * wps;
                printf("%s  %d wave(s)/SIMD: %8.2f ms  %7.1f TFLOP/s  in-kernel clock %.0f MHz\n", shape ? "16x16x32" : "32x32x16", wps, ms, flop / ms / 1e9,
                       (double)h[0] / (double)h[1] * 100.0);
            }
    return 0;
}
