// Probe (round 5): the round-4 probe mfma_bf16_shapes.hip compared the two bf16 MFMA shapes with a compiler-generated 16x16x32 loop that
// carried 60 v_accvgpr_mov + 31 s_nop per 32 MFMAs (hipcc renamed the accumulators every iteration), so its "1.55 PFLOP/s at 2.4 GHz" was an
// issue-bound loop, not the shape's rate.  Here both loops are inline asm on accumulators pinned in place ("+a"), operands in registers,
// random data, 1 or 2 waves per SIMD, long enough for the governor to settle; in-kernel clock = s_memtime / s_memrealtime.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_bf16_shapes2.hip -o mfma_bf16_shapes2
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
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = rnd(threadIdx.x * 7 + 1 + 977 * i);
    for (int i = 0; i < 2; ++i) b[i] = rnd(threadIdx.x * 3 + 9 + 131 * i);
    float s = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if (SHAPE == 0) {
        f32x16 acc[8];
        for (int j = 0; j < 8; ++j)
            for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a[j >> 1]), "v"(b[j & 1]));
        }
        for (int j = 0; j < 8; ++j)
            for (int v = 0; v < 16; ++v) s += acc[j][v];
    } else {
        f32x4 acc[32];
        for (int j = 0; j < 32; ++j)
            for (int v = 0; v < 4; ++v) acc[j][v] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 32; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a[j & 3]), "v"(b[(j >> 2) & 1]));
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
                const double cyc = (double)h[0] / iters / (shape ? 32 : 16);
                printf("%s  %d wave(s)/SIMD: %8.2f ms  %7.1f TFLOP/s  in-kernel clock %.0f MHz  %.1f cycles per MFMA per wave\n", shape ? "16x16x32" : "32x32x16", wps, ms,
                       flop / ms / 1e9, (double)h[0] / (double)h[1] * 100.0, cyc);
            }
    return 0;
}
