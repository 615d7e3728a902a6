// Sustained v_mfma_f32_32x32x2_f32 rate of the whole chip with nothing else going on: the practical ceiling the
// convolution kernels are measured against (the nominal 157.3 TFLOP/s assumes 2.4 GHz under full MFMA load).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float *out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j)
        for (int v = 0; v < 16; ++v) s += acc[j][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run(int blocks_per_cu, int iters) {
    int ncu = 256;
    float *out;
    hipMalloc(&out, (size_t)ncu * blocks_per_cu * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(ncu * blocks_per_cu), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flop = 2.0 * 32 * 32 * 2 * (double)NACC * iters * 4.0 * ncu * blocks_per_cu;
        printf("acc=%d waves/SIMD=%d iters=%d: %.3f ms, %.1f TFLOP/s\n", NACC, blocks_per_cu, iters, ms, flop / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    run<4>(1, 200000);
    run<4>(2, 100000);
    run<8>(1, 100000);
    return 0;
}
