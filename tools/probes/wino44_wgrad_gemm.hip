// Probe (round 5, for round 6): the matrix part of a weight gradient in F(4x4, 3x3)-tile form,
//   dU[xi][ci][co] = sum_tiles V[xi][tile][ci] Z[xi][tile][co]      (36 positions xi; V = B^T d B of the inputs, Z = A dY A^T of the output gradients),
// 135 GFLOP for the ConvLSTM cell's weight gradient at config 2 (128 x 256 channels, 7 x 8192 tiles) where the F(3x3, 2x2)-tile kernel of today
// (wino_wgrad_half_kernel, 2.11 ms) executes 240.  Would it run near its MFMA time (0.98 ms)?
// Design under test: K = the tile index.  A workgroup = 8 waves = FOUR positions x 2 halves of the 128 input channels, one 128-column block, one
// K split: a wave holds 64 x 128 outputs of one position (2 x 4 blocks of 32 x 32: 128 accumulator registers) - few positions per workgroup, so
// that the output block is large (21 FLOP per operand byte; all 36 positions in a workgroup would leave 32 x 64 blocks, 10.7).  Operands tile-major
// and blocked so that a wave's load is one contiguous kilobyte, [xi][channel block of 32][tile / 8][32 channels][8 tiles]: a lane's 16 bytes are 4
// k-steps of v_mfma_f32_32x32x2_f32; straight from L2 into a register ring, no LDS, no barrier.  Partial sums per K split go to memory (a reduction
// + G^T dU G would follow).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/wino44_wgrad_gemm.hip -o /tmp/w44wg && /tmp/w44wg
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <type_traits>
#include <utility>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define INL __attribute__((always_inline))

template <int... I, class F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F &&f) {
    sfor_impl(std::make_integer_sequence<int, N>(), f);
}

__device__ __forceinline__ i32x4 hdesc(const void *p) {
    const unsigned long long u = (unsigned long long)p;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    d[1] = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
    d[2] = 0x7fffffff;
    d[3] = 0x00020000;
    return d;
}

constexpr int RING = 3;          // k8-steps in flight: 6 loads each

// Vt [36][CI / 32][T8][32][8], Zt [36][CO / 32][T8][32][8], out [S][36][CI][CO]; T8 = tiles / 8, a multiple of S * RING
__global__ void __launch_bounds__(512, 1) wgrad44_gemm(const float *Vt, const float *Zt, float *out, const int CI, const int CO, const int T8, const int S) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pos_in = wave & 3, cih = wave >> 2;
    const int ncb = CO / 128;
    int b = blockIdx.x;
    const int split = b % S; b /= S;
    const int cb = b % ncb, pgrp = b / ncb;
    const int xi = pgrp * 4 + pos_in;
    const int k8n = T8 / S, k80 = split * k8n;
    // a block of 32 channels of one position: T8 kilobytes
    const i32x4 adesc = hdesc(Vt + ((long)xi * (CI / 32) + cih * 2) * T8 * 256);
    const i32x4 bdesc = hdesc(Zt + ((long)xi * (CO / 32) + cb * 4) * T8 * 256);
    const int lvoff = (l31 * 8 + kh * 4) * 4;
    const int blkstride = T8 * 1024;                                          // bytes between two channel blocks
    f32x4 a[RING][2], bq[RING][4];
    auto load = [&, &adesc = adesc, &bdesc = bdesc, &lvoff = lvoff, &blkstride = blkstride](auto r_tag, int k8) INL {
        constexpr int r = decltype(r_tag)::value;
        auto &ar = a;
        auto &br = bq;
        const int soff = __builtin_amdgcn_readfirstlane(k8 * 1024);
        const int v1 = lvoff + blkstride, v2 = lvoff + 2 * blkstride, v3 = lvoff + 3 * blkstride;
        asm volatile("s_nop 4\n\t"
                     "buffer_load_dwordx4 %0, %6, %10, %12 offen\n\t"
                     "buffer_load_dwordx4 %1, %7, %10, %12 offen\n\t"
                     "buffer_load_dwordx4 %2, %6, %11, %12 offen\n\t"
                     "buffer_load_dwordx4 %3, %7, %11, %12 offen\n\t"
                     "buffer_load_dwordx4 %4, %8, %11, %12 offen\n\t"
                     "buffer_load_dwordx4 %5, %9, %11, %12 offen"
                     : "=&v"(ar[r][0]), "=&v"(ar[r][1]), "=&v"(br[r][0]), "=&v"(br[r][1]), "=&v"(br[r][2]), "=&v"(br[r][3])
                     : "v"(lvoff), "v"(v1), "v"(v2), "v"(v3), "s"(adesc), "s"(bdesc), "s"(soff)
                     : "memory");
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    sfor<RING - 1>([&](auto r) INL { load(r, k80 + decltype(r)::value); });
    for (int k = 0; k < k8n; k += RING) {
        sfor<RING>([&, &acc = acc, &a = a, &bq = bq](auto r_tag) INL {
            constexpr int r = decltype(r_tag)::value;
            constexpr int rn = (r + RING - 1) % RING;
            // (past the end: a valid request nobody uses - the counts stay static)
            const int kn = k + r + RING - 1 < k8n ? k80 + k + r + RING - 1 : k80;
            load(std::integral_constant<int, rn>(), kn);
            asm volatile("s_waitcnt vmcnt(%c6)" : "+v"(a[r][0]), "+v"(a[r][1]), "+v"(bq[r][0]), "+v"(bq[r][1]), "+v"(bq[r][2]), "+v"(bq[r][3]) : "i"(6 * (RING - 1)) : "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t"
                                 "v_mfma_f32_32x32x2_f32 %0, %3, %4, %0\n\t"
                                 "v_mfma_f32_32x32x2_f32 %0, %5, %6, %0\n\t"
                                 "v_mfma_f32_32x32x2_f32 %0, %7, %8, %0"
                                 : "+v"(acc[i][j])
                                 : "v"(a[r][i].x), "v"(bq[r][j].x), "v"(a[r][i].y), "v"(bq[r][j].y), "v"(a[r][i].z), "v"(bq[r][j].z), "v"(a[r][i].w), "v"(bq[r][j].w));
        });
    }
    // (the clamped requests of the last steps are still in flight: their target registers must stay allocated until they land - without the
    // operands hipcc computed the store addresses into them in front of this wait, and the late data overwrote an address: a memory fault)
    sfor<RING>([&, &a = a, &bq = bq](auto r_tag) INL {
        constexpr int r = decltype(r_tag)::value;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[r][0]), "+v"(a[r][1]), "+v"(bq[r][0]), "+v"(bq[r][1]), "+v"(bq[r][2]), "+v"(bq[r][3])::"memory");
    });
    // acc[i][j][v]: row (input channel) = (v & 3) + 8 (v >> 2) + 4 kh of block i, column = l31 of block j
    float *o = out + ((long)split * 36 + xi) * CI * CO;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int ci = cih * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * kh, co = cb * 128 + j * 32 + l31;
                o[(long)ci * CO + co] = acc[i][j][v];
            }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static int run(int CI, int CO, int tiles, int S, bool check, int reps) {
    const int T8 = tiles / 8;
    std::vector<float> V((size_t)36 * tiles * CI), Z((size_t)36 * tiles * CO), Vt(V.size()), Zt(Z.size());
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    if (check) {
        for (auto &v : V) v = rnd();
        for (auto &v : Z) v = rnd();
        // logical [xi][tile][c] -> [xi][c / 32][tile / 8][c % 32][tile % 8]
        for (int xi = 0; xi < 36; ++xi)
            for (int t = 0; t < tiles; ++t) {
                for (int c = 0; c < CI; ++c) Vt[((((size_t)xi * (CI / 32) + c / 32) * T8 + t / 8) * 32 + c % 32) * 8 + t % 8] = V[((size_t)xi * tiles + t) * CI + c];
                for (int c = 0; c < CO; ++c) Zt[((((size_t)xi * (CO / 32) + c / 32) * T8 + t / 8) * 32 + c % 32) * 8 + t % 8] = Z[((size_t)xi * tiles + t) * CO + c];
            }
    }
    float *dV, *dZ, *dO;
    const size_t osz = (size_t)S * 36 * CI * CO;
    CK(hipMalloc(&dV, Vt.size() * 4)); CK(hipMalloc(&dZ, Zt.size() * 4)); CK(hipMalloc(&dO, osz * 4));
    CK(hipMemcpy(dV, Vt.data(), Vt.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dZ, Zt.data(), Zt.size() * 4, hipMemcpyHostToDevice));
    const int grid = 9 * (CO / 128) * S;
    hipLaunchKernelGGL(wgrad44_gemm, dim3(grid), dim3(512), 0, 0, dV, dZ, dO, CI, CO, T8, S);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    if (check) {
        std::vector<float> O(osz);
        CK(hipMemcpy(O.data(), dO, osz * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0;
        for (int xi = 0; xi < 36; xi += 5)
            for (int ci = 0; ci < CI; ci += 7)
                for (int co = 0; co < CO; co += 11) {
                    double want = 0, got = 0;
                    for (int t = 0; t < tiles; ++t) want += (double)V[((size_t)xi * tiles + t) * CI + ci] * Z[((size_t)xi * tiles + t) * CO + co];
                    for (int sp = 0; sp < S; ++sp) got += O[(((size_t)sp * 36 + xi) * CI + ci) * CO + co];
                    worst = fmax(worst, fabs(got - want));
                    scale = fmax(scale, fabs(want));
                }
        printf("check CI=%d CO=%d tiles=%d S=%d: worst |diff| %.3e (largest value %.3f)\n", CI, CO, tiles, S, worst, scale);
        if (worst > 1e-3 * fmax(scale, 1.0)) return 2;
    }
    if (reps > 0) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(wgrad44_gemm, dim3(grid), dim3(512), 0, 0, dV, dZ, dO, CI, CO, T8, S);
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wgrad44_gemm, dim3(grid), dim3(512), 0, 0, dV, dZ, dO, CI, CO, T8, S);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / reps, flop = 2.0 * 36 * (double)tiles * CI * CO;
        printf("time CI=%d CO=%d tiles=%d S=%d (%d workgroups): %.1f us per launch, %.1f TFLOP/s = %.3f of the 157.3 fp32 MFMA peak; operands %.2f GB\n", CI, CO, tiles, S,
               grid, us, flop / us * 1e-6, flop / us * 1e-6 / 157.3, 36.0 * tiles * (CI + CO) * 4 / 1e9);
    }
    hipFree(dV); hipFree(dZ); hipFree(dO);
    return 0;
}

int main() {
    if (int rc = run(128, 128, 96, 2, true, 0)) return rc;
    if (int rc = run(128, 256, 240, 5, true, 0)) return rc;
    // the ConvLSTM cell's weight gradient at config 2: 7 frames x 8 images x 1024 tiles, 128 -> 256 channels; 14 K splits = 252 workgroups
    if (int rc = run(128, 256, 7 * 8192 - 7 * 8192 % (8 * 14 * RING), 14, false, 50)) return rc;
    // refine conv1's: 640 input channels (5 slots x 2 directions x 64) -> 128 columns: as five 128-channel problems
    return run(128, 128, 7 * 8192 - 7 * 8192 % (8 * 28 * RING), 28, false, 50);
}
