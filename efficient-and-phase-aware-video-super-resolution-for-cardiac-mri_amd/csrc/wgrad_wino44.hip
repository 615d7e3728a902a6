// Weight gradient of a 3x3 convolution in Winograd form F(4x4, 3x3) over 4x4 output tiles on fp32 MFMA for gfx950 - rnh_wino44_tmajor,
// rnh_wino44_wgrad_gemm, rnh_wino44_wgrad_finish.
//
//   dg = G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G        per (input channel, output channel): 36 GEMMs over the tiles,
//   dU[xi][ci][co] = sum_t V[xi][t][ci] Z[xi][t][co]          2.25 multiplications per (pixel, ci, co) where the F(3x3, 2x2)-tile form of
//                                                             csrc/wgrad_wino.hip needs 4 and the direct form 9.
// Three kernels (tools/probes/wino44_wgrad_gemm.hip measured the middle one: 0.92 of the fp32 MFMA peak on the ConvLSTM cell's problem):
//   tmajor   V = B^T d B of the inputs (6x6 patches, zero padding of 1) or Z = A dY A^T of the output gradients (4x4 tiles), written TILE-MAJOR and
//            blocked, [xi][channel block of 32][tile / 8][32 channels][8 tiles]: the K dimension of the GEMMs is the tile index, a wave's operand load
//            is one contiguous kilobyte, a lane's 16 bytes are four k-steps of v_mfma_f32_32x32x2_f32.  A tensor holds all frames of a source, so
//            that window slot j of refine conv1 (frame f + j against the gradient of window f) is the same tensor j frames further on;
//   gemm     a workgroup = 8 waves = FOUR positions x the 2 halves of a problem's 128 input channels (each half = one source tensor: h_fwd | h_bwd,
//            x | h), one block of 128 output channels, one K split; a wave holds 64 x 128 outputs of one position (128 accumulators); operands
//            straight from L2 into a register ring, no LDS, no barrier; partial sums per K split to memory (fixed order: no atomics);
//   finish   sums the K splits, dg = G^T dU G, scatters into dw[co][ci][3][3] (OIHW, accumulating or not); the bias gradient is the tile sum of
//            Z at position (1, 1): 1^T dY 1 = w^T Z w with A^T w = 1 for w = e_1.
// Replaces the weight part of aten::convolution_backward of refine conv1 over the hidden states (reference src/model/nets/refine_net.py:149,
// loss.backward() at acdc_vsr_refinenet_trainer.py:46) where hipvsr/hip_ops.py selects it; results differ from rnh_wino_wgrad / rnh_conv_wgrad by
// fp32 rounding of the transforms and the summation order.
#include "rnh_common.h"
#include <type_traits>
#include <utility>

namespace {

typedef float f32x4w __attribute__((ext_vector_type(4)));
#define WG4_INL __attribute__((always_inline))

template <int... I, class F>
__device__ __forceinline__ void wg4_sfor_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void wg4_sfor(F &&f) {
    wg4_sfor_impl(std::make_integer_sequence<int, N>(), f);
}

__device__ __forceinline__ i32x4 wg4_desc(const void *p) {
    const unsigned long long u = (unsigned long long)p;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    d[1] = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
    d[2] = 0x7fffffff;
    d[3] = 0x00020000;
    return d;
}

// tile t of the list -> (image, tile row, tile column): the order of csrc/conv_wino44.hip (any order would do: V and Z use the same)
__device__ __forceinline__ void wg4_tile_xy(int t, int TX, int TY, int &img, int &ty, int &tx) {
    img = t / (TY * TX);
    const int rem = t - img * TY * TX;
    ty = rem / TX;
    tx = rem - ty * TX;
}

// MODE 0: V = B^T d B (6x6 patch around the tile, zero padding); MODE 1: Z = A dY A^T (the 4x4 tile).  A wave = 8 tiles x 32 channels = one
// (channel block, k8) kilobyte per position: lane = (tile & 7) + 8 * (channel quad)
template <int MODE>
__global__ void __launch_bounds__(256) wino44_tmajor_kernel(const float *x, const int C, const int c0, const int nblk, const int B, const int H, const int W,
                                                           const int TX, const int TY, const long K8, float *out) {
    const int lane = threadIdx.x & 63;
    const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= K8 * nblk) return;
    const int blk = (int)(gw % nblk);
    const long k8 = gw / nblk;
    const int t8 = lane & 7, cq = lane >> 3;
    const int t = (int)(k8 * 8 + t8);
    int img, ty, tx;
    wg4_tile_xy(t, TX, TY, img, ty, tx);
    const float *xp = x + c0 + blk * 32 + cq * 4;
    // [xi][blk][k8][32 channels][8 tiles]
    float *o = out + ((long)blk * K8 + k8) * 256 + cq * 32 + t8;
    const long xstride = (long)nblk * K8 * 256;
    auto put = [&](int xi, const f32x4w u) WG4_INL {
        float *p = o + xi * xstride;
        p[0] = u.x; p[8] = u.y; p[16] = u.z; p[24] = u.w;
    };
    if constexpr (MODE == 0) {
        f32x4w d[6][6];
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int y = y0 + i, xx = x0 + j;
                const bool in = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
                const long pix = ((long)img * H + (in ? y : 0)) * W + (in ? xx : 0);
                const f32x4w ld = *reinterpret_cast<const f32x4w *>(xp + pix * C);
                d[i][j] = in ? ld : f32x4w{0.f, 0.f, 0.f, 0.f};
            }
        auto bt6 = [](const f32x4w d0, const f32x4w d1, const f32x4w d2, const f32x4w d3, const f32x4w d4, const f32x4w d5, f32x4w *r) WG4_INL {
            const f32x4w a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = d3 - d1;
            r[0] = 4.f * d0 - 5.f * d2 + d4;
            r[1] = a + b;
            r[2] = a - b;
            r[3] = c + 2.f * e;
            r[4] = c - 2.f * e;
            r[5] = 4.f * d1 - 5.f * d3 + d5;
        };
        f32x4w tq[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            f32x4w r[6];
            bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], r);
#pragma unroll
            for (int i = 0; i < 6; ++i) tq[i][j] = r[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            f32x4w r[6];
            bt6(tq[i][0], tq[i][1], tq[i][2], tq[i][3], tq[i][4], tq[i][5], r);
#pragma unroll
            for (int j = 0; j < 6; ++j) put(6 * i + j, r[j]);
        }
    } else {
        // A = [1 0 0 0; 1 1 1 1; 1 -1 1 -1; 1 2 4 8; 1 -2 4 -8; 0 0 0 1]  (the transpose of the forward's A^T)
        f32x4w d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f32x4w *>(xp + (((long)img * H + 4 * ty + i) * W + 4 * tx + j) * C);
        auto a6 = [](const f32x4w y0, const f32x4w y1, const f32x4w y2, const f32x4w y3, f32x4w *r) WG4_INL {
            const f32x4w s02 = y0 + y2, s13 = y1 + y3, e = y0 + 4.f * y2, o = 2.f * y1 + 8.f * y3;
            r[0] = y0;
            r[1] = s02 + s13;
            r[2] = s02 - s13;
            r[3] = e + o;
            r[4] = e - o;
            r[5] = y3;
        };
        f32x4w tq[6][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4w r[6];
            a6(d[0][j], d[1][j], d[2][j], d[3][j], r);
#pragma unroll
            for (int i = 0; i < 6; ++i) tq[i][j] = r[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            f32x4w r[6];
            a6(tq[i][0], tq[i][1], tq[i][2], tq[i][3], r);
#pragma unroll
            for (int j = 0; j < 6; ++j) put(6 * i + j, r[j]);
        }
    }
}

constexpr int WG4_RING = 2;             // k8-steps in flight (6 requests each)
constexpr int WG4_MAX_PROB = 8;

struct wg4_gemm_args {
    const float *a[WG4_MAX_PROB][2];    // per problem: the two 64-channel halves of its 128 input channels, each [36][2][K8a][32][8], at the problem's first k8
    long a_k8[WG4_MAX_PROB][2];         // K8 of those tensors (their block stride in kilobytes)
    const float *z;                     // [36][CO / 32][K8z][32][8]
    long z_k8;
    float *out;                         // [S][nprob][36][128][CO]
    int nprob, CO, T8, S;
};

__global__ void __launch_bounds__(512, 1) wino44_wgrad_gemm_kernel(const wg4_gemm_args P) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pos_in = wave & 3, cih = wave >> 2;
    const int ncb = P.CO / 128;
    int b = blockIdx.x;
    const int split = b % P.S; b /= P.S;
    const int cb = b % ncb; b /= ncb;
    const int pgrp = b % 9, prob = b / 9;
    const int xi = pgrp * 4 + pos_in;
    const int k8n = P.T8 / P.S, k80 = split * k8n;
    const long ak8 = P.a_k8[prob][cih];
    const i32x4 adesc = wg4_desc(P.a[prob][cih] + (long)xi * 2 * ak8 * 256);
    const i32x4 bdesc = wg4_desc(P.z + ((long)xi * (P.CO / 32) + cb * 4) * P.z_k8 * 256);
    const int lvoff = (l31 * 8 + kh * 4) * 4;
    const int va1 = lvoff + (int)(ak8 * 1024);
    const int zs = (int)(P.z_k8 * 1024);
    const int vb1 = lvoff + zs, vb2 = lvoff + 2 * zs, vb3 = lvoff + 3 * zs;
    f32x4w a[WG4_RING][2], bq[WG4_RING][4];
    auto load = [&, &adesc = adesc, &bdesc = bdesc, &lvoff = lvoff, &va1 = va1, &vb1 = vb1, &vb2 = vb2, &vb3 = vb3](auto r_tag, int k8) WG4_INL {
        constexpr int r = decltype(r_tag)::value;
        auto &ar = a;
        auto &br = bq;
        const int soff = __builtin_amdgcn_readfirstlane(k8 * 1024);
        asm volatile("s_nop 4\n\t"
                     "buffer_load_dwordx4 %0, %6, %11, %13 offen\n\t"
                     "buffer_load_dwordx4 %1, %7, %11, %13 offen\n\t"
                     "buffer_load_dwordx4 %2, %6, %12, %13 offen\n\t"
                     "buffer_load_dwordx4 %3, %8, %12, %13 offen\n\t"
                     "buffer_load_dwordx4 %4, %9, %12, %13 offen\n\t"
                     "buffer_load_dwordx4 %5, %10, %12, %13 offen"
                     : "=&v"(ar[r][0]), "=&v"(ar[r][1]), "=&v"(br[r][0]), "=&v"(br[r][1]), "=&v"(br[r][2]), "=&v"(br[r][3])
                     : "v"(lvoff), "v"(va1), "v"(vb1), "v"(vb2), "v"(vb3), "s"(adesc), "s"(bdesc), "s"(soff)
                     : "memory");
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    wg4_sfor<WG4_RING - 1>([&](auto r) WG4_INL { load(r, k80 + decltype(r)::value); });
    for (int k = 0; k < k8n; k += WG4_RING) {
        wg4_sfor<WG4_RING>([&, &acc = acc, &a = a, &bq = bq](auto r_tag) WG4_INL {
            constexpr int r = decltype(r_tag)::value;
            constexpr int rn = (r + WG4_RING - 1) % WG4_RING;
            // (past the end: a valid request nobody uses - the counts stay static)
            const int kn = k + r + WG4_RING - 1 < k8n ? k80 + k + r + WG4_RING - 1 : k80;
            load(std::integral_constant<int, rn>(), kn);
            asm volatile("s_waitcnt vmcnt(%c6)"
                         : "+v"(a[r][0]), "+v"(a[r][1]), "+v"(bq[r][0]), "+v"(bq[r][1]), "+v"(bq[r][2]), "+v"(bq[r][3])
                         : "i"(6 * (WG4_RING - 1))
                         : "memory");
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
    // (the clamped requests of the last steps are still in flight: their target registers stay allocated until they land - a wait without these
    // operands let hipcc compute the store addresses into them, and the late data overwrote an address: tools/probes/wino44_wgrad_gemm.hip)
    wg4_sfor<WG4_RING>([&, &a = a, &bq = bq](auto r_tag) WG4_INL {
        constexpr int r = decltype(r_tag)::value;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[r][0]), "+v"(a[r][1]), "+v"(bq[r][0]), "+v"(bq[r][1]), "+v"(bq[r][2]), "+v"(bq[r][3])::"memory");
    });
    // acc[i][j][v]: row (input channel) = (v & 3) + 8 (v >> 2) + 4 kh of block i, column = l31 of block j
    float *o = P.out + (((long)split * P.nprob + prob) * 36 + xi) * 128 * P.CO;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int ci = cih * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * kh, co = cb * 128 + j * 32 + l31;
                o[(long)ci * P.CO + co] = acc[i][j][v];
            }
}

// the K splits summed in order, in place into split 0: thread = four consecutive elements of [nprob][36][128][CO]
__global__ void __launch_bounds__(256) wino44_wgrad_splitsum_kernel(float *part, const int S, const long n4) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    f32x4w *p = reinterpret_cast<f32x4w *>(part) + idx;
    f32x4w u = p[0];
    for (int s = 1; s < S; ++s) u += p[s * n4];
    p[0] = u;
}

// dw[co][rowbase[prob] + ci][a][b] (+)= sum_{i, j} G[i][a] G[j][b] dU[prob][6 i + j][ci][co]; thread = (prob, ci, co)
__global__ void __launch_bounds__(256) wino44_wgrad_finish_kernel(const float *part, const int nprob, const int CO, const int *rowbase, const int ncol, const int Cin,
                                                                 float *dw, const int accumulate) {
    const long total = (long)nprob * 128 * CO;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int co = (int)(idx % CO), ci = (int)((idx / CO) % 128), prob = (int)(idx / ((long)CO * 128));
    if (co >= ncol) return;
    const float G[6][3] = {{0.25f, 0.f, 0.f},          {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    float dg[3][3] = {};
    const long pstride = 128L * CO;
    const float *p0 = part + (long)prob * 36 * pstride + (long)ci * CO + co;
    float u[36];
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) u[xi] = p0[xi * pstride];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float rowg[3] = {0.f, 0.f, 0.f};                                     // sum_j dU[i][j] G[j][b]
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) rowg[bb] = __builtin_fmaf(u[6 * i + j], G[j][bb], rowg[bb]);
#pragma unroll
        for (int aa = 0; aa < 3; ++aa)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) dg[aa][bb] = __builtin_fmaf(G[i][aa], rowg[bb], dg[aa][bb]);
    }
    float *o = dw + ((long)co * Cin + rowbase[prob] + ci) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) o[t] = accumulate ? o[t] + dg[t / 3][t % 3] : dg[t / 3][t % 3];
}

// bias gradient: db[co] (+)= sum over the K8 kilobyte rows of Z at position xi = 7 = (1, 1); two stages, fixed order
__global__ void __launch_bounds__(256) wino44_zsum_kernel(const float *z, const int CO, const long K8, const int nchunk, float *part) {
    // block = (chunk, channel block); thread = (channel 32, 8 k8 lanes): partial sums over the chunk's k8 range
    const int blk = blockIdx.x % (CO / 32), chunk = blockIdx.x / (CO / 32);
    const int c = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const long per = (K8 + nchunk - 1) / nchunk, k0 = chunk * per, k1 = k0 + per < K8 ? k0 + per : K8;
    const float *p = z + ((7L * (CO / 32) + blk) * K8) * 256 + c * 8;
    float s = 0.f;
    for (long k = k0 + sub; k < k1; k += 8) {
        const f32x4w u0 = *reinterpret_cast<const f32x4w *>(p + k * 256), u1 = *reinterpret_cast<const f32x4w *>(p + k * 256 + 4);
        s += ((u0.x + u0.y) + (u0.z + u0.w)) + ((u1.x + u1.y) + (u1.z + u1.w));
    }
    __shared__ float red[8][32];
    red[sub][c] = s;
    __syncthreads();
    if (sub == 0) {
        float t = 0.f;
        for (int q = 0; q < 8; ++q) t += red[q][c];
        part[(long)chunk * CO + blk * 32 + c] = t;
    }
}
__global__ void wino44_zsum_finish_kernel(const float *part, const int CO, const int nchunk, const int ncol, float *db, const int accumulate) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= ncol) return;
    float t = 0.f;
    for (int q = 0; q < nchunk; ++q) t += part[(long)q * CO + co];
    db[co] = accumulate ? db[co] + t : t;
}

}  // namespace

extern "C" int64_t rnh_wino44_tmajor_floats(int B, int H, int W, int nch) {
    if (B < 1 || H < 4 || W < 4 || (H & 3) || (W & 3) || nch < 32 || (nch & 31)) return 0;
    const long tiles = (long)B * (H / 4) * (W / 4);
    if (tiles & 7) return 0;
    return 36L * (nch / 32) * (tiles / 8) * 256;
}

extern "C" int rnh_wino44_tmajor(const float *x, int C, int c0, int nch, int B, int H, int W, int mode, float *out, void *stream) {
    if (!x || !out || C < 1 || c0 < 0 || nch < 32 || c0 + nch > C || B < 1 || (mode != 0 && mode != 1)) RNH_FAIL(RNH_E_ARG, "rnh_wino44_tmajor: bad arguments");
    if ((C & 3) || (c0 & 3) || (nch & 31)) RNH_FAIL(RNH_E_ALIGN, "rnh_wino44_tmajor: C, c0 multiples of 4, nch a multiple of 32");
    if (H < 4 || W < 4 || (H & 3) || (W & 3)) RNH_FAIL(RNH_E_RANGE, "rnh_wino44_tmajor: H and W must be multiples of 4");
    const int TX = W / 4, TY = H / 4;
    const long tiles = (long)B * TY * TX;
    if (tiles & 7) RNH_FAIL(RNH_E_RANGE, "rnh_wino44_tmajor: the number of tiles must be a multiple of 8");
    if ((long)B * H * W >= (1L << 27)) RNH_FAIL(RNH_E_RANGE, "rnh_wino44_tmajor: too many pixels for 32-bit tile indices");
    const long K8 = tiles / 8, waves = K8 * (nch / 32);
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (mode == 0) hipLaunchKernelGGL((wino44_tmajor_kernel<0>), grid, dim3(256), 0, (hipStream_t)stream, x, C, c0, nch / 32, B, H, W, TX, TY, K8, out);
    else hipLaunchKernelGGL((wino44_tmajor_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, x, C, c0, nch / 32, B, H, W, TX, TY, K8, out);
    RNH_CHECK_LAUNCH("rnh_wino44_tmajor");
    return 0;
}

extern "C" int rnh_wino44_wgrad_gemm(const rnh_wino44_wgrad_args_t *args, void *stream) {
    if (!args) RNH_FAIL(RNH_E_ARG, "rnh_wino44_wgrad_gemm: null args");
    const rnh_wino44_wgrad_args_t &a = *args;
    if (a.nprob < 1 || a.nprob > WG4_MAX_PROB || !a.z || !a.part || a.CO < 128 || (a.CO & 127) || a.T8 < 1 || a.S < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_wino44_wgrad_gemm: bad arguments");
    if (a.T8 % (a.S * WG4_RING)) RNH_FAIL(RNH_E_RANGE, "rnh_wino44_wgrad_gemm: T8 must be a multiple of S * %d", WG4_RING);
    wg4_gemm_args p = {};
    for (int i = 0; i < a.nprob; ++i)
        for (int h = 0; h < 2; ++h) {
            if (!a.a[i][h] || a.a_k8[i][h] < a.a_k80[i][h] + a.T8 || a.a_k80[i][h] < 0) RNH_FAIL(RNH_E_ARG, "rnh_wino44_wgrad_gemm: bad input operand %d.%d", i, h);
            if (a.a_k8[i][h] * 1024 * 2 >= (1L << 31)) RNH_FAIL(RNH_E_RANGE, "rnh_wino44_wgrad_gemm: an input operand of at most 2^20 k8 rows");
            p.a[i][h] = a.a[i][h] + a.a_k80[i][h] * 256;                      // (the problem's first k8 row inside every block: blocks are a_k8 rows apart)
            p.a_k8[i][h] = a.a_k8[i][h];
        }
    if (a.z_k8 < a.T8 || a.z_k8 * 1024 * 4 >= (1L << 31)) RNH_FAIL(RNH_E_RANGE, "rnh_wino44_wgrad_gemm: bad gradient operand size");
    p.z = a.z, p.z_k8 = a.z_k8, p.out = a.part, p.nprob = a.nprob, p.CO = a.CO, p.T8 = a.T8, p.S = a.S;
    const unsigned grid = (unsigned)(a.nprob * 9 * (a.CO / 128) * a.S);
    hipLaunchKernelGGL(wino44_wgrad_gemm_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
    RNH_CHECK_LAUNCH("rnh_wino44_wgrad_gemm");
    return 0;
}

extern "C" int rnh_wino44_wgrad_finish(const rnh_wino44_wgrad_args_t *args, const int32_t *rowbase /* device [nprob] */, int ncol, int Cin, float *dw, float *db,
                                       float *zpart /* nchunk * CO floats or 0 */, int nchunk, int accumulate, void *stream) {
    if (!args || !rowbase || !dw || ncol < 1 || Cin < 1) RNH_FAIL(RNH_E_ARG, "rnh_wino44_wgrad_finish: bad arguments");
    const rnh_wino44_wgrad_args_t &a = *args;
    if (ncol > a.CO) RNH_FAIL(RNH_E_ARG, "rnh_wino44_wgrad_finish: more columns than the problem has");
    const long total = (long)a.nprob * 128 * a.CO, n4 = total * 36 / 4;
    if (a.S > 1) hipLaunchKernelGGL(wino44_wgrad_splitsum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a.part, a.S, n4);
    hipLaunchKernelGGL(wino44_wgrad_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a.part, a.nprob, a.CO, rowbase, ncol, Cin, dw,
                       accumulate);
    RNH_CHECK_LAUNCH("rnh_wino44_wgrad_finish");
    if (db) {
        if (!zpart || nchunk < 1) RNH_FAIL(RNH_E_ARG, "rnh_wino44_wgrad_finish: the bias gradient needs its workspace");
        hipLaunchKernelGGL(wino44_zsum_kernel, dim3((unsigned)(nchunk * (a.CO / 32))), dim3(256), 0, (hipStream_t)stream, a.z, a.CO, (long)a.z_k8, nchunk, zpart);
        hipLaunchKernelGGL(wino44_zsum_finish_kernel, dim3((unsigned)((ncol + 127) / 128)), dim3(128), 0, (hipStream_t)stream, zpart, a.CO, nchunk, ncol, db, accumulate);
        RNH_CHECK_LAUNCH("rnh_wino44_wgrad_finish (bias)");
    }
    return 0;
}
