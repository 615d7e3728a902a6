// The collapsed upsampler tail (csrc/uptail.hip; reference src/model/nets/refine_net.py:199-205 and its backward) for the
// bf16-storage path: the tail's input Y1 (B, Hm, Wm, 64) and the gradient dY1 that leaves it are bf16 in HBM, everything is
// contracted on v_mfma_f32_16x16x32_bf16 with fp32 accumulators.  Built for what the x4 / x8 nets need: r = 2, C1 = 64,
// out_channels = 1 (rnh_uptail_bf16_supported); everything else stays on the fp32 kernels of uptail.hip.
//
// The fp32 kernels are VALU kernels at 60-80 TFLOP/s (1.8 / 1.2 / 1.9 ms per stage at BASELINE config 2: 16 % of the bf16
// step).  On the matrix cores the same algebra is a few hundred GFLOP: all three kernels become HBM-bound - Y1 / dY1 are
// 1.4 GB each way per stage in bf16 (2.8 GB in fp32), d_o / out 176 MB.
//
//   forward   out[(2qy+i, 2qx+j)] = b3 + bsum[ij] + sum_{u in 5x5, c} Y1[q + u - 2][c] Kf[u][c][ij]          (uptail.hip)
//       MFMA rows m = (dy, ij): FOUR consecutive output rows qy0 + dy share one accumulator tile, the contraction runs over
//       u'y = uy + dy in 0..7, ux, c (K = 2560) with A[m][(u'y, ux, c)] = Kf[u'y - dy][ux][c][ij] (zero outside 0..4) read per
//       lane from a compact table in LDS; columns n = 16 consecutive pixels of a row, B = the halo image read at a shifted
//       address.  80 MFMAs per 64 output pixels instead of 200 for one output row per tile: every B fragment read feeds four
//       output rows, and all 64 lanes end up with the four sub-positions of one output pixel.
//   dgrad     dY1[q][c] = sum_{off in [-3,4]^2} d_o[2q + off] Kd[off][c]                                       (uptail.hip)
//       rows = channels (A = Kd, resident in registers), columns = 16 pixels, K = the 64 offsets.  The column operand is
//       Toeplitz in x: for a fixed offset row and column parity the 16 pixels read a sliding window of 4 values.  d_o is split
//       once per tile into bf16 hi + lo parts (the loss gradient is a constant +-c for L1: one bf16 would put the same relative
//       rounding error on every gradient of a stage; hi + lo carries 16 bits) and de-interleaved by column parity into planes,
//       each stored 4 times with a shift of 0..3 elements, so that every lane finds its window at an 8-byte aligned address
//       (two ds_read_b64 per fragment, no im2col).
//   xcorr     X[off][c] = sum_q Y1[q][c] d_o[2q + off],  SX[off] = sum_q d_o[2q + off]                          (uptail.hip)
//       rows = offsets (A from the same planes: 8 consecutive pixels of one offset), columns = channels (B = Y1 transposed by
//       ds_read_b64_tr_b16 from a swizzled [pixel][64] image), K = pixels; a fifth column tile of ones yields SX.
//       Partial sums per workgroup go to the slab layout of uptail_xcorr_kernel; border terms and the fixed-order reduction are
//       uptail.hip's.
//
// MFMA operand maps (cdna_hip_programming.md section 3), v_mfma_f32_16x16x32_bf16: lane l holds A[row l & 15][k = 8 (l >> 4) + j]
// and B[k = 8 (l >> 4) + j][col l & 15], j = 0..7; C/D: col = l & 15, row = 4 (l >> 4) + reg.
#include "rnh_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ unsigned short bf_bits(float v) {
    const __bf16 b = (__bf16)v;                                   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf_val(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ unsigned pk2(float a, float b) { return (unsigned)bf_bits(a) | ((unsigned)bf_bits(b) << 16); }

// ---------------------------------------------------------------------------------------------------------------------
// d_o planes of an 8 x 32 tile of mid-resolution pixels (dgrad and xcorr).  HR rows 2 y0 - 3 .. 2 y0 + 18 (22), HR columns
// 2 x0 - 3 .. 2 x0 + 66 (70) -> per row two parity planes indexed by i = floor(px / 2) - (x0 - 2): odd HR columns i = 0..34,
// even ones i = 1..35.  plane[hi/lo][parity][shift s][row][slot], element i at slot i + s: a window of 4 starting at i0 is
// read from copy s = -i0 & 3 at the 8-byte aligned slot i0 + s.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TH = 8, TW = 32;
constexpr int PROWS = 2 * TH + 6, PCOLS = 2 * TW + 6;
constexpr int PROW_B = 40 * 2;                                    // 40 slots of 2 bytes
constexpr int PCOPY_B = PROWS * PROW_B, PPAR_B = 4 * PCOPY_B, PHL_B = 2 * PPAR_B, PLANES_B = 2 * PHL_B;      // 1760 / 7040 / 14080 / 28160

__device__ __forceinline__ void fill_planes(unsigned char *pl, const float *__restrict__ dO, int b, int y0, int x0, int Hh, int Wh, int tid) {
    for (int e = tid; e < PROWS * PCOLS; e += 256) {
        const int pr = e / PCOLS, pc = e - pr * PCOLS;
        const int py = 2 * y0 - 3 + pr, px = 2 * x0 - 3 + pc;
        float v = 0.f;
        if ((unsigned)py < (unsigned)Hh && (unsigned)px < (unsigned)Wh) v = dO[((long)b * Hh + py) * Wh + px];
        const unsigned short hi = bf_bits(v), lo = bf_bits(v - bf_val(hi));
        const int par = (pc + 1) & 1, i = (pc + 1) >> 1;          // 2 x0 - 3 is odd: pc even <-> odd HR column
        unsigned char *base = pl + par * PPAR_B + pr * PROW_B + 2 * i;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            *reinterpret_cast<unsigned short *>(base + s * (PCOPY_B + 2)) = hi;
            *reinterpret_cast<unsigned short *>(base + s * (PCOPY_B + 2) + PHL_B) = lo;
        }
    }
}

// byte offset (inside one hi / lo half) of the 4-element window of plane `par`, row `pr`, starting at element i0
__device__ __forceinline__ int win_off(int par, int pr, int i0) {
    const int s = (-i0) & 3;
    return par * PPAR_B + s * PCOPY_B + pr * PROW_B + 2 * (i0 + s);
}

// ---------------------------------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------------------------------
// KfT[(uy*5 + ux)*4 + ij][c] = fp16(Kf[(uy*5 + ux)][c][ij])  (Kf of uptail_compose_fwd2_kernel, C1p = 64, NOP = 4).  IEEE half, not bf16 (round 6):
// the forward contracts on v_mfma_f32_16x16x32_f16 - the same rate - with Y1's bf16 values converted exactly (8 mantissa bits fit 11; below 2^-14
// they lose bits, absolute error < 2^-25), so the composed weights carry 11 bits instead of 8.  The outputs are one linear map away from the PSNR
// the contract is stated in: bf16 weights HERE moved it by up to 0.009 dB at trained weights (profiles/r06_a_*, r06_b_*), fp16 ones by 0.001
__global__ void pack_kft_kernel(const float *__restrict__ Kf, unsigned short *__restrict__ KfT) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 25 * 4 * 64) return;
    const int c = e & 63, ij = (e >> 6) & 3, u = e >> 8;
    const _Float16 h = (_Float16)Kf[((long)u * 64 + c) * 4 + ij];     // v_cvt_f16_f32: RNE
    KfT[e] = __builtin_bit_cast(unsigned short, h);
}
// eight bf16 -> eight fp16 (exact above 2^-14: v_cvt_pkrtz_f16_f32 truncates nothing there)
__device__ __forceinline__ unsigned bf2h2(unsigned x) {
    const float lo = __builtin_bit_cast(float, x << 16), hi = __builtin_bit_cast(float, x & 0xffff0000u);
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi));
}
__device__ __forceinline__ uint4 bf2h8(uint4 v) { return make_uint4(bf2h2(v.x), bf2h2(v.y), bf2h2(v.z), bf2h2(v.w)); }
// KdP[(kc*4 + mt)*64 + lane][j] = bf16(Kd[off(k)][channel(mt, lane & 15)]), k = 32 kc + 8 (lane >> 4) + j <-> offset row
// oy = 4 kc + (lane >> 4) and offset column ox = 2 j (j < 4: odd HR columns) / 2 (j - 4) + 1 (even HR columns);
// channel(mt, R) = 32 (mt >> 1) + 8 (R >> 2) + 4 (mt & 1) + (R & 3): a lane's accumulators are two runs of 8 channels
__global__ void pack_kd_kernel(const float *__restrict__ Kd, unsigned short *__restrict__ KdP) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 8 * 64 * 8) return;
    const int j = e & 7, lane = (e >> 3) & 63, fm = e >> 9, kc = fm >> 2, mt = fm & 3;
    const int R = lane & 15, g = lane >> 4;
    const int oy = 4 * kc + g, ox = j < 4 ? 2 * j : 2 * (j - 4) + 1;
    const int ch = 32 * (mt >> 1) + 8 * (R >> 2) + 4 * (mt & 1) + (R & 3);
    KdP[e] = bf_bits(Kd[(long)(oy * 8 + ox) * 64 + ch]);
}

// ---------------------------------------------------------------------------------------------------------------------
// dgrad: dY1 (bf16) from d_o
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) uptail_dgrad_bf16_kernel(const float *__restrict__ dO, const unsigned short *__restrict__ KdP,
                                                                unsigned short *__restrict__ dY1, int B, int Hm, int Wm, int TXn, int TYn) {
    __shared__ __attribute__((aligned(16))) unsigned char pl[PLANES_B];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n16 = lane & 15, g = lane >> 4;
    const int Hh = 2 * Hm, Wh = 2 * Wm;
    bf16x8 ka[2][4];                                              // the composed weights: resident for the whole kernel
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            ka[kc][mt] = *reinterpret_cast<const bf16x8 *>(KdP + ((kc * 4 + mt) * 64 + lane) * 8);
    const int ntiles = B * TYn * TXn;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / (TYn * TXn), trem = t - b * TYn * TXn, ty = trem / TXn, tx = trem - ty * TXn;
        const int y0 = ty * TH, x0 = tx * TW;
        __syncthreads();                                          // the previous tile's planes have been read
        fill_planes(pl, dO, b, y0, x0, Hh, Wh, tid);
        __syncthreads();
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int ly = 2 * wave + (gi >> 1), n = (gi & 1) * 16 + n16;
            f32x4 acc[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const int pr = 2 * ly + 4 * kc + g;
                const int oo = win_off(1, pr, n), oe = win_off(0, pr, n + 1);
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    const uint2 vo = *reinterpret_cast<const uint2 *>(__builtin_assume_aligned(pl + hl * PHL_B + oo, 8));    // ds_read_b64
                    const uint2 ve = *reinterpret_cast<const uint2 *>(__builtin_assume_aligned(pl + hl * PHL_B + oe, 8));
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, make_uint4(vo.x, vo.y, ve.x, ve.y));
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[kc][mt], bf, acc[mt], 0, 0, 0);
                }
            }
            const int qy = y0 + ly, qx = x0 + n;
            if (qy < Hm && qx < Wm) {
                unsigned short *o = dY1 + (((long)b * Hm + qy) * Wm + qx) * 64 + 8 * g;
                *reinterpret_cast<uint4 *>(o) = make_uint4(pk2(acc[0][0], acc[0][1]), pk2(acc[0][2], acc[0][3]),
                                                           pk2(acc[1][0], acc[1][1]), pk2(acc[1][2], acc[1][3]));
                *reinterpret_cast<uint4 *>(o + 32) = make_uint4(pk2(acc[2][0], acc[2][1]), pk2(acc[2][2], acc[2][3]),
                                                                pk2(acc[3][0], acc[3][1]), pk2(acc[3][2], acc[3][3]));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// xcorr: X[off][c] and SX[off] per workgroup from Y1 (bf16) and d_o
// ---------------------------------------------------------------------------------------------------------------------
constexpr int YT_B = TH * TW * 128;                               // the Y1 tile: 256 pixels x 64 bf16, 32-byte chunks swizzled
constexpr int XSLAB = 64 * 64 + 64;                               // uptail_xcorr_kernel's slab for r = 2: X[off][c], then SX[off]

__global__ void __launch_bounds__(256) uptail_xcorr_bf16_kernel(const unsigned short *__restrict__ y1, const float *__restrict__ dO,
                                                                float *__restrict__ Xs, int B, int Hm, int Wm, int TXn, int TYn) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[PLANES_B + YT_B];
    unsigned char *pl = sm, *yt = sm + PLANES_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n16 = lane & 15, g = lane >> 4;
    const int Hh = 2 * Hm, Wh = 2 * Wm;
    f32x4 acc[4][5];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A operand (rows = offsets): row R of tile mt is offset 16 mt + R = (oy, ox) = (2 mt + (R >> 3), R & 7); its 8 pixels
    // 8 g .. 8 g + 7 of a tile row start at element i0 = 8 g + ((ox + 1) >> 1) of parity plane (ox + 1) & 1
    const int ox = n16 & 7, oyl = n16 >> 3;
    const int a_i0 = 8 * g + ((ox + 1) >> 1), a_par = (ox + 1) & 1;
    const int a_s = (-a_i0) & 3;
    const int a_base = a_par * PPAR_B + a_s * PCOPY_B + 2 * (a_i0 + a_s);                    // + row * PROW_B
    // B operand (columns = channels), transposed read: lane 4 q + p of a 16-lane group supplies block row q (a pixel), 8 bytes
    // p of the 32-byte channel chunk; chunk nt of pixel r of a tile row lives at 32 (nt ^ sw(r)), sw(r) = (r >> 1 & 1) | (r >> 3 & 1) << 1
    const int tq = n16 >> 2, tp = n16 & 3;
    int b_off[2], b_sw[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = 8 * g + 4 * h + tq;
        b_off[h] = r * 128 + 8 * tp;
        b_sw[h] = ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
    }
    const int ntiles = B * TYn * TXn;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / (TYn * TXn), trem = t - b * TYn * TXn, ty = trem / TXn, tx = trem - ty * TXn;
        const int y0 = ty * TH, x0 = tx * TW;
        __syncthreads();
        fill_planes(pl, dO, b, y0, x0, Hh, Wh, tid);
        for (int e = tid; e < TH * TW * 8; e += 256) {            // Y1 tile: 16-byte pieces, zero outside the image
            const int pc = e & 7, p = e >> 3, ly = p >> 5, r = p & 31;
            const int qy = y0 + ly, qx = x0 + r;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (qy < Hm && qx < Wm) v = *reinterpret_cast<const uint4 *>(y1 + (((long)b * Hm + qy) * Wm + qx) * 64 + pc * 8);
            const int sw = ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
            *reinterpret_cast<uint4 *>(yt + p * 128 + 32 * ((pc >> 1) ^ sw) + 16 * (pc & 1)) = v;
        }
        __syncthreads();
        // ones column: 1.0 for the lane's pixels that lie inside the image (SX counts those only)
        unsigned ones[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int qx = x0 + 8 * g + 2 * q;
            ones[q] = (qx < Wm ? 0x3f80u : 0u) | (qx + 1 < Wm ? 0x3f800000u : 0u);
        }
        const bf16x8 onesf = __builtin_bit_cast(bf16x8, make_uint4(ones[0], ones[1], ones[2], ones[3]));
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int ly = 2 * wave + rr;
            if (y0 + ly >= Hm) break;                             // (wave-uniform: the transposed reads below need all lanes)
            bf16x8 bfr[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(yt + ly * 32 * 128 + b_off[0] + 32 * (nt ^ b_sw[0])));
                s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(yt + ly * 32 * 128 + b_off[1] + 32 * (nt ^ b_sw[1])));
                bfr[nt] = __builtin_bit_cast(bf16x8, make_uint4(__builtin_bit_cast(uint2, lo4).x, __builtin_bit_cast(uint2, lo4).y,
                                                                 __builtin_bit_cast(uint2, hi4).x, __builtin_bit_cast(uint2, hi4).y));
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int pr = 2 * ly + 2 * mt + oyl;
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    const unsigned char *ap = reinterpret_cast<const unsigned char *>(__builtin_assume_aligned(pl + hl * PHL_B + a_base + pr * PROW_B, 8));
                    const uint2 v0 = *reinterpret_cast<const uint2 *>(ap), v1 = *reinterpret_cast<const uint2 *>(ap + 8);
                    const bf16x8 af = __builtin_bit_cast(bf16x8, make_uint4(v0.x, v0.y, v1.x, v1.y));
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, onesf, acc[mt][4], 0, 0, 0);
                }
            }
        }
    }
    // cross-wave sum in LDS in wave order (fixed), then the block's slab: X[off][c] (off = 16 mt + 4 g + reg, c = 16 nt + n16), SX[off]
    float *red = reinterpret_cast<float *>(sm);
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int off = 16 * mt + 4 * g + v;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        float *p = red + off * 64 + 16 * nt + n16;
                        *p = (w == 0 ? 0.f : *p) + acc[mt][nt][v];
                    }
                    if (n16 == 0) {
                        float *p = red + 64 * 64 + off;
                        *p = (w == 0 ? 0.f : *p) + acc[mt][4][v];
                    }
                }
        }
    }
    __syncthreads();
    float *slab = Xs + (long)blockIdx.x * XSLAB;
    for (int e = tid; e < XSLAB; e += 256) slab[e] = red[e];
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: out (fp32) from Y1 (bf16)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int FT = 16, FH = FT + 4;                               // 16 x 16 output pixels per tile, 20 x 20 halo
constexpr int FPITCH = 160;                                       // bytes per halo pixel / per weight row (128 data + 32 pad: fragment reads conflict-free)
constexpr int FHALO_B = FH * FH * FPITCH, FKF_B = 100 * FPITCH, FZERO_B = 64;

__global__ void __launch_bounds__(256, 2) uptail_fwd_bf16_kernel(const unsigned short *__restrict__ y1, const unsigned short *__restrict__ KfT,
                                                                 const float *__restrict__ bsum, const float *__restrict__ b3,
                                                                 float *__restrict__ out, int B, int Hm, int Wm, int TXn, int TYn) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[FHALO_B + FKF_B + FZERO_B];
    unsigned char *halo = sm, *kf = sm + FHALO_B, *zero = kf + FKF_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n16 = lane & 15, g = lane >> 4;
    for (int e = tid; e < 100 * 8; e += 256)                     // the composed weights, once per workgroup
        *reinterpret_cast<uint4 *>(kf + (e >> 3) * FPITCH + (e & 7) * 16) = *reinterpret_cast<const uint4 *>(KfT + (e >> 3) * 64 + (e & 7) * 8);
    if (tid < FZERO_B / 16) *reinterpret_cast<uint4 *>(zero + tid * 16) = make_uint4(0, 0, 0, 0);
    const int dy = n16 >> 2, ij = n16 & 3;                        // A row m = (dy, ij)
    const int a_lane = ((-dy * 5) * 4 + ij) * FPITCH + g * 16;    // + ((u'y * 5 + ux) * 4) * FPITCH + kc * 64
    const int b_lane = (4 * wave * FH + n16) * FPITCH + g * 16;   // + (u'y * FH + ux) * FPITCH + kc * 64
    const float bias[4] = {bsum[0] + b3[0], bsum[1] + b3[0], bsum[2] + b3[0], bsum[3] + b3[0]};
    const int Hh = 2 * Hm, Wh = 2 * Wm;
    const int ntiles = B * TYn * TXn;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / (TYn * TXn), trem = t - b * TYn * TXn, ty = trem / TXn, tx = trem - ty * TXn;
        const int y0 = ty * FT, x0 = tx * FT;
        __syncthreads();
        for (int e = tid; e < FH * FH * 8; e += 256) {
            const int pc = e & 7, p = e >> 3, hy = p / FH, hx = p - hy * FH;
            const int gy = y0 - 2 + hy, gx = x0 - 2 + hx;
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((unsigned)gy < (unsigned)Hm && (unsigned)gx < (unsigned)Wm)
                v = bf2h8(*reinterpret_cast<const uint4 *>(y1 + (((long)b * Hm + gy) * Wm + gx) * 64 + pc * 8));
            *reinterpret_cast<uint4 *>(halo + p * FPITCH + pc * 16) = v;
        }
        __syncthreads();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int uy = 0; uy < 8; ++uy) {                          // u'y = uy of the kernel + dy of the row
            const bool ok = uy >= dy && uy <= dy + 4;
#pragma unroll
            for (int ux = 0; ux < 5; ++ux)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const unsigned char *ap = ok ? kf + a_lane + ((uy * 5 + ux) * 4) * FPITCH + kc * 64 : zero;
                    const f16x8 af = *reinterpret_cast<const f16x8 *>(ap);
                    const f16x8 bf = *reinterpret_cast<const f16x8 *>(halo + b_lane + (uy * FH + ux) * FPITCH + kc * 64);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc, 0, 0, 0);
                }
        }
        // lane (column n16 = pixel, g = dy): the four sub-positions of output pixel (y0 + 4 wave + g, x0 + n16)
        const int qy = y0 + 4 * wave + g, qx = x0 + n16;
        if (qy < Hm && qx < Wm) {
            float *o = out + ((long)b * Hh + 2 * qy) * Wh + 2 * qx;
            *reinterpret_cast<float2 *>(o) = make_float2(acc[0] + bias[0], acc[1] + bias[1]);
            *reinterpret_cast<float2 *>(o + Wh) = make_float2(acc[2] + bias[2], acc[3] + bias[3]);
        }
    }
}

inline int grid1d(long n) { return (int)((n + 255) / 256); }

}  // namespace

extern "C" int rnh_uptail_bf16_supported(int C1, int r, int Co) { return C1 == 64 && r == 2 && Co == 1; }

extern "C" int64_t rnh_uptail_fwd_bf16_ws_floats(int C1, int Cq, int r, int Co) {
    return ((rnh_uptail_fwd_ws_floats(C1, Cq, r, Co) + 3) & ~(int64_t)3) + 25 * 4 * 64 / 2 + 64;
}

extern "C" int rnh_uptail_fwd_bf16(const void *y1, const float *w2, const float *b2, const float *w3, const float *b3, float *out, float *ws,
                                   int B, int Hm, int Wm, int C1, int Cq, int r, int Co, void *stream) {
    if (!y1 || !w2 || !b2 || !w3 || !b3 || !out || !ws || B < 1 || Hm < 1 || Wm < 1 || Cq < 1) RNH_FAIL(RNH_E_ARG, "rnh_uptail_fwd_bf16: bad arguments");
    if (!rnh_uptail_bf16_supported(C1, r, Co)) RNH_FAIL(RNH_E_RANGE, "rnh_uptail_fwd_bf16: built for r = 2, C1 = 64, out_channels = 1");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = rnh_uptail_fwd_compose_(w2, b2, w3, ws, C1, Cq, r, Co, st)) return rc;
    const int r2 = r * r;
    const float *Kf = ws + (long)9 * r2 * 9 * C1 + r2 * 9, *bsum = Kf + (long)25 * 64 * 4;
    unsigned short *KfT = reinterpret_cast<unsigned short *>(ws + ((rnh_uptail_fwd_ws_floats(C1, Cq, r, Co) + 3) & ~(int64_t)3));
    hipLaunchKernelGGL(pack_kft_kernel, dim3(grid1d(25 * 4 * 64)), dim3(256), 0, st, Kf, KfT);
    RNH_CHECK_LAUNCH("rnh_uptail_fwd_bf16(pack)");
    const int TXn = (Wm + FT - 1) / FT, TYn = (Hm + FT - 1) / FT;
    const long ntiles = (long)B * TXn * TYn;
    if (ntiles >= (1L << 31)) RNH_FAIL(RNH_E_RANGE, "rnh_uptail_fwd_bf16: too many tiles");
    const int grid = (int)(ntiles < 512 ? ntiles : 512);          // two workgroups of 80 KB per CU, each walks its share of the tiles
    hipLaunchKernelGGL(uptail_fwd_bf16_kernel, dim3(grid), dim3(256), 0, st, (const unsigned short *)y1, KfT, bsum, b3, out, B, Hm, Wm, TXn, TYn);
    RNH_CHECK_LAUNCH("rnh_uptail_fwd_bf16");
    return rnh_uptail_fwd_border_bf16_(y1, ws, out, B, Hm, Wm, C1, r, Co, st);
}

extern "C" int64_t rnh_uptail_dgrad_bf16_ws_floats(void) { return 8 * 64 * 8 / 2 + 64; }

extern "C" int rnh_uptail_dgrad_bf16(const float *d_o, const float *G, void *dy1, float *ws, int B, int Hm, int Wm, int C1, int Co, int r,
                                     void *stream) {
    if (!d_o || !G || !dy1 || !ws || B < 1 || Hm < 1 || Wm < 1) RNH_FAIL(RNH_E_ARG, "rnh_uptail_dgrad_bf16: bad arguments");
    if (!rnh_uptail_bf16_supported(C1, r, Co)) RNH_FAIL(RNH_E_RANGE, "rnh_uptail_dgrad_bf16: built for r = 2, C1 = 64, out_channels = 1");
    hipStream_t st = (hipStream_t)stream;
    const float *Kd = G + (long)9 * (r + 2) * (r + 2) * C1;      // behind G (rnh_uptail_compose)
    unsigned short *KdP = reinterpret_cast<unsigned short *>(ws);
    hipLaunchKernelGGL(pack_kd_kernel, dim3(grid1d(8 * 64 * 8)), dim3(256), 0, st, Kd, KdP);
    RNH_CHECK_LAUNCH("rnh_uptail_dgrad_bf16(pack)");
    const int TXn = (Wm + TW - 1) / TW, TYn = (Hm + TH - 1) / TH;
    const long ntiles = (long)B * TXn * TYn;
    if (ntiles >= (1L << 31)) RNH_FAIL(RNH_E_RANGE, "rnh_uptail_dgrad_bf16: too many tiles");
    const int grid = (int)(ntiles < 1024 ? ntiles : 1024);
    hipLaunchKernelGGL(uptail_dgrad_bf16_kernel, dim3(grid), dim3(256), 0, st, d_o, KdP, (unsigned short *)dy1, B, Hm, Wm, TXn, TYn);
    RNH_CHECK_LAUNCH("rnh_uptail_dgrad_bf16");
    return rnh_uptail_dgrad_border_bf16_(d_o, G, dy1, B, Hm, Wm, C1, r, st);
}

extern "C" int rnh_uptail_xcorr_bf16(const void *y1, const float *d_o, float *M, float *S, float *ws, int B, int Hm, int Wm, int C1, int r,
                                     void *stream) {
    if (!y1 || !d_o || !M || !S || !ws || B < 1 || Hm < 1 || Wm < 1) RNH_FAIL(RNH_E_ARG, "rnh_uptail_xcorr_bf16: bad arguments");
    if (!rnh_uptail_bf16_supported(C1, r, 1)) RNH_FAIL(RNH_E_RANGE, "rnh_uptail_xcorr_bf16: built for r = 2, C1 = 64");
    int TX, TY, nblk, nchunk, NT;
    rnh_uptail_xcorr_shape_(B, Hm, Wm, r, &TX, &TY, &nblk, &nchunk, &NT);     // tiles of 8 x 32, nblk = min(tiles, 512) slabs: ws as for rnh_uptail_xcorr
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(uptail_xcorr_bf16_kernel, dim3(nblk), dim3(256), 0, st, (const unsigned short *)y1, d_o, ws, B, Hm, Wm, TX, TY);
    RNH_CHECK_LAUNCH("rnh_uptail_xcorr_bf16");
    return rnh_uptail_xcorr_finish_bf16_(y1, d_o, M, S, ws, B, Hm, Wm, C1, r, st);
}
