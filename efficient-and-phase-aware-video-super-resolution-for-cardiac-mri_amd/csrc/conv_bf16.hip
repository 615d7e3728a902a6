// 3x3 (padding 1) / 1x1 convolution as an implicit GEMM on bf16 MFMA for gfx950 - rnh_conv_bf16: the bf16-storage
// form of the call sites of rnh_conv_igemm (reference src/model/nets/refine_net.py:149-154, :199-205, :235-265 and their
// data gradients; the reference itself is fp32 throughout - BASELINE.json configs[2] asks for this path).
//
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32 MFMA, so this kernel is not MFMA-bound but bound by how fast
// operands reach the matrix cores and by everything that is NOT an MFMA; the design therefore moves every input byte as
// few times as possible and keeps two workgroups on every CU so that one's non-MFMA phases run under the other's MFMAs:
//
//   * one workgroup (4 waves) = 8 x 32 output pixels of one image x NCOLS = 128 (64) output columns;
//     wave = (pixel half: 4 rows of 32 pixels) x (column half): 4 x NB accumulator tiles of 32 x 32;
//   * A operand: per 16-channel chunk of the K dimension the 10 x 34 pixel HALO of the tile is staged ONCE in LDS (bf16;
//     fp32 sources are converted on the way with v_cvt_pk_bf16_f32) and serves all 9 taps - the fragment of tap (dy, dx)
//     is the same LDS image read at a shifted address.  48-byte pitch per pixel (32 B of data + 16 B pad): a 16-byte
//     fragment read of 32 consecutive pixels touches every bank exactly once per 16-lane group.  Two halo buffers, one
//     barrier per chunk; the staging loads are raw buffer loads without a branch (out-of-range offset = zero padding);
//   * B operand: straight from L2 into registers, two taps ahead, in a ring of three fragment sets (see the kernel);
//   * epilogue through LDS: the fp32 accumulators (+ bias) are parked as a [pixel][column] tile (128 pixels at a time for
//     128-column tiles), then all 256 threads finish 8 columns of a pixel per step with whole-row 16-byte global accesses -
//     STORE (segments, optional accumulate, fp32 or bf16), PS (PixelShuffle fused into the store) or LSTM (the four gates
//     of 8 hidden channels meet in one thread: sigmoid / tanh, c' = f c + i g, h' = o tanh c'; c stays fp32).
//
// MFMA operand maps (cdna_hip_programming.md section 3): lane l, r = l & 31, h = l >> 5 holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r], j = 0..7; C/D: col = l & 31, row = (v & 3) + 8 (v >> 2) + 4 h.  Rows = the 32 pixels of one
// image-row segment, k = channel inside the chunk.
#include "rnh_common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int TW = 32, HPW = TW + 2;                                         // tile width; halo width
constexpr int PITCH = 48;                                                    // bytes per halo pixel / weight column

__device__ __forceinline__ unsigned pk2(float a, float b) {
    const bf16x2 r = {(__bf16)a, (__bf16)b};                                 // v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ uint4 pack8(const float4 a, const float4 b) {
    return make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ void unpack8(const uint4 u, float *f) {
    f[0] = bf_lo(u.x); f[1] = bf_hi(u.x); f[2] = bf_lo(u.y); f[3] = bf_hi(u.y);
    f[4] = bf_lo(u.z); f[5] = bf_hi(u.z); f[6] = bf_lo(u.w); f[7] = bf_hi(u.w);
}
// 8 consecutive elements of a tensor of type dt at element index e -> fp32
__device__ __forceinline__ void load8(const void *p, int dt, long e, float *f) {
    if (dt == RNH_DT_BF16) {
        unpack8(*reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned short *>(p) + e), f);
    } else {
        const float4 a = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + e);
        const float4 b = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + e + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
}
__device__ __forceinline__ void store8(void *p, int dt, long e, const float *f) {
    if (dt == RNH_DT_BF16) {
        *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned short *>(p) + e) =
            make_uint4(pk2(f[0], f[1]), pk2(f[2], f[3]), pk2(f[4], f[5]), pk2(f[6], f[7]));
    } else {
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p) + e) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p) + e + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
}

// the activations of conv_wino.hip (v_exp_f32 / v_rcp_f32, 1 ulp each; both forms of tanh computed and selected)
__device__ __forceinline__ float b_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float b_tanh(float x) {
    const float ax = fabsf(x);
    const float big = __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * ax)), 1.f);
    const float small = ax * __builtin_fmaf(-0.33333334f * ax, ax, 1.f);
    return copysignf(ax < 0.04f ? small : big, x);
}

// tanh as 2 sigmoid(2x) - 1: one exp, one rcp, two FMAs; absolute error <= 2 ulp of 1 (as the candidate gate of
// conv_wino.hip) - in the bf16-storage path h' = o tanh(c') is rounded to 8 bits anyway
__device__ __forceinline__ float b_tanh_fast(float x) { return __builtin_fmaf(2.f, __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)), -1.f); }

// the lane halves trade: afterwards lanes 0..31 hold (their x, the x of lane + 32), lanes 32..63 (the y of lane - 32, their y)
__device__ __forceinline__ void swap_halves(unsigned &x, unsigned &y) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);      // v_permlane32_swap_b32: x[32..63] <-> y[0..31]
    x = r[0];
    y = r[1];
}
__device__ __forceinline__ void swap_halves(float &x, float &y) {
    unsigned a = __builtin_bit_cast(unsigned, x), b = __builtin_bit_cast(unsigned, y);
    swap_halves(a, b);
    x = __builtin_bit_cast(float, a);
    y = __builtin_bit_cast(float, b);
}

// wp[ks][n][kk] = W[o][i][tap] as bf16, index conventions of rnh_pack_weights (conv_igemm.hip), kk in natural order
__global__ void pack_bf16_kernel(const float *w, const float *bias, unsigned short *wp, float *biasp, const int *kbase, const int *knv,
                                 const int *ktap, const int *kcoff, const int *colmap, int nk, int Npad, int Cout, int Cin, int ntaps,
                                 int kstride, int transposed) {
    const long total = (long)nk * Npad * 16;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total + Npad; e += (long)gridDim.x * blockDim.x) {
        if (e >= total) {
            const int n = (int)(e - total);
            if (biasp) biasp[n] = (bias && !transposed && colmap[n] >= 0) ? bias[colmap[n]] : 0.f;
            continue;
        }
        const int kk = (int)(e & 15), n = (int)((e >> 4) % Npad), ks = (int)(e / (16 * (long)Npad));
        const int col = colmap[n];
        float v = 0.f;
        if (col >= 0 && kk < knv[ks]) {
            const int k = kbase[ks] + kk * kstride, c = col + (kcoff ? kcoff[ks] : 0), t = ktap[ks];
            v = transposed ? w[((long)k * Cin + c) * ntaps + (ntaps - 1 - t)] : w[((long)c * Cin + k) * ntaps + t];
        }
        const __bf16 b = (__bf16)v;
        wp[e] = __builtin_bit_cast(unsigned short, b);
    }
}

// wave-uniform raw-buffer descriptor: base + 2 GiB window; an offset of 0xFFFFFFFF is out of range and reads as 0 -
// zero padding and absent channels cost no branch (the convention of conv_igemm.hip / conv_wino.hip)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bdesc(const void *p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(u & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint4 bld16(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

// Diagnostic ablations (tools/bf16_ablate.sh; never defined in the product build): -DRNH_EXP=<mask> removes one cost of the main
// loop at a time - results are WRONG, only the launch time is of interest.  1: weight fragments loaded once, 2: halo fragments read
// once per chunk, 4: no halo staging after the prologue, 8: no barrier in the loop, 16: epilogue skipped; 32 / 64 / 128 keep the
// results: column-tile-major block order, second workgroup of a CU delayed by ~6 / ~12 us
#ifndef RNH_EXP
#define RNH_EXP 0
#endif

#ifdef RNH_STAMPS
__device__ unsigned long long g_bf16_stamps[64];
#define BSTAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_bf16_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define BSTAMP(i)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// The kernel.  ("Variant D" of this round's experiments; variant L staged the weights of a chunk through LDS beside the
// halo - 143 KB, one workgroup per CU, 125 us for the ConvLSTM cell; as 4 x 32 pixel tiles with two workgroups per CU 115 us;
// this one 95 us: profiles/README.md, r02_c.)  The B operand never touches LDS.  The packed weights [chunk*tap][Npad][16] are laid out
// such that the fragment of (chunk, tap, 32-column block) is ONE contiguous kilobyte (lane l reads bytes 32 (l & 31) +
// 16 (l >> 5) .. +16 of it), the whole weight set (<= 1.3 MB) lives in L2, and a wave needs each fragment exactly once: so
// every wave streams its fragments with raw buffer loads straight into registers, two taps ahead of their MFMAs, in a
// ring of three register sets that runs across chunk boundaries and knows nothing of the barriers.  LDS then holds only
// the halo (16 KB per buffer, double-buffered), the staging phase per chunk shrinks from 12 to 3 16-byte LDS writes per
// thread, and the epilogue parks 128 pixels at a time for 128-column tiles (two rounds): 68 KB of LDS per workgroup, TWO
// workgroups of 8 x 32 pixels per CU - one's prologue, barriers and epilogue run under the other's MFMAs.
// KC = channels per chunk (= per barrier).  16: any mix of fp32 / bf16 sources, channel counts in multiples of 8.  32 (round 3): bf16
// sources of 32-channel multiples only (the ConvLSTM cell, its data gradient, the PixelShuffle convolutions): half the barriers and
// halo-staging events per MFMA, 64 instead of 32 bytes of every 128-byte line per staging load, no fp32 staging registers - which
// pays for a weight-fragment ring of SIX sets: the fragments are requested five steps (40 MFMAs) ahead instead of two, so that a
// wait for them no longer sits out the younger-than-them halo loads from HBM (vector-memory loads return in order)
template <int NCOLS, int KC = 16>
struct GeoD {
    static constexpr int TH = 8, MB = 4, NB = NCOLS / 64;
    static constexpr int HPH = TH + 2, HP = HPW * HPH;
    static constexpr int KS = KC / 16;                          // MFMA k steps per tap
    static constexpr int APITCH = KC == 32 ? 80 : PITCH;        // bytes per halo pixel: data + 16 B pad (fragment reads conflict-free for both)
    static constexpr int PPP = 2 * KS;                          // 16-byte pieces per halo pixel
    static constexpr int A_BYTES = HP * APITCH, A_PIECES = PPP * HP, A_ITERS = (A_PIECES + 255) / 256;
    static constexpr int SMEM = 2 * A_BYTES;                    // the two halo buffers; the epilogue stays in registers
    static_assert(2 * SMEM <= 160 * 1024, "two workgroups per CU");
};

template <int EPI, int NCOLS, int NTAPS, int KC = 16>
__global__ void __launch_bounds__(256, 2) conv_bf16d_kernel(const rnh_conv_bf16_args_t P, const int TYn, const int TXn, const int NT) {
    using G = GeoD<NCOLS, KC>;
    constexpr int TH = G::TH, NB = G::NB, MB = G::MB, A_BYTES = G::A_BYTES, A_PIECES = G::A_PIECES, A_ITERS = G::A_ITERS;
    constexpr int KS = G::KS, APITCH = G::APITCH, PPP = G::PPP;
    static_assert(NTAPS % 3 == 0 || NTAPS == 1, "the fragment ring has three sets");
    static_assert(KC == 16 || (KC == 32 && NTAPS == 9), "32-channel chunks serve the 3x3 kernels");
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::SMEM];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5, wave = tid >> 6;
    const int ph = wave & 1, chalf = wave >> 1;
    const int bid = rnh_xcd_remap(blockIdx.x, P.B * TYn * TXn * NT);
    // (RNH_EXP & 32, experiment: column-tile-major block order - the workgroups that run together stream the SAME weight fragments)
    const int nt = (RNH_EXP & 32) ? bid / (P.B * TYn * TXn) : bid % NT, mt = (RNH_EXP & 32) ? bid % (P.B * TYn * TXn) : bid / NT;
    if ((RNH_EXP & (64 | 128)) && blockIdx.x >= 256 && blockIdx.x < 512) {      // experiment: the second workgroup of every CU starts late
        for (int i = 0; i < ((RNH_EXP & 64) ? 2 : 0) + ((RNH_EXP & 128) ? 4 : 0); ++i) __builtin_amdgcn_s_sleep(100);
    }
    const int img = mt / (TYn * TXn), trem = mt - img * (TYn * TXn), ty = trem / TXn, tx = trem - ty * TXn;
    const int y0 = ty * TH, x0 = tx * TW;
    const int H = P.H, W = P.W;
    const int sc = P.src[0].scale, Hs = H * sc, Ws = W * sc;

    int apix[A_ITERS], alds[A_ITERS], ahalf[A_ITERS];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int p = tid + 256 * i, px = p / PPP, hr = px / HPW, hc = px - hr * HPW;
        const int y = y0 - 1 + hr, x = x0 - 1 + hc;
        const bool in = p < A_PIECES && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        ahalf[i] = p % PPP;
        alds[i] = p < A_PIECES ? px * APITCH + (p % PPP) * 16 : -1;
        apix[i] = in ? (y * sc) * Ws + x * sc : -1;
    }

    uint4 ra[A_ITERS];
    [[maybe_unused]] uint4 rh[KC == 16 ? A_ITERS : 1];           // second halves of fp32 pieces (16-channel chunks only)
    int ra_f32 = 0;
    int si = 0, cc = 0;
    auto load_chunk = [&]() {
        const rnh_msrc_t &S = P.src[si];
        const int es = (KC == 32 || S.dtype == RNH_DT_BF16) ? 2 : 4;
        const char *base = reinterpret_cast<const char *>(S.ptr) +
                           ((((long)(img + S.img_off) * Hs + S.sub_y) * Ws + S.sub_x) * S.C + S.c0 + cc * KC) * es;
        const __amdgpu_buffer_rsrc_t rs = bdesc(base);
        const int pstride = S.C * es, left = S.nch - cc * KC;
        if constexpr (KC == 16) ra_f32 = S.dtype != RNH_DT_BF16;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int ch = ahalf[i] * 8;
            const bool ok = apix[i] >= 0 && ch < left;
            const int off = apix[i] * pstride + ch * es;
            ra[i] = bld16(rs, ok ? off : -1);
            if constexpr (KC == 16) rh[i] = bld16(rs, ok && ra_f32 && ch + 4 < left ? off + 16 : -1);
        }
        if (++cc * KC >= S.nch) {
            cc = 0;
            ++si;
        }
    };
    auto store_chunk = [&](int buf) {
        unsigned char *Ab = smem + buf * A_BYTES;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            uint4 v = ra[i];
            if constexpr (KC == 16) v = ra_f32 ? pack8(__builtin_bit_cast(float4, ra[i]), __builtin_bit_cast(float4, rh[i])) : ra[i];
            if (alds[i] >= 0) *reinterpret_cast<uint4 *>(Ab + alds[i]) = v;
        }
    };

    // weight fragments: descriptor over the packed weights, per-lane offset inside a (chunk, tap) slab, slab stride
    const __amdgpu_buffer_rsrc_t wrs = bdesc(P.wp);
    // the wave's 32-column blocks of the tile: block(n) = BLK0 + BSTR * n.  Column halves (blocks NB chalf + n) everywhere but in the
    // LSTM-backward epilogue, whose tile is [input gradient | dh_rec]: there the waves take the blocks in turn (2 n + chalf), so that
    // every wave finishes one block of each kind - the gate backward (11 loads and 5 stores per item) is spread over all four waves
    constexpr int BSTR = (EPI == RNH_EPI_LSTM_BWD && NB == 2) ? 2 : 1;
    const int BLK0 = (EPI == RNH_EPI_LSTM_BWD && NB == 2) ? chalf : NB * chalf;
    const int wlane = ((nt * NCOLS + BLK0 * 32 + l31) * 16 + kh * 8) * 2;
    const int slab = P.Npad * 32;                               // bytes of one (chunk, tap) slab
    const int nslabs = P.nchunks * NTAPS;
    constexpr int RING = NTAPS == 1 ? 1 : (KC == 32 ? 6 : 3), AHEAD = RING - 1;
    constexpr int NSTEP = NTAPS * KS;                            // (tap, k step) pairs per chunk; NSTEP % RING == 0: the set of a step is static
    static_assert(NTAPS == 1 || NSTEP % RING == 0, "the fragment ring must divide the steps of a chunk");
    uint4 bq[RING][NB];
    // fragments of step g = chunk * NSTEP + tap * KS + ks of the launch: slab (16-channel chunk KS * chunk + ks, tap); beyond the end: zeros, unused
    auto bload = [&](int g, int set) {
        const int c = g / NSTEP, r = g - c * NSTEP, tap = r / KS, ks = r - tap * KS;
        const bool ok = g < nslabs;
        const int base = ((KS * c + ks) * NTAPS + tap) * slab;
#pragma unroll
        for (int n = 0; n < NB; ++n) bq[set][n] = bld16(wrs, ok ? base + wlane + n * BSTR * 32 * 32 : -1);
    };

    // accumulators of the transposed product (see the epilogue): register v of block n = column (v & 3) + 8 (v >> 2) + 4 kh; they start
    // from the bias of their column
    f32x16 acc[MB][NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bv = P.bias ? *reinterpret_cast<const float4 *>(P.bias + nt * NCOLS + (BLK0 + BSTR * n) * 32 + 8 * q + 4 * kh)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                acc[m][n][4 * q] = bv.x;
                acc[m][n][4 * q + 1] = bv.y;
                acc[m][n][4 * q + 2] = bv.z;
                acc[m][n][4 * q + 3] = bv.w;
            }
        }
    const int nch = P.nchunks / KS;

    // One chunk.  The halo of chunk c + 1 sits in registers since step 2 KS of chunk c - 1 (a whole chunk of MFMAs ago, so the
    // wait in store_chunk costs nothing); at step 2 KS it goes to the other LDS buffer - nobody reads that one before the
    // barrier at the end of this chunk - and the loads of chunk c + 2 are issued into the same registers.
    auto compute = [&](int buf, int c) {
        const unsigned char *Ab = smem + buf * A_BYTES + (MB * ph * HPW + l31) * APITCH + kh * 16;
        bf16x8 a[2][MB];
        auto afrags = [&](int step, int set) {
            const int tap = step / KS, ks = step - tap * KS;
            const int dy = NTAPS == 9 ? tap / 3 : 1, dx = NTAPS == 9 ? tap % 3 : 1;
#pragma unroll
            for (int m = 0; m < MB; ++m) a[set][m] = *reinterpret_cast<const bf16x8 *>(Ab + ((m + dy) * HPW + dx) * APITCH + ks * 32);
        };
        afrags(0, 0);
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            if (step + 1 < NSTEP && !(RNH_EXP & 2)) afrags(step + 1, (step + 1) & 1);
            if constexpr (NTAPS == 9 && !(RNH_EXP & 1)) bload(c * NSTEP + step + AHEAD, (step + AHEAD) % RING);      // AHEAD steps ahead
            if (step == (NTAPS == 9 ? 2 * KS : 0) && !(RNH_EXP & 4)) {
                if (c + 1 < nch) store_chunk(buf ^ 1);
                if (c + 2 < nch) load_chunk();
            }
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bq[NTAPS == 9 ? step % RING : 0][n]),
                                                                        a[(RNH_EXP & 2) ? 0 : (step & 1)][m], acc[m][n], 0, 0, 0);
            if constexpr (NTAPS == 1) bload(c + 1, 0);
            if constexpr (NTAPS == 9) __builtin_amdgcn_sched_barrier(0);     // pin the issue order: hipcc otherwise sinks the weight loads to their first use
        }
    };

    // ---- K loop: double-buffered halo, one barrier per chunk ------------------------------------------------------------
    BSTAMP(0);
    bload(0, 0);
    if constexpr (NTAPS == 9) {
#pragma unroll
        for (int g = 1; g < AHEAD; ++g) bload(g, g);
    }
    load_chunk();
    store_chunk(0);
    if (nch > 1) load_chunk();
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        BSTAMP(8 + 3 * (c & 15));
        compute(c & 1, c);
        BSTAMP(9 + 3 * (c & 15));
        if (!(RNH_EXP & 8) && c + 1 < nch) __syncthreads();     // (behind the last chunk nobody writes LDS any more)
    }
    BSTAMP(1);
    if (RNH_EXP & 16) {                                         // keep the accumulators alive, skip the epilogue
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int v = 0; v < 16; ++v) s += acc[m][n][v];
        if (s == 1.2345e-30f) reinterpret_cast<float *>(smem)[tid] = s;
        return;
    }

    // ---- epilogue, in registers (round 4; until then the accumulators were parked in LDS as a [pixel][column] tile, two barriers
    // and 128 four-byte LDS stores per lane: 9.5 k cycles per workgroup for a plain store, 15 k with the gate math).  The MFMAs
    // computed the TRANSPOSED product - weights as the A operand, pixels as B - so lane l holds pixel x0 + (l & 31) of each of its
    // MB image rows and, per 32-column block, the columns (v & 3) + 8 (v >> 2) + 4 kh: FOUR consecutive columns per register quad,
    // and after one v_permlane32_swap_b32 per register (the lane halves trade quads) EIGHT consecutive ones.  Every global
    // access is a 16-byte piece of one pixel, nothing goes through LDS, no barrier follows the main loop: a wave enters its
    // epilogue behind its own last MFMA, under the other waves' MFMAs.
    const int xg = x0 + l31;
    const bool xin = xg < W;
    if constexpr (EPI == RNH_EPI_LSTM) {
        // columns (plans.lstm_colmap8): block bb = 2 chalf + n of the tile, register quad q = gate, so v = 4 gate + e is gate `gate` of
        // hidden channel nt * 32 + bb * 8 + 4 kh + e - the four gates of four hidden channels of a pixel in ONE lane.
        const int hd = P.hd;
        const float *csrc = P.c_prev ? P.c_prev : P.c_out;
        typedef float f32x4q __attribute__((ext_vector_type(4)));       // (a plain vector type: HIP's float4 struct would be passed to the asm through memory)
        f32x4q cpq[MB][NB];
        long pix[MB];
        // the previous cell state of all the lane's items: unconditional asm loads from clamped (always valid) addresses, requested
        // before the gate activations and awaited behind them (left to hipcc each load sits right in front of its use)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int y = y0 + MB * ph + m;
            pix[m] = ((long)img * H + min(y, H - 1)) * W + min(xg, W - 1);
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int hcl = min(nt * 32 + (2 * chalf + n) * 8 + 4 * kh, hd - 4);
                const float *p = csrc + pix[m] * hd + hcl;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(cpq[m][n]) : "v"(p) : "memory");
            }
        }
        // activations in place: 8 of the 10 transcendental pairs per (pixel, channel), no memory operation - they cover the loads above
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[m][n][v] = (v >> 2) == 3 ? b_tanh_fast(acc[m][n][v]) : b_sigmoid(acc[m][n][v]);
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n) asm volatile("s_waitcnt vmcnt(0)" : "+v"(cpq[m][n]));
        BSTAMP(2);
        const bool has_c = P.c_prev != nullptr;
        float *cout = P.c_out;
        unsigned short *hout16 = reinterpret_cast<unsigned short *>(P.h_out), *gout16 = reinterpret_cast<unsigned short *>(P.gates_out);
        float *hout32 = reinterpret_cast<float *>(P.h_out), *gout32 = reinterpret_cast<float *>(P.gates_out);
        const bool h16 = P.h_dtype == RNH_DT_BF16, g16 = P.gates_dtype == RNH_DT_BF16, wantg = P.gates_out != nullptr;
        const int hc8 = nt * 32 + (2 * chalf + kh) * 8;               // after the swap: the lane's eight hidden channels
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const bool valid = xin && y0 + MB * ph + m < H;
            float hn[NB][4];
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int hc = nt * 32 + (2 * chalf + n) * 8 + 4 * kh;
                float cn[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float cp = has_c ? cpq[m][n][e] : 0.f;
                    cn[e] = acc[m][n][4 + e] * cp + acc[m][n][e] * acc[m][n][12 + e];
                    hn[n][e] = acc[m][n][8 + e] * b_tanh_fast(cn[e]);
                }
                if (valid && hc < hd) {
                    *reinterpret_cast<float4 *>(cout + pix[m] * hd + hc) = make_float4(cn[0], cn[1], cn[2], cn[3]);
                    if (!h16) *reinterpret_cast<float4 *>(hout32 + pix[m] * hd + hc) = make_float4(hn[n][0], hn[n][1], hn[n][2], hn[n][3]);
                    if (wantg && !g16) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *reinterpret_cast<float4 *>(gout32 + pix[m] * 4 * hd + g * hd + hc) =
                                make_float4(acc[m][n][4 * g], acc[m][n][4 * g + 1], acc[m][n][4 * g + 2], acc[m][n][4 * g + 3]);
                    }
                }
            }
            // bf16 destinations: pack the two blocks' quads and let the lane halves trade them - lane half kh then owns the eight
            // channels of block 2 chalf + kh: one 16-byte store per tensor row piece
            if constexpr (NB == 2) {
                if (h16) {
                    unsigned a0 = pk2(hn[0][0], hn[0][1]), a1 = pk2(hn[0][2], hn[0][3]), b0 = pk2(hn[1][0], hn[1][1]), b1 = pk2(hn[1][2], hn[1][3]);
                    swap_halves(a0, b0);
                    swap_halves(a1, b1);
                    if (valid && hc8 < hd) *reinterpret_cast<uint4 *>(hout16 + pix[m] * hd + hc8) = make_uint4(a0, a1, b0, b1);
                }
                if (wantg && g16) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        unsigned a0 = pk2(acc[m][0][4 * g], acc[m][0][4 * g + 1]), a1 = pk2(acc[m][0][4 * g + 2], acc[m][0][4 * g + 3]);
                        unsigned b0 = pk2(acc[m][1][4 * g], acc[m][1][4 * g + 1]), b1 = pk2(acc[m][1][4 * g + 2], acc[m][1][4 * g + 3]);
                        swap_halves(a0, b0);
                        swap_halves(a1, b1);
                        if (valid && hc8 < hd) *reinterpret_cast<uint4 *>(gout16 + pix[m] * 4 * hd + g * hd + hc8) = make_uint4(a0, a1, b0, b1);
                    }
                }
            }
        }
    } else {
        // STORE / PS / LSTM_BWD: per (block n, quad pair p) the lane halves trade quads; lane half kh then holds the eight consecutive
        // columns from c8 = 32 block(n) + 16 p + 8 kh of the tile.  What those columns are is looked up once per (n, p).
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int c8 = (BLK0 + BSTR * n) * 32 + 16 * p + 8 * kh, n0 = nt * NCOLS + c8;
                bool live;
                void *dptr;
                int ddt, dacc = 0;
                long dimg = 0, dch = 0, dC = 0;                      // element = ((img' * H + y) * W' + x') * dC + dch with the strides below
                int ps_r = 1, ps_i = 0, ps_j = 0;
                [[maybe_unused]] int rec = -1;                       // LSTM_BWD: first hidden channel of a dh_rec group, -1 for a column group of the input gradient
                if constexpr (EPI == RNH_EPI_PS) {
                    const int rr = P.ps_r, cq = P.ps_cq;
                    live = n0 < cq * rr * rr;
                    const int sub = n0 / cq;
                    ps_r = rr, ps_i = sub / rr, ps_j = sub - (sub / rr) * rr;
                    dptr = P.dst[0].ptr, ddt = P.dst[0].dtype, dC = cq, dch = n0 - sub * cq, dimg = img;
                } else if constexpr (EPI == RNH_EPI_LSTM_BWD) {
                    // the tile's columns: [input gradient: dst[0].ncols | dh_rec: hd]
                    const rnh_mdst_t &D = P.dst[0];
                    live = c8 < D.ncols;
                    rec = !live && c8 - D.ncols < P.hd ? c8 - D.ncols : -1;
                    dptr = D.ptr, ddt = D.dtype, dacc = D.accumulate, dC = D.C, dch = D.c0 + (live ? c8 : 0), dimg = img + D.img_off;
                } else {
                    int seg = -1, cbase = 0;
                    for (int d = 0; d < P.ndst; ++d) {
                        if (seg < 0 && n0 < cbase + P.dst[d].ncols) seg = d;
                        if (seg < 0) cbase += P.dst[d].ncols;
                    }
                    live = seg >= 0;
                    const rnh_mdst_t &D = P.dst[live ? seg : 0];
                    dptr = D.ptr, ddt = D.dtype, dacc = D.accumulate, dC = D.C, dch = D.c0 + (n0 - cbase), dimg = img + D.img_off;
                }
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const int y = y0 + MB * ph + m;
                    const bool valid = xin && y < H;
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f[e] = acc[m][n][8 * p + e];
                        f[4 + e] = acc[m][n][8 * p + 4 + e];
                        swap_halves(f[e], f[4 + e]);
                    }
                    if (live && valid) {
                        const long e = ((dimg * H + y) * ps_r + ps_i) * ((long)W * ps_r) * dC + ((long)xg * ps_r + ps_j) * dC + dch;
                        if (dacc) {
                            float old[8];
                            load8(dptr, ddt, e, old);
#pragma unroll
                            for (int q = 0; q < 8; ++q) f[q] += old[q];
                        }
                        store8(dptr, ddt, e, f);
                    }
                    if constexpr (EPI == RNH_EPI_LSTM_BWD) {
                        // item (pixel, 8 hidden channels): the body of gates_bwd_m_kernel (mixed_kernels.hip), expression by expression, with
                        // dh2 := dh_rec out of the accumulators, rounded to the element type the unfused path would have stored it in
                        if (rec >= 0 && valid) {
                            const int hd = P.hd;
                            const long pp = ((long)img * H + y) * W + xg;
                            const long o = pp * hd + rec, og = pp * 4 * hd + rec;
                            const float *dcn = P.bw_dc_next, *cprev = P.bw_c_prev;
                            const int gdt = P.gates_dtype, dgdt = P.bw_dgates_dtype;
                            float vdh[8], vdc[8], vcp[8], vcn[8], gi[8], gf[8], go[8], gg[8];
                            load8(P.bw_dh, P.bw_dh_dtype, o, vdh);
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                float t = f[q];
                                if (P.bw_rec_dtype == RNH_DT_BF16) t = (float)(__bf16)t;
                                vdh[q] += t;
                            }
                            if (dcn) load8(dcn, RNH_DT_F32, o, vdc);
                            if (cprev) load8(cprev, RNH_DT_F32, o, vcp);
                            load8(P.bw_c_next, RNH_DT_F32, o, vcn);
                            load8(P.bw_gates, gdt, og, gi);
                            load8(P.bw_gates, gdt, og + hd, gf);
                            load8(P.bw_gates, gdt, og + 2 * hd, go);
                            load8(P.bw_gates, gdt, og + 3 * hd, gg);
                            float di[8], df[8], dgo[8], dg[8], dcp[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const float th = tanhf(vcn[q]);
                                const float d_o = vdh[q] * th;
                                const float dct = (dcn ? vdc[q] : 0.f) + vdh[q] * go[q] * (1.f - th * th);
                                di[q] = dct * gg[q] * gi[q] * (1.f - gi[q]);
                                df[q] = dct * (cprev ? vcp[q] : 0.f) * gf[q] * (1.f - gf[q]);
                                dgo[q] = d_o * go[q] * (1.f - go[q]);
                                dg[q] = dct * gi[q] * (1.f - gg[q] * gg[q]);
                                dcp[q] = dct * gf[q];
                            }
                            store8(P.bw_dgates, dgdt, og, di);
                            store8(P.bw_dgates, dgdt, og + hd, df);
                            store8(P.bw_dgates, dgdt, og + 2 * hd, dgo);
                            store8(P.bw_dgates, dgdt, og + 3 * hd, dg);
                            if (P.bw_dc_prev) store8(P.bw_dc_prev, RNH_DT_F32, o, dcp);
                        }
                    }
                }
            }
    }
    BSTAMP(3);
}

inline int bgrid_for(long n, int cap = 8192) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

int check_msrc(const rnh_msrc_t &s, const char *who) {
    if (!s.ptr) RNH_FAIL(RNH_E_ARG, "%s: null source pointer", who);
    if (s.dtype != RNH_DT_F32 && s.dtype != RNH_DT_BF16) RNH_FAIL(RNH_E_ARG, "%s: bad source element type", who);
    const int g = s.dtype == RNH_DT_BF16 ? 7 : 3;
    if (s.C <= 0 || s.nch <= 0 || s.c0 < 0 || s.c0 + s.nch > s.C) RNH_FAIL(RNH_E_ARG, "%s: bad channel range", who);
    if ((s.C & g) || (s.c0 & g) || (s.nch & g)) RNH_FAIL(RNH_E_ALIGN, "%s: channels must be multiples of %d", who, g + 1);
    if (s.scale < 1 || s.sub_y < 0 || s.sub_x < 0 || s.sub_y >= s.scale || s.sub_x >= s.scale) RNH_FAIL(RNH_E_ARG, "%s: bad scale / sub-pixel", who);
    return 0;
}

}  // namespace

int rnh_check_msrc(const rnh_msrc_t &s, const char *who) { return check_msrc(s, who); }

#ifdef RNH_STAMPS
extern "C" int rnh_debug_bf16_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bf16_stamps), sizeof(g_bf16_stamps));
}
#endif

extern "C" int rnh_pack_weights_bf16(const float *w, const float *bias, void *wp, float *biasp, const int32_t *kbase, const int32_t *knv,
                                     const int32_t *ktap, const int32_t *kcoff, const int32_t *colmap, int nk, int Npad, int Cout, int Cin,
                                     int ntaps, int kstride, int transposed, void *stream) {
    if (!w || !wp || !kbase || !knv || !ktap || !colmap || nk < 1 || Npad < 1 || Cout < 1 || Cin < 1 || kstride < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_pack_weights_bf16: bad arguments");
    if (ntaps != 9 && ntaps != 1) RNH_FAIL(RNH_E_RANGE, "rnh_pack_weights_bf16: ntaps must be 9 or 1");
    if (Npad % 64) RNH_FAIL(RNH_E_RANGE, "rnh_pack_weights_bf16: Npad must be a multiple of 64");
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(bgrid_for((long)nk * Npad * 16 + Npad)), dim3(256), 0, (hipStream_t)stream, w, bias,
                       (unsigned short *)wp, biasp, kbase, knv, ktap, kcoff, colmap, nk, Npad, Cout, Cin, ntaps, kstride, transposed);
    RNH_CHECK_LAUNCH("rnh_pack_weights_bf16");
    return 0;
}

extern "C" int rnh_conv_bf16(const rnh_conv_bf16_args_t *args, void *stream) {
    if (!args) RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: null args");
    const rnh_conv_bf16_args_t &a = *args;
    if (a.nsrc < 1 || a.nsrc > RNH_MAX_SRC || a.B < 1 || a.H < 1 || a.W < 1 || !a.wp) RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: bad arguments");
    if (a.ntaps != 9 && a.ntaps != 1) RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: ntaps must be 9 or 1");
    if (a.Npad < 64 || a.Npad % 64) RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: Npad must be a multiple of 64");
    int chunks = 0;
    for (int i = 0; i < a.nsrc; ++i) {
        if (int rc = check_msrc(a.src[i], "rnh_conv_bf16")) return rc;
        if (a.src[i].scale != a.src[0].scale) RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: one scale for all sources");
        chunks += (a.src[i].nch + 15) / 16;
    }
    if (chunks != a.nchunks) RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: nchunks = %d but the sources hold %d chunks of 16 channels", a.nchunks, chunks);
    if ((long)a.B * a.H * a.W * a.src[0].scale * a.src[0].scale >= (1L << 31)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: too many pixels");
    const int ncols = a.Npad % 128 ? 64 : 128;
    constexpr int TH = 8;
    const int TYn = (a.H + TH - 1) / TH, TXn = (a.W + TW - 1) / TW, NT = a.Npad / ncols;
    const long blocks = (long)a.B * TYn * TXn * NT;
    if (blocks >= (1L << 31)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: grid too large");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(256);
    // 32-channel chunks (deeper weight-fragment ring, half the barriers) where every source is bf16 with a multiple of 32 channels
    bool k32 = a.ntaps == 9 && !(getenv("RNH_BF16_KC") && getenv("RNH_BF16_KC")[0] == '1');
    for (int i = 0; i < a.nsrc; ++i) k32 = k32 && a.src[i].dtype == RNH_DT_BF16 && a.src[i].nch % 32 == 0;
#define RNH_LAUNCH(EPI, NC, NTP) hipLaunchKernelGGL((conv_bf16d_kernel<EPI, NC, NTP>), grid, block, 0, st, a, TYn, TXn, NT)
#define RNH_LAUNCH9(EPI, NC)                                                                                             \
    do {                                                                                                                 \
        if (k32) hipLaunchKernelGGL((conv_bf16d_kernel<EPI, NC, 9, 32>), grid, block, 0, st, a, TYn, TXn, NT);          \
        else hipLaunchKernelGGL((conv_bf16d_kernel<EPI, NC, 9, 16>), grid, block, 0, st, a, TYn, TXn, NT);              \
    } while (0)
    switch (a.epilogue) {
        case RNH_EPI_STORE:
            if (a.ndst < 1 || a.ndst > RNH_MAX_DST) RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: bad destination count");
            for (int d = 0; d < a.ndst; ++d) {
                const rnh_mdst_t &D = a.dst[d];
                if (!D.ptr || D.ncols < 1 || (D.dtype != RNH_DT_F32 && D.dtype != RNH_DT_BF16)) RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: bad destination %d", d);
                if ((D.C & 7) || (D.c0 & 7) || (D.ncols & 7)) RNH_FAIL(RNH_E_ALIGN, "rnh_conv_bf16: destination channels must be multiples of 8");
            }
            if (a.ntaps == 9) {
                if (ncols == 128) RNH_LAUNCH9(RNH_EPI_STORE, 128);
                else RNH_LAUNCH9(RNH_EPI_STORE, 64);
            } else {
                if (ncols == 128) RNH_LAUNCH(RNH_EPI_STORE, 128, 1);
                else RNH_LAUNCH(RNH_EPI_STORE, 64, 1);
            }
            break;
        case RNH_EPI_PS:
            if (a.ntaps != 9) RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: the pixel-shuffle epilogue serves 3x3 convolutions");
            if (a.ndst != 1 || !a.dst[0].ptr || a.ps_r < 1 || a.ps_cq < 8 || (a.ps_cq & 7) || a.ps_cq * a.ps_r * a.ps_r > a.Npad)
                RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: bad pixel-shuffle destination");
            if (ncols == 128) RNH_LAUNCH9(RNH_EPI_PS, 128);
            else RNH_LAUNCH9(RNH_EPI_PS, 64);
            break;
        case RNH_EPI_LSTM_BWD: {
            if (a.ntaps != 9) RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: the LSTM-backward epilogue serves 3x3 convolutions");
            const rnh_mdst_t &D = a.dst[0];
            if (a.ndst != 1 || !D.ptr || D.ncols < 8 || (D.dtype != RNH_DT_F32 && D.dtype != RNH_DT_BF16) || (D.C & 7) || (D.c0 & 7) || (D.ncols & 7))
                RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: the LSTM-backward epilogue stores the input gradient to dst[0] (channels in multiples of 8)");
            if (a.hd < 8 || (a.hd & 7) || NT != 1 || D.ncols + a.hd > a.Npad)
                RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: LSTM-backward epilogue: input-gradient + hd columns must fit ONE column tile (Npad %d)", a.Npad);
            if (!a.bw_dh || !a.bw_gates || !a.bw_c_next || !a.bw_dgates) RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: LSTM-backward epilogue needs bw_dh, bw_gates, bw_c_next, bw_dgates");
            if ((a.bw_dh_dtype != RNH_DT_F32 && a.bw_dh_dtype != RNH_DT_BF16) || (a.bw_dgates_dtype != RNH_DT_F32 && a.bw_dgates_dtype != RNH_DT_BF16) ||
                (a.bw_rec_dtype != RNH_DT_F32 && a.bw_rec_dtype != RNH_DT_BF16) ||
                (a.gates_dtype != RNH_DT_F32 && a.gates_dtype != RNH_DT_BF16))
                RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: LSTM-backward epilogue: bad element type");
            if (ncols == 128) RNH_LAUNCH9(RNH_EPI_LSTM_BWD, 128);
            else RNH_LAUNCH9(RNH_EPI_LSTM_BWD, 64);
            break;
        }
        case RNH_EPI_LSTM:
            if (a.ntaps != 9) RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: the LSTM epilogue serves 3x3 convolutions");
            if (!a.h_out || !a.c_out || a.hd < 8 || (a.hd & 7) || !a.bias) RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: LSTM epilogue needs h_out, c_out, hd % 8 == 0, bias");
            if (a.Npad != 128 * ((a.hd + 31) / 32)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: LSTM column layout (plans.lstm_colmap)");
            RNH_LAUNCH9(RNH_EPI_LSTM, 128);
            break;
        default:
            RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: epilogue %d not available", a.epilogue);
    }
#undef RNH_LAUNCH9
#undef RNH_LAUNCH
    RNH_CHECK_LAUNCH("rnh_conv_bf16");
    return 0;
}
