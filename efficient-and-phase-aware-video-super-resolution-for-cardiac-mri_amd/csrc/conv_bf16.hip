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
#include "conv_bf16_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// wp[ks][n][kk] = W[o][i][tap] as bf16, index conventions of rnh_pack_weights (conv_igemm.hip), kk in natural order
template <bool F16>
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
        if constexpr (F16) {
            const _Float16 h = (_Float16)v;                         // v_cvt_f16_f32 (RNE)
            wp[e] = __builtin_bit_cast(unsigned short, h);
        } else {
            const __bf16 b = (__bf16)v;
            wp[e] = __builtin_bit_cast(unsigned short, b);
        }
    }
}

// Diagnostic ablations (tools/bf16_ablate.sh; never defined in the product build): -DRNH_EXP=<mask> removes one cost of the main
// loop at a time - results are WRONG, only the launch time is of interest.  1: weight fragments loaded once, 2: halo fragments read
// once per chunk, 4: no halo staging after the prologue, 8: no barrier in the loop, 16: epilogue skipped; 32 / 64 / 128 keep the
// results: column-tile-major block order, second workgroup of a CU delayed by ~6 / ~12 us; 256: one workgroup per CU (LDS padded); 512: the generic item loop of the LSTM-backward epilogue instead of the prefetching one
#ifndef RNH_EXP
#define RNH_EXP 0
#endif
// RNH_M16 = 1 builds the 32-channel-chunk kernels on v_mfma_f32_16x16x32_bf16 with the halo brought in by LDS-DMA (the "M16" blocks below).
// Measured in round 5 and NOT the product build: a bare loop of that MFMA shape sustains 2.07 PFLOP/s against 1.86 for 32x32x16 on this board
// (tools/probes/mfma_bf16_shapes2.hip, profiles/r05_f_*: the chip holds 2.1 instead of 1.87 GHz under it), but inside this kernel the two forms
// are equal - ConvLSTM cell 78.4-78.9 against 79.1-80.5 us, bf16 step 80.78 against 80.71 ms, the fused backward launch 5 % slower (profiles/r05_i_*,
// r05_j_*): the launch is not bound by its main loop's matrix instructions.  Kept as an A/B build (124 bf16 tests green with it).
#ifndef RNH_M16
#define RNH_M16 0
#endif
// RNH_W8 = 1 (A/B build, NOT the product): the waves of a 128-column 3x3 workgroup are its four 32-column groups, each over ALL eight tile rows (8 x 1
// accumulator blocks) instead of (pixel half) x (column half) with 4 x 2 blocks: a weight fragment then feeds eight MFMAs instead of four and no two waves
// of a workgroup stream the same fragment (L2 -> register traffic per MFMA halves), at twice the halo-fragment LDS reads per MFMA.  Measured in round 5:
// bit-identical, 1-2 % SLOWER (cell 79.4 against 78.0 us, refine conv1 2.91 against 2.85 ms, bf16 step 79.0 against 78.5 ms; profiles/r05_x_*): operand
// bytes per MFMA go from 0.75 to 1.125 KB - the 4 x 2 block shape is the one that moves the fewest, and the launch's time follows the bytes, not where they come from.
#ifndef RNH_W8
#define RNH_W8 0
#endif

// (experiment, tools/experiments/r05_prio.sh: s_setprio RNH_PRIO for the main loop, RNH_PRIO_EPI for the epilogue - every combination of
// 0..3 measured within noise of no s_setprio at all, profiles/r05_d_setprio.txt; the product build issues none)
#if defined(RNH_PRIO) || defined(RNH_PRIO_EPI)
#ifndef RNH_PRIO
#define RNH_PRIO 0
#endif
#ifndef RNH_PRIO_EPI
#define RNH_PRIO_EPI 0
#endif
#define RNH_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define RNH_SETPRIO(x)
#endif

#ifdef RNH_STAMPS
__device__ unsigned long long g_bf16_stamps[64];
#define BSTAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_bf16_stamps[i] = __builtin_readcyclecounter(); } while (0)
// per-workgroup trace (tools/bf16_wgtrace.py): wall clock (100 MHz) and shader cycles at the start / park / end of every workgroup, and where it ran
__device__ unsigned long long g_bf16_wgtrace[8 * 4096];
#define WGTRACE(slot) do { if (threadIdx.x == 0 && blockIdx.x < 4096) { g_bf16_wgtrace[8 * blockIdx.x + 2 * (slot)] = __builtin_amdgcn_s_memrealtime(); \
        g_bf16_wgtrace[8 * blockIdx.x + 2 * (slot) + 1] = __builtin_amdgcn_s_memtime(); \
        if ((slot) == 0) g_bf16_wgtrace[8 * blockIdx.x + 6] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4); } } while (0)
#else
#define BSTAMP(i)
#define WGTRACE(slot)
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
    // KC == 32 (round 5): v_mfma_f32_16x16x32_bf16 - a lane supplies 8 channels of one of 16 pixels, the four 8-channel groups of the chunk sit in
    // four lane groups.  The halo is stored as four PLANES [channel group][pixel][16 bytes] (plane stride a multiple of 256 bytes): the 16 lanes
    // that a ds_read_b128 serves per LDS cycle (MI355X_MICROARCH.md, LDS table: {0-3, 12-15, 20-27}, ...) are 8 pixels of one plane and the 8
    // pixels between them of the next plane = 16 different 16-byte bank groups.  (Pixel-major with any pitch cannot be conflict-free for this
    // access: the second plane's slots would have to be the first's shifted by one.)  A plane is 64 consecutive pixels per wave-instruction of
    // an LDS-DMA load (buffer_load_dwordx4 ... lds: LDS address = M0 + 16 lane), so the halo goes from memory to LDS without staging registers
    // or ds_write: wave w fills plane w of the chunk, six instructions (tools/probes/lds_dma_oob.hip: lanes with an out-of-range offset write
    // zeros - the zero padding and the 44 slots behind the 340 pixels cost no branch)
    static constexpr bool M16 = KC == 32 && RNH_M16;
    static constexpr int PLANE = (HP + 63) / 64 * 1024;          // whole LDS-DMA wave-instructions (64 slots of 16 bytes); a multiple of 256 bytes
    static constexpr int A_BYTES = M16 ? 4 * PLANE : HP * APITCH, A_PIECES = PPP * HP, A_ITERS = (A_PIECES + 255) / 256;
    static constexpr int OPITCH = NCOLS + 4;                    // floats per parked pixel
    static constexpr int PXR = 2 * TW;                          // pixels parked per epilogue round: one 32-pixel row block of each pixel half
    static constexpr int OUT_BYTES = 2 * PXR * OPITCH * 4;      // two park images
    static constexpr int SMEM0 = 2 * A_BYTES > OUT_BYTES ? 2 * A_BYTES : OUT_BYTES;
    static constexpr int SMEM = (RNH_EXP & 256) ? 100 * 1024 : SMEM0;      // (experiment 256: ONE workgroup per CU)
    static_assert(2 * SMEM0 <= 160 * 1024, "two workgroups per CU");
};

struct rnh_conv_bf16_pair_t {
    rnh_conv_bf16_args_t call[2];
};

// PP, nA: ONE launch may serve TWO calls of equal geometry (rnh_conv_bf16_pair: the ConvLSTM cells of the two directions at small images, where a
// call alone leaves half the chip idle): workgroups [0, nA) belong to PP.call[0], the rest to PP.call[1].  A single call passes nA = its workgroup count (call[1] is never read).
// F16 (round 6; the upsampler's PixelShuffle convolutions in the forward): the packed weights are IEEE half (rnh_pack_weights_f16: 11 bits instead of
// 8), the halo's bf16 values are converted to half on their way to LDS (exact) and the contraction runs on v_mfma_f32_32x32x16_f16 - the same rate.
template <int EPI, int NCOLS, int NTAPS, int KC = 16, bool F16 = false>
__global__ void __launch_bounds__(256, 2) conv_bf16d_kernel(const rnh_conv_bf16_pair_t PP, const int nA, const int TYn, const int TXn, const int NT) {
    const int second = (int)blockIdx.x >= nA;
    const rnh_conv_bf16_args_t &P = PP.call[second];                 // (an offset into the kernel-argument segment: no copy)
    const int bx = second ? (int)blockIdx.x - nA : (int)blockIdx.x;
    using G = GeoD<NCOLS, KC>;
    constexpr int TH = G::TH, NB = G::NB, MB = G::MB, A_BYTES = G::A_BYTES, A_PIECES = G::A_PIECES, A_ITERS = G::A_ITERS;
    constexpr int KS = G::KS, APITCH = G::APITCH, PPP = G::PPP;
    constexpr bool M16 = G::M16;
    constexpr int PLANE = G::PLANE, NB16 = NCOLS / 32;               // 16-column blocks of a wave (its NCOLS / 2 columns)
    constexpr bool W8 = RNH_W8 && NCOLS == 128 && NTAPS == 9 && !M16;   // wave = (all 8 tile rows) x (32-column group `wave`)
    static_assert(NTAPS % 3 == 0 || NTAPS == 1, "the fragment ring has three sets");
    static_assert(KC == 16 || (KC == 32 && NTAPS == 9), "32-channel chunks serve the 3x3 kernels");
    static_assert(!F16 || (KC == 32 && !M16 && !W8), "the f16 form is the 32-channel-chunk kernel's (bf16 sources through staging registers)");
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::SMEM];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5, wave = tid >> 6;
    [[maybe_unused]] const int l15 = lane & 15, kg = lane >> 4;
    const int ph = wave & 1, chalf = wave >> 1;
    const int bid = rnh_xcd_remap(bx, P.B * TYn * TXn * NT);
    // (RNH_EXP & 32, experiment: column-tile-major block order - the workgroups that run together stream the SAME weight fragments)
    const int nt = (RNH_EXP & 32) ? bid / (P.B * TYn * TXn) : bid % NT, mt = (RNH_EXP & 32) ? bid % (P.B * TYn * TXn) : bid / NT;
    if ((RNH_EXP & (64 | 128)) && bx >= 256 && bx < 512) {      // experiment: the second workgroup of every CU starts late
        for (int i = 0; i < ((RNH_EXP & 64) ? 2 : 0) + ((RNH_EXP & 128) ? 4 : 0); ++i) __builtin_amdgcn_s_sleep(100);
    }
    const int img = mt / (TYn * TXn), trem = mt - img * (TYn * TXn), ty = trem / TXn, tx = trem - ty * TXn;
    const int y0 = ty * TH, x0 = tx * TW;
    const int H = P.H, W = P.W;
    const int sc = P.src[0].scale, Hs = H * sc, Ws = W * sc;

    // halo staging: piece p = tid + 256 i of the chunk's A_PIECES 16-byte pieces = (halo pixel p / PPP, 8-channel group p % PPP).  256 is a multiple of
    // PPP, so the group is the same for every i and the pixel advances by 256 / PPP: one LDS address and one group per thread, the rest are constants
    constexpr int A_BLOCKS = PLANE / 1024;                           // (M16) LDS-DMA instructions per plane
    int apix[M16 ? A_BLOCKS : A_ITERS];
    const int ahalf0 = tid % PPP;
    const int alds0 = M16 ? ahalf0 * PLANE + (tid / PPP) * 16 : (tid / PPP) * APITCH + ahalf0 * 16;
    constexpr int ALDS_STEP = (256 / PPP) * (M16 ? 16 : APITCH);
    const bool alast = tid < A_PIECES - 256 * (A_ITERS - 1);         // the last round of pieces is a partial one
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < A_BLOCKS; ++i) {                        // halo pixel 64 i + lane (the same in every wave; the wave picks the plane)
            const int px = 64 * i + lane, hr = px / HPW, hc = px - hr * HPW;
            const int y = y0 - 1 + hr, x = x0 - 1 + hc;
            const bool in = px < G::HP && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            apix[i] = in ? (y * sc) * Ws + x * sc : -1;
        }
    } else {
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int p = tid + 256 * i, px = p / PPP, hr = px / HPW, hc = px - hr * HPW;
            const int y = y0 - 1 + hr, x = x0 - 1 + hc;
            const bool in = p < A_PIECES && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            apix[i] = in ? (y * sc) * Ws + x * sc : -1;
        }
    }

    uint4 ra[M16 ? 1 : A_ITERS];
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
            const int ch = ahalf0 * 8;
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
            if constexpr (F16) v = bf2h8(v);
            if (i + 1 < A_ITERS || alast) *reinterpret_cast<uint4 *>(Ab + alds0 + i * ALDS_STEP) = v;
        }
    };

    // (M16) the halo of the next 32-channel chunk of the source list, by LDS-DMA into halo buffer buf: wave w requests channel group w of every halo
    // pixel.  Inline asm (hipcc would drain every ordinary load behind an LDS-DMA builtin with vmcnt(0): cdna_hip_programming.md, "Pipelining
    // across barriers"); M0 is written in the statement that reads it.  The requests count in vmcnt like any load: the chunk loop waits for them with
    // a COUNTED vmcnt that leaves the weight-fragment loads issued behind them in flight.
    typedef int i32x4q __attribute__((ext_vector_type(4)));
    [[maybe_unused]] const unsigned smem_lds = (unsigned)(unsigned long long)(&smem[0]);
    [[maybe_unused]] auto dma_chunk = [&](int buf) {
        const rnh_msrc_t &S = P.src[si];
        const char *base = reinterpret_cast<const char *>(S.ptr) +
                           ((((long)(img + S.img_off) * Hs + S.sub_y) * Ws + S.sub_x) * S.C + S.c0 + cc * KC + wave * 8) * 2;
        const unsigned long long u = (unsigned long long)base;
        i32x4q rs;
        rs[0] = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
        rs[1] = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
        rs[2] = 0x7fffffff;
        rs[3] = 0x00020000;
        const int pstride = S.C * 2;
        const unsigned l0 = __builtin_amdgcn_readfirstlane(smem_lds + buf * A_BYTES + wave * PLANE);
#pragma unroll
        for (int i = 0; i < A_BLOCKS; ++i) {
            const int voff = apix[i] >= 0 ? apix[i] * pstride : -1;
            const unsigned ld = l0 + i * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(ld), "v"(voff), "s"(rs) : "memory");
        }
        if (++cc * KC >= S.nch) {
            cc = 0;
            ++si;
        }
    };

    // weight fragments: descriptor over the packed weights, per-lane offset inside a (chunk, tap) slab, slab stride
    const __amdgpu_buffer_rsrc_t wrs = bdesc(P.wp);
    const int wlane = ((nt * NCOLS + (W8 ? wave * 32 : chalf * (NCOLS / 2)) + l31) * 16 + kh * 8) * 2;
    const int slab = P.Npad * 32;                               // bytes of one (chunk, tap) slab
    constexpr int RING = NTAPS == 1 ? 1 : (M16 ? 1 : (KC == 32 ? 6 : 3)), AHEAD = RING - 1;
    constexpr int NSTEP = NTAPS * KS;                            // (tap, k step) pairs per chunk; NSTEP % RING == 0: the set of a step is static
    static_assert(NTAPS == 1 || NSTEP % RING == 0, "the fragment ring must divide the steps of a chunk");
    constexpr int NBW = W8 ? 1 : NB;                                // 32-column blocks of a wave
    uint4 bq[RING][NBW];
    const int nch = P.nchunks / KS;
    // fragments of step sa of chunk c (sa >= NSTEP: of the chunks behind it): slab (16-channel chunk KS * chunk + ks, tap).  sa is a compile-time
    // constant at every call site, so tap, k step and chunk increment are too: the slab's byte offset is ONE scalar multiply-add and goes into the
    // load's scalar offset, the per-lane offsets never change (round 4; until then a division by NSTEP per step - 14 SALU + 4 VALU instructions
    // in front of every 8 MFMAs).  Behind the last chunk the fragments are never used: the last chunk's are read again.
    int wvo[NBW];
#pragma unroll
    for (int n = 0; n < NBW; ++n) wvo[n] = wlane + n * 32 * 32;
    auto bload = [&](int c, int sa, int set) {
        const int q = sa / NSTEP, r = sa - q * NSTEP, tap = r / KS, ks = r - tap * KS;
        const int cc = c + q < nch ? c + q : nch - 1;
        const int base = (cc * KS + ks) * NTAPS * slab + tap * slab;
#pragma unroll
        for (int n = 0; n < NBW; ++n) bq[set][n] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo[n], base, 0));
    };

    // (W8) accumulator block m = tile row m, the wave's 32 columns; a HALF step = (tap, k step, tile rows 0-3 / 4-7): 4 A fragments (ring of three
    // sets, requested two half steps = 8 MFMAs ahead), 4 MFMAs; the step's one B fragment serves both half steps
    [[maybe_unused]] f32x16 acc8[W8 ? 2 * MB : 1];
    if constexpr (W8) {
#pragma unroll
        for (int m = 0; m < 2 * MB; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc8[m][v] = 0.f;
    }
    [[maybe_unused]] auto compute8 = [&](int buf, int c) {
        const unsigned char *Ab = smem + buf * A_BYTES + l31 * APITCH + kh * 16;
        bf16x8 a[3][MB];
        auto afrags = [&](int hs, int set) {
            const int step = hs >> 1, half = hs & 1, tap = step / KS, ks = step - tap * KS, dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int m = 0; m < MB; ++m) a[set][m] = *reinterpret_cast<const bf16x8 *>(Ab + ((half * MB + m + dy) * HPW + dx) * APITCH + ks * 32);
        };
        afrags(0, 0);
        afrags(1, 1);
#pragma unroll
        for (int hs = 0; hs < 2 * NSTEP; ++hs) {
            const int step = hs >> 1, half = hs & 1;
            if (hs + 2 < 2 * NSTEP) afrags(hs + 2, (hs + 2) % 3);
            if (half == 0) bload(c, step + AHEAD, (step + AHEAD) % RING);
            if (hs == 4 * KS) {
                if (c + 1 < nch) store_chunk(buf ^ 1);
                if (c + 2 < nch) load_chunk();
            }
#pragma unroll
            for (int m = 0; m < MB; ++m)
                acc8[W8 ? half * MB + m : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[hs % 3][m], __builtin_bit_cast(bf16x8, bq[step % RING][0]),
                                                                                      acc8[W8 ? half * MB + m : 0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    f32x16 acc[(M16 || W8) ? 1 : MB][(M16 || W8) ? 1 : NB];
    if constexpr (!M16 && !W8) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[m][n][v] = 0.f;
    }
    // ---- the 16x16x32 form (KC == 32): accumulator block (b, j) = pixels 16 (b & 1) .. + 15 of tile row MB ph + (b >> 1) x columns 16 j .. + 15 of the
    // wave's NCOLS / 2; a step = (tap, pair of tile rows): 4 A fragments (double-buffered), the tap's NB16 B fragments, 4 NB16 MFMAs.  The B
    // fragment of (tap, 16 columns) comes out of the SAME packed weights: lane group kg reads the 16-byte half (kg & 1) of 16-channel slab
    // 2 chunk + (kg >> 1) - its channels 8 kg .. 8 kg + 7 of the chunk, the ones A's lane group kg holds.  Ring of three taps, requested two taps
    // (four steps, 64 MFMAs) ahead, across chunk boundaries.
    typedef float f32x4m __attribute__((ext_vector_type(4)));
    [[maybe_unused]] f32x4m acc16[M16 ? 2 * MB : 1][M16 ? NB16 : 1];
    [[maybe_unused]] uint4 bq16[M16 ? 3 : 1][M16 ? NB16 : 1];
    [[maybe_unused]] int wvo16 = 0;                                  // per-lane byte offset of the first 16-column block; block j is 512 j bytes further (an immediate)
    if constexpr (M16) {
#pragma unroll
        for (int b = 0; b < 2 * MB; ++b)
#pragma unroll
            for (int j = 0; j < NB16; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc16[b][j][v] = 0.f;
        wvo16 = ((nt * NCOLS + chalf * (NCOLS / 2) + l15) * 16 + (kg & 1) * 8) * 2 + (kg >> 1) * NTAPS * slab;
    }
    [[maybe_unused]] auto bload16 = [&](int c, int ta, int set) {        // fragments of tap ta of chunk c (ta >= NTAPS: of the chunks behind it)
        const int q = ta / NTAPS, tap = ta - q * NTAPS;
        const int cc = c + q < nch ? c + q : nch - 1;
        const int base = (cc * 2 * NTAPS + tap) * slab;
#pragma unroll
        for (int j = 0; j < (M16 ? NB16 : 1); ++j) bq16[set][j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo16 + j * 512, base, 0));
    };
    [[maybe_unused]] auto compute16 = [&](int buf, int c) {
        const unsigned char *Ab = smem + buf * A_BYTES + kg * PLANE + ((MB * ph) * HPW + l15) * 16;
        bf16x8 a[2][4];
        auto afrags = [&](int step, int set) {
            const int tap = step >> 1, half = step & 1, dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                a[set][q] = *reinterpret_cast<const bf16x8 *>(Ab + ((half * 2 + (q >> 1) + dy) * HPW + (q & 1) * 16 + dx) * 16);
        };
        afrags(0, 0);
#pragma unroll
        for (int step = 0; step < 2 * NTAPS; ++step) {
            const int tap = step >> 1, half = step & 1;
            if (step + 1 < 2 * NTAPS) afrags(step + 1, (step + 1) & 1);
            if (half == 0) bload16(c, tap + 2, (tap + 2) % 3);
            if (step == 0 && c + 1 < nch) dma_chunk(buf ^ 1);      // (nobody reads that buffer before the barrier at the end of this chunk)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < (M16 ? NB16 : 1); ++j)
                    acc16[M16 ? half * 4 + q : 0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[step & 1][q], __builtin_bit_cast(bf16x8, bq16[tap % 3][j]),
                                                                                               acc16[M16 ? half * 4 + q : 0][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // One chunk.  The halo of chunk c + 1 sits in registers since step 2 KS of chunk c - 1 (a whole chunk of MFMAs ago, so the
    // wait in store_chunk costs nothing); at step 2 KS it goes to the other LDS buffer - nobody reads that one before the
    // barrier at the end of this chunk - and the loads of chunk c + 2 are issued into the same registers.
    [[maybe_unused]] auto compute = [&](int buf, int c) {
        if constexpr (!M16 && !W8) {
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
            if constexpr (NTAPS == 9 && !(RNH_EXP & 1)) bload(c, step + AHEAD, (step + AHEAD) % RING);      // AHEAD steps ahead
            if (step == (NTAPS == 9 ? 2 * KS : 0) && !(RNH_EXP & 4)) {
                if (c + 1 < nch) store_chunk(buf ^ 1);
                if (c + 2 < nch) load_chunk();
            }
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    if constexpr (F16)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[step & 1][m]),
                                                                           __builtin_bit_cast(f16x8, bq[NTAPS == 9 ? step % RING : 0][n]), acc[m][n], 0, 0, 0);
                    else
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(RNH_EXP & 2) ? 0 : (step & 1)][m],
                                                                            __builtin_bit_cast(bf16x8, bq[NTAPS == 9 ? step % RING : 0][n]), acc[m][n], 0, 0, 0);
            if constexpr (NTAPS == 1) bload(c + 1, 0, 0);
            if constexpr (NTAPS == 9) __builtin_amdgcn_sched_barrier(0);     // pin the issue order: hipcc otherwise sinks the weight loads to their first use
        }
        }
    };

    // ---- K loop: double-buffered halo, one barrier per chunk ------------------------------------------------------------
    BSTAMP(0);
    WGTRACE(0);
    RNH_SETPRIO(RNH_PRIO);
    if constexpr (M16) {
        dma_chunk(0);
        bload16(0, 0, 0);
        bload16(0, 1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB16) : "memory");        // the halo has landed, the weight fragments may still be on their way
        __syncthreads();
    } else {
        bload(0, 0, 0);
        if constexpr (NTAPS == 9) {
#pragma unroll
            for (int g = 1; g < AHEAD; ++g) bload(0, g, g);
        }
    }
    if constexpr (!M16) {
        load_chunk();
        store_chunk(0);
        if (nch > 1) load_chunk();
        __syncthreads();
    }
    for (int c = 0; c < nch; ++c) {
        BSTAMP(8 + 3 * (c & 15));
        if constexpr (M16) compute16(c & 1, c);
        else if constexpr (W8) compute8(c & 1, c);
        else compute(c & 1, c);
        BSTAMP(9 + 3 * (c & 15));
        BSTAMP(10 + 3 * (c & 15));
        // (M16) this wave's halo requests of step 0 have landed: behind them it has issued the fragments of 8 taps (steps 2 .. 16)
        if constexpr (M16) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * NB16) : "memory");
        if (!(RNH_EXP & 8)) __syncthreads();
    }
    RNH_SETPRIO(RNH_PRIO_EPI);
    BSTAMP(1);
    WGTRACE(1);
    if (RNH_EXP & 16) {                                         // keep the accumulators alive, skip the epilogue
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < ((M16 || W8) ? 1 : MB); ++m)
#pragma unroll
            for (int n = 0; n < ((M16 || W8) ? 1 : NB); ++n)
#pragma unroll
                for (int v = 0; v < 16; ++v) s += acc[m][n][v];
        if constexpr (W8) {
#pragma unroll
            for (int m = 0; m < 2 * MB; ++m)
#pragma unroll
                for (int v = 0; v < 16; ++v) s += acc8[m][v];
        }
        if constexpr (M16) {
#pragma unroll
            for (int b = 0; b < 2 * MB; ++b)
#pragma unroll
                for (int j = 0; j < NB16; ++j)
#pragma unroll
                    for (int v = 0; v < 4; ++v) s += acc16[b][j][v];
        }
        if (s == 1.2345e-30f) reinterpret_cast<float *>(smem)[tid] = s;
        return;
    }

    // ---- epilogue (round 5 form): MB = 4 rounds, one accumulator row block of EVERY wave per round.  Round r parks tile rows r and
    // 4 + r (64 pixels x NCOLS columns, + bias) as an fp32 [pixel][column] image, then all 256 threads finish 8 columns of a pixel
    // per item with whole-row 16-byte accesses.  Two such images: the block of round r + 1 is parked BEFORE the items of round r
    // are read, so a round costs one barrier, every wave parks and finishes the same amount in every round, and a wave's LDS
    // writes run beside its LDS reads.  (Until round 4: two rounds of 128 pixels, parked by the two waves that own that pixel half
    // while the other two waited at the barrier, two barriers per round - per-workgroup traces, tools/bf16_wgtrace.py, showed the
    // finishing phase at 38 % of a workgroup's lifetime; the values parked and the arithmetic of an item are unchanged, so are
    // the results, bit for bit.)
    float *const ot0 = reinterpret_cast<float *>(smem);
    constexpr int PXR = G::PXR, ROUNDS = MB, OBUF = PXR * G::OPITCH;
    static_assert(PXR == 2 * TW && TH == 2 * MB, "a round = one row block of the two pixel halves");
    // ConvLSTM: the previous cell state of ALL the thread's items (one per round: 8 channels, fp32) is requested here, before the
    // accumulators are parked, as unconditional asm loads from clamped (always valid) addresses; left to hipcc each load sat
    // right in front of its use, one exposed memory round trip per item (tools/bf16_stamps.py, round 3).  No previous state: any
    // valid address, zeros behind the wait.
    constexpr int NIT = EPI == RNH_EPI_LSTM ? ROUNDS : 1;
    typedef float f32x4q __attribute__((ext_vector_type(4)));       // (a plain vector type: HIP's float4 struct would be passed to the asm through memory)
    [[maybe_unused]] f32x4q cpq[NIT][2];
    if constexpr (EPI == RNH_EPI_LSTM) {
        const int hd = P.hd, hcl = min(nt * 32 + (tid & 3) * 8, hd - 8);
        const float *csrc = P.c_prev ? P.c_prev : P.c_out;
        const int px = tid >> 2;
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
            const int y = min(y0 + (px >> 5) * MB + q, H - 1), x = min(x0 + (px & (TW - 1)), W - 1);
            const float *p = csrc + (((long)img * H + y) * W + x) * hd + hcl;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(cpq[q][0]) : "v"(p) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(cpq[q][1]) : "v"(p) : "memory");
        }
    }
    // ConvLSTM backward, the shape the bf16 engine launches (hd = 64 hidden channels, 64 input-gradient columns, bf16 dh / gates): a thread has
    // exactly two gate items per round - pixels (tid >> 3) and (tid >> 3) + 32 of the image, channels 8 (tid & 7) .. - and their eleven 16-byte
    // operands (dh, dc_next, c_prev, c_next, the four gates) are requested a ROUND AHEAD as unconditional asm loads from clamped addresses, like
    // the forward cell's previous state above: in the generic loop below every item issued its loads right in front of their use, eleven
    // exposed round trips per item with two waves per SIMD to hide them (125 us per launch at config 2's size against 78 + 60 for the two
    // separate launches that move 20 % more bytes).  Values and arithmetic are those of the generic loop, bit for bit.
    typedef unsigned u32x4q __attribute__((ext_vector_type(4)));
    struct BwItem { u32x4q dh, g[4]; f32x4q dc[2], cp[2], cn[2]; };
    [[maybe_unused]] BwItem bwq[EPI == RNH_EPI_LSTM_BWD ? 2 : 1];
    [[maybe_unused]] bool bw_fast = false;
    [[maybe_unused]] auto bw_issue = [&](int r) {
        const int g = tid & 7;
        const char *dhp = reinterpret_cast<const char *>(P.bw_dh), *gp = reinterpret_cast<const char *>(P.bw_gates);
        const float *cnext = P.bw_c_next, *dcn = P.bw_dc_next ? P.bw_dc_next : cnext, *cprev = P.bw_c_prev ? P.bw_c_prev : cnext;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int y = min(y0 + k * MB + r, H - 1), x = min(x0 + ((tid >> 3) & (TW - 1)), W - 1);
            const long p = ((long)img * H + y) * W + x;
            const long o = p * 64 + g * 8, og = p * 256 + g * 8;
            const char *a_dh = dhp + o * 2, *a_g = gp + og * 2;
            const float *a_dc = dcn + o, *a_cp = cprev + o, *a_cn = cnext + o;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bwq[k].dh) : "v"(a_dh) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bwq[k].dc[0]) : "v"(a_dc) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(bwq[k].dc[1]) : "v"(a_dc) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bwq[k].cp[0]) : "v"(a_cp) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(bwq[k].cp[1]) : "v"(a_cp) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bwq[k].cn[0]) : "v"(a_cn) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(bwq[k].cn[1]) : "v"(a_cn) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bwq[k].g[0]) : "v"(a_g) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:128" : "=v"(bwq[k].g[1]) : "v"(a_g) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:256" : "=v"(bwq[k].g[2]) : "v"(a_g) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:384" : "=v"(bwq[k].g[3]) : "v"(a_g) : "memory");
        }
    };
    if constexpr (EPI == RNH_EPI_LSTM_BWD) {
        bw_fast = NCOLS == 128 && P.hd == 64 && P.dst[0].ncols == 64 && P.bw_dh_dtype == RNH_DT_BF16 && P.gates_dtype == RNH_DT_BF16 && !(RNH_EXP & 512);
    }
    float bv[M16 ? NB16 : NB];
    if constexpr (W8) {
        bv[0] = P.bias ? P.bias[nt * NCOLS + wave * 32 + l31] : 0.f;
    } else if constexpr (M16) {
#pragma unroll
        for (int j = 0; j < NB16; ++j) bv[j] = P.bias ? P.bias[nt * NCOLS + chalf * (NCOLS / 2) + j * 16 + l15] : 0.f;
    } else {
#pragma unroll
        for (int n = 0; n < NB; ++n) bv[n] = P.bias ? P.bias[nt * NCOLS + chalf * (NCOLS / 2) + n * 32 + l31] : 0.f;
    }
    auto park = [&](int m, float *ob) {                             // row block m of this wave -> pixels 32 ph .. 32 ph + 31 of the image
        if constexpr (W8) {                                         // tile rows m and MB + m of the wave's 32 columns -> pixels 0 .. 31 and 32 .. 63
            const int col = wave * 32 + l31;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int v = 0; v < 16; ++v) ob[(hh * TW + (v & 3) + 8 * (v >> 2) + 4 * kh) * G::OPITCH + col] = acc8[W8 ? hh * MB + m : 0][v] + bv[0];
        } else if constexpr (M16) {                                        // C / D of 16x16x32: column l & 15, rows 4 (l >> 4) + v
#pragma unroll
            for (int xh = 0; xh < 2; ++xh)
#pragma unroll
                for (int j = 0; j < NB16; ++j) {
                    const int col = chalf * (NCOLS / 2) + j * 16 + l15;
#pragma unroll
                    for (int v = 0; v < 4; ++v) ob[(ph * TW + xh * 16 + 4 * kg + v) * G::OPITCH + col] = acc16[M16 ? 2 * m + xh : 0][M16 ? j : 0][v] + bv[j];
                }
        } else {
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int col = chalf * (NCOLS / 2) + n * 32 + l31;
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    ob[(ph * TW + (v & 3) + 8 * (v >> 2) + 4 * kh) * G::OPITCH + col] = acc[(M16 || W8) ? 0 : m][(M16 || W8) ? 0 : n][v] + bv[n];
            }
        }
    };
    park(0, ot0);
    __syncthreads();
    BSTAMP(2);
    if constexpr (EPI == RNH_EPI_LSTM) {
#pragma unroll
        for (int q = 0; q < NIT; ++q) asm volatile("s_waitcnt vmcnt(0)" : "+v"(cpq[q][0]), "+v"(cpq[q][1]));
    }

    // per-thread constants of the items (the same in every round)
    [[maybe_unused]] bool live = false;
    [[maybe_unused]] void *dptr = nullptr;
    [[maybe_unused]] int ddt = 0, dacc = 0, ps_r = 1, ps_i = 0, ps_j = 0;
    [[maybe_unused]] long dimg = 0, dch = 0, dC = 0;                 // element = ((img' * H + y) * W' + x') * dC + dch with the strides below
    constexpr int G8 = NCOLS / 8;
    if constexpr (EPI == RNH_EPI_PS) {
        const int n0 = nt * NCOLS + (tid % G8) * 8, rr = P.ps_r, cq = P.ps_cq;
        live = n0 < cq * rr * rr;
        const int sub = n0 / cq;
        ps_r = rr, ps_i = sub / rr, ps_j = sub - (sub / rr) * rr;
        dptr = P.dst[0].ptr, ddt = P.dst[0].dtype, dC = cq, dch = n0 - sub * cq, dimg = img;
    } else if constexpr (EPI == RNH_EPI_STORE) {
        const int n0 = nt * NCOLS + (tid % G8) * 8;
        int seg = -1, cbase = 0;
        for (int d = 0; d < P.ndst; ++d) {
            if (seg < 0 && n0 < cbase + P.dst[d].ncols) seg = d;
            if (seg < 0) cbase += P.dst[d].ncols;
        }
        live = seg >= 0;
        const rnh_mdst_t &D = P.dst[live ? seg : 0];
        dptr = D.ptr, ddt = D.dtype, dacc = D.accumulate, dC = D.C, dch = D.c0 + (n0 - cbase), dimg = img + D.img_off;
    }

    // The four rounds.  `fast` (LSTM_BWD only) selects the gate-item form at COMPILE time inside the loop; the run-time choice between the two forms
    // is made once, around the whole loop - so that on the prefetching path no other path's code sits between an asm load and its wait (the compiler
    // reuses the registers of in-flight prefetches on a path where they are dead: tests/test_isa_guards.py follows the control flow to check that the
    // path that waits for them never does)
    auto rounds = [&](auto fast_) {
    constexpr bool FAST = decltype(fast_)::value;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const float *ot = ot0 + (r & 1) * OBUF;
        BSTAMP(40 + 4 * r);
        if constexpr (FAST) {
            if (r == 0) bw_issue(0);                                // (round 0's operands: under the park of round 1; the later rounds' a whole round ahead)
        }
        if (r + 1 < ROUNDS) park(r + 1, ot0 + ((r + 1) & 1) * OBUF);   // (its image was last read in round r - 1, in front of the last barrier)
        BSTAMP(41 + 4 * r);
        // pixel p of the image = tile row (p >> 5) * MB + r, column p & 31
        if constexpr (EPI == RNH_EPI_LSTM) {
            // column = gate * 32 + j of the tile's 32 hidden channels nt * 32 + j (plans.lstm_colmap).  One item per thread and round:
            // pixel tid >> 2, channels 8 (tid & 3) ..
            const int hd = P.hd, j0 = (tid & 3) * 8, hc = nt * 32 + j0;
            const float *cprev = P.c_prev;
            float *cout = P.c_out;
            void *hout = P.h_out, *gout = P.gates_out;
            const int hdt = P.h_dtype, gdt = P.gates_dtype;
            const int px = tid >> 2;
            const int y = y0 + (px >> 5) * MB + r, x = x0 + (px & (TW - 1));
            if (hc < hd && y < H && x < W) {
                const float *o = ot + px * G::OPITCH + j0;
                const long pe = ((long)img * H + y) * W + x;
                float cp[8], cn[8], hn[8], gi[8], gf[8], go[8], gg[8];
                const f32x4q ca = cpq[r][0], cb = cpq[r][1];
                cp[0] = ca.x; cp[1] = ca.y; cp[2] = ca.z; cp[3] = ca.w; cp[4] = cb.x; cp[5] = cb.y; cp[6] = cb.z; cp[7] = cb.w;
                if (!cprev) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) cp[e] = 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    gi[e] = b_sigmoid(o[e]);
                    gf[e] = b_sigmoid(o[32 + e]);
                    go[e] = b_sigmoid(o[64 + e]);
                    gg[e] = b_tanh_fast(o[96 + e]);
                    cn[e] = __builtin_fmaf(gf[e], cp[e], gi[e] * gg[e]);            // (spelled out, so that every build of the cell - csrc/experiments/conv_bf16p.hip too - rounds the same way)
                    hn[e] = go[e] * b_tanh_fast(cn[e]);
                }
                store8(cout, RNH_DT_F32, pe * hd + hc, cn);
                store8(hout, hdt, pe * hd + hc, hn);
                if (gout) {
                    store8(gout, gdt, pe * 4 * hd + hc, gi);
                    store8(gout, gdt, pe * 4 * hd + hd + hc, gf);
                    store8(gout, gdt, pe * 4 * hd + 2 * hd + hc, go);
                    store8(gout, gdt, pe * 4 * hd + 3 * hd + hc, gg);
                }
            }
        } else if constexpr (EPI == RNH_EPI_LSTM_BWD) {
            // Data gradient of a ConvLSTM cell + gate backward of the frame its chain processes next (include/refinenet_hip.h).  The parked
            // image holds the input gradient in columns [0, ncx) and dh_rec, the recurrent part of that frame's dh, in the next hd columns.
            const rnh_mdst_t &D = P.dst[0];
            const int ncx = D.ncols, ncx8 = ncx >> 3, hd = P.hd, hd8 = hd >> 3;
            if constexpr (FAST) {
                // the gate items of this round from the operands requested a round ago (bw_issue), then the requests of the next round, then the
                // input-gradient items
                const float *dcn = P.bw_dc_next, *cprev = P.bw_c_prev;
                void *dgp = P.bw_dgates;
                float *dcprev = P.bw_dc_prev;
                const int dgdt = P.bw_dgates_dtype, rdt = P.bw_rec_dtype;
                constexpr int gdt = RNH_DT_BF16;                    // (bw_fast)
                const int g = tid & 7;
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    asm volatile("s_waitcnt vmcnt(0)"
                                 : "+v"(bwq[k].dh), "+v"(bwq[k].g[0]), "+v"(bwq[k].g[1]), "+v"(bwq[k].g[2]), "+v"(bwq[k].g[3]), "+v"(bwq[k].dc[0]), "+v"(bwq[k].dc[1]),
                                   "+v"(bwq[k].cp[0]), "+v"(bwq[k].cp[1]), "+v"(bwq[k].cn[0]), "+v"(bwq[k].cn[1]));
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int px = (tid >> 3) + 32 * k;
                    const int y = y0 + k * MB + r, x = x0 + (px & (TW - 1));
                    if (y >= H || x >= W) continue;
                    const long p = ((long)img * H + y) * W + x;
                    const long o = p * 64 + g * 8, og = p * 256 + g * 8;
                    float vdh[8], vdc[8], vcp[8], vcn[8], gi[8], gf[8], go[8], gg[8];
                    unpack8(__builtin_bit_cast(uint4, bwq[k].dh), vdh);
                    unpack8(__builtin_bit_cast(uint4, bwq[k].g[0]), gi);
                    unpack8(__builtin_bit_cast(uint4, bwq[k].g[1]), gf);
                    unpack8(__builtin_bit_cast(uint4, bwq[k].g[2]), go);
                    unpack8(__builtin_bit_cast(uint4, bwq[k].g[3]), gg);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vdc[e] = bwq[k].dc[0][e], vdc[4 + e] = bwq[k].dc[1][e];
                        vcp[e] = bwq[k].cp[0][e], vcp[4 + e] = bwq[k].cp[1][e];
                        vcn[e] = bwq[k].cn[0][e], vcn[4 + e] = bwq[k].cn[1][e];
                    }
                    const float *rec = ot + px * G::OPITCH + ncx + g * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = rec[e];
                        if (rdt == RNH_DT_BF16) t = (float)(__bf16)t;
                        vdh[e] += t;
                    }
                    float di[8], df[8], dgo[8], dg[8], dcp[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float th = gdt == RNH_DT_BF16 ? b_tanh_fast(vcn[e]) : tanhf(vcn[e]);   // (as the forward cell's h' = o tanh(c'); gates_bwd_m_kernel alike)
                        const float d_o = vdh[e] * th;
                        const float dct = (dcn ? vdc[e] : 0.f) + vdh[e] * go[e] * (1.f - th * th);
                        di[e] = dct * gg[e] * gi[e] * (1.f - gi[e]);
                        df[e] = dct * (cprev ? vcp[e] : 0.f) * gf[e] * (1.f - gf[e]);
                        dgo[e] = d_o * go[e] * (1.f - go[e]);
                        dg[e] = dct * gi[e] * (1.f - gg[e] * gg[e]);
                        dcp[e] = dct * gf[e];
                    }
                    store8(dgp, dgdt, og, di);
                    store8(dgp, dgdt, og + 64, df);
                    store8(dgp, dgdt, og + 128, dgo);
                    store8(dgp, dgdt, og + 192, dg);
                    if (dcprev) store8(dcprev, RNH_DT_F32, o, dcp);
                }
                if (r + 1 < ROUNDS) bw_issue(r + 1);
#pragma unroll
                for (int k = 0; k < 2; ++k) {                       // items (pixel, 8 columns of the input gradient): the same pixels and column groups
                    const int px = (tid >> 3) + 32 * k;
                    const int y = y0 + k * MB + r, x = x0 + (px & (TW - 1));
                    if (y >= H || x >= W) continue;
                    const float *o = ot + px * G::OPITCH + g * 8;
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = o[e];
                    const long e = ((((long)img + D.img_off) * H + y) * W + x) * D.C + D.c0 + g * 8;
                    if (D.accumulate) {
                        float old[8];
                        load8(D.ptr, D.dtype, e, old);
#pragma unroll
                        for (int q = 0; q < 8; ++q) f[q] += old[q];
                    }
                    store8(D.ptr, D.dtype, e, f);
                }
            } else {
            for (int it = tid; it < PXR * ncx8; it += 256) {        // items (pixel, 8 columns of the input gradient)
                const int px = it / ncx8, c8 = it - px * ncx8;
                const int y = y0 + (px >> 5) * MB + r, x = x0 + (px & (TW - 1));
                if (y >= H || x >= W) continue;
                const float *o = ot + px * G::OPITCH + c8 * 8;
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = o[e];
                const long e = ((((long)img + D.img_off) * H + y) * W + x) * D.C + D.c0 + c8 * 8;
                if (D.accumulate) {
                    float old[8];
                    load8(D.ptr, D.dtype, e, old);
#pragma unroll
                    for (int q = 0; q < 8; ++q) f[q] += old[q];
                }
                store8(D.ptr, D.dtype, e, f);
            }
            // items (pixel, 8 hidden channels): the body of gates_bwd_m_kernel (mixed_kernels.hip), expression by expression, with
            // dh2 := dh_rec out of LDS, rounded to the element type the unfused path would have stored it in
            const void *dhp = P.bw_dh, *gp = P.bw_gates;
            const float *dcn = P.bw_dc_next, *cprev = P.bw_c_prev, *cnext = P.bw_c_next;
            void *dgp = P.bw_dgates;
            float *dcprev = P.bw_dc_prev;
            const int hdt = P.bw_dh_dtype, gdt = P.gates_dtype, dgdt = P.bw_dgates_dtype, rdt = P.bw_rec_dtype;
            for (int it = tid; it < PXR * hd8; it += 256) {
                const int px = it / hd8, g = it - px * hd8;
                const int y = y0 + (px >> 5) * MB + r, x = x0 + (px & (TW - 1));
                if (y >= H || x >= W) continue;
                const long p = ((long)img * H + y) * W + x;
                const long o = p * hd + g * 8, og = p * 4 * hd + g * 8;
                float vdh[8], t[8], vdc[8], vcp[8], vcn[8], gi[8], gf[8], go[8], gg[8];
                load8(dhp, hdt, o, vdh);
                const float *rec = ot + px * G::OPITCH + ncx + g * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    t[e] = rec[e];
                    if (rdt == RNH_DT_BF16) t[e] = (float)(__bf16)t[e];
                    vdh[e] += t[e];
                }
                if (dcn) load8(dcn, RNH_DT_F32, o, vdc);
                if (cprev) load8(cprev, RNH_DT_F32, o, vcp);
                load8(cnext, RNH_DT_F32, o, vcn);
                load8(gp, gdt, og, gi);
                load8(gp, gdt, og + hd, gf);
                load8(gp, gdt, og + 2 * hd, go);
                load8(gp, gdt, og + 3 * hd, gg);
                float di[8], df[8], dgo[8], dg[8], dcp[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float th = gdt == RNH_DT_BF16 ? b_tanh_fast(vcn[e]) : tanhf(vcn[e]);   // (as the forward cell's h' = o tanh(c'); gates_bwd_m_kernel alike)
                    const float d_o = vdh[e] * th;
                    const float dct = (dcn ? vdc[e] : 0.f) + vdh[e] * go[e] * (1.f - th * th);
                    di[e] = dct * gg[e] * gi[e] * (1.f - gi[e]);
                    df[e] = dct * (cprev ? vcp[e] : 0.f) * gf[e] * (1.f - gf[e]);
                    dgo[e] = d_o * go[e] * (1.f - go[e]);
                    dg[e] = dct * gi[e] * (1.f - gg[e] * gg[e]);
                    dcp[e] = dct * gf[e];
                }
                store8(dgp, dgdt, og, di);
                store8(dgp, dgdt, og + hd, df);
                store8(dgp, dgdt, og + 2 * hd, dgo);
                store8(dgp, dgdt, og + 3 * hd, dg);
                if (dcprev) store8(dcprev, RNH_DT_F32, o, dcp);
            }
            }
        } else {
            // a thread's 8 columns are the same in every item (256 % G8 == 0): destination segment / sub-pixel found once, above
            if (live) {
                const int c8 = tid % G8;
#pragma unroll
                for (int px = tid / G8; px < PXR; px += 256 / G8) {
                    const int y = y0 + (px >> 5) * MB + r, x = x0 + (px & (TW - 1));
                    if (y >= H || x >= W) continue;
                    const float *o = ot + px * G::OPITCH + c8 * 8;
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = o[e];
                    const long e = ((dimg * H + y) * ps_r + ps_i) * ((long)W * ps_r) * dC + ((long)x * ps_r + ps_j) * dC + dch;
                    if (dacc) {
                        float old[8];
                        load8(dptr, ddt, e, old);
#pragma unroll
                        for (int q = 0; q < 8; ++q) f[q] += old[q];
                    }
                    store8(dptr, ddt, e, f);
                }
            }
        }
        BSTAMP(42 + 4 * r);
        if (r + 1 < ROUNDS) __syncthreads();                        // image (r + 1) & 1 is written, image r & 1 is read
        BSTAMP(43 + 4 * r);
    }
    };
    if constexpr (EPI == RNH_EPI_LSTM_BWD) {
        if (bw_fast) rounds(std::true_type{});
        else rounds(std::false_type{});
    } else {
        rounds(std::false_type{});
    }
    BSTAMP(3);
    WGTRACE(2);
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
#ifdef RNH_WITH_PERSISTENT
int rnh_conv_bf16_persistent(const rnh_conv_bf16_args_t &a, int TYn, int TXn, int NT, hipStream_t st);   // csrc/experiments/conv_bf16p.hip (not in the product build)
#else
static inline int rnh_conv_bf16_persistent(const rnh_conv_bf16_args_t &, int, int, int, hipStream_t) { return 0; }
#endif

#ifdef RNH_STAMPS
extern "C" int rnh_debug_bf16_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bf16_stamps), sizeof(g_bf16_stamps));
}
extern "C" int rnh_debug_bf16_wgtrace(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bf16_wgtrace), sizeof(g_bf16_wgtrace));
}
#endif

extern "C" int rnh_pack_weights_bf16(const float *w, const float *bias, void *wp, float *biasp, const int32_t *kbase, const int32_t *knv,
                                     const int32_t *ktap, const int32_t *kcoff, const int32_t *colmap, int nk, int Npad, int Cout, int Cin,
                                     int ntaps, int kstride, int transposed, void *stream) {
    if (!w || !wp || !kbase || !knv || !ktap || !colmap || nk < 1 || Npad < 1 || Cout < 1 || Cin < 1 || kstride < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_pack_weights_bf16: bad arguments");
    if (ntaps != 9 && ntaps != 1) RNH_FAIL(RNH_E_RANGE, "rnh_pack_weights_bf16: ntaps must be 9 or 1");
    if (Npad % 64) RNH_FAIL(RNH_E_RANGE, "rnh_pack_weights_bf16: Npad must be a multiple of 64");
    hipLaunchKernelGGL(pack_bf16_kernel<false>, dim3(bgrid_for((long)nk * Npad * 16 + Npad)), dim3(256), 0, (hipStream_t)stream, w, bias,
                       (unsigned short *)wp, biasp, kbase, knv, ktap, kcoff, colmap, nk, Npad, Cout, Cin, ntaps, kstride, transposed);
    RNH_CHECK_LAUNCH("rnh_pack_weights_bf16");
    return 0;
}

extern "C" int rnh_pack_weights_f16(const float *w, const float *bias, void *wp, float *biasp, const int32_t *kbase, const int32_t *knv,
                                    const int32_t *ktap, const int32_t *kcoff, const int32_t *colmap, int nk, int Npad, int Cout, int Cin,
                                    int ntaps, int kstride, int transposed, void *stream) {
    if (!w || !wp || !kbase || !knv || !ktap || !colmap || nk < 1 || Npad < 1 || Cout < 1 || Cin < 1 || kstride < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_pack_weights_f16: bad arguments");
    if (ntaps != 9) RNH_FAIL(RNH_E_RANGE, "rnh_pack_weights_f16: the f16 form serves 3x3 convolutions");
    if (Npad % 64) RNH_FAIL(RNH_E_RANGE, "rnh_pack_weights_f16: Npad must be a multiple of 64");
    hipLaunchKernelGGL(pack_bf16_kernel<true>, dim3(bgrid_for((long)nk * Npad * 16 + Npad)), dim3(256), 0, (hipStream_t)stream, w, bias,
                       (unsigned short *)wp, biasp, kbase, knv, ktap, kcoff, colmap, nk, Npad, Cout, Cin, ntaps, kstride, transposed);
    RNH_CHECK_LAUNCH("rnh_pack_weights_f16");
    return 0;
}

namespace {

struct BfGeo {                  // what the launch of a validated call needs
    int ncols, TYn, TXn, NT;
    long blocks;
    bool k32;
};

// every check of a call's arguments (who = the entry point's name, for the messages)
int conv_bf16_check(const rnh_conv_bf16_args_t &a, BfGeo &g, const char *who) {
    if (a.nsrc < 1 || a.nsrc > RNH_MAX_SRC || a.B < 1 || a.H < 1 || a.W < 1 || !a.wp) RNH_FAIL(RNH_E_ARG, "%s: bad arguments", who);
    if (a.ntaps != 9 && a.ntaps != 1) RNH_FAIL(RNH_E_RANGE, "%s: ntaps must be 9 or 1", who);
    if (a.Npad < 64 || a.Npad % 64) RNH_FAIL(RNH_E_RANGE, "%s: Npad must be a multiple of 64", who);
    int chunks = 0;
    for (int i = 0; i < a.nsrc; ++i) {
        if (int rc = check_msrc(a.src[i], who)) return rc;
        if (a.src[i].scale != a.src[0].scale) RNH_FAIL(RNH_E_RANGE, "%s: one scale for all sources", who);
        chunks += (a.src[i].nch + 15) / 16;
    }
    if (chunks != a.nchunks) RNH_FAIL(RNH_E_ARG, "%s: nchunks = %d but the sources hold %d chunks of 16 channels", who, a.nchunks, chunks);
    if ((long)a.B * a.H * a.W * a.src[0].scale * a.src[0].scale >= (1L << 31)) RNH_FAIL(RNH_E_RANGE, "%s: too many pixels", who);
    g.ncols = a.Npad % 128 ? 64 : 128;
    constexpr int TH = 8;
    g.TYn = (a.H + TH - 1) / TH, g.TXn = (a.W + TW - 1) / TW, g.NT = a.Npad / g.ncols;
    g.blocks = (long)a.B * g.TYn * g.TXn * g.NT;
    if (g.blocks >= (1L << 30)) RNH_FAIL(RNH_E_RANGE, "%s: grid too large", who);
    // 32-channel chunks (deeper weight-fragment ring, half the barriers) where every source is bf16 with a multiple of 32 channels
    g.k32 = a.ntaps == 9 && !(getenv("RNH_BF16_KC") && getenv("RNH_BF16_KC")[0] == '1');
    for (int i = 0; i < a.nsrc; ++i) g.k32 = g.k32 && a.src[i].dtype == RNH_DT_BF16 && a.src[i].nch % 32 == 0;
    const int NT = g.NT;
    if (a.wp_f16 != 0 && a.wp_f16 != 1) RNH_FAIL(RNH_E_ARG, "%s: wp_f16 must be 0 or 1", who);
    if (a.wp_f16 && !(a.epilogue == RNH_EPI_PS && g.k32))
        RNH_FAIL(RNH_E_RANGE, "%s: f16 weights serve the pixel-shuffle epilogue over bf16 sources of 32-channel multiples", who);
    switch (a.epilogue) {
        case RNH_EPI_STORE:
            if (a.ndst < 1 || a.ndst > RNH_MAX_DST) RNH_FAIL(RNH_E_ARG, "%s: bad destination count", who);
            for (int d = 0; d < a.ndst; ++d) {
                const rnh_mdst_t &D = a.dst[d];
                if (!D.ptr || D.ncols < 1 || (D.dtype != RNH_DT_F32 && D.dtype != RNH_DT_BF16)) RNH_FAIL(RNH_E_ARG, "%s: bad destination %d", who, d);
                if ((D.C & 7) || (D.c0 & 7) || (D.ncols & 7)) RNH_FAIL(RNH_E_ALIGN, "%s: destination channels must be multiples of 8", who);
            }
            break;
        case RNH_EPI_PS:
            if (a.ntaps != 9) RNH_FAIL(RNH_E_RANGE, "%s: the pixel-shuffle epilogue serves 3x3 convolutions", who);
            if (a.ndst != 1 || !a.dst[0].ptr || a.ps_r < 1 || a.ps_cq < 8 || (a.ps_cq & 7) || a.ps_cq * a.ps_r * a.ps_r > a.Npad)
                RNH_FAIL(RNH_E_ARG, "%s: bad pixel-shuffle destination", who);
            break;
        case RNH_EPI_LSTM_BWD: {
            if (a.ntaps != 9) RNH_FAIL(RNH_E_RANGE, "%s: the LSTM-backward epilogue serves 3x3 convolutions", who);
            const rnh_mdst_t &D = a.dst[0];
            if (a.ndst != 1 || !D.ptr || D.ncols < 8 || (D.dtype != RNH_DT_F32 && D.dtype != RNH_DT_BF16) || (D.C & 7) || (D.c0 & 7) || (D.ncols & 7))
                RNH_FAIL(RNH_E_ARG, "%s: the LSTM-backward epilogue stores the input gradient to dst[0] (channels in multiples of 8)", who);
            if (a.hd < 8 || (a.hd & 7) || NT != 1 || D.ncols + a.hd > a.Npad)
                RNH_FAIL(RNH_E_RANGE, "%s: LSTM-backward epilogue: input-gradient + hd columns must fit ONE column tile (Npad %d)", who, a.Npad);
            if (!a.bw_dh || !a.bw_gates || !a.bw_c_next || !a.bw_dgates) RNH_FAIL(RNH_E_ARG, "%s: LSTM-backward epilogue needs bw_dh, bw_gates, bw_c_next, bw_dgates", who);
            if ((a.bw_dh_dtype != RNH_DT_F32 && a.bw_dh_dtype != RNH_DT_BF16) || (a.bw_dgates_dtype != RNH_DT_F32 && a.bw_dgates_dtype != RNH_DT_BF16) ||
                (a.bw_rec_dtype != RNH_DT_F32 && a.bw_rec_dtype != RNH_DT_BF16) ||
                (a.gates_dtype != RNH_DT_F32 && a.gates_dtype != RNH_DT_BF16))
                RNH_FAIL(RNH_E_ARG, "%s: LSTM-backward epilogue: bad element type", who);
            break;
        }
        case RNH_EPI_LSTM:
            if (a.ntaps != 9) RNH_FAIL(RNH_E_RANGE, "%s: the LSTM epilogue serves 3x3 convolutions", who);
            if (!a.h_out || !a.c_out || a.hd < 8 || (a.hd & 7) || !a.bias) RNH_FAIL(RNH_E_ARG, "%s: LSTM epilogue needs h_out, c_out, hd %% 8 == 0, bias", who);
            if (a.Npad != 128 * ((a.hd + 31) / 32)) RNH_FAIL(RNH_E_RANGE, "%s: LSTM column layout (plans.lstm_colmap)", who);
            break;
        default:
            RNH_FAIL(RNH_E_RANGE, "%s: epilogue %d not available", who, a.epilogue);
    }
    return 0;
}

// launch a validated call (pair: a validated second call of the same kernel instantiation and geometry rides in the same launch)
int conv_bf16_launch(const rnh_conv_bf16_args_t &a, const rnh_conv_bf16_args_t &b, bool pair, const BfGeo &g, hipStream_t st, const char *who) {
    const int ncols = g.ncols, TYn = g.TYn, TXn = g.TXn, NT = g.NT, nA = (int)g.blocks;
    const bool k32 = g.k32;
    const dim3 grid((unsigned)(pair ? 2 * g.blocks : g.blocks)), block(256);
    rnh_conv_bf16_pair_t pp;
    pp.call[0] = a;
    pp.call[1] = b;
#define RNH_LAUNCH(EPI, NC, NTP) hipLaunchKernelGGL((conv_bf16d_kernel<EPI, NC, NTP>), grid, block, 0, st, pp, nA, TYn, TXn, NT)
#define RNH_LAUNCH9(EPI, NC)                                                                                                 \
    do {                                                                                                                     \
        if (k32) hipLaunchKernelGGL((conv_bf16d_kernel<EPI, NC, 9, 32>), grid, block, 0, st, pp, nA, TYn, TXn, NT);        \
        else hipLaunchKernelGGL((conv_bf16d_kernel<EPI, NC, 9, 16>), grid, block, 0, st, pp, nA, TYn, TXn, NT);            \
    } while (0)
    switch (a.epilogue) {
        case RNH_EPI_STORE:
            if (a.ntaps == 9) {
                if (!pair && ncols == 128 && k32 && rnh_conv_bf16_persistent(a, TYn, TXn, NT, st)) break;
                if (ncols == 128) RNH_LAUNCH9(RNH_EPI_STORE, 128);
                else RNH_LAUNCH9(RNH_EPI_STORE, 64);
            } else {
                if (ncols == 128) RNH_LAUNCH(RNH_EPI_STORE, 128, 1);
                else RNH_LAUNCH(RNH_EPI_STORE, 64, 1);
            }
            break;
        case RNH_EPI_PS:
            if (a.wp_f16) {                                           // (conv_bf16_check: 32-channel chunks)
                if (ncols == 128) hipLaunchKernelGGL((conv_bf16d_kernel<RNH_EPI_PS, 128, 9, 32, true>), grid, block, 0, st, pp, nA, TYn, TXn, NT);
                else hipLaunchKernelGGL((conv_bf16d_kernel<RNH_EPI_PS, 64, 9, 32, true>), grid, block, 0, st, pp, nA, TYn, TXn, NT);
            } else if (ncols == 128) RNH_LAUNCH9(RNH_EPI_PS, 128);
            else RNH_LAUNCH9(RNH_EPI_PS, 64);
            break;
        case RNH_EPI_LSTM_BWD:
            if (ncols == 128) RNH_LAUNCH9(RNH_EPI_LSTM_BWD, 128);
            else RNH_LAUNCH9(RNH_EPI_LSTM_BWD, 64);
            break;
        default:            // RNH_EPI_LSTM (conv_bf16_check has refused everything else)
            if (!pair && k32 && rnh_conv_bf16_persistent(a, TYn, TXn, NT, st)) break;
            RNH_LAUNCH9(RNH_EPI_LSTM, 128);
            break;
    }
#undef RNH_LAUNCH9
#undef RNH_LAUNCH
    RNH_CHECK_LAUNCH(who);
    return 0;
}

}  // namespace

extern "C" int rnh_conv_bf16(const rnh_conv_bf16_args_t *args, void *stream) {
    if (!args) RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: null args");
    BfGeo g;
    if (int rc = conv_bf16_check(*args, g, "rnh_conv_bf16")) return rc;
    return conv_bf16_launch(*args, *args, false, g, (hipStream_t)stream, "rnh_conv_bf16");
}

extern "C" int rnh_conv_bf16_pair(const rnh_conv_bf16_args_t *args_a, const rnh_conv_bf16_args_t *args_b, void *stream) {
    if (!args_a || !args_b) RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16_pair: null args");
    BfGeo ga, gb;
    if (int rc = conv_bf16_check(*args_a, ga, "rnh_conv_bf16_pair (first call)")) return rc;
    if (int rc = conv_bf16_check(*args_b, gb, "rnh_conv_bf16_pair (second call)")) return rc;
    const rnh_conv_bf16_args_t &a = *args_a, &b = *args_b;
    if (a.B != b.B || a.H != b.H || a.W != b.W || a.Npad != b.Npad || a.ntaps != b.ntaps || a.epilogue != b.epilogue || ga.k32 != gb.k32 ||
        a.src[0].scale != b.src[0].scale || a.wp_f16 != b.wp_f16)
        RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16_pair: the two calls must agree in B, H, W, Npad, ntaps, epilogue, source scale and chunk size");
    return conv_bf16_launch(a, b, true, ga, (hipStream_t)stream, "rnh_conv_bf16_pair");
}
