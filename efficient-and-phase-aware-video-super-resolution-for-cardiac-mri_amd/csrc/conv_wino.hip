// 3x3 convolution (padding 1) in Winograd form F(2x2, 3x3) on fp32 MFMA for gfx950, two workgroups per CU - rnh_conv_wino.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 independent GEMMs (one per position xi of the 4x4 transform domain) of [tiles x C] x [C x N].  the round-1 kernel (git history: csrc/conv_wino.hip before round 2's last commits)
// gave one wave all 16 positions of 32 tiles x 32 columns: 256 accumulators + 256 working registers = the whole register
// file of a SIMD, ONE wave per SIMD, so every cycle that wave spent outside the MFMA stream (set-up, staging transform,
// barrier skew, the gate epilogue, stores) was a cycle of matrix-core idle time: 0.56 of the fp32 peak.
// Here a wave owns HALF the transform domain (the positions xi = 4 i + j with j in {2h, 2h+1}: 8 of the 16) of 32 tiles x 32
// columns - 128 accumulators, 256 registers in all - and a workgroup is 2 halves x 2 column groups = 32 tiles x 64 columns:
// two workgroups are resident per CU, every SIMD holds one wave of each, and part of what one of them does beside its MFMAs
// hides behind the MFMAs of the other.  The price:
//   * the output transform needs all four j: the two halves exchange partial 2x2 outputs through LDS (each wave keeps the
//     tiles of 8 accumulator registers and sends the other 8 to its partner: 8 ds_write_b128 + 8 ds_read_b128 per lane);
//   * the staged input transform of 32 tiles feeds 64 columns instead of 128 (LDS and L2 traffic per MFMA as before,
//     staging loads and transform adds per MFMA doubled for convolutions wider than 64 columns).
// Measured on the ConvLSTM cell (N = 8, 128 x 128, same box): 0.437 -> 0.412 ms, its data gradient 0.351 -> 0.320 ms;
// MFMA pipe busy 0.55 -> 0.62 / 0.67 -> 0.76 (rocprofv3 PMC); the training step 371 -> 355 ms.
// The kernel has a second geometry (template argument NW = 8, `tile` = RNH_WINO_COLS128): 8 waves = 2 halves x 4 column
// groups = 32 tiles x 128 columns, 32-channel chunks (every thread still stages one (tile, channel pair) per chunk), one
// workgroup per CU in 139 KB of LDS.  The staged input transform then feeds twice the columns - half the staging work per
// MFMA - and both waves of a SIMD are always in the same phase: the main loop runs at 0.90 of the MFMA rate (73 k cycles for
// 65.5 k of MFMA per block), but nothing hides set-up (8 k) and epilogue (8 k).  Where the column count is a multiple of 128
// and every source one of 32 channels (the ConvLSTM cell and its data gradient, refine conv1, the PixelShuffle convolutions)
// it wins: cell 0.408 -> 0.387 ms, data gradient 0.325 -> 0.304 ms, refine conv1 11.0 -> 10.3 ms, step 355.6 -> 341.8 ms on
// one box; the plans (hipvsr/plans.py) use it there and the 64-column geometry everywhere else.
//
// What the experiments on this kernel say about the machine (tools/wino_stamps.py; the three shelved kernels live in the
// history only: `git show 6671234:tools/experiments/<file>`):
//   * issue arbitration between the two waves of a SIMD is STRICT, not round-robin: the wave in the lower slot (the
//     workgroup that arrived first) wins whenever it has an instruction ready; its blocks take ~90 k cycles, the other
//     workgroup's ~120 k (HWMAP=1 tools/wino_stamps.py; s_setprio on the second workgroup reverses it).  The hardware's
//     workgroup dispatcher evens that out - a freed slot gets the next block - which is why a PERSISTENT variant of this
//     kernel (conv_wino2_persistent.hip: static block lists, the first chunk of the next block staged
//     under the last chunk of this one) was 7 % slower: the favoured workgroup finishes its list early and the CU runs
//     half empty at the end;
//   * a vector instruction of wave B is served about once per MFMA of wave A (34 cycles per instruction measured for a
//     pure producer wave beside a pure MFMA wave, conv_wino3_specialised_waves.hip): VALU work does not
//     run "under" the fp32 MFMAs of the other wave, it interleaves with them, and inside one wave every non-MFMA
//     instruction costs 8-10 cycles of matrix-core time.  With ~1 500 such instructions per 512 MFMAs of a wave
//     (operand loads 450, staging 650, epilogue 470) the ceiling of this formulation is about 0.70 of the fp32 peak;
//   * wave priorities do not help either: s_setprio 3 outside the main loop (set-up, first chunk, epilogue) and 0 inside it
//     - so that a workgroup's short non-MFMA phases are served at once - left both the ConvLSTM cell and its data gradient
//     where they were (0.404 against 0.398 ms, 0.322 against 0.322 ms);
//   * persistent workgroups for the 8-wave geometry (one per CU: no partner to be unfair to) did not pay either
//     (conv_wino_persistent_8wave.hip: data gradient 0.299 against 0.301 ms).
// Operands (64-column geometry; the 8-wave one below doubles chunk and columns): the input transform B^T d B of a 16-channel
// chunk is computed once per workgroup (thread =
// (tile, channel pair): 16 raw 8-byte buffer loads - out-of-image lanes carry offset -1 and read the zero padding - 32
// packed adds, 8 16-byte LDS writes into [xi / 2][tile][36]), the weights arrive pre-transformed from
// rnh_wino_pack_weights as U[step][xi / 2][n][lane half][xi & 1][2]: one buffer_load_dwordx4 and one ds_read_b128 feed
// four MFMAs.  With the ConvLSTM column order plans.lstm_colmap64 a workgroup's 64 columns are the 4 gates of 16 hidden
// channels: after the exchange every wave activates 2 gates of 16 tiles, the gates meet in LDS and the 256 threads finish
// (tile, pixel, channel) items (c' = f c + i g, h' = o tanh c').
//
// Same operand conventions as rnh_conv_igemm (rnh_conv_args_t: multi-source K without concatenation, destination
// segments, packed bias, pixel-unshuffled sources of one common scale).  Epilogues: STORE, PS, LSTM.
#include "rnh_common.h"
#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4w __attribute__((ext_vector_type(4)));

// v_exp_f32 / v_rcp_f32 (1 ulp each) and no branch
__device__ __forceinline__ float h_tanh(float x) {
    const float ax = fabsf(x);
    const float big = __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * ax)), 1.f);
    const float small = ax * __builtin_fmaf(-0.33333334f * ax, ax, 1.f);      // |x| < 0.04: the exp form cancels
    const float t = ax < 0.04f ? small : big;
    return copysignf(t, x);
}

// the buffer descriptor (base + 2 GiB window, raw buffer) as a plain SGPR quadruple for the asm loads
__device__ __forceinline__ i32x4 hdesc(const float *p) {
    const unsigned long long u = (unsigned long long)p;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    d[1] = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
    d[2] = 0x7fffffff;
    d[3] = 0x00020000;
    return d;
}

// U[s][xi / 2][n][kh][xi & 1][c] = (G g G^T)[xi] for input channel kbase[s] + (2 kh + c) * kstride and output column n
__global__ void wino_pack_kernel(const float *w, const float *bias, float *wp, float *biasp, const int *kbase, const int *knv,
                                 const int *kcoff, const int *colmap, int ns, int Npad, int Cout, int Cin, int kstride, int transposed) {
    const long total = (long)ns * 16 * Npad * 4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total + Npad; e += (long)gridDim.x * blockDim.x) {
        if (e >= total) {
            const int n = (int)(e - total);
            if (biasp) biasp[n] = (bias && !transposed && colmap[n] >= 0) ? bias[colmap[n]] : 0.f;
            continue;
        }
        // e = ((((s * 8 + pair) * Npad + n) * 2 + kh) * 2 + odd) * 2 + c:  xi = 2 pair + odd, channel q = 2 kh + c of the step
        const int c = (int)(e & 1), odd = (int)((e >> 1) & 1), kh = (int)((e >> 2) & 1), n = (int)((e >> 3) % Npad);
        const int pr = (int)((e / (8 * (long)Npad)) & 7), s = (int)(e / (64 * (long)Npad));
        const int q = 2 * kh + c, xi = 2 * pr + odd;
        const int col = colmap[n];
        float v = 0.f;
        if (col >= 0 && q < knv[s]) {
            const int k = kbase[s] + q * kstride, c = col + (kcoff ? kcoff[s] : 0);
            const float *g = transposed ? w + ((long)k * Cin + c) * 9 : w + ((long)c * Cin + k) * 9;
            const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
            const int xy = xi >> 2, xx = xi & 3;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    const int t = a * 3 + b;
                    v += G[xy][a] * G[xx][b] * g[transposed ? 8 - t : t];
                }
        }
        wp[e] = v;
    }
}

#ifdef RNH_STAMPS
__device__ unsigned long long g_wino_stamps[8];
__device__ unsigned long long g_wino_hw[4096 * 3];          // per block: HW_ID, start, end
#ifndef RNH_STAMP_BLOCK
#define RNH_STAMP_BLOCK 0
#endif
#define HSTAMP(i) do { if (blockIdx.x == RNH_STAMP_BLOCK && threadIdx.x == 0) g_wino_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define HSTAMP(i)
#endif

constexpr int H_TILES = 32;
// Geometry of a workgroup of NW waves (2 halves of the transform domain x NW / 2 column groups of 32): NW = 4: 32 tiles x 64
// columns, 16-channel chunks, two workgroups per CU; NW = 8: 32 tiles x 128 columns, 32-channel chunks, one workgroup per CU
// (the staged input transform feeds twice the columns: half the staging work per MFMA; both waves of a SIMD are in the same
// phase, nothing hides set-up and epilogue).  Every thread stages one (tile, channel pair) per chunk either way.
template <int NW>
struct WinoGeo {
    static constexpr int CG = NW / 2;                           // column groups (waves per half)
    static constexpr int CPC = 2 * NW, CH = 2 * CPC, SPC = CH / 4;   // channel pairs / channels / 4-channel steps per chunk
    static constexpr int CHS = 4 * CPC + 4;                      // LDS row of one (xi pair, tile): [channel pair][xi & 1][2] + 4 pad
    static constexpr int BUF = 8 * H_TILES * CHS;                // floats per staging buffer
    static constexpr int PART = NW * 8 * 64 * 4;                 // floats of the partial-output exchange: [wave][entry][lane][4]
    static constexpr int CW = 8 * CG;                            // hidden channels of a ConvLSTM block (its 32 * CG columns = 4 gates x CW)
    // gate exchange [gate][tile][pixel][CW]: strides that keep the lane groups of a wave's store in different bank ranges
    static constexpr int TS = 4 * CW + 8, GS = 32 * TS + 16;
    static_assert(PART + 4 * GS <= 2 * BUF, "the epilogue's exchange areas live in the staging buffers");
};

// PP, nA: ONE launch may serve TWO calls of equal geometry (rnh_conv_wino_pair: the ConvLSTM cells of the two directions at small images, where a call
// alone leaves half the chip idle): workgroups [0, nA) belong to PP.call[0], the rest to PP.call[1] (an offset into the kernel-argument segment, no
// copy).  A single call passes nA = its workgroup count and call[1] is never read.
struct rnh_conv_pair_t {
    rnh_conv_args_t call[2];
};

template <int EPI, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) conv_winoh_kernel(const rnh_conv_pair_t PP, const int nA, const int MT, const int NT, const int TX,
                                                                               const int TY) {
    const int second = (int)blockIdx.x >= nA;
    const rnh_conv_args_t &P = PP.call[second];
    const int bx = second ? (int)blockIdx.x - nA : (int)blockIdx.x;
    using G = WinoGeo<NW>;
    constexpr int TILES = H_TILES, CPC = G::CPC, CH = G::CH, CHS = G::CHS, BUF = G::BUF, SPC = G::SPC, CG = G::CG;
    constexpr int H_PART = G::PART, H_TS = G::TS, H_GS = G::GS, CW = G::CW;
    __shared__ __attribute__((aligned(16))) float stage[2 * BUF];   // 73.7 KB (two workgroups per CU) / 139 KB
    __shared__ int tpix[TILES];                               // top-left output pixel of the block's tiles (epilogue)
    __shared__ int tcoord[TILES];                             // the same as (image << 20 | y << 10 | x), -1: no such tile
    HSTAMP(0);
#ifdef RNH_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        g_wino_hw[blockIdx.x * 3] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        g_wino_hw[blockIdx.x * 3 + 1] = __builtin_readcyclecounter();
    }
#endif
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = wave & 1, cg = wave >> 1;                   // half of the transform domain, column group
    const int bid = rnh_xcd_remap(bx, MT * NT);
    const int mt = bid / NT, nt = bid - mt * NT;
    const int H = P.H, W = P.W, ntiles = P.B * TY * TX;
    const int m0 = mt * TILES;

    // ---- staging: thread = (tile ts, channel pair cp of the chunk) ------------------------------------------------
    const int ts = threadIdx.x / CPC, cp = threadIdx.x % CPC;
    // Tile t of the list -> (image, tile row, tile column).  Where the tile grid allows it (TX % 8 == 0, TY % 4 == 0) the
    // list runs over 8 x 4 blocks of tiles, so that a workgroup's 32 tiles are a 16 x 8 pixel rectangle whose 4x4 patches
    // cover 18 x 10 input pixels (1.4x the outputs) instead of a 64 x 2 strip that needs 66 x 4 (2.1x): fewer bytes through
    // the vector cache per chunk and through L2 per launch.
    const bool blocked = !(TX & 7) && !(TY & 3);
    auto tile_xy = [&](int t, int &img, int &ty, int &tx) {
        img = t / (TY * TX);
        const int rem = t - img * TY * TX;
        if (blocked) {
            const int bi = rem >> 5, wi = rem & 31, bpr = TX >> 3, by = bi / bpr, bx = bi - by * bpr;
            ty = by * 4 + (wi >> 3);
            tx = bx * 8 + (wi & 7);
        } else {
            ty = rem / TX;
            tx = rem - ty * TX;
        }
    };
    const int t0 = m0 < ntiles ? m0 : 0;
    int img0, ty0, tx0_;
    tile_xy(t0, img0, ty0, tx0_);
    const int sc = P.src[0].scale, Hs = H * sc, Ws = W * sc;
    const int base_pix = (img0 * Hs + (2 * ty0 - 1) * sc) * Ws - sc;   // at or before every pixel the block touches
    // the thread's 4x4 patch: pixel offset of its top-left corner (relative to base_pix) and a 16-bit mask of the pixels
    // inside the image (the 16 offsets are rebuilt from these two per source - they would cost 16 registers to keep)
    int pix00, okmask = 0;
    {
        const int t = m0 + ts;
        const bool tok = t < ntiles;
        const int tt = tok ? t : t0;
        int img, ty, tx;
        tile_xy(tt, img, ty, tx);
        pix00 = (img * Hs + (2 * ty - 1) * sc) * Ws + (2 * tx - 1) * sc - base_pix;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int y = 2 * ty - 1 + (p >> 2), x = 2 * tx - 1 + (p & 3);
            okmask |= (tok && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? 1 << p : 0;
        }
    }
    int si = 0, cchunk = 0, nchunk = P.src[0].nch / CH;
    int voff[16];
    i32x4 adesc;
    auto setup_src = [&](int sidx) {
        const rnh_src_t &S = P.src[sidx];
        adesc = hdesc(S.ptr + S.c0 + ((long)S.img_off * Hs * Ws + base_pix + S.sub_y * Ws + S.sub_x) * S.C);
        const int C4 = S.C * 4;
        // (opaque copy: the 16 pixel offsets are loop invariants that hipcc would hoist out of the chunk loop - and spill)
        int pb = pix00;
        asm volatile("" : "+v"(pb));
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int off = __mul24(pb + ((p >> 2) * Ws + (p & 3)) * sc, C4) + cp * 8;      // < 2^24 pixels per block, < 2^24 bytes per pixel
            const int inside = __builtin_amdgcn_sbfe(okmask, p, 1);                       // -1 inside the image, 0 outside
            voff[p] = off | ~inside;
        }
        nchunk = S.nch / CH;
    };
    setup_src(0);

    // All vector-memory and LDS reads of the loop are volatile asm with hand-counted waits (hipcc sinks plain loads to their first use and waits for each with a full s_waitcnt between two MFMAs); every wait
    // names the registers it covers as "+v" operands, which orders their uses behind it.  The s_nop covers the 5 wait
    // states between an SALU / v_readfirstlane write of an SGPR and a VMEM instruction reading it.
    auto ld8 = [&](f32x2 *dst, const int *vo, const i32x4 &desc, int soff) {
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dwordx2 %0, %8, %16, %17 offen\n\t"
            "buffer_load_dwordx2 %1, %9, %16, %17 offen\n\t"
            "buffer_load_dwordx2 %2, %10, %16, %17 offen\n\t"
            "buffer_load_dwordx2 %3, %11, %16, %17 offen\n\t"
            "buffer_load_dwordx2 %4, %12, %16, %17 offen\n\t"
            "buffer_load_dwordx2 %5, %13, %16, %17 offen\n\t"
            "buffer_load_dwordx2 %6, %14, %16, %17 offen\n\t"
            "buffer_load_dwordx2 %7, %15, %16, %17 offen"
            : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]), "=&v"(dst[4]), "=&v"(dst[5]), "=&v"(dst[6]), "=&v"(dst[7])
            : "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "v"(vo[4]), "v"(vo[5]), "v"(vo[6]), "v"(vo[7]), "s"(desc), "s"(soff)
            : "memory");
    };
    f32x2 stg[16];
    auto gload = [&]() {                                   // next chunk of the source list -> registers (16 loads)
        const int soff = __builtin_amdgcn_readfirstlane(cchunk * CH * 4);
        ld8(stg, voff, adesc, soff);
        ld8(stg + 8, voff + 8, adesc, soff);
        if (++cchunk == nchunk) {
            cchunk = 0;
            if (++si < P.nsrc) setup_src(si);
        }
    };
    auto xform_store = [&](int buf) {                       // V = B^T d B on the thread's two channels, to LDS
        asm volatile("s_waitcnt vmcnt(4)"
                     : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[3]), "+v"(stg[4]), "+v"(stg[5]), "+v"(stg[6]), "+v"(stg[7]),
                       "+v"(stg[8]), "+v"(stg[9]), "+v"(stg[10]), "+v"(stg[11]), "+v"(stg[12]), "+v"(stg[13]), "+v"(stg[14]),
                       "+v"(stg[15]));
        auto sub = [&](f32x2 a, f32x2 b) {                 // one v_pk_add_f32 (hipcc scalarises packed adds / subtractions)
            f32x2 r;
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            return r;
        };
        auto add = [&](f32x2 a, f32x2 b) {
            f32x2 r;
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            return r;
        };
        f32x2 tq[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tq[0 * 4 + j] = sub(stg[0 * 4 + j], stg[2 * 4 + j]);
            tq[1 * 4 + j] = add(stg[1 * 4 + j], stg[2 * 4 + j]);
            tq[2 * 4 + j] = sub(stg[2 * 4 + j], stg[1 * 4 + j]);
            tq[3 * 4 + j] = sub(stg[1 * 4 + j], stg[3 * 4 + j]);
        }
        float *o = stage + buf * BUF + ts * CHS + 4 * cp;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 v0 = sub(tq[i * 4 + 0], tq[i * 4 + 2]), v1 = add(tq[i * 4 + 1], tq[i * 4 + 2]);
            const f32x2 v2 = sub(tq[i * 4 + 2], tq[i * 4 + 1]), v3 = sub(tq[i * 4 + 1], tq[i * 4 + 3]);
            *reinterpret_cast<f32x4w *>(o + (i * 2 + 0) * TILES * CHS) = __builtin_shufflevector(v0, v1, 0, 1, 2, 3);
            *reinterpret_cast<f32x4w *>(o + (i * 2 + 1) * TILES * CHS) = __builtin_shufflevector(v2, v3, 0, 1, 2, 3);
        }
    };

    // this wave's pairs of transform positions: pr = 2 i + h (row i of the 4x4 domain, columns 2h and 2h + 1)
    const i32x4 bdesc = hdesc(P.wp + (long)((nt * CG + cg) * 32) * 8);
    const int pstride = P.Npad * 32;                        // bytes between two PAIRS of transform positions of one step
    int boffx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) boffx[i] = (l31 * 2 + kh) * 16 + (2 * i + h) * pstride;
    auto loadb = [&](f32x4w *u, int sb) {                   // transformed weights of step sb: 4 loads of 16 bytes
        const int soff = __builtin_amdgcn_readfirstlane(sb * 8 * pstride);
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dwordx4 %0, %4, %8, %9 offen\n\t"
            "buffer_load_dwordx4 %1, %5, %8, %9 offen\n\t"
            "buffer_load_dwordx4 %2, %6, %8, %9 offen\n\t"
            "buffer_load_dwordx4 %3, %7, %8, %9 offen"
            : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3])
            : "v"(boffx[0]), "v"(boffx[1]), "v"(boffx[2]), "v"(boffx[3]), "s"(bdesc), "s"(soff)
            : "memory");
    };
    const unsigned lds0 = (unsigned)(size_t)stage;
    const unsigned vlane = lds0 + ((h * TILES + l31) * CHS + 4 * kh) * 4;
    // the lane's tile, channels 4q + 2kh, +1 of the staged chunk, pairs 2a and 2a + 1 of this wave's four (half a step: 8 MFMAs)
    auto loadv = [&](f32x4w *V, int buf, int q, auto a_tag) {
        constexpr int a = decltype(a_tag)::value;
        const unsigned adr = vlane + buf * BUF * 4 + q * 32;
        asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(V[0]) : "v"(adr), "i"((4 * a) * TILES * CHS * 4) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(V[1]) : "v"(adr), "i"((4 * a + 2) * TILES * CHS * 4) : "memory");
    };
    auto wait_lds = [&](f32x4w *V) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V[0]), "+v"(V[1])); };
    auto wait_vm = [&](f32x4w *u, auto keep) {
        asm volatile("s_waitcnt vmcnt(%c4)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]) : "i"(decltype(keep)::value));
    };

    f32x16 acc[8];                                          // acc[2 i + odd] = position (row i, column 2h + odd)

    auto compute = [&](const f32x4w *V, const f32x4w *u, auto a_tag) {      // V: the two pairs of half a, u: all four pairs of the step
        constexpr int a = decltype(a_tag)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = 2 * a + j;
            acc[2 * i] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[j].x, u[i].x, acc[2 * i], 0, 0, 0);
            acc[2 * i] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[j].y, u[i].y, acc[2 * i], 0, 0, 0);
            acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[j].z, u[i].z, acc[2 * i + 1], 0, 0, 0);
            acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[j].w, u[i].w, acc[2 * i + 1], 0, 0, 0);
        }
    };

    // ---- main loop over chunks of CH channels (SPC steps of 4 channels, 16 MFMAs each); chunk c+1 travels global -> registers
    // under the MFMAs of chunk c and is transformed into the other LDS buffer in its next-to-last step -------------------
    int nchunks_total = 0;
    for (int i = 0; i < P.nsrc; ++i) nchunks_total += P.src[i].nch / CH;
    f32x4w Va[2], Vb[2], u0[4], u1[4];                        // staged operands by half steps, weights by steps
    if (threadIdx.x < TILES) {
        const int tr = m0 + threadIdx.x, tq = tr < ntiles ? tr : t0;
        int im, yy, xx;
        tile_xy(tq, im, yy, xx);
        tpix[threadIdx.x] = (im * H + 2 * yy) * W + 2 * xx;
        tcoord[threadIdx.x] = tr < ntiles ? (im << 20) | (2 * yy << 10) | (2 * xx) : -1;
    }
    HSTAMP(1);
    gload();
    loadb(u0, 0);                                            // 4 loads younger than the staging loads: vmcnt(4) in xform_store
    xform_store(0);
    __syncthreads();
    HSTAMP(2);
    int s = 0;                                               // global 4-channel step index (weights)
#pragma unroll
    for (int a = 0; a < 8; ++a)                              // (zeroed here, behind the prologue's transform: 128 registers)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[a][v] = 0.f;
    using A0 = std::integral_constant<int, 0>;
    using A1 = std::integral_constant<int, 1>;
    loadv(Va, 0, 0, A0());
    using K4 = std::integral_constant<int, 4>;               // the 4 weight loads of ONE step may stay in flight
    using K0 = std::integral_constant<int, 0>;
    // ---- what the epilogue needs from memory: bias of the lane's column and, for the ConvLSTM, the previous cell state of
    // the thread's two items (tile, pixel, 4 hidden channels) q * threads + tid - 16 bytes of every gate, of c and of h per
    // item.  Requested in the LAST chunk right behind its last weight request (vector-memory loads complete in order: any
    // earlier and the chunk's counted waits for weights would have to sit out these loads first).  The state comes from
    // HBM behind the gate stores of the whole chip: measured 54 us of a 387 us launch while it was requested behind the
    // main loop and awaited before the gate math; now it has the last two steps, the exchange and the gate math to arrive.
    const int ncol = (nt * CG + cg) * 32 + l31;
    float bv;
    [[maybe_unused]] f32x4w cpv[2];
    [[maybe_unused]] bool lstm_full = false;
    [[maybe_unused]] long item_o[2];                            // pixel index of the two items
    [[maybe_unused]] int item_x[2];                             // their float offset in the gate exchange area
    constexpr int NEPI = EPI == RNH_EPI_LSTM ? 3 : 1;           // loads of the request
    auto epi_request = [&]() {
        // (unconditional asm loads from clamped, always valid addresses: a load under an `if` would give its target register
        // a second definition and hipcc a reason to copy it in flight; no bias / no previous state: zeros behind the wait)
        asm volatile("global_load_dword %0, %1, off" : "=v"(bv) : "v"((P.bias ? P.bias : P.wp) + ncol) : "memory");
        if constexpr (EPI == RNH_EPI_LSTM) {
            lstm_full = m0 + TILES <= ntiles && !(H & 1) && !(W & 1) && nt * CW + CW <= P.hd;
            const float *csrc = P.c_prev ? P.c_prev : P.c_out;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int it = q * (64 * NW) + (int)threadIdx.x, t = it / CW, p = (it / (CW / 4)) & 3, c4 = (it % (CW / 4)) * 4;
                item_o[q] = (long)tpix[t] + (p >> 1) * W + (p & 1);
                item_x[q] = t * H_TS + p * CW + c4;
                const int hcl = min(nt * CW + c4, P.hd - 4);    // (the tile list is clamped in tpix, the channel here)
                const long pl = lstm_full ? item_o[q] : (long)tpix[t];
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(cpv[q]) : "v"(csrc + pl * P.hd + hcl) : "memory");
            }
        }
    };
    // The loop body (every chunk but the last) has no branch: a register that is the target of an asynchronous asm load
    // must have exactly one definition per iteration (tests/test_isa_guards.py); the last chunk is peeled off.
    // LDS operands run half a step (8 MFMAs) ahead in two register pairs, weights one step (16 MFMAs) ahead.
    // One chunk = SPC steps of 16 MFMAs.  Step q multiplies out of (Va, Vb) and the weight set of its parity (u0 for even q);
    // the staging loads of the next chunk go out in step 0, its transform into the other LDS buffer happens in step
    // SPC - 2, the chunk's barrier sits in front of the last 8 MFMAs of step SPC - 1.
    auto chunk = [&](const int buf, auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
        auto step = [&](auto q_tag, f32x4w *ucur, f32x4w *unext) {
            constexpr int q = decltype(q_tag)::value;
            constexpr bool last = q == SPC - 1;
            wait_lds(Va);
            loadv(Vb, buf, q, A1());
            if constexpr (q == 0) {
                // (the staging loads go behind the wait: a staging load whose 64 lanes are all outside the image never goes
                // to memory and returns ahead of older loads, so it must not be among the loads a counted wait leaves in flight)
                wait_vm(ucur, K0());
                if constexpr (more) gload();
                loadb(unext, s + 1);
            } else if constexpr (!last) {
                loadb(unext, s + q + 1);
                if constexpr (q == SPC - 2 && !more) {
                    epi_request();                          // younger than every weight load: the waits below leave it in flight
                    wait_vm(ucur, std::integral_constant<int, 4 + NEPI>());
                } else {
                    wait_vm(ucur, K4());
                }
                // the transform of the staged chunk into the other LDS buffer (its loads are older than the weight loads the
                // wait leaves in flight)
                if constexpr (q == SPC - 2 && more) xform_store(buf ^ 1);
            } else {
                if constexpr (more) {
                    loadb(unext, s + SPC);
                    wait_vm(ucur, K4());
                } else {
                    wait_vm(ucur, std::integral_constant<int, NEPI>());
                }
            }
            compute(Va, ucur, A0());
            wait_lds(Vb);
            if constexpr (!last) {
                loadv(Va, buf, q + 1, A0());
            } else {
                // the chunk's barrier in front of its last 8 MFMAs (all reads of this buffer issued and landed, all writes of
                // the other one done), so the first operands of the next chunk are fetched under their cover
                asm volatile("s_barrier" ::: "memory");
                if constexpr (more) loadv(Va, buf ^ 1, 0, A0());
            }
            compute(Vb, ucur, A1());
        };
        step(std::integral_constant<int, 0>(), u0, u1);
        step(std::integral_constant<int, 1>(), u1, u0);
        step(std::integral_constant<int, 2>(), u0, u1);
        step(std::integral_constant<int, 3>(), u1, u0);
        if constexpr (SPC == 8) {
            step(std::integral_constant<int, 4>(), u0, u1);
            step(std::integral_constant<int, 5>(), u1, u0);
            step(std::integral_constant<int, 6>(), u0, u1);
            step(std::integral_constant<int, 7>(), u1, u0);
        }
        s += SPC;
    };
    for (int c = 0; c + 1 < nchunks_total; ++c) chunk(c & 1, std::true_type());
    chunk((nchunks_total - 1) & 1, std::false_type());
    HSTAMP(3);

    // ---- output transform: this half's share of Y = A^T M A, exchange with the partner wave --------------------------
    // (every wave is past the last chunk's barrier, i.e. nobody reads the staging buffers any more)
    f32x4w *px = reinterpret_cast<f32x4w *>(stage);
    float Yf[8][4];                                             // entries 8h .. 8h + 7: tiles 16h .. 16h + 15 of the block
    // (the half is a template argument: accumulator registers cannot be indexed at run time)
    auto exchange = [&](auto h_tag) {
        constexpr int hh = decltype(h_tag)::value;
        auto part4 = [&](int v, float *Y) {
            float s0[2], s1[2];
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                s0[o] = acc[0 + o][v] + acc[2 + o][v] + acc[4 + o][v];
                s1[o] = acc[2 + o][v] - acc[4 + o][v] - acc[6 + o][v];
            }
            if constexpr (hh == 0) {                            // columns 0, 1 of the domain
                Y[0] = s0[0] + s0[1]; Y[1] = s0[1]; Y[2] = s1[0] + s1[1]; Y[3] = s1[1];
            } else {                                            // columns 2, 3
                Y[0] = s0[0]; Y[1] = -s0[0] - s0[1]; Y[2] = s1[0]; Y[3] = -s1[0] - s1[1];
            }
        };
#pragma unroll
        for (int e = 0; e < 8; ++e) {                           // the partner's entries
            float Y[4];
            part4(8 * (1 - hh) + e, Y);
            const f32x4w y4 = {Y[0], Y[1], Y[2], Y[3]};
            px[(wave * 8 + e) * 64 + lane] = y4;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) part4(8 * hh + e, Yf[e]);
    };
    if (h == 0) exchange(std::integral_constant<int, 0>());
    else exchange(std::integral_constant<int, 1>());
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(%c1)" : "+v"(bv) : "i"(NEPI - 1));       // the bias has landed, the state may still be on its way
    if (!P.bias) bv = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const f32x4w y4 = px[((wave ^ 1) * 8 + e) * 64 + lane];
        Yf[e][0] += y4.x + bv; Yf[e][1] += y4.y + bv; Yf[e][2] += y4.z + bv; Yf[e][3] += y4.w + bv;
    }
    HSTAMP(4);
    // tile row (0..31) of entry e of this wave
    auto trl_of = [&](int e) { const int v = 8 * h + e; return (v & 3) + 8 * (v >> 2) + 4 * kh; };

    if constexpr (EPI == RNH_EPI_LSTM) {
        const int hd = P.hd;
        const bool full = lstm_full;
        float *xg = stage + H_PART;
        // phase 1: lanes 0..15 / 16..31 of a row block hold gates 2cg / 2cg + 1 (i, f | o, g) of 16 hidden channels;
        // sigmoid, and tanh as 2 sigmoid(2x) - 1 for the candidate gate, in one form: m rcp(1 + exp(-m x)) + b
        const int gate = (cg * 32 + l31) / CW, ch = (cg * 32 + l31) % CW, hc = nt * CW + ch;
        const float gm = gate == 3 ? 2.f : 1.f, gb = gate == 3 ? -1.f : 0.f;
        float *xw = xg + gate * H_GS + ch;
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int p = 0; p < 4; ++p) Yf[e][p] = __builtin_fmaf(gm, __builtin_amdgcn_rcpf(1.f + __expf(-gm * Yf[e][p])), gb);
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int p = 0; p < 4; ++p) xw[trl_of(e) * H_TS + p * CW] = Yf[e][p];
        if (P.gates_out && !full) {                             // (whole blocks store the gates from LDS in phase 2, 16 bytes per lane)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int trl = trl_of(e), tc = tcoord[trl];
                const bool ok = tc >= 0 && hc < hd;
                const int yy = (tc >> 10) & 1023, xx = tc & 1023;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (ok && yy + (p >> 1) < H && xx + (p & 1) < W)
                        P.gates_out[((long)tpix[trl] + (p >> 1) * W + (p & 1)) * 4 * hd + gate * hd + hc] = Yf[e][p];
            }
        }
        // the gates are in LDS: wait for the LDS writes only (not for the gates_out stores)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        HSTAMP(5);
        // (on every path: the registers must not be reused while the loads are in flight.  On the partial-block path this is
        // also a wait for its gate stores.)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(cpv[0]), "+v"(cpv[1]));
        if (!P.c_prev) cpv[0] = cpv[1] = f32x4w{0.f, 0.f, 0.f, 0.f};
        if (full) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c4 = ((q * (64 * NW) + (int)threadIdx.x) % (CW / 4)) * 4;
                const float *xi = xg + item_x[q];
                const f32x4w gi = *reinterpret_cast<const f32x4w *>(xi), gf = *reinterpret_cast<const f32x4w *>(xi + H_GS);
                const f32x4w go = *reinterpret_cast<const f32x4w *>(xi + 2 * H_GS), gg = *reinterpret_cast<const f32x4w *>(xi + 3 * H_GS);
                if (P.gates_out) {
                    // (plain stores: with the nt bit, 16-byte pieces of one 128-byte line written by the workgroups of four
                    // column blocks came out corrupted now and then - tools/debug/lstm_mismatch.py)
                    f32x4w *gp = reinterpret_cast<f32x4w *>(P.gates_out + item_o[q] * 4 * hd + nt * CW + c4);
#ifdef RNH_GATES_NT      // experiment (tools/nt_store_probe.hip found nothing wrong with nt pieces of shared lines): gates past the caches
                    __builtin_nontemporal_store(gi, gp); __builtin_nontemporal_store(gf, gp + hd / 4);
                    __builtin_nontemporal_store(go, gp + 2 * (hd / 4)); __builtin_nontemporal_store(gg, gp + 3 * (hd / 4));
#else
                    gp[0] = gi; gp[hd / 4] = gf; gp[2 * (hd / 4)] = go; gp[3 * (hd / 4)] = gg;
#endif
                }
                f32x4w cn, hn;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cn[j] = gf[j] * cpv[q][j] + gi[j] * gg[j];
                    hn[j] = go[j] * h_tanh(cn[j]);
                }
                const long o = item_o[q] * hd + nt * CW + c4;
                *reinterpret_cast<f32x4w *>(P.c_out + o) = cn;
                *reinterpret_cast<f32x4w *>(P.h_out + o) = hn;
            }
        } else {
            // partial blocks: items (tile, pixel, channel) e = k * threads + tid one by one
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = k * (64 * NW) + (int)threadIdx.x, ch2 = e % CW, p2 = (e / CW) & 3, t = e / (4 * CW);
                const int tc = tcoord[t], hc2 = nt * CW + ch2;
                if (tc < 0 || hc2 >= hd) continue;
                const int yy = (tc >> 10) & 1023, xx = tc & 1023;
                if (yy + (p2 >> 1) >= H || xx + (p2 & 1) >= W) continue;
                const float *xr = xg + t * H_TS + p2 * CW + ch2;
                const float gi = xr[0 * H_GS], gf = xr[1 * H_GS], go = xr[2 * H_GS], gg = xr[3 * H_GS];
                const long o = ((long)tpix[t] + (p2 >> 1) * W + (p2 & 1)) * hd + hc2;
                const float cp = P.c_prev ? P.c_prev[o] : 0.f;
                const float cn = gf * cp + gi * gg;
                P.c_out[o] = cn;
                P.h_out[o] = go * h_tanh(cn);
            }
        }
        HSTAMP(6);
#ifdef RNH_STAMPS
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_wino_hw[blockIdx.x * 3 + 2] = __builtin_readcyclecounter();
#endif
    } else if constexpr (EPI == RNH_EPI_PS) {
        // column n = (i*r + j)*cq + c  ->  pixel (r*y + i, r*x + j), channel c of the (B, rH, rW, cq) destination
        const int r = P.ps_r, cq = P.ps_cq;
        if (ncol >= cq * r * r) return;
        const int sub = ncol / cq, c = ncol - sub * cq, pi = sub / r, pj = sub - pi * r;
        float *dp = P.dst[0].ptr + c;
        const long Wr = (long)W * r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int tc = tcoord[trl_of(e)];
            if (tc < 0) continue;
            const int im = tc >> 20, yy = (tc >> 10) & 1023, xx = tc & 1023;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int y = yy + (p >> 1), x = xx + (p & 1);
                if (y >= H || x >= W) continue;
                dp[(((long)im * H + y) * r + pi) * Wr * cq + ((long)x * r + pj) * cq] = Yf[e][p];
            }
        }
    } else {
        // destination segment of this lane's column
        int seg = -1, cbase = 0;
        for (int d = 0; d < P.ndst; ++d) {
            if (seg < 0 && ncol < cbase + P.dst[d].ncols) seg = d;
            if (seg < 0) cbase += P.dst[d].ncols;
        }
        if (seg < 0) return;
        const rnh_dst_t &D = P.dst[seg];
        float *dp = D.ptr + (long)D.img_off * H * W * D.C + D.c0 + (ncol - cbase);
        const bool full = m0 + TILES <= ntiles && !(H & 1) && !(W & 1);
        if (full) {                                          // no per-element predicates
            const long rowC = (long)W * D.C;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float *o = dp + (long)tpix[trl_of(e)] * D.C;
                if (D.accumulate) {
                    const float a0 = o[0], a1 = o[D.C], a2 = o[rowC], a3 = o[rowC + D.C];
                    o[0] = a0 + Yf[e][0]; o[D.C] = a1 + Yf[e][1]; o[rowC] = a2 + Yf[e][2]; o[rowC + D.C] = a3 + Yf[e][3];
                } else {
                    o[0] = Yf[e][0]; o[D.C] = Yf[e][1]; o[rowC] = Yf[e][2]; o[rowC + D.C] = Yf[e][3];
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int trl = trl_of(e), tc = tcoord[trl];
                if (tc < 0) continue;
                const int yy = (tc >> 10) & 1023, xx = tc & 1023;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (yy + (p >> 1) >= H || xx + (p & 1) >= W) continue;
                    float *o = dp + ((long)tpix[trl] + (p >> 1) * W + (p & 1)) * D.C;
                    *o = D.accumulate ? *o + Yf[e][p] : Yf[e][p];
                }
            }
        }
    }
}

inline int wgrid_for(long n, int cap = 8192) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

#ifdef RNH_STAMPS
extern "C" int rnh_debug_wino_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino_stamps), sizeof(g_wino_stamps));
}
extern "C" int rnh_debug_wino_hw(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino_hw), sizeof(g_wino_hw));
}
#endif

extern "C" int rnh_wino_pack_weights(const float *w, const float *bias, float *wp, float *biasp, const int32_t *kbase, const int32_t *knv,
                                     const int32_t *kcoff, const int32_t *colmap, int ns, int Npad, int Cout, int Cin, int kstride,
                                     int transposed, void *stream) {
    if (!w || !wp || !kbase || !knv || !colmap || ns < 1 || Npad < 1 || Cout < 1 || Cin < 1 || kstride < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_wino_pack_weights: bad arguments");
    if (Npad % 64) RNH_FAIL(RNH_E_RANGE, "rnh_wino_pack_weights: Npad must be a multiple of 64");
    hipLaunchKernelGGL(wino_pack_kernel, dim3(wgrid_for((long)ns * 16 * Npad * 4 + Npad)), dim3(256), 0, (hipStream_t)stream, w, bias, wp,
                       biasp, kbase, knv, kcoff, colmap, ns, Npad, Cout, Cin, kstride, transposed);
    RNH_CHECK_LAUNCH("rnh_wino_pack_weights");
    return 0;
}

template <int NW>
static int launch_wino(const rnh_conv_args_t &a, const rnh_conv_args_t &b, bool pair, int MT, int NT, int TX, int TY, hipStream_t st) {
    using G = WinoGeo<NW>;
    const int nA = MT * NT;
    const dim3 grid((unsigned)(pair ? 2 * nA : nA)), block(64 * NW);
    rnh_conv_pair_t pp;
    pp.call[0] = a;
    pp.call[1] = b;
    switch (a.epilogue) {
        case RNH_EPI_STORE:
            hipLaunchKernelGGL((conv_winoh_kernel<RNH_EPI_STORE, NW>), grid, block, 0, st, pp, nA, MT, NT, TX, TY);
            break;
        case RNH_EPI_PS:
            hipLaunchKernelGGL((conv_winoh_kernel<RNH_EPI_PS, NW>), grid, block, 0, st, pp, nA, MT, NT, TX, TY);
            break;
        case RNH_EPI_LSTM:
            if (a.Npad != 32 * G::CG * ((a.hd + G::CW - 1) / G::CW) || b.Npad != 32 * G::CG * ((b.hd + G::CW - 1) / G::CW))
                RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: LSTM column layout (%d-column blocks = the four gates of %d hidden channels)", 32 * G::CG, G::CW);
            hipLaunchKernelGGL((conv_winoh_kernel<RNH_EPI_LSTM, NW>), grid, block, 0, st, pp, nA, MT, NT, TX, TY);
            break;
        default:
            RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: epilogue %d not available", a.epilogue);
    }
    return 0;
}

struct WinoGeom {               // what the launch of a validated call needs
    int wide, MT, NT, TX, TY;
};

static int wino_check(const rnh_conv_args_t &a, WinoGeom &g, const char *who) {
    if (a.nsrc < 1 || a.nsrc > RNH_MAX_SRC || a.B < 1 || a.H < 1 || a.W < 1 || !a.wp) RNH_FAIL(RNH_E_ARG, "%s: bad arguments", who);
    if (a.ntaps != 9) RNH_FAIL(RNH_E_RANGE, "%s: 3x3 convolutions only", who);
    const int wide = a.tile == RNH_WINO_COLS128;                // 128-column blocks (8 waves, 32-channel chunks), else 64-column blocks
    const int bc = wide ? 128 : 64, cm = wide ? 32 : 16;
    if (a.Npad < bc || a.Npad % bc) RNH_FAIL(RNH_E_RANGE, "%s: Npad must be a multiple of %d", who, bc);
    int steps = 0;
    for (int i = 0; i < a.nsrc; ++i) {
        if (int rc = rnh_check_src(a.src[i], who)) return rc;
        if (a.src[i].scale != a.src[0].scale || a.src[i].ptr2) RNH_FAIL(RNH_E_RANGE, "%s: one scale for all sources, no second pointer", who);
        if (a.src[i].nch % cm) RNH_FAIL(RNH_E_ALIGN, "%s: source channel counts must be multiples of %d", who, cm);
        steps += a.src[i].nch / 4;
    }
    if (steps != a.nk) RNH_FAIL(RNH_E_ARG, "%s: nk = %d but the sources hold %d steps of 4 channels", who, a.nk, steps);
    const int TY = (a.H + 1) / 2, TX = (a.W + 1) / 2;
    const long ntiles = (long)a.B * TY * TX;
    if (a.H > 1023 || a.W > 1023 || a.B > 2047) RNH_FAIL(RNH_E_RANGE, "%s: at most 2047 images of 1023 x 1023", who);
    if (ntiles * 4 >= (1L << 29)) RNH_FAIL(RNH_E_RANGE, "%s: too many pixels for 32-bit offsets", who);
    // pixel offsets inside a block (it may straddle two images) go through 24-bit multiplies
    if ((long)a.H * a.W * a.src[0].scale * a.src[0].scale >= (1L << 22)) RNH_FAIL(RNH_E_RANGE, "%s: source images of at most 2^22 pixels", who);
    g.wide = wide, g.TX = TX, g.TY = TY;
    g.MT = (int)((ntiles + H_TILES - 1) / H_TILES), g.NT = a.Npad / bc;
    if (a.epilogue == RNH_EPI_STORE) {
        if (a.ndst < 1 || a.ndst > RNH_MAX_DST) RNH_FAIL(RNH_E_ARG, "%s: bad destination count", who);
        for (int d = 0; d < a.ndst; ++d)
            if (!a.dst[d].ptr || a.dst[d].ncols < 1) RNH_FAIL(RNH_E_ARG, "%s: bad destination %d", who, d);
    } else if (a.epilogue == RNH_EPI_PS) {
        if (a.ndst != 1 || !a.dst[0].ptr || a.ps_r < 1 || a.ps_cq < 1 || a.ps_cq * a.ps_r * a.ps_r > a.Npad)
            RNH_FAIL(RNH_E_ARG, "%s: bad pixel-shuffle destination", who);
    } else if (a.epilogue == RNH_EPI_LSTM) {
        if (!a.h_out || !a.c_out || a.hd < 1 || !a.bias) RNH_FAIL(RNH_E_ARG, "%s: LSTM epilogue needs h_out, c_out, hd, bias", who);
    }
    return 0;
}

extern "C" int rnh_conv_wino(const rnh_conv_args_t *args, void *stream) {
    if (!args) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: null args");
    const rnh_conv_args_t &a = *args;
    WinoGeom g;
    if (int rc = wino_check(a, g, "rnh_conv_wino")) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = g.wide ? launch_wino<8>(a, a, false, g.MT, g.NT, g.TX, g.TY, st) : launch_wino<4>(a, a, false, g.MT, g.NT, g.TX, g.TY, st)) return rc;
    RNH_CHECK_LAUNCH("rnh_conv_wino");
    return 0;
}

extern "C" int rnh_conv_wino_pair(const rnh_conv_args_t *args_a, const rnh_conv_args_t *args_b, void *stream) {
    if (!args_a || !args_b) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino_pair: null args");
    const rnh_conv_args_t &a = *args_a, &b = *args_b;
    WinoGeom ga, gb;
    if (int rc = wino_check(a, ga, "rnh_conv_wino_pair (first call)")) return rc;
    if (int rc = wino_check(b, gb, "rnh_conv_wino_pair (second call)")) return rc;
    if (a.B != b.B || a.H != b.H || a.W != b.W || a.Npad != b.Npad || a.epilogue != b.epilogue || a.tile != b.tile || a.src[0].scale != b.src[0].scale)
        RNH_FAIL(RNH_E_ARG, "rnh_conv_wino_pair: the two calls must agree in B, H, W, Npad, epilogue, tile and source scale");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = ga.wide ? launch_wino<8>(a, b, true, ga.MT, ga.NT, ga.TX, ga.TY, st) : launch_wino<4>(a, b, true, ga.MT, ga.NT, ga.TX, ga.TY, st)) return rc;
    RNH_CHECK_LAUNCH("rnh_conv_wino_pair");
    return 0;
}
