// EXPERIMENT, not built into librefinenet_hip.so (round 2): rnh_conv_wino2 with persistent workgroups.  Correct (it passed the Winograd parity
// tests of tests/test_hip_parity.py when it was wired in) but slower than csrc/conv_wino2.hip; kept for the measurements
// quoted in that file's header and in DESIGN.md section 7.  To try it: copy to csrc/, add to build.sh, declare the entry in lib.py.
//
// 3x3 convolution (padding 1) in Winograd form F(2x2, 3x3) on fp32 MFMA for gfx950, two workgroups per CU - rnh_conv_wino.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 independent GEMMs (one per position xi of the 4x4 transform domain) of [tiles x C] x [C x N].  conv_wino.hip (rounds 1-2)
// gave one wave all 16 positions of 32 tiles x 32 columns: 256 accumulators + 256 working registers = the whole register
// file of a SIMD, ONE wave per SIMD, so every cycle that wave spent outside the MFMA stream (set-up, staging transform,
// barrier skew, the 2 000-instruction gate epilogue, stores) was a cycle of matrix-core idle time: 0.56 of the fp32 peak.
// Here a wave owns HALF the transform domain (the positions xi = 4 i + j with j in {2h, 2h+1}: 8 of the 16) of 32 tiles x 32
// columns - 128 accumulators, 256 registers in all - and a workgroup is 2 halves x 2 column groups = 32 tiles x 64 columns:
// two workgroups are resident per CU, every SIMD holds one wave of each, and whatever one of them does beside its MFMAs
// hides behind the MFMAs of the other.  The price:
//   * the output transform needs all four j: the two halves exchange partial 2x2 outputs through LDS (each wave keeps the
//     tiles of 8 accumulator registers and sends the other 8 to its partner: 8 ds_write_b128 + 8 ds_read_b128 per lane);
//   * the staged input transform of 32 tiles feeds 64 columns instead of 128 (LDS and L2 traffic per MFMA as before,
//     staging loads and transform adds per MFMA doubled for convolutions wider than 64 columns - vector work that now
//     runs under the other workgroup's MFMAs).
// Operands as in conv_wino.hip: the input transform B^T d B of a 16-channel chunk is computed once per workgroup (thread =
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

#ifdef RNH_STAMPS
__device__ unsigned long long g_wino2_stamps[8];
#ifndef RNH_STAMP_BLOCK
#define RNH_STAMP_BLOCK 0
#endif
#ifndef RNH_STAMP_TILE
#define RNH_STAMP_TILE 0
#endif
#ifdef RNH_STAMP_TILES
#define HSTAMP(i)
#else
#define HSTAMP(i) do { if (blockIdx.x == RNH_STAMP_BLOCK && threadIdx.x == 0 && k == RNH_STAMP_TILE) g_wino2_stamps[i] = __builtin_readcyclecounter(); } while (0)
#endif
#else
#define HSTAMP(i)
#endif

constexpr int H_TILES = 32, H_CPC = 8, H_CH = 16, H_CHS = 4 * H_CPC + 4, H_BUF = 8 * H_TILES * H_CHS;   // floats per staging buffer
constexpr int H_PART = 4 * 8 * 64 * 4;             // floats of the partial-output exchange: [wave][entry][lane][4]
constexpr int H_TS = 68, H_GS = 32 * H_TS + 32;    // gate exchange [gate][tile][pixel][16 channels]: strides that keep the four
                                                   // (lane half, gate) groups of a wave's store in four different bank ranges
static_assert(H_PART <= H_BUF && 4 * H_GS <= H_BUF, "the epilogue's exchange areas live in ONE staging buffer (the other holds the next block's first chunk)");

// Persistent workgroups: the launch has (at most) two workgroups per CU and each walks over the blocks
// blockIdx.x, blockIdx.x + gridDim.x, ... of the (tile block, column block) list.  The chunk pipeline runs ACROSS blocks: the
// last chunk of a block stages the first chunk of the next one (other tile coordinates, other weight columns) like any
// other next chunk, so set-up, the first staging round trip to memory and the launch of a new workgroup - 13-15 k of the
// 98 k cycles a block took as a workgroup of its own - disappear behind the MFMAs; the epilogue works in the LDS buffer
// the last chunk has just finished reading.
template <int EPI>
__global__ void __launch_bounds__(256, 2) conv_winoh_kernel(const rnh_conv_args_t P, const int MT, const int NT, const int TX, const int TY) {
    constexpr int TILES = H_TILES, CPC = H_CPC, CH = H_CH, CHS = H_CHS, BUF = H_BUF;
    __shared__ __attribute__((aligned(16))) float stage[2 * BUF];   // 73.7 KB: two workgroups per CU
    __shared__ int tpix[TILES];                               // top-left output pixel of the block's tiles (epilogue)
    __shared__ int tcoord[TILES];                             // the same as (image << 20 | y << 10 | x), -1: no such tile
#ifdef RNH_WINO_SOLO                                           // experiments: one workgroup per CU
    __shared__ float solo_pad[5000];
    if (P.H < 0) solo_pad[threadIdx.x] = 1.f, tpix[0] = (int)solo_pad[threadIdx.x + 1];
#endif
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = wave & 1, cg = wave >> 1;                   // half of the transform domain, column group
    // The two workgroups of a CU start together and - sharing the MFMA pipe while both are in their chunk loops - would stay
    // in step for the whole launch: the epilogues of both coincide and nothing hides them.  The workgroup in the second
    // wave slot of its SIMDs (HW_ID.wave_id) starts RNH_WINO_SKEW cycles late; the offset then persists.
#ifndef RNH_WINO_SKEW
#define RNH_WINO_SKEW 40000
#endif
    if (RNH_WINO_SKEW > 0 && gridDim.x > 256 && (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 1)) {      // HW_REG_HW_ID
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < RNH_WINO_SKEW) __builtin_amdgcn_s_sleep(32);
    }
#ifdef RNH_X_PRIO
    if (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 1) __builtin_amdgcn_s_setprio(RNH_X_PRIO);
#endif
    const int H = P.H, W = P.W, ntiles = P.B * TY * TX, nblocks = MT * NT;
    const int my_blocks = (nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // >= 1: the grid never exceeds the list
    // block i of this workgroup (i wraps: the pipeline stages one block past the end, see below)
    auto block_of = [&](int i, int &mt_, int &nt_) {
        const int bid = rnh_xcd_remap((int)blockIdx.x + (i % my_blocks) * (int)gridDim.x, nblocks);
        mt_ = bid / NT;
        nt_ = bid - mt_ * NT;
    };

    // ---- staging: thread = (tile ts, channel pair cp of the chunk) ------------------------------------------------
    const int ts = threadIdx.x / CPC, cp = threadIdx.x % CPC;
    const int sc = P.src[0].scale, Hs = H * sc, Ws = W * sc;
    // the thread's 4x4 patch of the block being STAGED (one block ahead of the one being computed at block boundaries):
    // pixel offset of its top-left corner (relative to base_pix) and a 16-bit mask of the pixels inside the image (the 16
    // offsets are rebuilt from these two per source - they would cost 16 registers to keep)
    int ld_block = 0, base_pix = 0, pix00 = 0, okmask = 0;
    auto stage_block = [&](int i) {
        int mt_, nt_;
        block_of(i, mt_, nt_);
        const int m0_ = mt_ * TILES, t0 = m0_ < ntiles ? m0_ : 0;
        const int img0 = t0 / (TY * TX), r0 = t0 - img0 * TY * TX, ty0 = r0 / TX;
        base_pix = (img0 * Hs + (2 * ty0 - 1) * sc) * Ws - sc;              // at or before every pixel the block touches
        const int t = m0_ + ts;
        const bool tok = t < ntiles;
        const int tt = tok ? t : t0;
        const int img = tt / (TY * TX), trem = tt - img * TY * TX, ty = trem / TX, tx = trem - ty * TX;
        pix00 = (img * Hs + (2 * ty - 1) * sc) * Ws + (2 * tx - 1) * sc - base_pix;
        okmask = 0;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int y = 2 * ty - 1 + (p >> 2), x = 2 * tx - 1 + (p & 3);
            okmask |= (tok && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? 1 << p : 0;
        }
    };
    int si = 0, cchunk = 0, nchunk = P.src[0].nch / CH;
    int C4 = 0;                                               // bytes per pixel of the source being staged
    i32x4 adesc;
    auto setup_src = [&](int sidx) {
        const rnh_src_t &S = P.src[sidx];
        adesc = hdesc(S.ptr + S.c0 + ((long)S.img_off * Hs * Ws + base_pix + S.sub_y * Ws + S.sub_x) * S.C);
        C4 = S.C * 4;
        nchunk = S.nch / CH;
    };
    // byte offsets of patch pixels p0 .. p0 + 7 for the loads of one chunk, rebuilt from (pix00, okmask) each time: three
    // vector instructions per load under the MFMAs instead of 16 registers held through the loop and the epilogue
    auto patch_offsets = [&](int p0, int *vo) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int p = p0 + q;
            const int off = __mul24(pix00 + ((p >> 2) * Ws + (p & 3)) * sc, C4) + cp * 8;   // < 2^24 pixels per block, < 2^24 bytes per pixel
            const int inside = __builtin_amdgcn_sbfe(okmask, p, 1);                       // -1 inside the image, 0 outside
            vo[q] = off | ~inside;
        }
    };
    stage_block(0);
    setup_src(0);

    // All vector-memory and LDS reads of the loop are volatile asm with hand-counted waits (see conv_wino.hip); every wait
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
        int vo[8];
        patch_offsets(0, vo);
        ld8(stg, vo, adesc, soff);
        patch_offsets(8, vo);
        ld8(stg + 8, vo, adesc, soff);
        if (++cchunk == nchunk) {
            cchunk = 0;
            if (++si == P.nsrc) {                           // the source list of the block is through: on to the next block
                si = 0;
                stage_block(++ld_block);
            }
            setup_src(si);
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
    const int pstride = P.Npad * 32;                        // bytes between two PAIRS of transform positions of one step
    int boffx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) boffx[i] = (l31 * 2 + kh) * 16 + (2 * i + h) * pstride;
    auto wdesc_of = [&](int nt_) { return hdesc(P.wp + (long)((nt_ * 2 + cg) * 32) * 8); };   // this wave's 32 columns of column block nt_
    auto loadb = [&](f32x4w *u, const i32x4 &bd, int sb) {   // transformed weights of step sb: 4 loads of 16 bytes
#ifdef RNH_X_NOB
        if (sb > 0) return;
#endif
        const int soff = __builtin_amdgcn_readfirstlane(sb * 8 * pstride);
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dwordx4 %0, %4, %8, %9 offen\n\t"
            "buffer_load_dwordx4 %1, %5, %8, %9 offen\n\t"
            "buffer_load_dwordx4 %2, %6, %8, %9 offen\n\t"
            "buffer_load_dwordx4 %3, %7, %8, %9 offen"
            : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3])
            : "v"(boffx[0]), "v"(boffx[1]), "v"(boffx[2]), "v"(boffx[3]), "s"(bd), "s"(soff)
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

    // ---- the chunk pipeline: 16-channel chunks of 4 steps of 4 channels (16 MFMAs each); the next chunk travels global ->
    // registers under the MFMAs of this one and is transformed into the other LDS buffer in its third step ---------------
    int nchunks_block = 0;
    for (int i = 0; i < P.nsrc; ++i) nchunks_block += P.src[i].nch / CH;
    f32x4w Va[2], Vb[2], u0[4], u1[4];                        // staged operands by half steps, weights by steps
    int mt, nt, mt_n, nt_n;                                   // the block being computed and the one after it
    block_of(0, mt, nt);
    i32x4 bdesc = wdesc_of(nt);
    gload();
    loadb(u0, bdesc, 0);                                     // 4 loads younger than the staging loads: vmcnt(4) in xform_store
    xform_store(0);
    __syncthreads();
    using A0 = std::integral_constant<int, 0>;
    using A1 = std::integral_constant<int, 1>;
    loadv(Va, 0, 0, A0());
    using K4 = std::integral_constant<int, 4>;               // the 4 weight loads of ONE step may stay in flight
    using K0 = std::integral_constant<int, 0>;
    // The chunk body has no branch: a register that is the target of an asynchronous asm load must have exactly one
    // definition per iteration (tests/test_isa_guards.py).  There is no "last chunk" form either: the last chunk of the
    // last block stages and prefetches a block that is never computed (block 0 of this workgroup again).
    // LDS operands run half a step (8 MFMAs) ahead in two register pairs, weights one step (16 MFMAs) ahead.
    // s: step index of the chunk's first step in the block's weights; (bd_next, s_next): where the step after the chunk is.
    auto chunk = [&](const int buf, const int s, const i32x4 &bd_next, const int s_next) {
        // step 0: [staging loads of the next chunk] [weights of step 1] | MFMAs of step 0
        wait_lds(Va);
        loadv(Vb, buf, 0, A1());
        // (the staging loads go behind the wait: a staging load whose 64 lanes are all outside the image never goes to
        // memory and returns ahead of older loads, so it must not be among the loads a counted wait leaves in flight)
        wait_vm(u0, K0());
#ifndef RNH_X_NOSTAGE
        gload();
#endif
        loadb(u1, bdesc, s + 1);
        compute(Va, u0, A0());
        wait_lds(Vb);
        loadv(Va, buf, 1, A0());
        compute(Vb, u0, A1());
        // step 1
        wait_lds(Va);
        loadv(Vb, buf, 1, A1());
        loadb(u0, bdesc, s + 2);
        wait_vm(u1, K4());
        compute(Va, u1, A0());
        wait_lds(Vb);
        loadv(Va, buf, 2, A0());
        compute(Vb, u1, A1());
        // step 2: also the transform of the staged chunk into the other LDS buffer (its loads are older than the weight
        // loads the wait leaves in flight)
        wait_lds(Va);
        loadv(Vb, buf, 2, A1());
        loadb(u1, bdesc, s + 3);
        wait_vm(u0, K4());
#ifndef RNH_X_NOSTAGE
        xform_store(buf ^ 1);
#endif
        compute(Va, u0, A0());
        wait_lds(Vb);
        loadv(Va, buf, 3, A0());
        compute(Vb, u0, A1());
        // last step: the chunk's barrier in front of its last 8 MFMAs (all reads of this buffer issued and landed, all
        // writes of the other one done), so the first operands of the next chunk are fetched under their cover
        wait_lds(Va);
        loadv(Vb, buf, 3, A1());
        loadb(u0, bd_next, s_next);
        wait_vm(u1, K4());
        compute(Va, u1, A0());
        wait_lds(Vb);
#ifndef RNH_X_NOBAR
        asm volatile("s_barrier" ::: "memory");
#endif
        loadv(Va, buf ^ 1, 0, A0());
        compute(Vb, u1, A1());
    };

    int g = 0;                                                // chunks done: chunk g lives in LDS buffer g & 1
    for (int k = 0; k < my_blocks; ++k) {
        HSTAMP(0);
#ifdef RNH_STAMP_TILES
        if (blockIdx.x == RNH_STAMP_BLOCK && threadIdx.x == 0 && k < 7) g_wino2_stamps[k] = __builtin_readcyclecounter(), g_wino2_stamps[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
        block_of(k + 1, mt_n, nt_n);
        const i32x4 bdesc_n = wdesc_of(nt_n);
        const int m0 = mt * TILES;
        if (threadIdx.x < TILES) {                            // (read in the epilogue, behind the barriers of the chunks)
            const int tr = m0 + threadIdx.x, tq = tr < ntiles ? tr : (m0 < ntiles ? m0 : 0);
            const int im = tq / (TY * TX), rr = tq - im * TY * TX, yy = rr / TX, xx = rr - yy * TX;
            tpix[threadIdx.x] = (im * H + 2 * yy) * W + 2 * xx;
            tcoord[threadIdx.x] = tr < ntiles ? (im << 20) | (2 * yy << 10) | (2 * xx) : -1;
        }
        // (an inline constant per register: hipcc zeroes ONE register and copies it 127 times - copies out of a register
        // that is a load target elsewhere in the loop, which tests/test_isa_guards.py cannot tell from a stale-operand bug)
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int v = 0; v < 16; ++v) asm volatile("v_mov_b32 %0, 0" : "=v"(acc[a][v]));
        HSTAMP(1);
        for (int c = 0; c + 1 < nchunks_block; ++c, ++g) chunk(g & 1, 4 * c, bdesc, 4 * c + 4);
        chunk(g & 1, 4 * (nchunks_block - 1), bdesc_n, 0);     // ... and on to the first step of the next block
        float *const ep = stage + (g & 1) * BUF;              // the epilogue's LDS: the buffer this chunk has finished reading
        ++g;
        HSTAMP(2);

        // ---- output transform: this half's share of Y = A^T M A, exchange with the partner wave ----------------------
        const int ncol = (nt * 2 + cg) * 32 + l31;
        // what the epilogue needs from memory is requested here and lands during the exchange: a wait behind the gate /
        // output stores would be a wait for those stores too (loads and stores share vmcnt and may retire out of order)
        float bv = 0.f;
        if (P.bias) asm volatile("global_load_dword %0, %1, off" : "=v"(bv) : "v"(P.bias + ncol) : "memory");
        [[maybe_unused]] float cpv[8];
        [[maybe_unused]] bool lstm_full = false;
        if constexpr (EPI == RNH_EPI_LSTM) {
            // phase 2 items of this thread: pixel (lane >> 4) of tiles wave, wave + 4, ..., hidden channel nt * 16 + (lane & 15)
            lstm_full = m0 + TILES <= ntiles && !(H & 1) && !(W & 1) && nt * 16 + 16 <= P.hd;
#pragma unroll
            for (int q = 0; q < 8; ++q) cpv[q] = 0.f;
            if (lstm_full && P.c_prev) {
                const int p2 = lane >> 4;
                const float *cpb = P.c_prev + (long)((p2 >> 1) * W + (p2 & 1)) * P.hd + nt * 16 + (lane & 15);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    asm volatile("global_load_dword %0, %1, off" : "=v"(cpv[q]) : "v"(cpb + (long)tpix[wave + 4 * q] * P.hd) : "memory");
            }
        }
        f32x4w *px = reinterpret_cast<f32x4w *>(ep);
        float Yf[8][4];                                         // entries 8h .. 8h + 7: tiles 16h .. 16h + 15 of the block
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
                if constexpr (hh == 0) {                        // columns 0, 1 of the domain
                    Y[0] = s0[0] + s0[1]; Y[1] = s0[1]; Y[2] = s1[0] + s1[1]; Y[3] = s1[1];
                } else {                                        // columns 2, 3
                    Y[0] = s0[0]; Y[1] = -s0[0] - s0[1]; Y[2] = s1[0]; Y[3] = -s1[0] - s1[1];
                }
            };
#pragma unroll
            for (int e = 0; e < 8; ++e) {                       // the partner's entries
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
        // (the prefetched weights of the next block's first step are older than these loads: they have landed too, and the
        // chunk's own wait for them will not have to sit out the stores of this epilogue)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bv), "+v"(cpv[0]), "+v"(cpv[1]), "+v"(cpv[2]), "+v"(cpv[3]), "+v"(cpv[4]), "+v"(cpv[5]), "+v"(cpv[6]), "+v"(cpv[7]),
                     "+v"(u0[0]), "+v"(u0[1]), "+v"(u0[2]), "+v"(u0[3]));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f32x4w y4 = px[((wave ^ 1) * 8 + e) * 64 + lane];
            Yf[e][0] += y4.x + bv; Yf[e][1] += y4.y + bv; Yf[e][2] += y4.z + bv; Yf[e][3] += y4.w + bv;
        }
        HSTAMP(3);
        // tile row (0..31) of entry e of this wave
        auto trl_of = [&](int e) { const int v = 8 * h + e; return (v & 3) + 8 * (v >> 2) + 4 * kh; };

        if constexpr (EPI == RNH_EPI_LSTM) {
            const int hd = P.hd;
            const bool full = lstm_full;
            float *xg = ep;                                     // the gates take the place of the partial outputs
            // phase 1: lanes 0..15 / 16..31 of a row block hold gates 2cg / 2cg + 1 (i, f | o, g) of 16 hidden channels;
            // sigmoid, and tanh as 2 sigmoid(2x) - 1 for the candidate gate, in one form: m rcp(1 + exp(-m x)) + b
            const int gate = 2 * cg + (l31 >> 4), ch = l31 & 15, hc = nt * 16 + ch;
            const float gm = gate == 3 ? 2.f : 1.f, gb = gate == 3 ? -1.f : 0.f;
            float *xw = xg + gate * H_GS + ch;
            const int ch2 = lane & 15, p2 = lane >> 4, hc2 = nt * 16 + ch2, poff2 = (p2 >> 1) * W + (p2 & 1);
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int p = 0; p < 4; ++p) Yf[e][p] = __builtin_fmaf(gm, __builtin_amdgcn_rcpf(1.f + __expf(-gm * Yf[e][p])), gb);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has read its partner's partial outputs
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int p = 0; p < 4; ++p) xw[trl_of(e) * H_TS + p * 16] = Yf[e][p];
            if (P.gates_out) {
                if (full) {
                    float *gb0 = P.gates_out + gate * hd + hc;
                    const long rowg = (long)W * 4 * hd;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float *gp = gb0 + (long)tpix[trl_of(e)] * 4 * hd;
                        gp[0] = Yf[e][0]; gp[4 * hd] = Yf[e][1]; gp[rowg] = Yf[e][2]; gp[rowg + 4 * hd] = Yf[e][3];
                    }
                } else {
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
            }
            // the gates are in LDS: wait for the LDS writes only (not for the gates_out stores)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            HSTAMP(4);
            const float *xr = xg + p2 * 16 + ch2;
            float gq[8][4];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int t = wave + 4 * q;
                gq[q][0] = xr[0 * H_GS + t * H_TS]; gq[q][1] = xr[1 * H_GS + t * H_TS]; gq[q][2] = xr[2 * H_GS + t * H_TS]; gq[q][3] = xr[3 * H_GS + t * H_TS];
            }
            int tp[8], tcq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) tp[q] = tpix[wave + 4 * q], tcq[q] = tcoord[wave + 4 * q];
            // the next chunk writes this buffer in its third step and the next block's set-up rewrites tpix / tcoord
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const long o = ((long)tp[q] + poff2) * hd + hc2;
                if (!full) {
                    const int yy = (tcq[q] >> 10) & 1023, xx = tcq[q] & 1023;
                    if (tcq[q] < 0 || hc2 >= hd || yy + (p2 >> 1) >= H || xx + (p2 & 1) >= W) continue;
                    cpv[q] = P.c_prev ? P.c_prev[o] : 0.f;
                }
                const float cn = gq[q][1] * cpv[q] + gq[q][0] * gq[q][3];
                P.c_out[o] = cn;
                P.h_out[o] = gq[q][2] * h_tanh(cn);
            }
            HSTAMP(5);
        } else {
            int tp[8], tcq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) tp[e] = tpix[trl_of(e)], tcq[e] = tcoord[trl_of(e)];
            // the next chunk writes this buffer in its third step and the next block's set-up rewrites tpix / tcoord
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if constexpr (EPI == RNH_EPI_PS) {
                // column n = (i*r + j)*cq + c  ->  pixel (r*y + i, r*x + j), channel c of the (B, rH, rW, cq) destination
                const int r = P.ps_r, cq = P.ps_cq;
                if (ncol < cq * r * r) {
                    const int sub = ncol / cq, c = ncol - sub * cq, pi = sub / r, pj = sub - pi * r;
                    float *dp = P.dst[0].ptr + c;
                    const long Wr = (long)W * r;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int tc = tcq[e];
                        if (tc < 0) continue;
                        const int im = tc >> 20, yy = (tc >> 10) & 1023, xx = tc & 1023;
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const int y = yy + (p >> 1), x = xx + (p & 1);
                            if (y >= H || x >= W) continue;
                            dp[(((long)im * H + y) * r + pi) * Wr * cq + ((long)x * r + pj) * cq] = Yf[e][p];
                        }
                    }
                }
            } else {
                // destination segment of this lane's column
                int seg = -1, cbase = 0;
                for (int d = 0; d < P.ndst; ++d) {
                    if (seg < 0 && ncol < cbase + P.dst[d].ncols) seg = d;
                    if (seg < 0) cbase += P.dst[d].ncols;
                }
                if (seg >= 0) {
                    const rnh_dst_t &D = P.dst[seg];
                    float *dp = D.ptr + (long)D.img_off * H * W * D.C + D.c0 + (ncol - cbase);
                    const bool full = m0 + TILES <= ntiles && !(H & 1) && !(W & 1);
                    if (full) {                                  // no per-element predicates
                        const long rowC = (long)W * D.C;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float *o = dp + (long)tp[e] * D.C;
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
                            const int tc = tcq[e];
                            if (tc < 0) continue;
                            const int yy = (tc >> 10) & 1023, xx = tc & 1023;
#pragma unroll
                            for (int p = 0; p < 4; ++p) {
                                if (yy + (p >> 1) >= H || xx + (p & 1) >= W) continue;
                                float *o = dp + ((long)tp[e] + (p >> 1) * W + (p & 1)) * D.C;
                                *o = D.accumulate ? *o + Yf[e][p] : Yf[e][p];
                            }
                        }
                    }
                }
            }
            HSTAMP(5);
        }
        mt = mt_n;
        nt = nt_n;
        bdesc = bdesc_n;
    }
    // the pipeline is one chunk ahead: let its loads land before the wave gives its registers back
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

}  // namespace

#ifdef RNH_STAMPS
extern "C" int rnh_debug_wino2_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino2_stamps), sizeof(g_wino2_stamps));
}
#endif

extern "C" int rnh_conv_wino2(const rnh_conv_args_t *args, void *stream) {
    if (!args) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: null args");
    const rnh_conv_args_t &a = *args;
    if (a.nsrc < 1 || a.nsrc > RNH_MAX_SRC || a.B < 1 || a.H < 1 || a.W < 1 || !a.wp) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: bad arguments");
    if (a.ntaps != 9) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: 3x3 convolutions only");
    if (a.Npad < 64 || a.Npad % 64) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: Npad must be a multiple of 64");
    int steps = 0;
    for (int i = 0; i < a.nsrc; ++i) {
        if (int rc = rnh_check_src(a.src[i], "rnh_conv_wino")) return rc;
        if (a.src[i].scale != a.src[0].scale || a.src[i].ptr2) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: one scale for all sources, no second pointer");
        if (a.src[i].nch & 15) RNH_FAIL(RNH_E_ALIGN, "rnh_conv_wino: source channel counts must be multiples of 16");
        steps += a.src[i].nch / 4;
    }
    if (steps != a.nk) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: nk = %d but the sources hold %d steps of 4 channels", a.nk, steps);
    const int TY = (a.H + 1) / 2, TX = (a.W + 1) / 2;
    const long ntiles = (long)a.B * TY * TX;
    if (a.H > 1023 || a.W > 1023 || a.B > 2047) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: at most 2047 images of 1023 x 1023");
    if (ntiles * 4 >= (1L << 29)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: too many pixels for 32-bit offsets");
    // pixel offsets inside a block (it may straddle two images) go through 24-bit multiplies
    if ((long)a.H * a.W * a.src[0].scale * a.src[0].scale >= (1L << 22)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: source images of at most 2^22 pixels");
    const int MT = (int)((ntiles + H_TILES - 1) / H_TILES), NT = a.Npad / 64;
    hipStream_t st = (hipStream_t)stream;
    // persistent workgroups: two per CU, each takes every gridDim.x-th block of the MT * NT list
    static int resident = 0;
    if (!resident) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: cannot read the CU count of the device");
        resident = 2 * cus;
    }
    const long nblocks = (long)MT * NT;
    const dim3 grid((unsigned)(nblocks < resident ? nblocks : resident)), block(256);
    switch (a.epilogue) {
        case RNH_EPI_STORE:
            if (a.ndst < 1 || a.ndst > RNH_MAX_DST) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: bad destination count");
            for (int d = 0; d < a.ndst; ++d)
                if (!a.dst[d].ptr || a.dst[d].ncols < 1) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: bad destination %d", d);
            hipLaunchKernelGGL((conv_winoh_kernel<RNH_EPI_STORE>), grid, block, 0, st, a, MT, NT, TX, TY);
            break;
        case RNH_EPI_PS:
            if (a.ndst != 1 || !a.dst[0].ptr || a.ps_r < 1 || a.ps_cq < 1 || a.ps_cq * a.ps_r * a.ps_r > a.Npad)
                RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: bad pixel-shuffle destination");
            hipLaunchKernelGGL((conv_winoh_kernel<RNH_EPI_PS>), grid, block, 0, st, a, MT, NT, TX, TY);
            break;
        case RNH_EPI_LSTM:
            if (!a.h_out || !a.c_out || a.hd < 1 || !a.bias) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: LSTM epilogue needs h_out, c_out, hd, bias");
            if (a.Npad != 64 * ((a.hd + 15) / 16)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: LSTM column layout (plans.lstm_colmap64)");
            hipLaunchKernelGGL((conv_winoh_kernel<RNH_EPI_LSTM>), grid, block, 0, st, a, MT, NT, TX, TY);
            break;
        default:
            RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: epilogue %d not available", a.epilogue);
    }
    RNH_CHECK_LAUNCH("rnh_conv_wino");
    return 0;
}
