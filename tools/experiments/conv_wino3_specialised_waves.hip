// EXPERIMENT, not built into librefinenet_hip.so (round 2): rnh_conv_wino2 with specialised waves.  Correct (it passed the Winograd parity
// tests of tests/test_hip_parity.py when it was wired in) but slower than csrc/conv_wino2.hip; kept for the measurements
// quoted in that file's header and in DESIGN.md section 7.  To try it: copy to csrc/, add to build.sh, declare the entry in lib.py.
//
// 3x3 convolution (padding 1) in Winograd form F(2x2, 3x3) on fp32 MFMA for gfx950 with specialised waves - rnh_conv_wino.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 independent GEMMs (one per position xi of the 4x4 transform domain) of [tiles x C] x [C x N].  In conv_wino.hip every
// wave does everything - stages and transforms the input, feeds the matrix cores, runs the epilogue - and owns a whole
// SIMD (512 registers): whatever it does beside its MFMAs is matrix-core idle time (0.56 of the fp32 peak on the ConvLSTM
// cell).  Two such workgroups per CU with half the accumulators each (conv_wino2.hip) hide part of that behind each other:
// 0.62.  Here the work is split by ROLE inside one persistent workgroup of 8 waves per CU:
//
//   * waves 0-3, one per SIMD, are CONSUMERS: wave (h, cg) owns half h of the transform domain (positions xi = 4 i + j with
//     j in {2h, 2h+1}) of 32 tiles x 32 columns (128 accumulators) - a workgroup computes 32 tiles x 64 columns per block.
//     Their instruction stream is ds_read_b128 (staged input transform), buffer_load_dwordx4 (pre-transformed weights from
//     L2, one step ahead) and MFMAs; per block they add their share of the output transform (partial 2x2 outputs, the
//     two halves are not yet summed) and park it in LDS - 16 ds_write_b128 - and go on with the next block.
//   * waves 4-7, the other wave of each SIMD, are PRODUCERS: they load the 4x4 patches of the next chunk (thread = (tile,
//     channel pair): 16 raw 8-byte buffer loads, out-of-image lanes carry offset -1 and read the zero padding), compute
//     B^T d B (32 packed adds) and write it to the LDS buffer the consumers will read next; and they run the EPILOGUE of
//     the previous block out of the parked partial outputs while the consumers are already multiplying the next one:
//     sum of the two halves, bias, then store / pixel-shuffle store / the ConvLSTM gate math (c' = f c + i g,
//     h' = o tanh c') with all its loads and stores.  They have a whole chunk (4 k cycles) for 600 cycles of work.
//
// One s_barrier per 16-channel chunk couples the roles (consumers: every read of the chunk's buffer has landed; producers:
// the next chunk is in the other buffer); the pipeline runs across blocks (the first chunk of the next block is staged
// during the last chunk of this one), so there is no per-block set-up in the consumers at all.  Operand layouts are those
// of conv_wino.hip: LDS [xi / 2][tile][36] with (xi even, xi odd) x 2 channels per 16 bytes, weights
// U[step][xi / 2][n][lane half][xi & 1][2] from rnh_wino_pack_weights; one ds_read_b128 and one buffer_load_dwordx4 feed
// four MFMAs.  ConvLSTM column order: plans.lstm_colmap64 (a block's 64 columns = the 4 gates of 16 hidden channels).
//
// Same operand conventions as rnh_conv_igemm (rnh_conv_args_t: multi-source K without concatenation, destination
// segments, packed bias, pixel-unshuffled sources of one common scale).  Epilogues: STORE, PS, LSTM.
#include "rnh_common.h"
#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4w __attribute__((ext_vector_type(4)));

// v_exp_f32 / v_rcp_f32 (1 ulp each) and no branch
__device__ __forceinline__ float s_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float s_tanh(float x) {
    const float ax = fabsf(x);
    const float big = __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * ax)), 1.f);
    const float small = ax * __builtin_fmaf(-0.33333334f * ax, ax, 1.f);      // |x| < 0.04: the exp form cancels
    const float t = ax < 0.04f ? small : big;
    return copysignf(t, x);
}

// the buffer descriptor (base + 2 GiB window, raw buffer) as a plain SGPR quadruple for the asm loads
__device__ __forceinline__ i32x4 sdesc3(const float *p) {
    const unsigned long long u = (unsigned long long)p;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    d[1] = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
    d[2] = 0x7fffffff;
    d[3] = 0x00020000;
    return d;
}

#ifdef RNH_STAMPS
__device__ unsigned long long g_wino3_stamps[20];
#ifndef RNH_STAMP_BLOCK
#define RNH_STAMP_BLOCK 0
#endif
#endif

constexpr int S_TILES = 32, S_CPC = 8, S_CH = 16, S_CHS = 4 * S_CPC + 4, S_BUF = 8 * S_TILES * S_CHS;   // floats per staging buffer
// hand-off of the partial outputs: [consumer wave][accumulator register v][lane][4 pixels], 260 floats per (wave, v): the
// consumers write 16 contiguous bytes per lane; a producer wave reads the lanes of 8 tiles x 8 column slots, 32 bytes
// apart inside a tile (two tiles per LDS pass, 1040 bytes apart: they interleave in the banks)
constexpr int S_EP = 260, S_HO = 4 * 16 * S_EP;

template <int EPI>
__global__ void __launch_bounds__(512, 1) conv_winos_kernel(const rnh_conv_args_t P, const int MT, const int NT, const int TX, const int TY) {
    constexpr int TILES = S_TILES, CPC = S_CPC, CH = S_CH, CHS = S_CHS, BUF = S_BUF;
    __shared__ __attribute__((aligned(16))) float stage[2 * BUF];   // 73.7 KB: the staged input transform, two chunks
    __shared__ __attribute__((aligned(16))) float ho[S_HO];          // 66.6 KB: partial outputs of the block just finished
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = P.H, W = P.W, ntiles = P.B * TY * TX, nblocks = MT * NT;
    const int my_blocks = (nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // >= 1: the grid never exceeds the list
    // block i of this workgroup (i wraps: the pipeline stages one block past the end)
    auto block_of = [&](int i, int &mt_, int &nt_) __attribute__((always_inline)) {
        const int bid = rnh_xcd_remap((int)blockIdx.x + (i % my_blocks) * (int)gridDim.x, nblocks);
        mt_ = bid / NT;
        nt_ = bid - mt_ * NT;
    };
    int nchunks_block = 0;
    for (int i = 0; i < P.nsrc; ++i) nchunks_block += P.src[i].nch / CH;       // >= 2 (checked by the launcher)
    const int pstride = P.Npad * 32;                        // bytes between two PAIRS of transform positions of one step

    if (wave < 4) {
        // =========================================== consumers ========================================================
#ifndef RNH_X_PRIO_C
#define RNH_X_PRIO_C 0
#endif
        __builtin_amdgcn_s_setprio(RNH_X_PRIO_C);
        const int l31 = lane & 31, kh = lane >> 5, h = wave & 1, cg = wave >> 1;
        // this wave's pairs of transform positions: pr = 2 i + h (row i of the 4x4 domain, columns 2h and 2h + 1)
        int boffx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) boffx[i] = (l31 * 2 + kh) * 16 + (2 * i + h) * pstride;
        auto wdesc_of = [&](int nt_) __attribute__((always_inline)) { return sdesc3(P.wp + (long)((nt_ * 2 + cg) * 32) * 8); };   // this wave's 32 columns of column block nt_
        // All vector-memory and LDS reads of the loop are volatile asm with hand-counted waits (see conv_wino.hip); every
        // wait names the registers it covers as "+v" operands, which orders their uses behind it.  The s_nop covers the 5
        // wait states between an SALU / v_readfirstlane write of an SGPR and a VMEM instruction reading it.
        auto loadb = [&](f32x4w *u, const i32x4 &bd, int sb) __attribute__((always_inline)) {   // transformed weights of step sb: 4 loads of 16 bytes
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
        auto loadv = [&](f32x4w *V, int buf, int q, auto a_tag) __attribute__((always_inline)) {
            constexpr int a = decltype(a_tag)::value;
            const unsigned adr = vlane + buf * BUF * 4 + q * 32;
            asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(V[0]) : "v"(adr), "i"((4 * a) * TILES * CHS * 4) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(V[1]) : "v"(adr), "i"((4 * a + 2) * TILES * CHS * 4) : "memory");
        };
        auto wait_lds = [&](f32x4w *V) __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V[0]), "+v"(V[1])); };
        auto wait_vm = [&](f32x4w *u, auto keep) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(%c4)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]) : "i"(decltype(keep)::value));
        };
        f32x16 acc[8];                                          // acc[2 i + odd] = position (row i, column 2h + odd)
        auto compute = [&](const f32x4w *V, const f32x4w *u, auto a_tag) __attribute__((always_inline)) {      // V: the two pairs of half a, u: all four pairs of the step
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
        f32x4w Va[2], Vb[2], u0[4], u1[4];                        // staged operands by half steps, weights by steps
#ifdef RNH_STAMPS
        unsigned long long t_barrier = 0;
#endif
        using A0 = std::integral_constant<int, 0>;
        using A1 = std::integral_constant<int, 1>;
        using K4 = std::integral_constant<int, 4>;               // the 4 weight loads of ONE step may stay in flight
        using K0 = std::integral_constant<int, 0>;
        // The chunk body has no branch: a register that is the target of an asynchronous asm load must have exactly one
        // definition per iteration (tests/test_isa_guards.py).  There is no "last chunk" form either: the last chunk of
        // the last block prefetches a chunk that is never computed (the producers stage block 0 of the workgroup again).
        // LDS operands run half a step (8 MFMAs) ahead in two register pairs, weights one step (16 MFMAs) ahead.
        // s: step index of the chunk's first step in the block's weights; (bd_next, s_next): where the step after the chunk is.
        auto chunk = [&](const int buf, const i32x4 &bd, const int s, const i32x4 &bd_next, const int s_next) __attribute__((always_inline)) {
            wait_lds(Va);
            loadv(Vb, buf, 0, A1());
            wait_vm(u0, K0());
            loadb(u1, bd, s + 1);
            compute(Va, u0, A0());
            wait_lds(Vb);
            loadv(Va, buf, 1, A0());
            compute(Vb, u0, A1());
            // step 1
            wait_lds(Va);
            loadv(Vb, buf, 1, A1());
            loadb(u0, bd, s + 2);
            wait_vm(u1, K4());
            compute(Va, u1, A0());
            wait_lds(Vb);
            loadv(Va, buf, 2, A0());
            compute(Vb, u1, A1());
            // step 2
            wait_lds(Va);
            loadv(Vb, buf, 2, A1());
            loadb(u1, bd, s + 3);
            wait_vm(u0, K4());
            compute(Va, u0, A0());
            wait_lds(Vb);
            loadv(Va, buf, 3, A0());
            compute(Vb, u0, A1());
            // last step: the chunk's barrier in front of its last 8 MFMAs (every read of this buffer has landed; the
            // producers arrive when the next chunk is in the other buffer), the first operands of the next chunk are
            // fetched under their cover
            wait_lds(Va);
            loadv(Vb, buf, 3, A1());
            loadb(u0, bd_next, s_next);
            wait_vm(u1, K4());
            compute(Va, u1, A0());
            wait_lds(Vb);
#ifdef RNH_STAMPS
            const unsigned long long tb0 = __builtin_readcyclecounter();
#endif
            asm volatile("s_barrier" ::: "memory");
#ifdef RNH_STAMPS
            t_barrier += __builtin_readcyclecounter() - tb0;
#endif
            loadv(Va, buf ^ 1, 0, A0());
            compute(Vb, u1, A1());
        };

        int mt, nt, mt_n, nt_n;
        block_of(0, mt, nt);
        i32x4 bdesc = wdesc_of(nt);
        loadb(u0, bdesc, 0);
        asm volatile("s_barrier" ::: "memory");                   // the first chunk is staged
        loadv(Va, 0, 0, A0());
        int g = 0;                                                // chunks done: chunk g lives in LDS buffer g & 1
        f32x4w *const hw = reinterpret_cast<f32x4w *>(ho) + wave * 16 * (S_EP / 4) + lane;
        for (int k = 0; k < my_blocks; ++k) {
#ifdef RNH_STAMPS
            if (blockIdx.x == RNH_STAMP_BLOCK && threadIdx.x == 0 && k < 8) g_wino3_stamps[k] = __builtin_readcyclecounter();
#endif
            block_of(k + 1, mt_n, nt_n);
            const i32x4 bdesc_n = wdesc_of(nt_n);
            // (an inline constant per register: hipcc zeroes ONE register and copies it 127 times - copies out of a register
            // that is a load target elsewhere in the loop, which tests/test_isa_guards.py cannot tell from a stale operand)
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int v = 0; v < 16; ++v) asm volatile("v_mov_b32 %0, 0" : "=v"(acc[a][v]));
            for (int c = 0; c + 1 < nchunks_block; ++c, ++g) chunk(g & 1, bdesc, 4 * c, bdesc, 4 * c + 4);
            chunk(g & 1, bdesc, 4 * (nchunks_block - 1), bdesc_n, 0);     // ... and on to the first step of the next block
            ++g;
#ifdef RNH_STAMPS
            if (blockIdx.x == RNH_STAMP_BLOCK && threadIdx.x == 0 && k < 8) g_wino3_stamps[8 + k] = __builtin_readcyclecounter();
            if (blockIdx.x == RNH_STAMP_BLOCK && threadIdx.x == 0 && k == 7) g_wino3_stamps[15] = t_barrier;
#endif
            // this half's share of Y = A^T M A for the 16 tiles x 4 pixels of a lane -> LDS.  (The producers have read the
            // previous block's partial outputs during the chunks of this one; the next chunk's barrier publishes these.)
            auto park = [&](auto h_tag) __attribute__((always_inline)) {
                constexpr int hh = decltype(h_tag)::value;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    float s0[2], s1[2];
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        s0[o] = acc[0 + o][v] + acc[2 + o][v] + acc[4 + o][v];
                        s1[o] = acc[2 + o][v] - acc[4 + o][v] - acc[6 + o][v];
                    }
                    f32x4w y4;
                    if constexpr (hh == 0) {                    // columns 0, 1 of the domain
                        y4 = f32x4w{s0[0] + s0[1], s0[1], s1[0] + s1[1], s1[1]};
                    } else {                                    // columns 2, 3
                        y4 = f32x4w{s0[0], -s0[0] - s0[1], s1[0], -s1[0] - s1[1]};
                    }
                    hw[v * (S_EP / 4)] = y4;
                }
            };
            if (h == 0) park(std::integral_constant<int, 0>());
            else park(std::integral_constant<int, 1>());
            mt = mt_n;
            nt = nt_n;
            bdesc = bdesc_n;
        }
        // the last block's partial outputs are in LDS; the pipeline is one chunk ahead: let its loads land
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        return;
    }

    // =============================================== producers ==========================================================
#ifndef RNH_X_PRIO_P
#define RNH_X_PRIO_P 3
#endif
    // the producers' short bursts go first: issue arbitration is strict (at equal priority the older wave - a consumer -
    // wins every time it has an instruction ready, and a producer behind it is served only while the consumer waits)
    __builtin_amdgcn_s_setprio(RNH_X_PRIO_P);
    const int ptid = threadIdx.x - 256;                       // 0..255
    const int ts = ptid / CPC, cp = ptid % CPC;               // staging: (tile, channel pair of the chunk); epilogue: (tile, column slot)
    const int sc = P.src[0].scale, Hs = H * sc, Ws = W * sc;
    // the thread's 4x4 patch of the block being STAGED: pixel offset of its top-left corner (relative to base_pix) and a
    // 16-bit mask of the pixels inside the image (the 16 byte offsets are rebuilt from these per chunk)
    int ld_block = 0, base_pix = 0, pix00 = 0, okmask = 0;
    auto stage_block = [&](int i) __attribute__((always_inline)) {
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
    int si = 0, cchunk = 0, nchunk = P.src[0].nch / CH, C4 = 0;
    i32x4 adesc;
    auto setup_src = [&](int sidx) __attribute__((always_inline)) {
        const rnh_src_t &S = P.src[sidx];
        adesc = sdesc3(S.ptr + S.c0 + ((long)S.img_off * Hs * Ws + base_pix + S.sub_y * Ws + S.sub_x) * S.C);
        C4 = S.C * 4;
        nchunk = S.nch / CH;
    };
    // Lanes outside the image load the patch's pixel (1, 1) - the tile's own top-left output pixel, always inside - and
    // are zeroed behind the wait: with the usual out-of-range offset a load whose 64 lanes are all outside never goes to
    // memory and returns AHEAD of older loads, which a counted wait cannot tolerate (the producers keep two chunks in flight).
    auto patch_offsets = [&](int p0, int *vo) __attribute__((always_inline)) {
        const int centre = __mul24(pix00 + (Ws + 1) * sc, C4) + cp * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int p = p0 + q;
            const int off = __mul24(pix00 + ((p >> 2) * Ws + (p & 3)) * sc, C4) + cp * 8;   // < 2^24 pixels per block, < 2^24 bytes per pixel
            vo[q] = (okmask >> p) & 1 ? off : centre;
        }
    };
    stage_block(0);
    setup_src(0);
    auto ld8 = [&](f32x2 *dst, const int *vo, const i32x4 &desc_, int soff) __attribute__((always_inline)) {
        // (hipcc's divergence analysis loses track of the descriptor across the producer loop and would hand a VGPR
        // quadruple to the "s" operand)
        i32x4 desc;
#pragma unroll
        for (int i = 0; i < 4; ++i) desc[i] = __builtin_amdgcn_readfirstlane(desc_[i]);
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
    // two chunks in flight: register sets A and B take turns (chunk g + 2 is requested when chunk g + 1 has been transformed
    // out of the same set), so a chunk has a whole chunk time (4 k cycles) to arrive.  mask: the set's in-image bits.
    f32x2 stgA[16], stgB[16];
    int maskA = 0, maskB = 0;
    auto gload = [&](f32x2 *stg, int &mask) __attribute__((always_inline)) {              // next chunk of the source list -> registers (16 loads)
        const int soff = __builtin_amdgcn_readfirstlane(cchunk * CH * 4);
        int vo[8];
        patch_offsets(0, vo);
        ld8(stg, vo, adesc, soff);
        patch_offsets(8, vo);
        ld8(stg + 8, vo, adesc, soff);
        mask = okmask;
        if (++cchunk == nchunk) {
            cchunk = 0;
            if (++si == P.nsrc) {                           // the source list of the block is through: on to the next block
                si = 0;
                stage_block(++ld_block);
            }
            setup_src(si);
        }
    };
    // Counted wait: at most the 16 youngest vector-memory operations stay in flight.  Loads complete in order, so whatever
    // the stores of an epilogue in between do, the set named here (older than the 16 youngest loads) has landed.
    auto wait_stage = [&](f32x2 *stg, auto keep) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%c16)"
                     : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[3]), "+v"(stg[4]), "+v"(stg[5]), "+v"(stg[6]), "+v"(stg[7]),
                       "+v"(stg[8]), "+v"(stg[9]), "+v"(stg[10]), "+v"(stg[11]), "+v"(stg[12]), "+v"(stg[13]), "+v"(stg[14]),
                       "+v"(stg[15])
                     : "i"(decltype(keep)::value));
    };
    using K16 = std::integral_constant<int, 16>;
    using K0 = std::integral_constant<int, 0>;
    auto xform_store = [&](const f32x2 *stg, int mask, int buf) __attribute__((always_inline)) {      // V = B^T d B on the thread's two channels, to LDS
        auto sub = [&](f32x2 a, f32x2 b) __attribute__((always_inline)) {                 // one v_pk_add_f32 (hipcc scalarises packed adds / subtractions)
            f32x2 r;
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            return r;
        };
        auto add = [&](f32x2 a, f32x2 b) __attribute__((always_inline)) {
            f32x2 r;
            asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            return r;
        };
        f32x2 d[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const bool in = (mask >> p) & 1;
            d[p] = f32x2{in ? stg[p].x : 0.f, in ? stg[p].y : 0.f};
        }
        f32x2 tq[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tq[0 * 4 + j] = sub(d[0 * 4 + j], d[2 * 4 + j]);
            tq[1 * 4 + j] = add(d[1 * 4 + j], d[2 * 4 + j]);
            tq[2 * 4 + j] = sub(d[2 * 4 + j], d[1 * 4 + j]);
            tq[3 * 4 + j] = sub(d[1 * 4 + j], d[3 * 4 + j]);
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

    // ---- epilogue of one block out of the parked partial outputs: thread = (tile ts of the block, column slot cp) -------
    // partial outputs of column (cg, l31) of tile ts: consumer waves 2 cg and 2 cg + 1, accumulator register v, lane lt + l31
    const int ev = (ts & 3) + 4 * (ts >> 3), lt = ((ts >> 2) & 1) * 32;
    auto partial = [&](int cg, int l31) __attribute__((always_inline)) {                    // 4 pixels of one output column: the two halves summed
        const f32x4w a = *reinterpret_cast<const f32x4w *>(ho + ((2 * cg) * 16 + ev) * S_EP + (lt + l31) * 4);
        const f32x4w b = *reinterpret_cast<const f32x4w *>(ho + ((2 * cg + 1) * 16 + ev) * S_EP + (lt + l31) * 4);
        return a + b;
    };
    // What the epilogue of a block needs from memory - bias of the thread's 8 columns, previous cell state of its 8 items -
    // is requested one chunk ahead (epi_request, behind the staging loads of that chunk) into eb[] / ec[]: asm loads from
    // clamped, always valid addresses; predicates are applied to the values.  (Plain loads here would make hipcc wait
    // vmcnt(0), i.e. for the chunk requested a moment ago, once per block.)
    float eb[8], ec[8];
    auto epi_request = [&](int kb) __attribute__((always_inline)) {
        int mt, nt;
        block_of(kb, mt, nt);
        const int tr = min(mt * TILES + ts, ntiles - 1);
        const int im = tr / (TY * TX), rr = tr - im * TY * TX, yy = 2 * (rr / TX), xx = 2 * (rr - (rr / TX) * TX);
        const long pix = ((long)im * H + yy) * W + xx;
        const float *bsrc = P.bias ? P.bias : P.wp;             // (no bias: any valid address, the values are not used)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // LSTM: gate j >> 1, channel 2 cp + (j & 1); otherwise column group j >> 2, column 4 cp + (j & 3)
            const int col = EPI == RNH_EPI_LSTM ? (nt * 2 + (j >> 2)) * 32 + ((j >> 1) & 1) * 16 + 2 * cp + (j & 1)
                                                : (nt * 2 + (j >> 2)) * 32 + 4 * cp + (j & 3);
            asm volatile("global_load_dword %0, %1, off" : "=v"(eb[j]) : "v"(bsrc + col) : "memory");
        }
        if constexpr (EPI == RNH_EPI_LSTM) {
            const int hd = P.hd, hc0 = nt * 16 + 2 * cp;
            const float *csrc = P.c_prev ? P.c_prev : P.c_out;  // (no previous state: any valid address)
#pragma unroll
            for (int j = 0; j < 8; ++j) {                       // item j: channel j >> 2, pixel j & 3
                const int p = j & 3, e = j >> 2;
                const bool okp = (!(p & 1) || xx + 1 < W) && (!(p >> 1) || yy + 1 < H);
                const int hc = hc0 + e < hd ? hc0 + e : 0;
                const long o = (pix + (okp ? (p >> 1) * W + (p & 1) : 0)) * hd + hc;
                asm volatile("global_load_dword %0, %1, off" : "=v"(ec[j]) : "v"(csrc + o) : "memory");
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) ec[j] = 0.f;
        }
    };
    auto epi_wait = [](float *b, float *c, auto keep) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%c16)"
                     : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]),
                       "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
                     : "i"(decltype(keep)::value));
    };
    auto epilogue = [&](int kb) __attribute__((always_inline)) {
        int mt, nt;
        block_of(kb, mt, nt);
        const int tr = mt * TILES + ts;
        if (tr >= ntiles) return;
        const int im = tr / (TY * TX), rr = tr - im * TY * TX, yy = 2 * (rr / TX), xx = 2 * (rr - (rr / TX) * TX);
        const long pix = ((long)im * H + yy) * W + xx;          // top-left output pixel of the tile
        const bool okx = xx + 1 < W, oky = yy + 1 < H;
        const bool okp[4] = {true, okx, oky, okx && oky};
        const long poff[4] = {0, 1, W, (long)W + 1};
        if constexpr (EPI == RNH_EPI_LSTM) {
            // the block's 64 columns are the 4 gates (cg = gate >> 1, l31 = (gate & 1) * 16 + channel) of 16 hidden
            // channels; this thread: channels 2 cp and 2 cp + 1 of the block
            const int hd = P.hd, hc = nt * 16 + 2 * cp;
            if (hc >= hd) return;
            const bool two = hc + 1 < hd;                       // (hd may be odd: the second channel is padding then)
            float cpv[2][4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                cpv[0][p] = P.c_prev && okp[p] ? ec[p] : 0.f;
                cpv[1][p] = P.c_prev && okp[p] && two ? ec[4 + p] : 0.f;
            }
            float gt[4][2][4];                                  // [gate][channel][pixel]
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int l31 = (gi & 1) * 16 + 2 * cp + e;
                    const f32x4w y = partial(gi >> 1, l31) + eb[2 * gi + e];
#pragma unroll
                    for (int p = 0; p < 4; ++p) gt[gi][e][p] = gi == 3 ? s_tanh(y[p]) : s_sigmoid(y[p]);
                }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (!okp[p]) continue;
                const long o = (pix + poff[p]) * hd + hc;
                float cn[2], hn[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    cn[e] = gt[1][e][p] * cpv[e][p] + gt[0][e][p] * gt[3][e][p];
                    hn[e] = gt[2][e][p] * s_tanh(cn[e]);
                }
                if (two && !(hd & 1)) {                        // 8-byte stores: 8 threads write 64 contiguous bytes
                    *reinterpret_cast<f32x2 *>(P.c_out + o) = f32x2{cn[0], cn[1]};
                    *reinterpret_cast<f32x2 *>(P.h_out + o) = f32x2{hn[0], hn[1]};
                    if (P.gates_out) {
                        float *gp = P.gates_out + (pix + poff[p]) * 4 * hd + hc;
#pragma unroll
                        for (int gi = 0; gi < 4; ++gi) *reinterpret_cast<f32x2 *>(gp + gi * hd) = f32x2{gt[gi][0][p], gt[gi][1][p]};
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if (e == 1 && !two) continue;
                        P.c_out[o + e] = cn[e];
                        P.h_out[o + e] = hn[e];
                        if (P.gates_out) {
#pragma unroll
                            for (int gi = 0; gi < 4; ++gi) P.gates_out[(pix + poff[p]) * 4 * hd + gi * hd + hc + e] = gt[gi][e][p];
                        }
                    }
                }
            }
        } else {
            // this thread: columns 4 cp .. 4 cp + 3 of both 32-column groups of the block
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                const int ncol0 = (nt * 2 + cg) * 32 + 4 * cp;
                float y[4][4];                                  // [column][pixel]
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4w v = partial(cg, 4 * cp + j) + (P.bias ? eb[4 * cg + j] : 0.f);
#pragma unroll
                    for (int p = 0; p < 4; ++p) y[j][p] = v[p];
                }
                if constexpr (EPI == RNH_EPI_PS) {
                    // column n = (i*r + j)*cq + c  ->  pixel (r*y + i, r*x + j), channel c of the (B, rH, rW, cq) destination
                    const int r = P.ps_r, cq = P.ps_cq;
                    const long Wr = (long)W * r;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ncol = ncol0 + j;
                        if (ncol >= cq * r * r) continue;
                        const int sub = ncol / cq, c = ncol - sub * cq, pi = sub / r, pj = sub - pi * r;
#pragma unroll
                        for (int p = 0; p < 4; ++p)
                            if (okp[p])
                                P.dst[0].ptr[(((long)im * H + yy + (p >> 1)) * r + pi) * Wr * cq + ((long)(xx + (p & 1)) * r + pj) * cq + c] = y[j][p];
                    }
                } else {
                    // destination segment of the four columns (segments start at multiples of 4 columns: checked by the launcher)
                    int seg = -1, cbase = 0;
                    for (int d = 0; d < P.ndst; ++d) {
                        if (seg < 0 && ncol0 < cbase + P.dst[d].ncols) seg = d;
                        if (seg < 0) cbase += P.dst[d].ncols;
                    }
                    if (seg < 0) continue;
                    const rnh_dst_t &D = P.dst[seg];
                    const int nv = min(4, cbase + D.ncols - ncol0);       // columns of the four that exist
                    float *dp = D.ptr + (long)D.img_off * H * W * D.C + D.c0 + (ncol0 - cbase);
                    const bool vec = nv == 4 && !((D.C | D.c0) & 3);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        if (!okp[p]) continue;
                        float *o = dp + (pix + poff[p]) * D.C;
                        if (vec) {
                            f32x4w v = {y[0][p], y[1][p], y[2][p], y[3][p]};
                            if (D.accumulate) v += *reinterpret_cast<const f32x4w *>(o);
                            *reinterpret_cast<f32x4w *>(o) = v;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (j < nv) o[j] = D.accumulate ? o[j] + y[j][p] : y[j][p];
                        }
                    }
                }
            }
        }
    };

    // prologue: chunk 0 of the first block -> LDS buffer 0, chunks 1 and 2 on their way
    gload(stgA, maskA);
    gload(stgB, maskB);
    wait_stage(stgA, K16());
    xform_store(stgA, maskA, 0);
    gload(stgA, maskA);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // iteration g (chunk g is being multiplied out of buffer g & 1): chunk g + 1 -> the other buffer, chunk g + 3 requested;
    // in the first chunk of a block the memory operands of the previous block's epilogue are requested, in the second
    // (the barrier of the first has published the parked partial outputs) that epilogue runs
    const int total = my_blocks * nchunks_block;
    int g = 0, k = 0, c = 0;
#ifdef RNH_STAMPS
    unsigned long long tp[4] = {0, 0, 0, 0}, tq0 = 0;
#define PSTAMP(i) do { const unsigned long long now = __builtin_readcyclecounter(); tp[i] += now - tq0; tq0 = now; } while (0)
    tq0 = __builtin_readcyclecounter();
#else
#define PSTAMP(i)
#endif
    auto iteration = [&](f32x2 *stg, int &mask) __attribute__((always_inline)) {
        wait_stage(stg, K16());
        PSTAMP(0);
        xform_store(stg, mask, (g + 1) & 1);
        PSTAMP(1);
        const bool first = c == 0 && k > 0, second = c == 1 && k > 0;
        if (first) epi_request(k - 1);                          // (older than the chunk requested next: it lands with the 16 youngest in flight)
        gload(stg, mask);
        if (second) {
            epi_wait(eb, ec, K16());
            epilogue(k - 1);
        }
        PSTAMP(2);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        PSTAMP(3);
        ++g;
        if (++c == nchunks_block) c = 0, ++k;
#ifdef RNH_STAMPS
        if (blockIdx.x == RNH_STAMP_BLOCK && threadIdx.x == 256 && g == 64)
            for (int i = 0; i < 4; ++i) g_wino3_stamps[16 + i] = tp[i];
#endif
    };
    while (g + 1 < total) {
        iteration(stgB, maskB);
        iteration(stgA, maskA);
    }
    if (g < total) iteration(stgB, maskB);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");              // the consumers have parked the last block
    epi_request(my_blocks - 1);
    epi_wait(eb, ec, K0());                                                // (also lets the chunks requested past the end land)
    epilogue(my_blocks - 1);
}

}  // namespace

#ifdef RNH_STAMPS
extern "C" int rnh_debug_wino3_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wino3_stamps), sizeof(g_wino3_stamps));
}
#endif

extern "C" int rnh_conv_wino3(const rnh_conv_args_t *args, void *stream) {
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
    if (steps < 8) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: at least 32 input channels (two chunks: the epilogue of a block runs during the second chunk of the next)");
    const int TY = (a.H + 1) / 2, TX = (a.W + 1) / 2;
    const long ntiles = (long)a.B * TY * TX;
    if (a.H > 1023 || a.W > 1023 || a.B > 2047) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: at most 2047 images of 1023 x 1023");
    if (ntiles * 4 >= (1L << 29)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: too many pixels for 32-bit offsets");
    // pixel offsets inside a block (it may straddle two images) go through 24-bit multiplies
    if ((long)a.H * a.W * a.src[0].scale * a.src[0].scale >= (1L << 22)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: source images of at most 2^22 pixels");
    const int MT = (int)((ntiles + S_TILES - 1) / S_TILES), NT = a.Npad / 64;
    hipStream_t st = (hipStream_t)stream;
    // persistent workgroups: one per CU, each takes every gridDim.x-th block of the MT * NT list
    static int resident = 0;
    if (!resident) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: cannot read the CU count of the device");
        resident = cus;
    }
    const long nblocks = (long)MT * NT;
    const dim3 grid((unsigned)(nblocks < resident ? nblocks : resident)), block(512);
    switch (a.epilogue) {
        case RNH_EPI_STORE: {
            if (a.ndst < 1 || a.ndst > RNH_MAX_DST) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: bad destination count");
            int cb = 0;
            for (int d = 0; d < a.ndst; ++d) {
                if (!a.dst[d].ptr || a.dst[d].ncols < 1) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: bad destination %d", d);
                if (cb & 3) RNH_FAIL(RNH_E_ALIGN, "rnh_conv_wino: destination segments start at multiples of 4 columns");
                cb += a.dst[d].ncols;
            }
            hipLaunchKernelGGL((conv_winos_kernel<RNH_EPI_STORE>), grid, block, 0, st, a, MT, NT, TX, TY);
            break;
        }
        case RNH_EPI_PS:
            if (a.ndst != 1 || !a.dst[0].ptr || a.ps_r < 1 || a.ps_cq < 1 || a.ps_cq * a.ps_r * a.ps_r > a.Npad)
                RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: bad pixel-shuffle destination");
            hipLaunchKernelGGL((conv_winos_kernel<RNH_EPI_PS>), grid, block, 0, st, a, MT, NT, TX, TY);
            break;
        case RNH_EPI_LSTM:
            if (!a.h_out || !a.c_out || a.hd < 1 || !a.bias) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: LSTM epilogue needs h_out, c_out, hd, bias");
            if (a.Npad != 64 * ((a.hd + 15) / 16)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: LSTM column layout (plans.lstm_colmap64)");
            hipLaunchKernelGGL((conv_winos_kernel<RNH_EPI_LSTM>), grid, block, 0, st, a, MT, NT, TX, TY);
            break;
        default:
            RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: epilogue %d not available", a.epilogue);
    }
    RNH_CHECK_LAUNCH("rnh_conv_wino");
    return 0;
}
