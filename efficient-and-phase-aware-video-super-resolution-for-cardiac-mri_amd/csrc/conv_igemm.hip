// Implicit-GEMM 3x3 / 1x1 convolution for gfx950 on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   out[m][n] = sum_k A[m][k] * Wp[k][n],  m = (image, y, x) linearised,  k = (source, 16-channel chunk, tap)
//
// * A is never materialised: every K step gathers a [BM pixels][16 channels] slab straight from the NHWC
//   source tensors (tap-shifted, zero outside the image).  Several sources concatenate along K, which is how
//   torch.cat([x, h]) of the ConvLSTM cell, the 5-frame window of the refine block and the pixel-unshuffle of
//   the upsampler backward disappear.
// * Both slabs are staged through LDS as rows of 16 floats (4 x 16-B slots, slot XOR ((row>>2)&3) => the
//   ds_read_b128 fragment reads are bank-conflict free) with register double buffering: the global loads of
//   step k+1 are in flight while the MFMAs of step k run; one barrier per step.
// * Inside a 16-channel chunk, MFMA step k (0..7) contracts channel 8*(k>>2) + 4*kh + (k&3) in lane-half kh:
//   each lane needs two 16-byte pieces (channels 4kh..4kh+3 and 8+4kh..8+4kh+3) of its pixel, and the packed
//   weights store a lane's 8 values contiguously ([ks][n][kh][8]).  A and B use the same permutation, so the
//   sum over K is unchanged.
// * DIRECT variant (RNH_TILE_DIRECT): no LDS and no barrier at all.  Every wave loads its own A and B
//   fragments straight from global memory / L2 into registers in MFMA layout (the two lane-halves of a pixel
//   read adjacent 16-byte pieces; a B fragment is one fully coalesced 2 KiB read), one K step ahead of the
//   MFMAs.  The fp32 MFMA is slow enough (64 cycles per 512 B of operands) that L1/L2 feed it directly, and
//   without workgroup barriers the waves of a SIMD never convoy behind each other's MFMA phases.
// * Epilogues: bias + store/accumulate into up to 4 channel segments; PixelShuffle fused into the store;
//   ConvLSTM gate math (the 4 gates of a hidden channel sit in the same lane of the 4 column tiles of a wave).
#include <type_traits>
#include "rnh_common.h"

namespace {

// sigmoid / tanh on the hardware exp2 and reciprocal (v_exp_f32, v_rcp_f32: about 1 ulp each; the epilogue runs
// them 5 times per hidden element, and the libm versions cost 4x the instructions for digits the fp32 GEMM in
// front of them does not have).  tanh(x) = 1 - 2 / (1 + e^{2x}) is exact at +-inf and loses nothing near 0 that the
// 1e-7 relative error of the gates would not already hide.
// v_exp_f32 / v_rcp_f32 (1 ulp each; __frcp_rn would be a correctly rounded division: two v_div_scale, v_rcp, four FMAs,
// v_div_fmas, v_div_fixup per value) and no branch: both forms of tanh are computed and selected (hipcc turned the
// ternary around the exp form into an exec-masked branch per value, 64 of them per lane in the gate epilogue).
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
    const float ax = fabsf(x);
    const float big = __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * ax)), 1.f);
    const float small = ax * __builtin_fmaf(-0.33333334f * ax, ax, 1.f);      // |x| < 0.04: the exp form cancels
    const float t = ax < 0.04f ? small : big;
    return copysignf(t, x);
}

// waves per SIMD the register allocator is asked to make room for (DIRECT: 3, or 2 for the 64x128 wave tile)
template <int MI, int NI, bool DIRECT>
constexpr int min_waves() { return DIRECT ? (MI * NI >= 8 ? 1 : 2) : 1; }

template <int WM, int WN, int MI, int NI, int EPI, bool DIRECT>
__global__ void __launch_bounds__(256, (min_waves<MI, NI, DIRECT>())) conv_igemm_kernel(const rnh_conv_args_t P, const int MT, const int NT) {
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    constexpr int AIT = BM / 64;
    constexpr int BIT = (BN * 4 + 255) / 256;
    constexpr int STAGE = DIRECT ? 1 : (BM + BN) * 4;          // float4 per stage
    __shared__ float4 lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int wg = rnh_xcd_remap(blockIdx.x, MT * NT);
    const int mt = wg / NT, nt = wg - mt * NT;
    const int H = P.H, W = P.W, HW = H * W;
    const int Mtot = P.B * HW;
    const int m0 = mt * BM, n0 = nt * BN;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (!DIRECT) {
    // ---- rows this thread stages ------------------------------------------------------------------
    int rb[AIT], ry[AIT], rx[AIT];
    bool rok[AIT];
    const int c4 = tid & 3;
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
        const int m = m0 + (tid >> 2) + 64 * it;
        rok[it] = m < Mtot;
        const int mm = rok[it] ? m : 0;
        const int b = mm / HW, rem = mm - b * HW;
        rb[it] = b;
        ry[it] = rem / W;
        rx[it] = rem - ry[it] * W;
    }

    // ---- K-step state: source s, chunk ch, tap t -----------------------------------------------------
    int s = 0, ch = 0, t = 0;
    int spix[AIT];                                   // pixel index of (b + img_off, y*scale+sy, x*scale+sx)
    auto setup_src = [&](int si) {
        const rnh_src_t &S = P.src[si];
        const int Hs = H * S.scale, Ws = W * S.scale;
#pragma unroll
        for (int it = 0; it < AIT; ++it)
            spix[it] = ((rb[it] + S.img_off) * Hs + ry[it] * S.scale + S.sub_y) * Ws + rx[it] * S.scale + S.sub_x;
    };
    setup_src(0);

    float4 ra[AIT], rw[BIT];
    auto load_stage = [&](int ks) {
        const rnh_src_t &S = P.src[s];
        int dy = 0, dx = 0;
        if (P.ntaps == 9) {
            dy = t / 3 - 1;
            dx = t - (dy + 1) * 3 - 1;
        }
        const int cc = ch * 16 + c4 * 4;
        const bool cok = cc < S.nch;
        const int tapoff = (dy * W * S.scale + dx) * S.scale;
        const float *p1 = S.ptr + S.c0 + cc;
        const float *p2 = S.ptr2 ? S.ptr2 + S.c0 + cc : nullptr;
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const bool v = rok[it] && cok && (unsigned)(ry[it] + dy) < (unsigned)H && (unsigned)(rx[it] + dx) < (unsigned)W;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v) {
                const long off = (long)(spix[it] + tapoff) * S.C;
                val = rnh_ld4(p1 + off);
                if (p2) val = val + rnh_ld4(p2 + off);
            }
            ra[it] = val;
        }
        const float *wb = P.wp + ((long)ks * P.Npad + n0) * 16;
#pragma unroll
        for (int it = 0; it < BIT; ++it) {
            const int idx = tid + 256 * it;
            if (BN * 4 % 256 == 0 || idx < BN * 4) rw[it] = rnh_ld4(wb + idx * 4);
        }
    };
    auto store_stage = [&](int buf) {
        float4 *As = lds + buf * STAGE;
        float4 *Bs = As + BM * 4;
#pragma unroll
        for (int it = 0; it < AIT; ++it) {
            const int r = (tid >> 2) + 64 * it;
            As[r * 4 + (c4 ^ ((r >> 2) & 3))] = ra[it];
        }
#pragma unroll
        for (int it = 0; it < BIT; ++it) {
            const int idx = tid + 256 * it;
            if (BN * 4 % 256 == 0 || idx < BN * 4) {
                const int n = idx >> 2, q = idx & 3;
                Bs[n * 4 + (q ^ ((n >> 2) & 3))] = rw[it];
            }
        }
    };
    auto advance = [&]() {
        if (++t == P.ntaps) {
            t = 0;
            if (++ch * 16 >= P.src[s].nch) {
                ch = 0;
                if (++s < P.nsrc) setup_src(s);
            }
        }
    };

    auto compute = [&](int buf) {
        const float4 *As = lds + buf * STAGE;
        const float4 *Bs = As + BM * 4;
        float a[MI][8], b[NI][8];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = (wm * MI + i) * 32 + l31, sw = (row >> 2) & 3;
            const float4 v0 = As[row * 4 + (kh ^ sw)], v1 = As[row * 4 + ((2 + kh) ^ sw)];
            a[i][0] = v0.x; a[i][1] = v0.y; a[i][2] = v0.z; a[i][3] = v0.w;
            a[i][4] = v1.x; a[i][5] = v1.y; a[i][6] = v1.z; a[i][7] = v1.w;
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int row = (wn * NI + j) * 32 + l31, sw = (row >> 2) & 3;
            const float4 v0 = Bs[row * 4 + ((2 * kh) ^ sw)], v1 = Bs[row * 4 + ((2 * kh + 1) ^ sw)];
            b[j][0] = v0.x; b[j][1] = v0.y; b[j][2] = v0.z; b[j][3] = v0.w;
            b[j][4] = v1.x; b[j][5] = v1.y; b[j][6] = v1.z; b[j][7] = v1.w;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][k], b[j][k], acc[i][j], 0, 0, 0);
    };

    // ---- main loop --------------------------------------------------------------------------------------
    const int nk = P.nk;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const bool more = ks + 1 < nk;
        if (more) {
            advance();
            load_stage(ks + 1);
        }
        compute(ks & 1);
        if (more) store_stage((ks + 1) & 1);
        __syncthreads();
    }

    } else {
    // ---- DIRECT: every lane owns MI pixel rows (A) and NI weight columns (B) -----------------------------
    // All loads are raw buffer loads: a wave-uniform descriptor (base + 2 GiB window) plus a 32-bit per-lane byte
    // offset; lanes outside the image / channel range get offset 0xFFFFFFFF, which the hardware range check turns
    // into zeros - no branch, no select, so the prefetch of step k+1 stays in flight under the MFMAs of step k.
    int rb[MI], ry[MI], rx[MI], rel[MI];
    bool rok[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + (wm * MI + i) * 32 + l31;
        rok[i] = m < Mtot;
        const int mm = rok[i] ? m : 0;
        const int b = mm / HW, rem = mm - b * HW;
        rb[i] = b;
        ry[i] = rem / W;
        rx[i] = rem - ry[i] * W;
    }
    int s = 0, ch = 0, t = 0;
    // Buffer descriptors as plain SGPR quadruples (base, base_hi, 2 GiB window, raw-buffer flags) for the asm loads
    auto make_desc = [&](const float *p) {
        const unsigned long long u = (unsigned long long)p;
        i32x4 d;
        d[0] = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
        d[1] = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
        d[2] = 0x7fffffff;
        d[3] = 0x00020000;
        return d;
    };
    i32x4 adesc;
    int aC4 = 0, anch = 0, asc = 1;
    auto setup_src = [&](int si) {
        const rnh_src_t &S = P.src[si];
        const int Hs = H * S.scale, Ws = W * S.scale;
        int spix[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
            spix[i] = ((rb[i] + S.img_off) * Hs + ry[i] * S.scale + S.sub_y) * Ws + rx[i] * S.scale + S.sub_x;
        // pixel index of the wave's first row minus one row and one column: every tap of every lane is at or after it
        const int base_pix = __builtin_amdgcn_readfirstlane(spix[0]) - (Ws + 1) * S.scale;
#pragma unroll
        for (int i = 0; i < MI; ++i) rel[i] = rok[i] ? spix[i] - base_pix : 0;
        adesc = make_desc(S.ptr + S.c0 + (long)base_pix * S.C);
        aC4 = S.C * 4;
        anch = S.nch;
        asc = S.scale;
    };
    setup_src(0);
    auto advance = [&]() {
        if (++t == P.ntaps) {
            t = 0;
            if (++ch * 16 >= anch) {
                ch = 0;
                if (++s < P.nsrc) setup_src(s);
            }
        }
    };
    const i32x4 bdesc = make_desc(P.wp + ((long)n0 + wn * NI * 32) * 16);
    const int boff0 = (l31 * 16 + kh * 8) * 4, boff1 = boff0 + 4096, boff2 = boff0 + 8192;   // column tiles 0-1, 2-3, 4 (imm <= 4095)
    const int kstride_bytes = P.Npad * 64;

    // The loads are volatile asm so that they stay where they are written: [a few loads of the NEXT K step] [MI*NI
    // MFMAs of the current one], eight times per step.  A wave's vector instructions issue in the shadow of its own
    // MFMAs (64 cycles each); a load-only phase beside another wave's MFMA stream would get one issue slot per MFMA
    // (measured on the weight-gradient kernel).  Waits are counted: vmcnt(L/2) leaves the younger half-step of loads
    // in flight, so every load has one whole K step of MFMAs as lead time.  The s_nop covers SALU->VMEM hazards on
    // the descriptor / soffset registers, which hipcc does not pad inside asm.
    struct Frag {
        f32x4 a[MI][2];
        f32x4 b[NI][2];
    };
    constexpr int LH = MI + NI;                       // loads per half step
    auto lda = [&](f32x4 &dst, int voff) {
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(adesc) : "memory");
    };
    auto ldb = [&](f32x4 &dst, int j, int q, int soff) {
        const int vo = j >= 4 ? boff2 : ((j & 2) ? boff1 : boff0);
        if ((j & 1) == 0 && q == 0)
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(vo), "s"(bdesc), "s"(soff) : "memory");
        else if ((j & 1) == 0)
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(dst) : "v"(vo), "s"(bdesc), "s"(soff) : "memory");
        else if (q == 0)
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen offset:2048" : "=v"(dst) : "v"(vo), "s"(bdesc), "s"(soff) : "memory");
        else
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen offset:2064" : "=v"(dst) : "v"(vo), "s"(bdesc), "s"(soff) : "memory");
    };
    // every register of the half is named exactly once: a duplicate "+v" operand would make hipcc copy a register
    // whose load has not landed yet
    auto wait_half = [&](Frag &F, int q, auto keep) {
        constexpr int KEEP = decltype(keep)::value;
        static_assert((MI == 1 || MI == 2) && (NI == 2 || NI == 4 || NI == 5), "wait_half lists its operands explicitly");
        if constexpr (MI == 2 && NI == 2)
            asm volatile("s_waitcnt vmcnt(%c4)" : "+v"(F.a[0][q]), "+v"(F.a[1][q]), "+v"(F.b[0][q]), "+v"(F.b[1][q]) : "i"(KEEP));
        else if constexpr (MI == 1 && NI == 4)
            asm volatile("s_waitcnt vmcnt(%c5)"
                         : "+v"(F.a[0][q]), "+v"(F.b[0][q]), "+v"(F.b[1][q]), "+v"(F.b[2][q]), "+v"(F.b[3][q])
                         : "i"(KEEP));
        else if constexpr (MI == 1 && NI == 5)
            asm volatile("s_waitcnt vmcnt(%c6)"
                         : "+v"(F.a[0][q]), "+v"(F.b[0][q]), "+v"(F.b[1][q]), "+v"(F.b[2][q]), "+v"(F.b[3][q]), "+v"(F.b[4][q])
                         : "i"(KEEP));
        else if constexpr (MI == 2 && NI == 4)
            asm volatile("s_waitcnt vmcnt(%c6)"
                         : "+v"(F.a[0][q]), "+v"(F.a[1][q]), "+v"(F.b[0][q]), "+v"(F.b[1][q]), "+v"(F.b[2][q]), "+v"(F.b[3][q])
                         : "i"(KEEP));
    };
    int voa[MI][2];                                   // byte offsets of the next step's A pieces (or -1: zeros)
    bool fm_cur = false, fm_prev = false;             // an A load with out-of-range lanes among this / the previous step's loads
    auto next_offsets = [&]() {
        int dy = 0, dx = 0;
        if (P.ntaps == 9) {
            dy = t / 3 - 1;
            dx = t - (dy + 1) * 3 - 1;
        }
        const int cc = ch * 16 + 4 * kh;
        const int tapoff = (dy * W * asc + dx) * asc;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const bool v = rok[i] && (unsigned)(ry[i] + dy) < (unsigned)H && (unsigned)(rx[i] + dx) < (unsigned)W;
            const int o = (rel[i] + tapoff) * aC4 + cc * 4;
            voa[i][0] = (v && cc < anch) ? o : -1;
            voa[i][1] = (v && cc + 8 < anch) ? o + 32 : -1;
        }
        // Loads with masked lanes.  Round 1 saw wrong workgroups on border rows and drained the queue whenever a load with ALL lanes out of range was
        // among the younger ones; round 4 widened that to any masked lane while hunting wrong steps - whose cause turned out to be the copied tail
        // registers described at the K loop below, most likely round 1's too.  Vector-memory operations of a wave complete in issue order, masked or
        // not (MI355X_MICROARCH.md; tools/probes/oob_order.hip, tools/probes/lds_dma_oob.hip), and the build without the drain is exact in 3000 cold
        // steps on one box (round 4, profiles/ARCHIVE/r04_at_*), 3700 on a second (round 5, profiles/r05_u_*) and passes the GPU suite: since round 5 the
        // counted waits stand alone.  -DRNH_IGEMM_MASK_DRAIN brings the drain back (diagnostic build).
        fm_prev = fm_cur;
        fm_cur = false;
#pragma unroll
        for (int i = 0; i < MI; ++i)
            fm_cur |= __builtin_amdgcn_ballot_w64(voa[i][0] == -1) != 0 || __builtin_amdgcn_ballot_w64(voa[i][1] == -1) != 0;
    };
    auto drain_if_unordered = [&]() {
#ifdef RNH_IGEMM_DRAIN_ALL                            // diagnostic build: a full drain in front of every half step
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif defined(RNH_IGEMM_MASK_DRAIN)                    // diagnostic build: the drain of rounds 1-4 whenever a masked lane is among the younger loads
        if (fm_cur || fm_prev) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    };
    auto issue_a = [&](Frag &FL, int q) {
#pragma unroll
        for (int i = 0; i < MI; ++i) lda(FL.a[i][q], voa[i][q]);
    };
    auto issue_b = [&](Frag &FL, int q, int soff, int j0, int j1) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (j >= j0 && j < j1) ldb(FL.b[j][q], j, q, soff);
    };
    auto mfma_group = [&](const Frag &FC, int q, int e) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FC.a[i][q][e], FC.b[j][q][e], acc[i][j], 0, 0, 0);
    };
    // one K step: multiply FC; if `issue`, load the next step (state s, ch, t already advanced) into FL meanwhile
    auto step = [&](Frag &FL, Frag &FC, int ks_next, auto issue_t) {
        constexpr bool ISSUE = decltype(issue_t)::value;
        const int soff = ks_next * kstride_bytes;
        if constexpr (ISSUE) next_offsets();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            drain_if_unordered();
            if (q == 0 || !ISSUE) {
                if (q == 0) wait_half(FC, 0, std::integral_constant<int, LH>());
                else wait_half(FC, 1, std::integral_constant<int, 0>());
            } else {
                wait_half(FC, 1, std::integral_constant<int, LH>());
            }
            mfma_group(FC, q, 0);
            if constexpr (ISSUE) issue_a(FL, q);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(FC, q, 1);
            if constexpr (ISSUE) issue_b(FL, q, soff, 0, (NI + 1) / 2);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(FC, q, 2);
            if constexpr (ISSUE) issue_b(FL, q, soff, (NI + 1) / 2, NI);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(FC, q, 3);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        Frag F0, F1;
        const int nk = P.nk;
        int ks = 0;
        // The tile ALWAYS ends with the same code: "multiply F0, load nothing".  (Until round 4 there were two tails - one step left: multiply F0;
        // two left: multiply F0 loading F1, then multiply F1 - which end in the same code on different registers; hipcc folded them into ONE
        // block behind copies of F0 / F1: v_mov of registers whose loads may still be in flight, in FRONT of the counted wait.  The last K step of
        // a tile then multiplied stale operands whenever its loads took longer than one step: round 4's "race", DESIGN.md 4d (e) - only under
        // cross-stream memory load, a whole XCD's tiles at a time.)  The parity is settled at the START instead: with an even number of K
        // steps the first one goes through F1.  tests/test_isa_guards.py checks that no load target is copied anywhere in this loop.
        next_offsets();                               // K step 0
        if (nk & 1) {
            issue_a(F0, 0);
            issue_b(F0, 0, 0, 0, NI);
            issue_a(F0, 1);
            issue_b(F0, 1, 0, 0, NI);
        } else {
            issue_a(F1, 0);
            issue_b(F1, 0, 0, 0, NI);
            issue_a(F1, 1);
            issue_b(F1, 1, 0, 0, NI);
            advance();
            step(F0, F1, 1, std::true_type());
            ks = 1;
        }
        for (; ks + 2 < nk; ks += 2) {                // an even number of steps is left behind step ks, which is in F0
            advance();
            step(F1, F0, ks + 1, std::true_type());
            advance();
            step(F0, F1, ks + 2, std::true_type());
        }
        step(F1, F0, 0, std::false_type());
    }
    }

    // ---- epilogue -----------------------------------------------------------------------------------------
    // accumulator register r of a 32x32 tile: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if constexpr (EPI == RNH_EPI_LSTM && NI == 2) {
        // 64-column gate groups: a wave's two column tiles are [i(16 ch) | f(16 ch)] and [o | g] of the same 16 hidden
        // channels, so lane c (< 16) holds i, o and lane c + 16 holds f, g of channel c.  The halves swap what the
        // other needs (lane c finishes accumulator registers 0-7, lane c + 16 registers 8-15), then each lane does
        // the gate math of 8 pixels.
        const int c16 = l31 & 15, hi = l31 >> 4;
        const int hc = (n0 + wn * 64) / 4 + c16;
        const int colA = n0 + wn * 64 + l31, colB = colA + 32;
        const float bA = P.bias[colA], bB = P.bias[colB];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float gi[8], gf[8], go[8], gg[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float ownA_lo = acc[i][0][r] + bA, ownA_hi = acc[i][0][r + 8] + bA;
                const float ownB_lo = acc[i][1][r] + bB, ownB_hi = acc[i][1][r + 8] + bB;
                // lane c sends its registers 8-15 (i, o) and receives f, g of registers 0-7; lane c + 16 the reverse
                const float gotA = __shfl_xor(hi ? ownA_lo : ownA_hi, 16, 64);
                const float gotB = __shfl_xor(hi ? ownB_lo : ownB_hi, 16, 64);
                gi[r] = hi ? gotA : ownA_lo;
                gf[r] = hi ? ownA_hi : gotA;
                go[r] = hi ? gotB : ownB_lo;
                gg[r] = hi ? ownB_hi : gotB;
            }
            if (hc >= P.hd) continue;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int rr = r + 8 * hi;                       // accumulator register this lane finishes
                const int m = m0 + (wm * MI + i) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * kh;
                if (m >= Mtot) continue;
                const float si = fast_sigmoid(gi[r]), sf = fast_sigmoid(gf[r]), so = fast_sigmoid(go[r]), tg = fast_tanh(gg[r]);
                const long o = (long)m * P.hd + hc;
                const float cp = P.c_prev ? P.c_prev[o] : 0.f;
                const float cn = sf * cp + si * tg;
                P.c_out[o] = cn;
                P.h_out[o] = so * fast_tanh(cn);
                if (P.gates_out) {
                    float *gp = P.gates_out + (long)m * 4 * P.hd + hc;
                    gp[0] = si;
                    gp[P.hd] = sf;
                    gp[2 * P.hd] = so;
                    gp[3 * P.hd] = tg;
                }
            }
        }
    } else if constexpr (EPI == RNH_EPI_LSTM) {
        static_assert(EPI != RNH_EPI_LSTM || (WN == 1 && NI == 4), "LSTM epilogue: 4x1 waves of 32x128, or 64-column tiles");
        const int hc = nt * 32 + l31;
        if (hc >= P.hd) return;
        const float bi = P.bias[n0 + l31], bf = P.bias[n0 + 32 + l31], bo = P.bias[n0 + 64 + l31],
                    bg = P.bias[n0 + 96 + l31];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m >= Mtot) continue;
                const float gi = fast_sigmoid(acc[i][0][r] + bi);
                const float gf = fast_sigmoid(acc[i][1][r] + bf);
                const float go = fast_sigmoid(acc[i][2][r] + bo);
                const float gg = fast_tanh(acc[i][3][r] + bg);
                const long o = (long)m * P.hd + hc;
                const float cp = P.c_prev ? P.c_prev[o] : 0.f;
                const float cn = gf * cp + gi * gg;
                P.c_out[o] = cn;
                P.h_out[o] = go * fast_tanh(cn);
                if (P.gates_out) {
                    float *gp = P.gates_out + (long)m * 4 * P.hd + hc;
                    gp[0] = gi;
                    gp[P.hd] = gf;
                    gp[2 * P.hd] = go;
                    gp[3 * P.hd] = gg;
                }
            }
    } else {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + (wn * NI + j) * 32 + l31;
            const float bv = P.bias ? P.bias[col] : 0.f;
            if constexpr (EPI == RNH_EPI_PS) {
                const int ncols = P.ps_cq * P.ps_r * P.ps_r;
                if (col >= ncols) continue;
                const int ij = col / P.ps_cq, c = col - ij * P.ps_cq;
                const int pi = ij / P.ps_r, pj = ij - pi * P.ps_r;
                const rnh_dst_t &D = P.dst[0];
                const int Ho = H * P.ps_r, Wo = W * P.ps_r;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        if (m >= Mtot) continue;
                        const int b = m / HW, rem = m - b * HW, y = rem / W, x = rem - y * W;
                        const long o = ((long)((b + D.img_off) * Ho + y * P.ps_r + pi) * Wo + x * P.ps_r + pj) * D.C + D.c0 + c;
                        const float v = acc[i][j][r] + bv;
                        D.ptr[o] = D.accumulate ? D.ptr[o] + v : v;
                    }
            } else {
                // destination segment of this column
                int seg = -1, cbase = 0;
#pragma unroll
                for (int d = 0; d < RNH_MAX_DST; ++d) {
                    if (d < P.ndst && seg < 0) {
                        if (col < cbase + P.dst[d].ncols) seg = d;
                        else cbase += P.dst[d].ncols;
                    }
                }
                if (seg < 0) continue;
                const rnh_dst_t &D = P.dst[seg];
                float *dp = D.ptr + (long)D.img_off * HW * D.C + D.c0 + (col - cbase);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        if (m >= Mtot) continue;
                        const long o = (long)m * D.C;
                        const float v = acc[i][j][r] + bv;
                        dp[o] = D.accumulate ? dp[o] + v : v;
                    }
            }
        }
    }
}

template <int WM, int WN, int MI, int NI, bool DIRECT>
int launch_tile(const rnh_conv_args_t &a, hipStream_t st) {
    constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
    if (a.Npad % BN) RNH_FAIL(RNH_E_RANGE, "rnh_conv_igemm: Npad %d not a multiple of the tile's %d columns", a.Npad, BN);
    const long M = (long)a.B * a.H * a.W;
    const int MT = (int)((M + BM - 1) / BM), NT = a.Npad / BN;
    const dim3 grid((unsigned)(MT * NT)), block(256);
    switch (a.epilogue) {
        case RNH_EPI_STORE:
            hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, MI, NI, RNH_EPI_STORE, DIRECT>), grid, block, 0, st, a, MT, NT);
            break;
        case RNH_EPI_PS:
            hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, MI, NI, RNH_EPI_PS, DIRECT>), grid, block, 0, st, a, MT, NT);
            break;
        case RNH_EPI_LSTM:
            if constexpr ((WN == 1 && NI == 4) || NI == 2) {
                hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, MI, NI, RNH_EPI_LSTM, DIRECT>), grid, block, 0, st, a, MT, NT);
            } else {
                RNH_FAIL(RNH_E_RANGE, "rnh_conv_igemm: the LSTM epilogue needs RNH_TILE_128x128_G");
            }
            break;
        default:
            RNH_FAIL(RNH_E_RANGE, "rnh_conv_igemm: unknown epilogue %d", a.epilogue);
    }
    RNH_CHECK_LAUNCH("rnh_conv_igemm");
    return 0;
}

}  // namespace

extern "C" int rnh_conv_igemm(const rnh_conv_args_t *args, void *stream) {
    if (!args) RNH_FAIL(RNH_E_ARG, "rnh_conv_igemm: null args");
    const rnh_conv_args_t &a = *args;
    if (a.nsrc < 1 || a.nsrc > RNH_MAX_SRC) RNH_FAIL(RNH_E_RANGE, "rnh_conv_igemm: nsrc %d", a.nsrc);
    if (a.B < 1 || a.H < 1 || a.W < 1) RNH_FAIL(RNH_E_ARG, "rnh_conv_igemm: bad geometry");
    if ((long)a.B * a.H * a.W >= (1L << 31) / 16) RNH_FAIL(RNH_E_RANGE, "rnh_conv_igemm: too many output pixels");
    if (a.ntaps != 9 && a.ntaps != 1) RNH_FAIL(RNH_E_RANGE, "rnh_conv_igemm: ntaps %d", a.ntaps);
    if (!a.wp) RNH_FAIL(RNH_E_ARG, "rnh_conv_igemm: null weights");
    int nk = 0;
    for (int i = 0; i < a.nsrc; ++i) {
        if (int e = rnh_check_src(a.src[i], "rnh_conv_igemm")) return e;
        nk += (a.src[i].nch + 15) / 16 * a.ntaps;
    }
    if (nk != a.nk) RNH_FAIL(RNH_E_ARG, "rnh_conv_igemm: nk %d does not match the sources (%d)", a.nk, nk);
    if (a.epilogue == RNH_EPI_LSTM) {
        if (!a.h_out || !a.c_out || !a.bias || a.hd < 1) RNH_FAIL(RNH_E_ARG, "rnh_conv_igemm: LSTM epilogue args");
        const int t_ = a.tile & ~RNH_TILE_DIRECT;
        const bool g64 = t_ == RNH_TILE_128x128 || t_ == RNH_TILE_256x64;       // 64-column gate groups
        if (!g64 && a.Npad != (a.hd + 31) / 32 * 128) RNH_FAIL(RNH_E_ARG, "rnh_conv_igemm: LSTM Npad");
        if (g64 && (a.Npad < (a.hd + 15) / 16 * 64)) RNH_FAIL(RNH_E_ARG, "rnh_conv_igemm: LSTM Npad (64-column groups)");
    } else {
        if (a.ndst < 1 || a.ndst > RNH_MAX_DST) RNH_FAIL(RNH_E_RANGE, "rnh_conv_igemm: ndst %d", a.ndst);
        for (int i = 0; i < a.ndst; ++i)
            if (!a.dst[i].ptr || a.dst[i].ncols < 1 || a.dst[i].C < 1) RNH_FAIL(RNH_E_ARG, "rnh_conv_igemm: bad destination");
        if (a.epilogue == RNH_EPI_PS && (a.ps_r < 2 || a.ps_cq < 1 || a.ps_cq * a.ps_r * a.ps_r > a.Npad))
            RNH_FAIL(RNH_E_ARG, "rnh_conv_igemm: pixel-shuffle args");
    }
    hipStream_t st = (hipStream_t)stream;
    if (a.tile & RNH_TILE_DIRECT) {
        for (int i = 0; i < a.nsrc; ++i)
            if (a.src[i].ptr2) RNH_FAIL(RNH_E_RANGE, "rnh_conv_igemm: the DIRECT variant does not take ptr2 sources");
        switch (a.tile & ~RNH_TILE_DIRECT) {
            case RNH_TILE_128x128:   return launch_tile<2, 2, 2, 2, true>(a, st);
            case RNH_TILE_128x128_G: return launch_tile<4, 1, 1, 4, true>(a, st);
            case RNH_TILE_256x64:    return launch_tile<4, 1, 2, 2, true>(a, st);
            case RNH_TILE_128x160:   return launch_tile<4, 1, 1, 5, true>(a, st);
            case RNH_TILE_256x128:   return launch_tile<4, 1, 2, 4, true>(a, st);
            default: RNH_FAIL(RNH_E_RANGE, "rnh_conv_igemm: unknown tile %d", a.tile);
        }
    }
    switch (a.tile) {
        case RNH_TILE_128x128:   return launch_tile<2, 2, 2, 2, false>(a, st);
        case RNH_TILE_128x128_G: return launch_tile<4, 1, 1, 4, false>(a, st);
        case RNH_TILE_256x64:    return launch_tile<4, 1, 2, 2, false>(a, st);
        case RNH_TILE_128x160:   return launch_tile<4, 1, 1, 5, false>(a, st);
        default: RNH_FAIL(RNH_E_RANGE, "rnh_conv_igemm: unknown tile %d", a.tile);
    }
}
