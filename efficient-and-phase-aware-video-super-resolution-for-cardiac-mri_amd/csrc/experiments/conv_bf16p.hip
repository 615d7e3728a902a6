// EXPERIMENT (round 5) - NOT part of the product build (csrc/build.sh compiles it only with RNH_PERSISTENT=1).  Correct (bit-identical to
// conv_bf16d_kernel: the N = 8 launch-geometry test and the bf16 suite pass with it) and SLOWER: ConvLSTM cell 89-95 us against 78-82, plain-store data
// gradient 79-81 against 76-78 (profiles/r05_l_*, r05_m_*).  Two things the design below did not reckon with: (1) a wave's vector-memory operations
// complete IN ORDER, so every store or state load that an epilogue slice issues inside the main loop holds up the weight fragments requested behind
// it (five to eight steps of look-ahead are less than a store's round trip); (2) an item slice sits in front of its step's MFMAs in program order
// and its dependent exp / rcp chains stall the in-order wave - hiding them needs instruction-level interleaving between single MFMAs, which
// hipcc does not do across a slice.  With one wave per SIMD nothing else covers either.
//
// Persistent form of the bf16 3x3 implicit-GEMM convolution (csrc/conv_bf16.hip) for its two heaviest call sites - the fused ConvLSTM cell
// (reference src/model/nets/refine_net.py:247-267) and the 128-column plain-store launches with >= 128 input channels (the cell's data gradient
// at the chain ends, autograd of :258-265) - with the EPILOGUE OF A TILE RUNNING INSIDE THE MAIN LOOP OF THE NEXT ONE.
//
// Why (round 5; tools/bf16_wgtrace.py, tools/bf16_stamps.py, profiles/r05_b_*, r05_c_*, r05_e_*): in conv_bf16d_kernel a workgroup lives
// 38 us of which 14 are its epilogue (LDS transposition, gate math, stores) and 3 its prologue (first halo from HBM); two workgroups share a CU
// so that one's epilogue runs under the other's MFMAs, but per-workgroup traces show that it does so badly - the finishing wave's vector
// instructions and the computing wave's matrix instructions on one SIMD slow each other down (items 2.1 k cycles alone, 4.5-8 k beside a main
// loop; the main loop 5.0-5.8 k cycles per chunk alone, 6-7 k beside an epilogue; s_setprio in either direction changes nothing), so the launch
// takes 78 us where its MFMAs need 50 and its main loops 62.  What does overlap on this hardware is a wave's OWN vector and matrix instructions
// (MI355X_MICROARCH.md: five single-issue instructions hide behind every 32x32x16 MFMA of the same wave).  Hence:
//
//   * ONE 256-thread workgroup per CU, one wave per SIMD with the whole register file: two accumulator sets (the tile being computed, the tile
//     being finished), a weight-fragment ring of nine sets (eight steps = 64 MFMAs ahead: with nobody else on the SIMD every wait is exposed);
//   * the workgroup walks a list of pixel tiles of ONE column tile (its weight-fragment stream never changes); halo requests, the LDS double
//     buffer and the fragment ring run across tile boundaries, so only the first tile of a workgroup has a prologue;
//   * the finished tile's four epilogue rounds (one 32-pixel row block of each pixel half: park, barrier, items - as in conv_bf16d_kernel) ride
//     in the four chunks of the next tile: round r is parked in chunk r and its items are computed in chunk r + 1 (the chunk barrier between
//     them is the barrier the round needs), a slice per step, between the step's MFMAs; LDS holds the halo double buffer AND the two park
//     images (119 KB).  The last tile of a workgroup is finished the serial way.
//
// Same arithmetic as conv_bf16d_kernel in the same order (accumulation over chunks, taps and k steps; bias added when parked; the item
// expressions): results are bit-identical to it.  All loads and stores are compiler-visible (no hand-counted waits): the step loop is pinned
// with sched_barrier(0), so a load stays in the step it was written in and hipcc counts its own vmcnt.
#include "../conv_bf16_common.h"
#include <stdlib.h>
#include <type_traits>

#define RNH_INL __attribute__((always_inline))

namespace {

constexpr int Q_TH = 8, Q_MB = 4, Q_NB = 2, Q_NCOLS = 128, Q_KC = 32, Q_KS = 2, Q_NTAPS = 9, Q_NSTEP = Q_NTAPS * Q_KS;
constexpr int Q_HP = HPW * (Q_TH + 2), Q_APITCH = 80, Q_A_BYTES = Q_HP * Q_APITCH, Q_PPP = 4, Q_A_PIECES = Q_PPP * Q_HP, Q_A_ITERS = (Q_A_PIECES + 255) / 256;
constexpr int Q_OPITCH = Q_NCOLS + 4, Q_PXR = 2 * TW, Q_OBUF = Q_PXR * Q_OPITCH;
constexpr int Q_SMEM = 2 * Q_A_BYTES + 2 * Q_OBUF * 4;
constexpr int Q_RING = 9, Q_AHEAD = Q_RING - 1;
static_assert(Q_NSTEP % Q_RING == 0, "the set of a step is static");
static_assert(Q_SMEM <= 160 * 1024, "one workgroup per CU");

typedef float f32x4p __attribute__((ext_vector_type(4)));

template <int N>
using IC = std::integral_constant<int, N>;
template <int... S, class F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, S...>, F &&f) {
    (f(IC<S>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F &&f) {                      // f(IC<0>{}), ..., f(IC<N - 1>{}): a loop whose index is a compile-time constant
    sfor_impl(std::make_integer_sequence<int, N>{}, f);
}

template <int EPI>
__global__ void __launch_bounds__(256, 1) conv_bf16p_kernel(const rnh_conv_bf16_args_t P, const int TYn, const int TXn, const int NT) {
    constexpr int MB = Q_MB, NB = Q_NB, KS = Q_KS, NTAPS = Q_NTAPS, NSTEP = Q_NSTEP, RING = Q_RING, AHEAD = Q_AHEAD, A_BYTES = Q_A_BYTES,
                  A_ITERS = Q_A_ITERS, APITCH = Q_APITCH, PPP = Q_PPP, OPITCH = Q_OPITCH, OBUF = Q_OBUF, NCOLS = Q_NCOLS, KC = Q_KC, TH = Q_TH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[Q_SMEM];
    float *const img0 = reinterpret_cast<float *>(smem + 2 * A_BYTES);      // the two park images behind the halo double buffer

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5, wave = tid >> 6;
    const int ph = wave & 1, chalf = wave >> 1;
    const int H = P.H, W = P.W;
    const int sc = P.src[0].scale, Hs = H * sc, Ws = W * sc;
    const int nch = P.nchunks / KS;

    // ---- the tile list of this workgroup: blocks b and b + 8 share an XCD (round-robin placement: speed only), so XCD x takes the contiguous
    // range [x Mx, (x + 1) Mx) of the pixel tiles and both column tiles of a pixel tile read its halo through the same L2; inside the XCD block
    // idx takes column tile idx % NT and every (G / NT)-th pixel tile from idx / NT on
    const int M = P.B * TYn * TXn, Mx = (M + 7) >> 3;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, G = gridDim.x >> 3;
    const int nt = idx % NT, slot = idx / NT, nslots = G / NT;
    const int mt_lo = xcd * Mx, mt_hi = min(M, mt_lo + Mx);
    const int ntiles = mt_lo + slot < mt_hi ? (mt_hi - mt_lo - slot + nslots - 1) / nslots : 0;
    if (ntiles == 0) return;
    const int total = ntiles * nch;                                  // chunks of the whole list
    struct Tile {
        int img, y0, x0;
    };
    auto tile_of = [&](int k) RNH_INL {
        const int mt = mt_lo + slot + k * nslots;
        const int img = mt / (TYn * TXn), trem = mt - img * (TYn * TXn), ty = trem / TXn, tx = trem - ty * TXn;
        return Tile{img, ty * TH, tx * TW};
    };

    // ---- halo staging (as conv_bf16d_kernel, 32-channel chunks of bf16 sources): a LOAD CURSOR two chunks ahead of the chunk being computed, which
    // walks the chunks of a tile's source list and then moves on to the next tile of the list
    int apix[A_ITERS];
    const int ahalf0 = tid % PPP, alds0 = (tid / PPP) * APITCH + ahalf0 * 16;
    constexpr int ALDS_STEP = (256 / PPP) * APITCH;
    const bool alast = tid < Q_A_PIECES - 256 * (A_ITERS - 1);
    int Lk = 0, Lc = 0, Limg = 0, si = 0, cc = 0;
    auto cursor_tile = [&](int k) RNH_INL {
        const Tile t = tile_of(k);
        Limg = t.img;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int p = tid + 256 * i, px = p / PPP, hr = px / HPW, hc = px - hr * HPW;
            const int y = t.y0 - 1 + hr, x = t.x0 - 1 + hc;
            const bool in = p < Q_A_PIECES && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            apix[i] = in ? (y * sc) * Ws + x * sc : -1;
        }
    };
    cursor_tile(0);
    uint4 ra[A_ITERS];
    auto load_chunk = [&]() RNH_INL {
        const rnh_msrc_t &S = P.src[si];
        const char *base = reinterpret_cast<const char *>(S.ptr) + ((((long)(Limg + S.img_off) * Hs + S.sub_y) * Ws + S.sub_x) * S.C + S.c0 + cc * KC) * 2;
        const __amdgpu_buffer_rsrc_t rs = bdesc(base);
        const int pstride = S.C * 2;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) ra[i] = bld16(rs, apix[i] >= 0 ? apix[i] * pstride + ahalf0 * 16 : -1);
        if (++cc * KC >= S.nch) {
            cc = 0;
            ++si;
        }
        if (++Lc == nch) {                                          // on to the next tile of the list
            Lc = 0, si = 0, cc = 0;
            if (++Lk < ntiles) cursor_tile(Lk);
        }
    };
    auto store_chunk = [&](int buf) RNH_INL {
        unsigned char *Ab = smem + buf * A_BYTES;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i)
            if (i + 1 < A_ITERS || alast) *reinterpret_cast<uint4 *>(Ab + alds0 + i * ALDS_STEP) = ra[i];
    };

    // ---- weight fragments (as conv_bf16d_kernel): the stream of a column tile is the same for every pixel tile, so behind a tile's last chunk
    // the ring simply continues with chunk 0
    const __amdgpu_buffer_rsrc_t wrs = bdesc(P.wp);
    const int wlane = ((nt * NCOLS + chalf * (NCOLS / 2) + l31) * 16 + kh * 8) * 2;
    const int slab = P.Npad * 32;
    uint4 bq[RING][NB];
    auto bload = [&](int c, int sa, int set) RNH_INL {
        const int q = sa / NSTEP, r = sa - q * NSTEP, tap = r / KS, ks = r - tap * KS;
        const int c2 = c + q < nch ? c + q : c + q - nch;
        const int base = (c2 * KS + ks) * NTAPS * slab + tap * slab;
#pragma unroll
        for (int n = 0; n < NB; ++n) bq[set][n] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane + n * 32 * 32, base, 0));
    };

    float bv[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) bv[n] = P.bias ? P.bias[nt * NCOLS + chalf * (NCOLS / 2) + n * 32 + l31] : 0.f;

    // ---- epilogue pieces ----------------------------------------------------------------------------------------------------------------
    // park row block m of a finished tile's accumulators (+ bias) as pixels 32 ph .. 32 ph + 31 of a [64][OPITCH] fp32 image
    auto park = [&](const f32x16 (&hacc)[MB][NB], auto mc, float *ob) RNH_INL {
        constexpr int m = decltype(mc)::value;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int col = chalf * (NCOLS / 2) + n * 32 + l31;
#pragma unroll
            for (int v = 0; v < 16; ++v) ob[(ph * TW + (v & 3) + 8 * (v >> 2) + 4 * kh) * OPITCH + col] = hacc[m][n][v] + bv[n];
        }
    };

    // per-thread destination constants of the plain-store items (found once; the same for every tile)
    [[maybe_unused]] bool live = false;
    [[maybe_unused]] void *dptr = nullptr;
    [[maybe_unused]] int ddt = 0, dacc = 0, dioff = 0;
    [[maybe_unused]] long dch = 0, dC = 0;
    constexpr int G8 = NCOLS / 8;
    if constexpr (EPI == RNH_EPI_STORE) {
        const int n0 = nt * NCOLS + (tid % G8) * 8;
        int seg = -1, cbase = 0;
        for (int d = 0; d < P.ndst; ++d) {
            if (seg < 0 && n0 < cbase + P.dst[d].ncols) seg = d;
            if (seg < 0) cbase += P.dst[d].ncols;
        }
        live = seg >= 0;
        const rnh_mdst_t &D = P.dst[live ? seg : 0];
        dptr = D.ptr, ddt = D.dtype, dacc = D.accumulate, dC = D.C, dch = D.c0 + (n0 - cbase), dioff = D.img_off;
    }

    // The items of one round, cut into slices 0 .. 15 (slice s runs in step s + 1 of a chunk when the round rides in a main loop; all of them in a
    // row when the last tile is finished).  State that lives from slice to slice:
    [[maybe_unused]] f32x4p cq[2];                                   // LSTM: previous cell state of the round's item (requested when the round was parked)
    [[maybe_unused]] float io[32], icp[8], icn[8], ihn[8], igi[8], igf[8], igo[8], igg[8];
    [[maybe_unused]] float sf[4][8];                                 // STORE: an item's 8 columns on their way from LDS to memory
    // LSTM: request the previous cell state of round r of tile t (clamped, always valid addresses; no previous state: zeros at the use)
    [[maybe_unused]] auto lstm_request = [&](const Tile &t, int r) RNH_INL {
        const int hd = P.hd, hcl = min(nt * 32 + (tid & 3) * 8, hd - 8), px = tid >> 2;
        const float *csrc = P.c_prev ? P.c_prev : P.c_out;
        const int y = min(t.y0 + (px >> 5) * MB + r, H - 1), x = min(t.x0 + (px & (TW - 1)), W - 1);
        const f32x4p *p = reinterpret_cast<const f32x4p *>(csrc + (((long)t.img * H + y) * W + x) * hd + hcl);
        cq[0] = p[0];
        cq[1] = p[1];
    };
    auto items = [&](const Tile &t, int r, const float *ot, auto sc_) RNH_INL {
        constexpr int s = decltype(sc_)::value;
        if constexpr (EPI == RNH_EPI_LSTM) {
            // one item per thread and round: pixel tid >> 2 of the image, hidden channels nt * 32 + 8 (tid & 3) ..; column = gate * 32 + j
            const int hd = P.hd, j0 = (tid & 3) * 8, hc = nt * 32 + j0, px = tid >> 2;
            if constexpr (s == 0) {
                const float *o = ot + px * OPITCH + j0;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 8; ++e) io[g * 8 + e] = o[g * 32 + e];
#pragma unroll
                for (int e = 0; e < 4; ++e) icp[e] = P.c_prev ? cq[0][e] : 0.f, icp[4 + e] = P.c_prev ? cq[1][e] : 0.f;
            } else if constexpr (s >= 1 && s <= 8) {
                constexpr int e = s - 1;
                igi[e] = b_sigmoid(io[e]);
                igf[e] = b_sigmoid(io[8 + e]);
                igo[e] = b_sigmoid(io[16 + e]);
                igg[e] = b_tanh_fast(io[24 + e]);
                icn[e] = __builtin_fmaf(igf[e], icp[e], igi[e] * igg[e]);
                ihn[e] = igo[e] * b_tanh_fast(icn[e]);
            } else if constexpr (s == 9) {
                const int y = t.y0 + (px >> 5) * MB + r, x = t.x0 + (px & (TW - 1));
                if (hc < hd && y < H && x < W) {
                    const long pe = ((long)t.img * H + y) * W + x;
                    store8(P.c_out, RNH_DT_F32, pe * hd + hc, icn);
                    store8(P.h_out, P.h_dtype, pe * hd + hc, ihn);
                    if (P.gates_out) {
                        store8(P.gates_out, P.gates_dtype, pe * 4 * hd + hc, igi);
                        store8(P.gates_out, P.gates_dtype, pe * 4 * hd + hd + hc, igf);
                        store8(P.gates_out, P.gates_dtype, pe * 4 * hd + 2 * hd + hc, igo);
                        store8(P.gates_out, P.gates_dtype, pe * 4 * hd + 3 * hd + hc, igg);
                    }
                }
            }
        } else {
            // four items per thread and round: pixels tid / 16 + 16 i, columns 8 (tid % 16) ..: item i is read from LDS in slice 2 i; ALL stores
            // (read-modify-write where the destination accumulates) go out in slice 8 - a wave's vector-memory operations complete in order, so
            // every store holds up the weight fragments requested behind it: one such event per chunk, not four
            const int c8 = tid % G8;
            if constexpr (s < 8 && (s & 1) == 0) {
                constexpr int i = s >> 1;
                const float *o = ot + (tid / G8 + (256 / G8) * i) * OPITCH + c8 * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) sf[i][e] = o[e];
            } else if constexpr (s == 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int px = tid / G8 + (256 / G8) * i;
                    const int y = t.y0 + (px >> 5) * MB + r, x = t.x0 + (px & (TW - 1));
                    if (live && y < H && x < W) {
                        const long e = ((((long)t.img + dioff) * H + y) * W + x) * dC + dch;
                        if (dacc) {
                            float old[8];
                            load8(dptr, ddt, e, old);
#pragma unroll
                            for (int q = 0; q < 8; ++q) sf[i][q] += old[q];
                        }
                        store8(dptr, ddt, e, sf[i]);
                    }
                }
            }
        }
    };
    constexpr int NSLICE = 10;                                       // slices with work (both kinds)

    // ---- one chunk: 18 steps of (4 A fragments from LDS, 2 B fragments into the ring five steps ahead, 8 MFMAs); the halo of the chunk after
    // next is requested at step 4, where the one requested a chunk ago goes to the other LDS buffer.  RIDE = which epilogue round of the previous
    // tile (accumulators in `held`, coordinates `tp`) rides along: -1 none; r = 0 .. 3: the items of round r (slices in steps 1 ..; the round
    // was parked a chunk ago), then round r + 1 is parked (step 13) and its requests go out
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc[MB][NB], held[MB][NB];
    auto chunk = [&](int c, int q, auto first_, auto ride_, const Tile &tp) RNH_INL {
        constexpr bool FIRST = decltype(first_)::value != 0;
        constexpr int RIDE = decltype(ride_)::value;
        const int buf = q & 1;
        const unsigned char *Ab = smem + buf * A_BYTES + (MB * ph * HPW + l31) * APITCH + kh * 16;
        bf16x8 a[2][MB];
        auto afrags = [&](int step, int set) RNH_INL {
            const int tap = step / KS, ks = step - tap * KS, dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int m = 0; m < MB; ++m) a[set][m] = *reinterpret_cast<const bf16x8 *>(Ab + ((m + dy) * HPW + dx) * APITCH + ks * 32);
        };
        afrags(0, 0);
        auto one_step = [&](auto step_) RNH_INL {
            constexpr int step = decltype(step_)::value;
            if constexpr (step + 1 < NSTEP) afrags(step + 1, (step + 1) & 1);
            bload(c, step + AHEAD, (step + AHEAD) % RING);
            if constexpr (step == 4) {
                if (q + 1 < total) store_chunk(buf ^ 1);
                if (q + 2 < total) load_chunk();
            }
            if constexpr (RIDE >= 0) {
                if constexpr (step >= 1 && step <= NSLICE) items(tp, RIDE, img0 + (RIDE & 1) * OBUF, IC<step - 1>{});
                if constexpr (step == 13 && RIDE + 1 < MB) {
                    park(held, IC<(RIDE + 1) & 3>{}, img0 + ((RIDE + 1) & 1) * OBUF);
                    if constexpr (EPI == RNH_EPI_LSTM) lstm_request(tp, RIDE + 1);
                }
            }
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[step & 1][m], __builtin_bit_cast(bf16x8, bq[step % RING][n]),
                                                                        (FIRST && step == 0) ? zero16 : acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        sfor<NSTEP>(one_step);
    };
    // behind a tile's last chunk, in front of its barrier: round 0 goes to image 0 straight from the accumulators (its requests go out), the other
    // three row blocks move to `held` (their rounds ride in the next tile's chunks, or are finished below)
    auto tail = [&](const Tile &t) RNH_INL {
        park(acc, IC<0>{}, img0);
        if constexpr (EPI == RNH_EPI_LSTM) lstm_request(t, 0);
#pragma unroll
        for (int m = 1; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n) held[m][n] = acc[m][n];
    };

    // ---- prologue of the list: fragment ring, the first two halos
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) bload(0, g, g);
    load_chunk();
    store_chunk(0);
    if (total > 1) load_chunk();
    __syncthreads();

    // ---- first tile: nothing rides
    Tile tcur = tile_of(0);
    chunk(0, 0, IC<1>{}, IC<-1>{}, tcur);
    for (int c = 1; c < nch; ++c) {
        __syncthreads();
        chunk(c, c, IC<0>{}, IC<-1>{}, tcur);
    }
    tail(tcur);
    __syncthreads();
    // ---- the others: the four rounds of tile k - 1 ride in chunks 0 .. 3 of tile k
    for (int k = 1; k < ntiles; ++k) {
        const Tile tp = tcur;
        tcur = tile_of(k);
        const int q0 = k * nch;
        chunk(0, q0, IC<1>{}, IC<0>{}, tp);
        __syncthreads();
        chunk(1, q0 + 1, IC<0>{}, IC<1>{}, tp);
        __syncthreads();
        chunk(2, q0 + 2, IC<0>{}, IC<2>{}, tp);
        __syncthreads();
        chunk(3, q0 + 3, IC<0>{}, IC<3>{}, tp);
        for (int c = 4; c < nch; ++c) {
            __syncthreads();
            chunk(c, q0 + c, IC<0>{}, IC<-1>{}, tp);
        }
        tail(tcur);
        __syncthreads();
    }
    // ---- the last tile is finished the serial way: items of round r, park of round r + 1, barrier
    sfor<MB>([&](auto r_) RNH_INL {
        constexpr int r = decltype(r_)::value;
        sfor<NSLICE>([&](auto s_) RNH_INL { items(tcur, r, img0 + (r & 1) * OBUF, s_); });
        if constexpr (r + 1 < MB) {
            park(held, IC<(r + 1) & 3>{}, img0 + ((r + 1) & 1) * OBUF);
            if constexpr (EPI == RNH_EPI_LSTM) lstm_request(tcur, r + 1);
            __syncthreads();
        }
    });
}

}  // namespace

// Launch the persistent kernel where it applies; returns 1 if it did, 0 if conv_bf16d_kernel should run (csrc/conv_bf16.hip rnh_conv_bf16).
// Applies to: 3x3, bf16 sources in multiples of 32 channels with at least four 32-channel chunks, 128-column tiles, the ConvLSTM and the
// plain-store epilogue, at least two tiles per workgroup.  RNH_BF16_PERSIST=0 switches it off (A/B runs).
int rnh_conv_bf16_persistent(const rnh_conv_bf16_args_t &a, int TYn, int TXn, int NT, hipStream_t st) {
    static const int enabled = [] {
        const char *e = getenv("RNH_BF16_PERSIST");
        return !(e && e[0] == '0');
    }();
    if (!enabled || a.ntaps != 9 || a.Npad % 128 || a.nchunks < 8 || a.nchunks % 2) return 0;
    if (a.epilogue != RNH_EPI_LSTM && a.epilogue != RNH_EPI_STORE) return 0;
    for (int i = 0; i < a.nsrc; ++i)
        if (a.src[i].dtype != RNH_DT_BF16 || a.src[i].nch % 32) return 0;
    const int M = a.B * TYn * TXn, Mx = (M + 7) / 8;
    if (NT > 2 || Mx * NT < 64) return 0;                           // fewer than two tiles per workgroup: the two-workgroups-per-CU kernel
    const int G = 32, grid = 8 * G;                                 // 256 workgroups: one per CU
    if (a.epilogue == RNH_EPI_LSTM) hipLaunchKernelGGL((conv_bf16p_kernel<RNH_EPI_LSTM>), dim3(grid), dim3(256), 0, st, a, TYn, TXn, NT);
    else hipLaunchKernelGGL((conv_bf16p_kernel<RNH_EPI_STORE>), dim3(grid), dim3(256), 0, st, a, TYn, TXn, NT);
    return 1;
}
