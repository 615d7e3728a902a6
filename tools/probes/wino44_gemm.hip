// Probe (round 5): would a Winograd F(4x4, 3x3) ConvLSTM cell beat the F(2x2, 3x3) one (0.3175 ms at config 2)?  This is the matrix part of
// such a kernel with a plain-store epilogue, on synthetic operands:
//   M[xi][tile][n] = sum_c V[xi][tile][c] U[xi][c][n]   (36 positions xi = 6 i + j of the 6x6 transform domain, fp32 MFMA 32x32x2)
//   Y = A^T M A                                          (4x4 outputs per tile and column)
// The transformed input V comes from HBM already in transform-domain form (a separate, memory-bound kernel would write it: B^T d B of 6x6
// patches, 2.25 x the bytes of the image) in the order the workgroup wants it in LDS, and goes there by LDS-DMA: no staging registers, no
// transform arithmetic in this kernel.  A workgroup = 8 waves = 4 position groups (3x3 blocks of the 6x6 domain) x 2 column groups of 32:
// 32 tiles (512 pixels) x 64 columns, 144 accumulator registers per wave, two waves per SIMD.  Weights stream from L2 through a register
// ring nine requests deep.  Output transform: every wave keeps a quarter of the (tile, column) entries and receives the other 27 positions of
// those from its three partners through LDS (two passes), then computes the 4x4 outputs and stores them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/wino44_gemm.hip -o /tmp/wino44 && /tmp/wino44
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
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

constexpr int TILES = 32, NPOS = 36, CH = 16, BUF = NPOS * TILES * CH;       // floats of one staged chunk: 73 728 bytes
constexpr int RB = 9, RA = 3, NQ = 18;                                        // weight ring, LDS operand ring, (position, 8-channel block) pairs per chunk

__device__ __forceinline__ i32x4 hdesc(const void *p) {
    const unsigned long long u = (unsigned long long)p;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    d[1] = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
    d[2] = 0x7fffffff;
    d[3] = 0x00020000;
    return d;
}

#ifndef W44_EPI
#define W44_EPI 1                // 0: no exchange / transform (every wave stores its raw accumulators' first entries) - timing only
#endif

__global__ void __launch_bounds__(512, 1) wino44_gemm(const float *V, const float *U, float *out, const int MT, const int NT, const int nchunks, const int Npad) {
    __shared__ __attribute__((aligned(16))) float stage[2 * BUF];            // 147 456 bytes
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3, cg = wave >> 2;
    // workgroups of one XCD (blockIdx % 8) take consecutive (tile block, column block) pairs, column block fastest: the NT workgroups that
    // read one tile block's V share an L2
    const int total = MT * NT, per = total / 8;
    const int bid = total % 8 == 0 ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int mt = bid / NT, nt = bid - mt * NT;

    const unsigned lds0 = (unsigned)(size_t)stage;
    // ---- V: LDS-DMA, nine 16-byte pieces per thread and chunk ----------------------------------------------------------------------------
    const i32x4 vdesc = hdesc(V);
    const int vvoff = tid * 16;
    auto dma = [&, &vvoff = vvoff, &vdesc = vdesc, &lds0 = lds0, &wave = wave, &mt = mt](int buf, int chunk, auto d_tag) INL {
        constexpr int d = decltype(d_tag)::value;
        const unsigned ld = __builtin_amdgcn_readfirstlane(lds0 + buf * BUF * 4 + (d * 512 + wave * 64) * 16);
        const int soff = __builtin_amdgcn_readfirstlane((mt * nchunks + chunk) * (BUF * 4) + d * 8192);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ld), "v"(vvoff), "s"(vdesc), "s"(soff) : "memory");
    };
    // ---- A operand: lane (tile l31, k half kh), position p of this wave's nine, 8-channel block kb: 16 bytes = channels 8 kb + kh + 2 m ------
    const int I0 = 3 * (pg >> 1), J0 = 3 * (pg & 1);
    const int sw = (l31 >> 2) & 3;
    unsigned avl[2][2];                                                       // [buffer][kb]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) avl[b][kb] = lds0 + b * BUF * 4 + (6 * I0 + J0) * 2048 + l31 * 64 + (((kb * 2 + kh) ^ sw) * 16);
    f32x4 aq[RA];
    auto loada = [&](int buf, auto g_tag) INL {                                // g = q of the chunk (0 .. 17)
        constexpr int g = decltype(g_tag)::value, p = g % 9, kb = g / 9;
        constexpr int off = ((p / 3) * 6 + (p % 3)) * 2048;
        auto &aqr = aq;                                                       // (a generic lambda captures only what a non-dependent expression names)
        auto &avlr = avl;
        asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(aqr[g % RA]) : "v"(avlr[buf][kb]), "i"(off) : "memory");
    };
    // ---- B operand: U[s8][xi][n][kh][4] --------------------------------------------------------------------------------------------------
    const i32x4 udesc = hdesc(U);
    const int pstride = Npad * 32;                                            // bytes of one (8-channel block, position) slab
    int bvoff[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) bvoff[p] = (6 * (I0 + p / 3) + J0 + p % 3) * pstride + ((nt * 64 + cg * 32 + l31) * 2 + kh) * 16;
    const int s8max = 2 * nchunks - 1;
    f32x4 bq[RB];
    auto loadb = [&, &udesc = udesc, &pstride = pstride, &s8max = s8max](int chunk, auto g_tag) INL {                              // g = q relative to the chunk's first pair (0 .. 25)
        constexpr int g = decltype(g_tag)::value, p = g % 9, kbr = g / 9;
        const int s8 = min(2 * chunk + kbr, s8max);
        const int soff = __builtin_amdgcn_readfirstlane(s8 * 36 * pstride);
        auto &bqr = bq;
        auto &bvr = bvoff;
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(bqr[g % RB]) : "v"(bvr[p]), "s"(udesc), "s"(soff) : "memory");
    };
    f32x16 acc[9];

    // ---- prologue -------------------------------------------------------------------------------------------------------------------------
    sfor<9>([&](auto d) INL { dma(0, 0, d); });
    sfor<8>([&](auto g) INL { loadb(0, g); });
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    loada(0, std::integral_constant<int, 0>());
    loada(0, std::integral_constant<int, 1>());
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[p][v] = 0.f;

    auto chunk_body = [&](const int c, auto buf_tag, auto more_tag) INL {
        constexpr bool more = decltype(more_tag)::value;
        constexpr int buf = decltype(buf_tag)::value;
        sfor<NQ>([&](auto q_tag) INL {
            constexpr int q = decltype(q_tag)::value;
            constexpr int p = q % 9;
            loadb(c, std::integral_constant<int, q + 8>());
            // DMA requests (end of steps 0 .. 8 of a chunk that has a successor) younger than this step's weights (requested 8 steps ago)
            constexpr int lo = q - 8 > 0 ? q - 8 : 0, hi = q - 1 < 8 ? q - 1 : 8;
            constexpr int nd = more && hi >= lo ? hi - lo + 1 : 0;
            if constexpr (q == 16) {
                // every DMA of the next chunk has landed (the 8 weight requests of steps 9 .. 16 are the only younger ones), all LDS reads of
                // this buffer are back; behind the barrier the other buffer is complete and this one free
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" : "+v"(bq[q % RB]), "+v"(aq[q % RA]), "+v"(aq[(q + 1) % RA])::"memory");
                if constexpr (more) loada(buf ^ 1, std::integral_constant<int, 0>());
            } else if constexpr (q == 17) {
                if constexpr (more) loada(buf ^ 1, std::integral_constant<int, 1>());
                if constexpr (more) asm volatile("s_waitcnt vmcnt(%c2) lgkmcnt(2)" : "+v"(bq[q % RB]), "+v"(aq[q % RA]) : "i"(8 + nd) : "memory");
                else asm volatile("s_waitcnt vmcnt(%c2) lgkmcnt(0)" : "+v"(bq[q % RB]), "+v"(aq[q % RA]) : "i"(8 + nd) : "memory");
            } else {
                loada(buf, std::integral_constant<int, q + 2>());
                asm volatile("s_waitcnt vmcnt(%c2) lgkmcnt(2)" : "+v"(bq[q % RB]), "+v"(aq[q % RA]) : "i"(8 + nd) : "memory");
            }
            // (asm: hipcc's scheduler otherwise moves the MFMAs across the requests and waits and copies ring registers to do so)
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, %3, %4, %0\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, %5, %6, %0\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, %7, %8, %0"
                         : "+v"(acc[p])
                         : "v"(aq[q % RA].x), "v"(bq[q % RB].x), "v"(aq[q % RA].y), "v"(bq[q % RB].y), "v"(aq[q % RA].z), "v"(bq[q % RB].z), "v"(aq[q % RA].w),
                           "v"(bq[q % RB].w));
            if constexpr (more && q < 9) dma(buf ^ 1, c + 1, std::integral_constant<int, q>());
        });
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    for (int c = 0; c + 2 < nchunks; c += 2) {                                // (nchunks is even: the buffer of a chunk is a compile-time constant)
        chunk_body(c, B0(), std::true_type());
        chunk_body(c + 1, B1(), std::true_type());
    }
    chunk_body(nchunks - 2, B0(), std::true_type());
    chunk_body(nchunks - 1, B1(), std::false_type());
    // (the clamped weight requests of the last eight steps are still in flight: their registers must not be reused before they land)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(bq[4]), "+v"(bq[5]), "+v"(bq[6]), "+v"(bq[7]), "+v"(bq[8])::"memory");

    const int ncol = nt * 64 + cg * 32 + l31;
#if W44_EPI == 0
    // timing variant: no exchange, no output transform
    {
        float sacc = 0.f;
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int v = 0; v < 16; ++v) sacc += acc[p][v];
        out[((long)(mt * 32 + pg * 8 + kh * 4) * 16) * Npad + ncol] = sacc;
    }
#else
    // ---- exchange: wave pg keeps the entries v = 4 pg .. 4 pg + 3 (tiles 8 pg + e + 4 kh) of every position ----------------------------------
    asm volatile("s_barrier" ::: "memory");                                  // (nobody reads the staging buffers any more)
    f32x4 *px = reinterpret_cast<f32x4 *>(stage);
    float M[36][4];
    auto epilogue = [&](auto pg_tag) INL {
        constexpr int PG = decltype(pg_tag)::value;
        constexpr int PI0 = 3 * (PG >> 1), PJ0 = 3 * (PG & 1);
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) M[6 * (PI0 + p / 3) + PJ0 + p % 3][e] = acc[p][4 * PG + e];
        auto pass = [&](auto p0_tag, auto np_tag) INL {
            constexpr int P0 = decltype(p0_tag)::value, NP = decltype(np_tag)::value;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r == PG) continue;
                const int sidx = PG < r ? PG : PG - 1;                        // this wave's slot among r's three senders
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const f32x4 v4 = {acc[P0 + p][4 * r], acc[P0 + p][4 * r + 1], acc[P0 + p][4 * r + 2], acc[P0 + p][4 * r + 3]};
                    px[(((cg * 4 + r) * 3 + sidx) * NP + p) * 64 + lane] = v4;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s == PG) continue;
                const int sidx = s < PG ? s : s - 1;
                const int SI0 = 3 * (s >> 1), SJ0 = 3 * (s & 1);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const f32x4 v4 = px[(((cg * 4 + PG) * 3 + sidx) * NP + p) * 64 + lane];
                    const int pos = 6 * (SI0 + (P0 + p) / 3) + SJ0 + (P0 + p) % 3;
                    M[pos][0] = v4.x; M[pos][1] = v4.y; M[pos][2] = v4.z; M[pos][3] = v4.w;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        pass(std::integral_constant<int, 0>(), std::integral_constant<int, 5>());
        pass(std::integral_constant<int, 5>(), std::integral_constant<int, 4>());
    };
    if (pg == 0) epilogue(std::integral_constant<int, 0>());
    else if (pg == 1) epilogue(std::integral_constant<int, 1>());
    else if (pg == 2) epilogue(std::integral_constant<int, 2>());
    else epilogue(std::integral_constant<int, 3>());

    // ---- Y = A^T M A per entry, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1] ------------------------------------------------
    auto at6 = [&](const float m0, const float m1, const float m2, const float m3, const float m4, const float m5, float *y) INL {
        const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
        y[0] = m0 + s1 + s2;
        y[1] = __builtin_fmaf(2.f, d2, d1);
        y[2] = __builtin_fmaf(4.f, s2, s1);
        y[3] = __builtin_fmaf(8.f, d2, d1) + m5;
    };
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float R[4][6];                                                        // R[a][j] = sum_i At[a][i] M[i][j]
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float y[4];
            at6(M[j][e], M[6 + j][e], M[12 + j][e], M[18 + j][e], M[24 + j][e], M[30 + j][e], y);
#pragma unroll
            for (int a = 0; a < 4; ++a) R[a][j] = y[a];
        }
        const int tile = 8 * pg + e + 4 * kh;
        float *o = out + ((long)(mt * 32 + tile) * 16) * Npad + ncol;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float y[4];
            at6(R[a][0], R[a][1], R[a][2], R[a][3], R[a][4], R[a][5], y);
#pragma unroll
            for (int b = 0; b < 4; ++b) o[(long)(a * 4 + b) * Npad] = y[b];
        }
    }
#endif
}

// host: logical operands -> device layouts
static void pack_v(const std::vector<float> &Vl, std::vector<float> &Vd, int MT, int nchunks) {
    // Vl[mt][chunk][xi][tile][16]  ->  image [xi][tile][piece ^ swizzle][4], piece = 2 kb + kh holds channels 8 kb + kh + 2 m
    for (long blk = 0; blk < (long)MT * nchunks; ++blk)
        for (int xi = 0; xi < 36; ++xi)
            for (int t = 0; t < 32; ++t)
                for (int kb = 0; kb < 2; ++kb)
                    for (int kh = 0; kh < 2; ++kh)
                        for (int m = 0; m < 4; ++m) {
                            const int piece = (kb * 2 + kh) ^ ((t >> 2) & 3);
                            Vd[blk * BUF + (xi * 32 + t) * 16 + piece * 4 + m] = Vl[blk * BUF + (xi * 32 + t) * 16 + 8 * kb + kh + 2 * m];
                        }
}
static void pack_u(const std::vector<float> &Ul, std::vector<float> &Ud, int C, int Npad) {
    // Ul[xi][c][n] -> Ud[s8][xi][n][kh][m], channel 8 s8 + kh + 2 m
    for (int s8 = 0; s8 < C / 8; ++s8)
        for (int xi = 0; xi < 36; ++xi)
            for (int n = 0; n < Npad; ++n)
                for (int kh = 0; kh < 2; ++kh)
                    for (int m = 0; m < 4; ++m)
                        Ud[((((long)s8 * 36 + xi) * Npad + n) * 2 + kh) * 4 + m] = Ul[((long)xi * C + 8 * s8 + kh + 2 * m) * Npad + n];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static int run(int MT, int C, int Npad, bool check, int reps) {
    const int nchunks = C / 16, NT = Npad / 64;
    std::vector<float> Vl((size_t)MT * nchunks * BUF), Ul((size_t)36 * C * Npad), Vd(Vl.size()), Ud(Ul.size());
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto &v : Vl) v = rnd();
    for (auto &v : Ul) v = rnd() * 0.25f;
    pack_v(Vl, Vd, MT, nchunks);
    pack_u(Ul, Ud, C, Npad);
    float *dV, *dU, *dO;
    const size_t osz = (size_t)MT * 32 * 16 * Npad;
    CK(hipMalloc(&dV, Vd.size() * 4)); CK(hipMalloc(&dU, Ud.size() * 4)); CK(hipMalloc(&dO, osz * 4));
    CK(hipMemcpy(dV, Vd.data(), Vd.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dU, Ud.data(), Ud.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dO, 0, osz * 4));
    hipLaunchKernelGGL(wino44_gemm, dim3(MT * NT), dim3(512), 0, 0, dV, dU, dO, MT, NT, nchunks, Npad);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    if (check) {
        std::vector<float> O(osz);
        CK(hipMemcpy(O.data(), dO, osz * 4, hipMemcpyDeviceToHost));
        static const double At[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
        double worst = 0, scale = 0;
        long bad = 0;
        for (int mt = 0; mt < MT; ++mt)
            for (int t = 0; t < 32; ++t)
                for (int n = 0; n < Npad; ++n) {
                    double M[36];
                    for (int xi = 0; xi < 36; ++xi) {
                        double a = 0;
                        for (int c = 0; c < C; ++c)
                            a += (double)Vl[((size_t)(mt * nchunks + c / 16) * 36 + xi) * 32 * 16 + t * 16 + c % 16] * Ul[((size_t)xi * C + c) * Npad + n];
                        M[xi] = a;
                    }
                    for (int a = 0; a < 4; ++a)
                        for (int b = 0; b < 4; ++b) {
                            double y = 0;
                            for (int i = 0; i < 6; ++i)
                                for (int j = 0; j < 6; ++j) y += At[a][i] * M[6 * i + j] * At[b][j];
                            const double g = O[((size_t)(mt * 32 + t) * 16 + a * 4 + b) * Npad + n];
                            const double d = fabs(g - y);
                            if (d > worst) worst = d;
                            if (fabs(y) > scale) scale = fabs(y);
                            if (d > 1e-3 + 1e-4 * fabs(y)) ++bad;
                        }
                }
        printf("check MT=%d C=%d Npad=%d: worst |diff| %.3e (largest |y| %.3f), entries out of tolerance: %ld\n", MT, C, Npad, worst, scale, bad);
        if (bad) return 2;
    }
    if (reps > 0) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(wino44_gemm, dim3(MT * NT), dim3(512), 0, 0, dV, dU, dO, MT, NT, nchunks, Npad);
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wino44_gemm, dim3(MT * NT), dim3(512), 0, 0, dV, dU, dO, MT, NT, nchunks, Npad);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / reps, flop = 2.0 * MT * 32 * 36 * (double)C * Npad;
        printf("time MT=%d C=%d Npad=%d (EPI=%d): %.1f us per launch, %.1f TFLOP/s executed = %.3f of the 157.3 fp32 MFMA peak\n", MT, C, Npad, W44_EPI, us, flop / us * 1e-6,
               flop / us * 1e-6 / 157.3);
    }
    hipFree(dV); hipFree(dU); hipFree(dO);
    return 0;
}

int main(int argc, char **argv) {
    if (W44_EPI) {
        if (int rc = run(3, 32, 64, true, 0)) return rc;
        if (int rc = run(8, 64, 128, true, 0)) return rc;                  // (an even number of 16-channel chunks)
        if (int rc = run(5, 128, 256, true, 0)) return rc;
    }
    // the ConvLSTM cell of config 2: 8 x 128 x 128 pixels = 8192 tiles of 4x4, 128 input channels, 256 columns
    return run(256, 128, 256, false, 200);
}
