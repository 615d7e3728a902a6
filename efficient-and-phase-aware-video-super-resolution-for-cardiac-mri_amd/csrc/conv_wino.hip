// 3x3 convolution (padding 1) in Winograd form F(2x2, 3x3) on fp32 MFMA for gfx950 - rnh_conv_wino.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 independent GEMMs (one per position xi of the 4x4 transform domain) of [tiles x C] x [C x N]: 4 MACs per output
// pixel and (c, n) pair instead of 9, i.e. 2.25x fewer MFMA passes than the implicit GEMM of conv_igemm.hip.  The
// price is vector work next to the matrix cores (measured with tools/issue_density.hip: a wave sustains one 8-byte
// load and one packed add per v_mfma_f32_32x32x2_f32 at about 75 % of the MFMA peak), so everything is fused:
//
//   * one wave = 32 tiles (rows of the MFMA) x 32 output columns x all 16 xi: 256 accumulator registers (the
//     unified 512-register file of a wave that has its SIMD to itself), so the output transform happens in registers
//     and the transform domain never touches memory;
//   * the input transform B^T d B is computed on the fly: per step of 4 channels a lane loads the 4x4 patch of its
//     tile as 16 raw 8-byte buffer loads (channels 2kh, 2kh+1 of the step for lane-half kh; lanes outside the image
//     carry offset 0xFFFFFFFF and the range check returns the zero padding) and spends 32 packed adds;
//   * the weights arrive pre-transformed from rnh_wino_pack_weights as U[step][xi / 2][n][lane half][xi & 1][2]: 16 bytes
//     per lane and PAIR of transform positions, and the staged input transform has the same pairing in LDS - one
//     buffer_load_dwordx4 / ds_read_b128 feeds four MFMAs (round 1 used 8-byte operands: twice the operand instructions
//     beside the MFMA stream, each worth about 8 cycles of matrix-core idle time);
//   * the 4 waves of a workgroup take 4 column groups of the same 32 tiles.  With the ConvLSTM column order
//     (plans.lstm_colmap) these are the 4 gates of 32 hidden channels: every wave activates its gate, the gates meet
//     in LDS and each wave finishes one pixel of every tile (c' = f c + i g, h' = o tanh c').
//
// Same operand conventions as rnh_conv_igemm (rnh_conv_args_t: multi-source K without concatenation, destination
// segments, packed bias, pixel-unshuffled sources of one common scale).  Epilogues: STORE, PS, LSTM.
#include "rnh_common.h"
#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4w __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// v_exp_f32 / v_rcp_f32 (1 ulp each; __frcp_rn would be a correctly rounded division: two v_div_scale, v_rcp, four FMAs,
// v_div_fmas, v_div_fixup per value) and no branch: both forms of tanh are computed and selected (hipcc turned the
// ternary around the exp form into an exec-masked branch per value, 64 of them per lane in the gate epilogue).
__device__ __forceinline__ float w_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float w_tanh(float x) {
    const float ax = fabsf(x);
    const float big = __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * ax)), 1.f);
    const float small = ax * __builtin_fmaf(-0.33333334f * ax, ax, 1.f);      // |x| < 0.04: the exp form cancels
    const float t = ax < 0.04f ? small : big;
    return copysignf(t, x);
}
// tanh for the candidate gate g: 2 sigmoid(2x) - 1, the same instruction count as the sigmoid of the other three gate
// waves (which wait for this one at the barrier).  Absolute error <= 2 ulp of 1 - what c' = f c + i g needs; h = o tanh(c')
// keeps the form above, which is also relatively accurate near 0.
__device__ __forceinline__ float w_tanh_gate(float x) { return __builtin_fmaf(2.f, __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)), -1.f); }

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

// wave-uniform descriptor: base + 2 GiB window, raw buffer (the readfirstlanes keep it in SGPRs - no waterfall loop)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wdesc(const float *p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(u & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
// the same descriptor as a plain SGPR quadruple for the asm loads
__device__ __forceinline__ i32x4 sdesc(const float *p) {
    const unsigned long long u = (unsigned long long)p;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    d[1] = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
    d[2] = 0x7fffffff;
    d[3] = 0x00020000;
    return d;
}
__device__ __forceinline__ f32x2 wld2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

#ifdef RNH_STAMPS
__device__ unsigned long long g_wino_stamps[8];
#define WSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_wino_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define WSTAMP(i)
#endif

// TG = tile groups per workgroup.  TG = 1: 32 tiles x 128 columns (4 waves = 4 column groups), 16-channel chunks.
// TG = 2 (convolutions with 64-column multiples, e.g. the 64-channel data gradients): 64 tiles x 64 columns, wave =
// (tile group, column group), 8-channel chunks so that the staging still is one (tile, channel pair) per thread.
template <int EPI, int TG>
__global__ void __launch_bounds__(256, 1) conv_wino_kernel(const rnh_conv_args_t P, const int MT, const int NT, const int TX, const int TY) {
    constexpr int TILES = 32 * TG, CPC = 8 / TG, CH = 2 * CPC, SPC = 4 / TG, CG = 4 / TG;   // tiles, channel pairs / channels / steps per chunk, column groups
    constexpr int CHS = 4 * CPC + 4, BUF = 8 * TILES * CHS;                                 // LDS row of one (xi pair, tile): [channel pair][xi & 1][2] + 4 pad; floats per buffer
    static_assert(EPI != RNH_EPI_LSTM || TG == 1, "the gate exchange needs the four column groups of one tile group");
    __shared__ __attribute__((aligned(16))) float stage[2 * BUF > 16384 ? 2 * BUF : 16384];   // 73.7 / 81.9 KB; the LSTM gate exchange reuses it
    float *xch = stage;
    __shared__ int tpix[TILES];                               // top-left output pixel of the block's tiles (epilogue)
    __shared__ int tcoord[TILES];                             // the same as (image << 20 | y << 10 | x), -1: no such tile
    WSTAMP(0);
#ifdef RNH_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) g_wino_stamps[6] = 0;
#endif
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5, wave = threadIdx.x >> 6;
    const int bid = rnh_xcd_remap(blockIdx.x, MT * NT);
    const int mt = bid / NT, nt = bid - mt * NT;
    const int H = P.H, W = P.W, ntiles = P.B * TY * TX;
    const int m0 = mt * TILES;
    const int tg = wave / CG, cg = wave - tg * CG;            // this wave's tile group and column group

    // ---- staging: B^T d B of the block's 32 tiles, 16 channels at a time, through LDS ---------------------------
    // (Lanes of one MFMA row block sit 2 pixels = 512 B apart in memory: loading patches per lane would touch 32 cache
    // lines per instruction; and the four waves of the block need the same transformed patches.)  Thread = (tile ts,
    // channel pair cp of the chunk): 16 8-byte loads (the 8 threads of a tile read 64 contiguous bytes per pixel), the
    // input transform once per block, 8 16-byte LDS writes.  LDS layout [xi / 2][tile][36]: row = 8 channel pairs x
    // (xi even, xi odd) x 2 channels + 4 floats of pad; the 16-byte reads of 16 consecutive tiles (stride 36 dwords) touch
    // every bank once.
    const int ts = threadIdx.x / CPC, cp = threadIdx.x % CPC;
    const int t0 = m0 < ntiles ? m0 : 0;
    const int img0 = t0 / (TY * TX), r0 = t0 - img0 * TY * TX, ty0 = r0 / TX;
    // sources may be the (sub_y, sub_x) phase of a scale-times larger image (pixel-unshuffle fused into the load);
    // one scale for all sources, the phase goes into the descriptor base
    const int sc = P.src[0].scale, Hs = H * sc, Ws = W * sc;
    const int base_pix = (img0 * Hs + (2 * ty0 - 1) * sc) * Ws - sc;   // at or before every pixel the block touches
    int pixrel[16];
    {
        const int t = m0 + ts;
        const bool tok = t < ntiles;
        const int tt = tok ? t : t0;
        const int img = tt / (TY * TX), trem = tt - img * TY * TX, ty = trem / TX, tx = trem - ty * TX;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int y = 2 * ty - 1 + (p >> 2), x = 2 * tx - 1 + (p & 3);
            const bool ok = tok && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            pixrel[p] = ok ? (img * Hs + y * sc) * Ws + x * sc - base_pix : -1;
        }
    }
    // loader state: source and 16-channel chunk inside it
    int si = 0, cchunk = 0, nchunk = P.src[0].nch / CH;
    int voff[16];
    i32x4 adesc;
    auto setup_src = [&](int sidx) {
        const rnh_src_t &S = P.src[sidx];
        adesc = sdesc(S.ptr + S.c0 + ((long)S.img_off * Hs * Ws + base_pix + S.sub_y * Ws + S.sub_x) * S.C);
        const int C4 = S.C * 4;
#pragma unroll
        for (int p = 0; p < 16; ++p) voff[p] = pixrel[p] < 0 ? -1 : pixrel[p] * C4 + cp * 8;
        nchunk = S.nch / CH;
    };
    setup_src(0);

    // All vector-memory and LDS reads of the loop are volatile asm: they stay where they are written (hipcc sinks
    // plain loads to their first use and then waits for each one with vmcnt(0) / lgkmcnt(0) between two MFMAs), and the
    // waits are counted by hand.  Each wait names the registers it covers exactly once as "+v" operands, which is what
    // orders their uses behind it.
    // Eight 8-byte buffer loads in ONE asm statement: the SGPR operands (descriptor, offset) may have been written by
    // SALU / v_readfirstlane just before, and a VMEM instruction reading such a register needs 5 wait states that hipcc
    // does not add around inline asm; inside one statement nothing can be scheduled between the s_nop and the loads.
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
        asm volatile("s_waitcnt vmcnt(8)"
                     : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[3]), "+v"(stg[4]), "+v"(stg[5]), "+v"(stg[6]), "+v"(stg[7]),
                       "+v"(stg[8]), "+v"(stg[9]), "+v"(stg[10]), "+v"(stg[11]), "+v"(stg[12]), "+v"(stg[13]), "+v"(stg[14]),
                       "+v"(stg[15]));
        // a - b in ONE v_pk_add_f32 with negated second operand: hipcc scalarises a packed subtraction into two v_add_f32
        // (54 of the transform's 64 instructions were scalar; writing it as fma(b, -1, a) is folded back into the same)
        auto sub = [&](f32x2 a, f32x2 b) {
            f32x2 r;
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            return r;
        };
        auto add = [&](f32x2 a, f32x2 b) {                 // (packed additions are scalarised as well)
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
            // positions 4i, 4i + 1 and 4i + 2, 4i + 3: two pairs, 16 bytes each (xi even, xi odd)
            const f32x2 v0 = sub(tq[i * 4 + 0], tq[i * 4 + 2]), v1 = add(tq[i * 4 + 1], tq[i * 4 + 2]);
            const f32x2 v2 = sub(tq[i * 4 + 2], tq[i * 4 + 1]), v3 = sub(tq[i * 4 + 1], tq[i * 4 + 3]);
            *reinterpret_cast<f32x4w *>(o + (i * 2 + 0) * TILES * CHS) = __builtin_shufflevector(v0, v1, 0, 1, 2, 3);
            *reinterpret_cast<f32x4w *>(o + (i * 2 + 1) * TILES * CHS) = __builtin_shufflevector(v2, v3, 0, 1, 2, 3);
        }
    };

    const i32x4 bdesc = sdesc(P.wp + (long)((nt * CG + cg) * 32) * 8);
    const int pstride = P.Npad * 32;                        // bytes between two PAIRS of transform positions of one step
    int boffx[8];                                           // per-lane byte offset of the 8 pairs inside a step
#pragma unroll
    for (int pr = 0; pr < 8; ++pr) boffx[pr] = (l31 * 2 + kh) * 16 + pr * pstride;
    auto loadb = [&](f32x4w *u, int sb) {                   // transformed weights of step sb: 8 loads of 16 bytes
        const int soff = __builtin_amdgcn_readfirstlane(sb * 8 * pstride);
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dwordx4 %0, %8, %16, %17 offen\n\t"
            "buffer_load_dwordx4 %1, %9, %16, %17 offen\n\t"
            "buffer_load_dwordx4 %2, %10, %16, %17 offen\n\t"
            "buffer_load_dwordx4 %3, %11, %16, %17 offen\n\t"
            "buffer_load_dwordx4 %4, %12, %16, %17 offen\n\t"
            "buffer_load_dwordx4 %5, %13, %16, %17 offen\n\t"
            "buffer_load_dwordx4 %6, %14, %16, %17 offen\n\t"
            "buffer_load_dwordx4 %7, %15, %16, %17 offen"
            : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(u[5]), "=&v"(u[6]), "=&v"(u[7])
            : "v"(boffx[0]), "v"(boffx[1]), "v"(boffx[2]), "v"(boffx[3]), "v"(boffx[4]), "v"(boffx[5]), "v"(boffx[6]), "v"(boffx[7]), "s"(bdesc),
              "s"(soff)
            : "memory");
    };
    const unsigned lds0 = (unsigned)(size_t)stage;          // LDS byte address of the staging area
    const unsigned vlane = lds0 + ((tg * 32 + l31) * CHS + 4 * kh) * 4;
    auto loadv = [&](f32x4w *V, int buf, int q) {           // the lane's tile, channels 4q + 2kh, +1 of the staged chunk, 8 pairs of positions
        const unsigned adr = vlane + buf * BUF * 4 + q * 32;
#define RNH_DSR(pr) asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(V[pr]) : "v"(adr), "i"((pr) * TILES * CHS * 4) : "memory")
        RNH_DSR(0); RNH_DSR(1); RNH_DSR(2); RNH_DSR(3); RNH_DSR(4); RNH_DSR(5); RNH_DSR(6); RNH_DSR(7);
#undef RNH_DSR
    };
    auto wait_lds = [&](f32x4w *V) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(V[4]), "+v"(V[5]), "+v"(V[6]), "+v"(V[7]));
    };
    auto wait_vm = [&](f32x4w *u, auto keep) {
        asm volatile("s_waitcnt vmcnt(%c8)"
                     : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
                     : "i"(decltype(keep)::value));
    };

    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[xi][v] = 0.f;

    auto compute = [&](const f32x4w *V, const f32x4w *u) {
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) {
            acc[2 * pr] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[pr].x, u[pr].x, acc[2 * pr], 0, 0, 0);
            acc[2 * pr] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[pr].y, u[pr].y, acc[2 * pr], 0, 0, 0);
            acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[pr].z, u[pr].z, acc[2 * pr + 1], 0, 0, 0);
            acc[2 * pr + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[pr].w, u[pr].w, acc[2 * pr + 1], 0, 0, 0);
        }
    };

    // ---- main loop over 16-channel chunks (4 steps of 4 channels); chunk c+1 travels global -> registers under the
    // MFMAs of chunk c and is transformed into the other LDS buffer at its end ---------------------------------------
    int nchunks_total = 0;
    for (int i = 0; i < P.nsrc; ++i) nchunks_total += P.src[i].nch / CH;
    f32x4w V0[8], V1[8], u0[8], u1[8];
    if (threadIdx.x < TILES) {
        const int tr = m0 + threadIdx.x, tq = tr < ntiles ? tr : t0;
        const int im = tq / (TY * TX), rr = tq - im * TY * TX, yy = rr / TX, xx = rr - yy * TX;
        tpix[threadIdx.x] = (im * H + 2 * yy) * W + 2 * xx;
        tcoord[threadIdx.x] = tr < ntiles ? (im << 20) | (2 * yy << 10) | (2 * xx) : -1;
    }
    WSTAMP(1);
    gload();
    loadb(u0, 0);                                            // 8 loads younger than the staging loads: vmcnt(8) in xform_store
    xform_store(0);
    __syncthreads();
    WSTAMP(2);
    int s = 0;                                               // global 4-channel step index (weights)
    loadv(V0, 0, 0);
    using K16 = std::integral_constant<int, 8>;              // the 8 weight loads of ONE step may stay in flight
    using K0 = std::integral_constant<int, 0>;
    // One chunk = 4 steps.  The loop body (every chunk but the last) has no branch: a register that is the target of an
    // asynchronous asm load must have exactly one definition per iteration, or hipcc reconciles the definitions at the
    // join with v_mov copies - executed before the load has landed (observed: about one workgroup in 10^5 summed
    // stale operands).  The last chunk is peeled off through the same lambda.
    auto chunk = [&](const int buf, auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
        // step 0: [staging loads of the next chunk] [weights of step 1] | MFMAs of step 0
        wait_lds(V0);
        loadv(V1, buf, 1);
        // (the staging loads are issued behind the wait, not in front of it: a staging load whose 64 lanes are all
        // outside the image never goes to memory and returns ahead of older loads, so it must not be among the loads
        // a counted wait leaves in flight)
        wait_vm(u0, K0());
        if constexpr (more) gload();
        loadb(u1, s + 1);
        compute(V0, u0);
        if constexpr (SPC == 4) {
            // step 1
            wait_lds(V1);
            loadv(V0, buf, 2);
            loadb(u0, s + 2);
            wait_vm(u1, K16());
            compute(V1, u1);
            // step 2: also the transform of the staged chunk into the other LDS buffer (nobody reads it during this chunk;
            // its loads are older than the 16 weight loads the wait above leaves in flight): plain code in front of the
            // MFMAs, so that hipcc interleaves its packed adds and LDS writes with them
            wait_lds(V0);
            loadv(V1, buf, 3);
            loadb(u1, s + 3);
            wait_vm(u0, K16());
            if constexpr (more) xform_store(buf ^ 1);
            compute(V0, u0);
            // last step.  The chunk's barrier sits HERE, in front of the last 32 MFMAs, not behind them: every wave has
            // issued all its reads of this buffer and finished its writes of the other one (lgkmcnt(0)), so after the
            // barrier the first operands of the next chunk can be fetched from LDS under the cover of this step's MFMAs.
            // (Behind the MFMAs the barrier's skew and the LDS latency were exposed: 12.5 k of the loop's 96 k cycles, of
            // which this order recovers 4.5 k; timing experiments: no barrier at all 83.5 k, no weight loads 83.1 k.)
            // Not __syncthreads(): its fence would also wait (vmcnt(0)) for the weight prefetch in flight.
            wait_lds(V1);
            asm volatile("s_barrier" ::: "memory");
            if constexpr (more) {
                loadv(V0, buf ^ 1, 0);
                loadb(u0, s + SPC);
                wait_vm(u1, K16());
            } else {
                wait_vm(u1, K0());
            }
            compute(V1, u1);
            s += SPC;
        } else {
            // two-step chunks (TG = 2): the staging loads were issued one step ago, so the transform stays in the last
            // step and the barrier behind it.  Nothing is prefetched past the end.
            wait_lds(V1);
            if constexpr (more) {
                loadb(u0, s + SPC);
                wait_vm(u1, K16());
                xform_store(buf ^ 1);
            } else {
                wait_vm(u1, K0());
            }
            compute(V1, u1);
            s += SPC;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if constexpr (more) loadv(V0, buf ^ 1, 0);
        }
    };
    for (int c = 0; c + 1 < nchunks_total; ++c) chunk(c & 1, std::true_type());
    chunk((nchunks_total - 1) & 1, std::false_type());

    WSTAMP(3);
    // ---- output transform Y = A^T M A per accumulator register, then the epilogue -------------------------------
    const int ncol = (nt * CG + cg) * 32 + l31;
    const int trow0 = tg * 32;                                  // first row of this wave's tile group in tpix / tcoord
    const float bv = P.bias ? P.bias[ncol] : 0.f;
    auto out4 = [&](int v, float *Y) {
        float sq[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sq[0][j] = acc[0 * 4 + j][v] + acc[1 * 4 + j][v] + acc[2 * 4 + j][v];
            sq[1][j] = acc[1 * 4 + j][v] - acc[2 * 4 + j][v] - acc[3 * 4 + j][v];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            Y[a * 2 + 0] = sq[a][0] + sq[a][1] + sq[a][2] + bv;
            Y[a * 2 + 1] = sq[a][1] - sq[a][2] - sq[a][3] + bv;
        }
    };
    // pixel (top-left output of the tile) and validity of row v of the MFMA tile
    auto tile_of = [&](int v, int &pix, bool &okx, bool &oky) -> bool {
        const int tr = m0 + trow0 + (v & 3) + 8 * (v >> 2) + 4 * kh;
        if (tr >= ntiles) return false;
        const int im = tr / (TY * TX), rr = tr - im * TY * TX, yy = rr / TX, xx = rr - yy * TX;
        pix = (im * H + 2 * yy) * W + 2 * xx;
        oky = 2 * yy + 1 < H;
        okx = 2 * xx + 1 < W;
        return true;
    };

    if constexpr (EPI == RNH_EPI_LSTM) {
        const int hd = P.hd, hc = nt * 32 + l31;
        // every row of the MFMA tile is whole and inside (the usual case): straight-line code, no per-element predicates
        const bool full = m0 + 32 <= ntiles && !(H & 1) && !(W & 1) && nt * 32 + 32 <= hd;
        const int p2 = wave, poff2 = (p2 >> 1) * W + (p2 & 1);
        float cpv[16];
        // (the previous cell state of the pixels this lane finishes is fetched in phase 2, all 16 values in one batch behind
        // the barrier: requested here, hipcc sank the loads to the end of phase 1 anyway - no register is free during the
        // gate math - and their wait then also covered the 64 gate stores in front of them)
        // phase 1: every wave activates its gate (wave 0..3 = i, f, o, g) and parks it in LDS (and in gates_out).
        // GV accumulator registers at a time: 4 GV independent exp / rcp chains for the one wave on this SIMD (measured
        // per workgroup, gate phase: GV = 1: 25.2k cycles, 2: 21.6k, 4: 18.2k, 8: 17.4k, 16: 16.3k).
#ifndef RNH_WINO_GV
#define RNH_WINO_GV 16
#endif
        constexpr int GV = RNH_WINO_GV;
        if (full) {
#pragma unroll
            for (int v0 = 0; v0 < 16; v0 += GV) {
                float Y[GV][4], g[GV][4];
#pragma unroll
                for (int dv = 0; dv < GV; ++dv) out4(v0 + dv, Y[dv]);
                if (wave == 3) {
#pragma unroll
                    for (int e = 0; e < 4 * GV; ++e) g[e >> 2][e & 3] = w_tanh_gate(Y[e >> 2][e & 3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4 * GV; ++e) g[e >> 2][e & 3] = w_sigmoid(Y[e >> 2][e & 3]);
                }
#pragma unroll
                for (int dv = 0; dv < GV; ++dv) {
                    const int v = v0 + dv, trl = (v & 3) + 8 * (v >> 2) + 4 * kh;
                    float *gp = P.gates_out ? P.gates_out + (long)tpix[trl] * 4 * hd + wave * hd + hc : nullptr;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        xch[((wave * 32 + trl) * 4 + p) * 32 + l31] = g[dv][p];
                        if (gp) gp[(long)((p >> 1) * W + (p & 1)) * 4 * hd] = g[dv][p];
                    }
                }
            }
        } else {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float Y[4];
                out4(v, Y);
                const int trl = (v & 3) + 8 * (v >> 2) + 4 * kh;
                int pix;
                bool okx, oky;
                const bool ok = tile_of(v, pix, okx, oky) && hc < hd;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float g = wave == 3 ? w_tanh(Y[p]) : w_sigmoid(Y[p]);
                    xch[((wave * 32 + trl) * 4 + p) * 32 + l31] = g;
                    if (P.gates_out && ok && ((p & 1) == 0 || okx) && ((p >> 1) == 0 || oky))
                        P.gates_out[((long)pix + (p >> 1) * W + (p & 1)) * 4 * hd + wave * hd + hc] = g;
                }
            }
        }
        // the gates are in LDS: wait for the LDS writes only - __syncthreads() would also wait (vmcnt(0)) for the gates_out
        // stores just issued to be acknowledged by memory
        WSTAMP(7);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        WSTAMP(4);
        // phase 2: wave w finishes output pixel w of every tile
        if (full) {
            if (P.c_prev) {
                const float *cpb = P.c_prev + (long)poff2 * hd + hc;
#pragma unroll
                for (int v = 0; v < 16; ++v) cpv[v] = cpb[(long)tpix[(v & 3) + 8 * (v >> 2) + 4 * kh] * hd];
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v) cpv[v] = 0.f;
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int trl = (v & 3) + 8 * (v >> 2) + 4 * kh;
                const float gi = xch[((0 * 32 + trl) * 4 + p2) * 32 + l31], gf = xch[((1 * 32 + trl) * 4 + p2) * 32 + l31];
                const float go = xch[((2 * 32 + trl) * 4 + p2) * 32 + l31], gg = xch[((3 * 32 + trl) * 4 + p2) * 32 + l31];
                const long o = ((long)tpix[trl] + poff2) * hd + hc;
                const float cn = gf * cpv[v] + gi * gg;
                P.c_out[o] = cn;
                P.h_out[o] = go * w_tanh(cn);
            }
        } else {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int trl = (v & 3) + 8 * (v >> 2) + 4 * kh;
                int pix;
                bool okx, oky;
                if (!tile_of(v, pix, okx, oky) || hc >= hd) continue;
                if (((p2 & 1) && !okx) || ((p2 >> 1) && !oky)) continue;
                const float gi = xch[((0 * 32 + trl) * 4 + p2) * 32 + l31], gf = xch[((1 * 32 + trl) * 4 + p2) * 32 + l31];
                const float go = xch[((2 * 32 + trl) * 4 + p2) * 32 + l31], gg = xch[((3 * 32 + trl) * 4 + p2) * 32 + l31];
                const long o = ((long)pix + poff2) * hd + hc;
                const float cp = P.c_prev ? P.c_prev[o] : 0.f;
                const float cn = gf * cp + gi * gg;
                P.c_out[o] = cn;
                P.h_out[o] = go * w_tanh(cn);
            }
        }
        WSTAMP(5);
    } else {
        if constexpr (EPI == RNH_EPI_PS) {
            // column n = (i*r + j)*cq + c  ->  pixel (r*y + i, r*x + j), channel c of the (B, rH, rW, cq) destination
            const int r = P.ps_r, cq = P.ps_cq;
            if (ncol >= cq * r * r) return;
            const int sub = ncol / cq, c = ncol - sub * cq, pi = sub / r, pj = sub - pi * r;
            float *dp = P.dst[0].ptr + c;
            const long Wr = (long)W * r;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float Y[4];
                out4(v, Y);
                const int tc = tcoord[trow0 + (v & 3) + 8 * (v >> 2) + 4 * kh];
                if (tc < 0) continue;
                const int im = tc >> 20, yy = (tc >> 10) & 1023, xx = tc & 1023;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int y = yy + (p >> 1), x = xx + (p & 1);
                    if (y >= H || x >= W) continue;
                    dp[(((long)im * H + y) * r + pi) * Wr * cq + ((long)x * r + pj) * cq] = Y[p];
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
                for (int v = 0; v < 16; ++v) {
                    float Y[4];
                    out4(v, Y);
                    float *o = dp + (long)tpix[trow0 + (v & 3) + 8 * (v >> 2) + 4 * kh] * D.C;
                    if (D.accumulate) {
                        const float a0 = o[0], a1 = o[D.C], a2 = o[rowC], a3 = o[rowC + D.C];
                        o[0] = a0 + Y[0]; o[D.C] = a1 + Y[1]; o[rowC] = a2 + Y[2]; o[rowC + D.C] = a3 + Y[3];
                    } else {
                        o[0] = Y[0]; o[D.C] = Y[1]; o[rowC] = Y[2]; o[rowC + D.C] = Y[3];
                    }
                }
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    float Y[4];
                    out4(v, Y);
                    int pix;
                    bool okx, oky;
                    if (!tile_of(v, pix, okx, oky)) continue;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        if (((p & 1) && !okx) || ((p >> 1) && !oky)) continue;
                        float *o = dp + ((long)pix + (p >> 1) * W + (p & 1)) * D.C;
                        *o = D.accumulate ? *o + Y[p] : Y[p];
                    }
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

extern "C" int rnh_conv_wino(const rnh_conv_args_t *args, void *stream) {
    if (!args) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: null args");
    const rnh_conv_args_t &a = *args;
    if (a.nsrc < 1 || a.nsrc > RNH_MAX_SRC || a.B < 1 || a.H < 1 || a.W < 1 || !a.wp) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: bad arguments");
    if (a.ntaps != 9) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: 3x3 convolutions only");
    if (a.Npad < 64 || a.Npad % 64) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: Npad must be a multiple of 64");
    const int TG = a.Npad % 128 ? 2 : 1;              // 64-column multiples: two tile groups x two column groups per workgroup
    int steps = 0;
    for (int i = 0; i < a.nsrc; ++i) {
        if (int rc = rnh_check_src(a.src[i], "rnh_conv_wino")) return rc;
        if (a.src[i].scale != a.src[0].scale || a.src[i].ptr2) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: one scale for all sources, no second pointer");
        if (a.src[i].nch & (TG == 1 ? 15 : 7)) RNH_FAIL(RNH_E_ALIGN, "rnh_conv_wino: source channel counts must be multiples of %d", 16 / TG);
        steps += a.src[i].nch / 4;
    }
    if (steps != a.nk) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: nk = %d but the sources hold %d steps of 4 channels", a.nk, steps);
    const int TY = (a.H + 1) / 2, TX = (a.W + 1) / 2;
    const long ntiles = (long)a.B * TY * TX;
    if (a.H > 1023 || a.W > 1023 || a.B > 2047) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: at most 2047 images of 1023 x 1023");
    if (ntiles * 4 >= (1L << 29)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: too many pixels for 32-bit offsets");
    const int MT = (int)((ntiles + 32 * TG - 1) / (32 * TG)), NT = a.Npad / (TG == 1 ? 128 : 64);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(MT * NT)), block(256);
    switch (a.epilogue) {
        case RNH_EPI_STORE:
            if (a.ndst < 1 || a.ndst > RNH_MAX_DST) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: bad destination count");
            for (int d = 0; d < a.ndst; ++d)
                if (!a.dst[d].ptr || a.dst[d].ncols < 1) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: bad destination %d", d);
            if (TG == 1) hipLaunchKernelGGL((conv_wino_kernel<RNH_EPI_STORE, 1>), grid, block, 0, st, a, MT, NT, TX, TY);
            else hipLaunchKernelGGL((conv_wino_kernel<RNH_EPI_STORE, 2>), grid, block, 0, st, a, MT, NT, TX, TY);
            break;
        case RNH_EPI_PS:
            if (a.ndst != 1 || !a.dst[0].ptr || a.ps_r < 1 || a.ps_cq < 1 || a.ps_cq * a.ps_r * a.ps_r > a.Npad)
                RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: bad pixel-shuffle destination");
            if (TG == 1) hipLaunchKernelGGL((conv_wino_kernel<RNH_EPI_PS, 1>), grid, block, 0, st, a, MT, NT, TX, TY);
            else hipLaunchKernelGGL((conv_wino_kernel<RNH_EPI_PS, 2>), grid, block, 0, st, a, MT, NT, TX, TY);
            break;
        case RNH_EPI_LSTM:
            if (!a.h_out || !a.c_out || a.hd < 1 || !a.bias) RNH_FAIL(RNH_E_ARG, "rnh_conv_wino: LSTM epilogue needs h_out, c_out, hd, bias");
            if (a.Npad != 128 * ((a.hd + 31) / 32)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: LSTM column layout (plans.lstm_colmap)");
            hipLaunchKernelGGL((conv_wino_kernel<RNH_EPI_LSTM, 1>), grid, block, 0, st, a, MT, NT, TX, TY);
            break;
        default:
            RNH_FAIL(RNH_E_RANGE, "rnh_conv_wino: epilogue %d not available", a.epilogue);
    }
    RNH_CHECK_LAUNCH("rnh_conv_wino");
    return 0;
}
