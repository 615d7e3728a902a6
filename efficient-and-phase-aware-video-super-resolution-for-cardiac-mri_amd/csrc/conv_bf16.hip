// 3x3 (padding 1) / 1x1 convolution as an implicit GEMM on bf16 MFMA for gfx950 - rnh_conv_bf16: the bf16-storage
// form of the call sites of rnh_conv_igemm (reference src/model/nets/refine_net.py:149-154, :199-205, :235-265 and their
// data gradients; the reference itself is fp32 throughout - BASELINE.json configs[2] asks for this path).
//
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the fp32 MFMA, so this kernel is not MFMA-bound but bound by how fast
// operands reach the matrix cores; the design therefore moves every input byte as few times as possible:
//
//   * one workgroup (4 waves) = TH x TW = 8 x 32 output pixels of one image x NCOLS = 128 (64) output columns;
//     wave = (pixel half: 4 rows of 32 pixels) x (column half): 4 x NB accumulator tiles of 32 x 32;
//   * per 16-channel chunk of the K dimension the 10 x 34 pixel HALO of the tile is staged ONCE in LDS (bf16; fp32
//     sources are converted on the way with v_cvt_pk_bf16_f32) and serves all 9 taps - the A operand of tap (dy, dx)
//     is the same LDS image read at a shifted address; the chunk's packed weights of all taps (9 x NCOLS x 16) are staged
//     beside it.  Both images use a 48-byte pitch per pixel / column (32 B of data + 16 B pad): a 16-byte fragment read of
//     32 consecutive pixels then touches every bank exactly once per 16-lane group (12 i mod 64, i = 0..15, are distinct
//     multiples of 4);
//   * two LDS buffers, one barrier per chunk: chunk c+1 is written (from registers loaded a chunk earlier) while chunk c
//     is being multiplied, chunk c+2 is requested right behind the barrier;
//   * epilogue through LDS: the waves park their fp32 accumulators (+ bias) as a [256 pixels][NCOLS] tile, then all 256
//     threads finish 8 columns of a pixel at a time with 16-byte global accesses - STORE (segments, optional accumulate,
//     fp32 or bf16), PS (PixelShuffle fused into the store) or LSTM (the four gates of 8 hidden channels meet in one
//     thread: sigmoid / tanh, c' = f c + i g, h' = o tanh c'; c stays fp32).
//
// MFMA operand maps (cdna_hip_programming.md section 3): lane l, r = l & 31, h = l >> 5 holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r], j = 0..7; C/D: col = l & 31, row = (v & 3) + 8 (v >> 2) + 4 h.  Rows = the 32 pixels of one
// image-row segment, k = channel inside the chunk.
#include "rnh_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int TH = 8, TW = 32, HPW = TW + 2, HPH = TH + 2, HP = HPW * HPH;   // halo: 10 x 34 = 340 pixels
constexpr int PITCH = 48;                                                    // bytes per halo pixel / weight column
constexpr int A_BYTES = HP * PITCH;                                          // 16 320
constexpr int A_PIECES = 2 * HP;                                             // 16-byte pieces (8 channels) per chunk
constexpr int A_ITERS = (A_PIECES + 255) / 256;                              // 3

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

template <int NCOLS>
struct Geo {
    static constexpr int NB = NCOLS / 64;                       // 32-column MFMA blocks per wave
    static constexpr int B_BYTES = 9 * NCOLS * PITCH;
    static constexpr int BUF = A_BYTES + B_BYTES;
    static constexpr int OPITCH = NCOLS + 4;                    // floats per pixel of the parked output tile
    static constexpr int OUT_BYTES = TH * TW * OPITCH * 4;
    static constexpr int SMEM = 2 * BUF > OUT_BYTES ? 2 * BUF : OUT_BYTES;
    static constexpr int B_PIECES_TAP = NCOLS * 2;              // 16-byte pieces of one tap
};

template <int EPI, int NCOLS, int NTAPS>
__global__ void __launch_bounds__(256, 1) conv_bf16_kernel(const rnh_conv_bf16_args_t P, const int TYn, const int TXn, const int NT) {
    using G = Geo<NCOLS>;
    constexpr int NB = G::NB;
    constexpr int B_ITERS = (NTAPS * G::B_PIECES_TAP + 255) / 256;
    constexpr bool B_EXACT = (NTAPS * G::B_PIECES_TAP) % 256 == 0;          // every thread has a piece in every iteration
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::SMEM];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5, wave = tid >> 6;
    const int ph = wave & 1, chalf = wave >> 1;
    const int bid = rnh_xcd_remap(blockIdx.x, P.B * TYn * TXn * NT);
    const int nt = bid % NT, mt = bid / NT;
    const int img = mt / (TYn * TXn), trem = mt - img * (TYn * TXn), ty = trem / TXn, tx = trem - ty * TXn;
    const int y0 = ty * TH, x0 = tx * TW;
    const int H = P.H, W = P.W;
    const int sc = P.src[0].scale, Hs = H * sc, Ws = W * sc;

    // ---- this thread's pieces of the halo (fixed for the whole K loop) --------------------------------------------
    long apix[A_ITERS];           // source pixel index inside an image (scale applied), -1: zero padding / no piece
    int alds[A_ITERS], ahalf[A_ITERS];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int p = tid + 256 * i, px = p >> 1, hr = px / HPW, hc = px - hr * HPW;
        const int y = y0 - 1 + hr, x = x0 - 1 + hc;
        const bool in = p < A_PIECES && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        ahalf[i] = p & 1;
        alds[i] = p < A_PIECES ? px * PITCH + (p & 1) * 16 : -1;
        apix[i] = in ? (long)(y * sc) * Ws + x * sc : -1;
    }

    uint4 ra[A_ITERS], rb[B_ITERS];
    int si = 0, cc = 0;                                         // loader state: source, chunk inside it
    auto load_chunk = [&](int cg) {
        const rnh_msrc_t &S = P.src[si];
        const long ibase = ((long)(img + S.img_off) * Hs + S.sub_y) * Ws + S.sub_x;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int ch = cc * 16 + ahalf[i] * 8;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (apix[i] >= 0 && ch < S.nch) {
                const long e = (ibase + apix[i]) * S.C + S.c0 + ch;
                if (S.dtype == RNH_DT_BF16) {
                    v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned short *>(S.ptr) + e);
                } else {
                    const float *f = reinterpret_cast<const float *>(S.ptr) + e;
                    const float4 lo = *reinterpret_cast<const float4 *>(f);
                    const float4 hi = ch + 4 < S.nch ? *reinterpret_cast<const float4 *>(f + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    v = pack8(lo, hi);
                }
            }
            ra[i] = v;
        }
        const unsigned short *wb = reinterpret_cast<const unsigned short *>(P.wp) + ((long)cg * NTAPS * P.Npad + nt * NCOLS) * 16;
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int p = tid + 256 * i;
            rb[i] = make_uint4(0u, 0u, 0u, 0u);
            if (B_EXACT || p < NTAPS * G::B_PIECES_TAP) {
                const int tap = p / G::B_PIECES_TAP, rem = p - tap * G::B_PIECES_TAP;
                rb[i] = *reinterpret_cast<const uint4 *>(wb + (long)tap * P.Npad * 16 + rem * 8);
            }
        }
        if (++cc * 16 >= S.nch) {
            cc = 0;
            ++si;
        }
    };
    auto store_chunk = [&](int buf) {
        unsigned char *Ab = smem + buf * G::BUF, *Bb = Ab + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i)
            if (alds[i] >= 0) *reinterpret_cast<uint4 *>(Ab + alds[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int p = tid + 256 * i;
            if (B_EXACT || p < NTAPS * G::B_PIECES_TAP) {
                const int tap = p / G::B_PIECES_TAP, rem = p - tap * G::B_PIECES_TAP;
                *reinterpret_cast<uint4 *>(Bb + (tap * NCOLS + (rem >> 1)) * PITCH + (rem & 1) * 16) = rb[i];
            }
        }
    };

    f32x16 acc[4][NB];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[m][n][v] = 0.f;

    auto compute = [&](int buf) {
        const unsigned char *Ab = smem + buf * G::BUF, *Bb = Ab + A_BYTES;
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
            const int dy = NTAPS == 9 ? tap / 3 : 1, dx = NTAPS == 9 ? tap % 3 : 1;
            bf16x8 a[4], b[NB];
#pragma unroll
            for (int m = 0; m < 4; ++m)
                a[m] = *reinterpret_cast<const bf16x8 *>(Ab + ((4 * ph + m + dy) * HPW + l31 + dx) * PITCH + kh * 16);
#pragma unroll
            for (int n = 0; n < NB; ++n)
                b[n] = *reinterpret_cast<const bf16x8 *>(Bb + (tap * NCOLS + chalf * (NCOLS / 2) + n * 32 + l31) * PITCH + kh * 16);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    };

    // ---- K loop -----------------------------------------------------------------------------------------------------
    const int nch = P.nchunks;
    load_chunk(0);
    store_chunk(0);
    if (nch > 1) load_chunk(1);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        compute(c & 1);
        if (c + 1 < nch) store_chunk((c + 1) & 1);              // the registers hold chunk c + 1
        __syncthreads();
        if (c + 2 < nch) load_chunk(c + 2);
    }

    // ---- park the accumulators (+ bias) as an fp32 [pixel][column] tile in LDS ------------------------------------------
    float *ot = reinterpret_cast<float *>(smem);                // every wave is past its last fragment read (barrier above)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int col = chalf * (NCOLS / 2) + n * 32 + l31;
        const float bv = P.bias ? P.bias[nt * NCOLS + col] : 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int px = (4 * ph + m) * TW + (v & 3) + 8 * (v >> 2) + 4 * kh;
                ot[px * G::OPITCH + col] = acc[m][n][v] + bv;
            }
    }
    __syncthreads();

    // ---- cooperative finish: 8 columns of one pixel per step ------------------------------------------------------------
    if constexpr (EPI == RNH_EPI_LSTM) {
        // column = gate * 32 + j of the tile's 32 hidden channels nt * 32 + j (plans.lstm_colmap); NCOLS == 128
        const int hd = P.hd;
        for (int it = tid; it < TH * TW * 4; it += 256) {
            const int px = it >> 2, j0 = (it & 3) * 8, hc = nt * 32 + j0;
            const int y = y0 + px / TW, x = x0 + (px & (TW - 1));
            if (y >= H || x >= W || hc >= hd) continue;
            const float *o = ot + px * G::OPITCH + j0;
            const long pe = ((long)img * H + y) * W + x;
            float cp[8], cn[8], hn[8], gi[8], gf[8], go[8], gg[8];
            if (P.c_prev) {
                load8(P.c_prev, RNH_DT_F32, pe * hd + hc, cp);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) cp[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                gi[e] = b_sigmoid(o[e]);
                gf[e] = b_sigmoid(o[32 + e]);
                go[e] = b_sigmoid(o[64 + e]);
                gg[e] = b_tanh(o[96 + e]);
                cn[e] = gf[e] * cp[e] + gi[e] * gg[e];
                hn[e] = go[e] * b_tanh(cn[e]);
            }
            store8(P.c_out, RNH_DT_F32, pe * hd + hc, cn);
            store8(P.h_out, P.h_dtype, pe * hd + hc, hn);
            if (P.gates_out) {
                store8(P.gates_out, P.gates_dtype, pe * 4 * hd + hc, gi);
                store8(P.gates_out, P.gates_dtype, pe * 4 * hd + hd + hc, gf);
                store8(P.gates_out, P.gates_dtype, pe * 4 * hd + 2 * hd + hc, go);
                store8(P.gates_out, P.gates_dtype, pe * 4 * hd + 3 * hd + hc, gg);
            }
        }
    } else {
        constexpr int G8 = NCOLS / 8;
        for (int it = tid; it < TH * TW * G8; it += 256) {
            const int px = it / G8, c8 = it - px * G8;
            const int y = y0 + px / TW, x = x0 + (px & (TW - 1));
            if (y >= H || x >= W) continue;
            const int n0 = nt * NCOLS + c8 * 8;
            const float *o = ot + px * G::OPITCH + c8 * 8;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = o[e];
            if constexpr (EPI == RNH_EPI_PS) {
                const int r = P.ps_r, cq = P.ps_cq;
                if (n0 >= cq * r * r) continue;
                const int sub = n0 / cq, c = n0 - sub * cq, pi = sub / r, pj = sub - pi * r;
                const long e = ((((long)img * H + y) * r + pi) * ((long)W * r) + (long)x * r + pj) * cq + c;
                store8(P.dst[0].ptr, P.dst[0].dtype, e, f);
            } else {
                int seg = -1, cbase = 0;
                for (int d = 0; d < P.ndst; ++d) {
                    if (seg < 0 && n0 < cbase + P.dst[d].ncols) seg = d;
                    if (seg < 0) cbase += P.dst[d].ncols;
                }
                if (seg < 0) continue;
                const rnh_mdst_t &D = P.dst[seg];
                const long e = (((long)(img + D.img_off) * H + y) * W + x) * D.C + D.c0 + (n0 - cbase);
                if (D.accumulate) {
                    float old[8];
                    load8(D.ptr, D.dtype, e, old);
#pragma unroll
                    for (int q = 0; q < 8; ++q) f[q] += old[q];
                }
                store8(D.ptr, D.dtype, e, f);
            }
        }
    }
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
    const int TYn = (a.H + TH - 1) / TH, TXn = (a.W + TW - 1) / TW, NT = a.Npad / ncols;
    const long blocks = (long)a.B * TYn * TXn * NT;
    if (blocks >= (1L << 31)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: grid too large");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)blocks), block(256);
#define RNH_LAUNCH(EPI, NC, NTP) hipLaunchKernelGGL((conv_bf16_kernel<EPI, NC, NTP>), grid, block, 0, st, a, TYn, TXn, NT)
    switch (a.epilogue) {
        case RNH_EPI_STORE:
            if (a.ndst < 1 || a.ndst > RNH_MAX_DST) RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: bad destination count");
            for (int d = 0; d < a.ndst; ++d) {
                const rnh_mdst_t &D = a.dst[d];
                if (!D.ptr || D.ncols < 1 || (D.dtype != RNH_DT_F32 && D.dtype != RNH_DT_BF16)) RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: bad destination %d", d);
                if ((D.C & 7) || (D.c0 & 7) || (D.ncols & 7)) RNH_FAIL(RNH_E_ALIGN, "rnh_conv_bf16: destination channels must be multiples of 8");
            }
            if (a.ntaps == 9) {
                if (ncols == 128) RNH_LAUNCH(RNH_EPI_STORE, 128, 9);
                else RNH_LAUNCH(RNH_EPI_STORE, 64, 9);
            } else {
                if (ncols == 128) RNH_LAUNCH(RNH_EPI_STORE, 128, 1);
                else RNH_LAUNCH(RNH_EPI_STORE, 64, 1);
            }
            break;
        case RNH_EPI_PS:
            if (a.ntaps != 9) RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: the pixel-shuffle epilogue serves 3x3 convolutions");
            if (a.ndst != 1 || !a.dst[0].ptr || a.ps_r < 1 || a.ps_cq < 8 || (a.ps_cq & 7) || a.ps_cq * a.ps_r * a.ps_r > a.Npad)
                RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: bad pixel-shuffle destination");
            if (ncols == 128) RNH_LAUNCH(RNH_EPI_PS, 128, 9);
            else RNH_LAUNCH(RNH_EPI_PS, 64, 9);
            break;
        case RNH_EPI_LSTM:
            if (a.ntaps != 9) RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: the LSTM epilogue serves 3x3 convolutions");
            if (!a.h_out || !a.c_out || a.hd < 8 || (a.hd & 7) || !a.bias) RNH_FAIL(RNH_E_ARG, "rnh_conv_bf16: LSTM epilogue needs h_out, c_out, hd % 8 == 0, bias");
            if (a.Npad != 128 * ((a.hd + 31) / 32)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: LSTM column layout (plans.lstm_colmap)");
            RNH_LAUNCH(RNH_EPI_LSTM, 128, 9);
            break;
        default:
            RNH_FAIL(RNH_E_RANGE, "rnh_conv_bf16: epilogue %d not available", a.epilogue);
    }
#undef RNH_LAUNCH
    RNH_CHECK_LAUNCH("rnh_conv_bf16");
    return 0;
}
