// HBM-bound kernels of the RefineNet hot path (gfx950): the 1->C input convolution + PReLU and its backward,
// the C->out_channels last convolution of the upsampler (forward, data gradient, weight gradient), the
// ConvLSTM gate backward, the fused loss + gradient, element-wise sums, weight packing, phase-code plane.
// All of them move 16 bytes per lane along the channel (NHWC) axis; reductions are two-pass
// (per-block partials in a workspace, then one fixed-order sum) so results are bitwise reproducible.
#include <stdarg.h>
#include "rnh_common.h"

// ---------------------------------------------------------------------------------------------------------
// error string
// ---------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "no error";
void rnh_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *rnh_last_error(void) { return g_err; }
extern "C" int rnh_abi_version(void) { return RNH_ABI_VERSION; }
extern "C" void rnh_struct_sizes(int32_t out[4]) {
    out[0] = (int32_t)sizeof(rnh_src_t);
    out[1] = (int32_t)sizeof(rnh_dst_t);
    out[2] = (int32_t)sizeof(rnh_conv_args_t);
    out[3] = (int32_t)sizeof(rnh_wgrad_args_t);
}

namespace {

__device__ __forceinline__ float block_sum(float v, float *red) {
    // wave reduction (64 lanes) then across the waves of the block through LDS; result valid in thread 0
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

// ---------------------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float *w, const float *bias, float *wp, float *biasp, const int *kbase,
                                    const int *knv, const int *ktap, const int *kcoff, const int *colmap, int nk, int Npad,
                                    int Cout, int Cin, int ntaps, int kstride, int transposed) {
    const long total = (long)nk * Npad * 16;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int kp = (int)(e & 15);                                  // packed position = kh*8 + e8
        const int kk = 8 * ((kp & 7) >> 2) + 4 * (kp >> 3) + (kp & 3);   // channel within the 16-channel chunk
        const long r = e >> 4;
        const int n = (int)(r % Npad), ks = (int)(r / Npad);
        const int cm = colmap[n];
        float v = 0.f;
        if (cm >= 0 && kk < knv[ks]) {
            const int kidx = kbase[ks] + kk * kstride;
            const int cme = cm + (kcoff ? kcoff[ks] : 0);
            const int o = transposed ? kidx : cme, i = transposed ? cme : kidx;
            const int tp = transposed ? ntaps - 1 - ktap[ks] : ktap[ks];
            v = w[((long)o * Cin + i) * ntaps + tp];
        }
        wp[e] = v;
    }
    if (biasp) {
        for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < Npad; n += (long)gridDim.x * blockDim.x) {
            const int cm = colmap[n];
            biasp[n] = (bias && cm >= 0 && !transposed) ? bias[cm] : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// input block: y = PReLU(conv3x3(x) + b), x [B][H][W][Cin] -> y [B][H][W][Cout]
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) inconv_fwd_kernel(const float *x, const float *w, const float *bias, const float *slope,
                                                         float *y, int B, int H, int W, int Cin, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float sw[];      // [tap][ci][co]
    const int nw = 9 * Cin * Cout;
    for (int e = threadIdx.x; e < nw; e += blockDim.x) {
        const int co = e % Cout, r = e / Cout, ci = r % Cin, tap = r / Cin;
        sw[e] = w[(co * Cin + ci) * 9 + tap];
    }
    __syncthreads();
    const float a = slope[0];
    const int G = Cout >> 2;
    const long total = (long)B * H * W * G;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int g = (int)(e % G);
        const long p = e / G;
        const int xx = (int)(p % W);
        const long q = p / W;
        const int yy = (int)(q % H);
        float4 acc = rnh_ld4(bias + g * 4);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if ((unsigned)(yy + dy) >= (unsigned)H || (unsigned)(xx + dx) >= (unsigned)W) continue;
            const float *xp = x + (p + dy * W + dx) * Cin;
            for (int ci = 0; ci < Cin; ++ci) {
                const float xv = xp[ci];
                const float4 wv = rnh_ld4(sw + (tap * Cin + ci) * Cout + g * 4);
                acc.x += xv * wv.x; acc.y += xv * wv.y; acc.z += xv * wv.z; acc.w += xv * wv.w;
            }
        }
        acc.x = acc.x > 0.f ? acc.x : a * acc.x;
        acc.y = acc.y > 0.f ? acc.y : a * acc.y;
        acc.z = acc.z > 0.f ? acc.z : a * acc.z;
        acc.w = acc.w > 0.f ? acc.w : a * acc.w;
        rnh_st4(y + p * Cout + g * 4, acc);
    }
}

constexpr int INB_BLOCKS = 2048;    // 32 waves per CU: the pixel loop is a chain of dependent loads, only occupancy hides it (512 blocks: 1.17 ms at
                                    // BASELINE config 2, round 3)
constexpr int INB_MAXK = 36;     // 9 * Cin, Cin <= 4

// thread = (4 output channels g, pixel lane sub of 256 / (Cout / 4)); partial[block][co][9*Cin + 2] = {dW taps..., db, dslope}.
// (Round 3: one 16-byte load of dy per pixel and thread instead of a 4-byte one - the loop is bound by its load instructions, a wave
// now covers 4 x 64 (pixel, channel) pairs per iteration instead of 64: 1.17 -> see profiles/README.md.)
template <int INB_MAXK>      // 9 for in_channels == 1 (every reference YAML), 36 up to 4 input channels
__global__ void __launch_bounds__(256) inconv_bwd_kernel(const float *x, const float *w, const float *bias, const float *slope,
                                                         const float *dy, float *ws, int B, int H, int W, int Cin, int Cout) {
    __shared__ float red[1024];
    const int K = 9 * Cin;
    const int G = Cout >> 2, g = threadIdx.x % G, sub = threadIdx.x / G, nsub = blockDim.x / G;
    float wr[INB_MAXK][4], acc[INB_MAXK][4];
#pragma unroll
    for (int k = 0; k < INB_MAXK; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[k][c] = 0.f;
            wr[k][c] = 0.f;
            if (k < K) {
                const int tap = k / Cin, ci = k - tap * Cin;
                wr[k][c] = w[((g * 4 + c) * Cin + ci) * 9 + tap];
            }
        }
    float db[4] = {0.f, 0.f, 0.f, 0.f}, ds[4] = {0.f, 0.f, 0.f, 0.f};
    const float a = slope[0];
    const float4 bv = rnh_ld4(bias + g * 4);
    const float bvv[4] = {bv.x, bv.y, bv.z, bv.w};
    const long M = (long)B * H * W;
    const long per = (M + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per;
    long p1 = p0 + per;
    if (p1 > M) p1 = M;
    for (long p = p0 + sub; p < p1; p += nsub) {
        const int xx = (int)(p % W);
        const int yy = (int)((p / W) % H);
        float xv[INB_MAXK];
#pragma unroll
        for (int k = 0; k < INB_MAXK; ++k) {
            xv[k] = 0.f;
            if (k < K) {
                const int tap = k / Cin, ci = k - tap * Cin;
                const int dy_ = tap / 3 - 1, dx_ = tap % 3 - 1;
                if ((unsigned)(yy + dy_) < (unsigned)H && (unsigned)(xx + dx_) < (unsigned)W)
                    xv[k] = x[(p + dy_ * W + dx_) * Cin + ci];
            }
        }
        const float4 gv = rnh_ld4(dy + p * Cout + g * 4);
        const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float z = bvv[c];
#pragma unroll
            for (int k = 0; k < INB_MAXK; ++k)
                if (k < K) z += xv[k] * wr[k][c];
            const float dz = z > 0.f ? gg[c] : a * gg[c];
            ds[c] += z > 0.f ? 0.f : gg[c] * z;
            db[c] += dz;
#pragma unroll
            for (int k = 0; k < INB_MAXK; ++k)
                if (k < K) acc[k][c] += dz * xv[k];
        }
    }
    // reduce over sub through LDS, one quantity at a time, in fixed order
    for (int k = 0; k < K + 2; ++k) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < INB_MAXK; ++kk)
            if (kk == k) { v[0] = acc[kk][0]; v[1] = acc[kk][1]; v[2] = acc[kk][2]; v[3] = acc[kk][3]; }
        if (k == K) { v[0] = db[0]; v[1] = db[1]; v[2] = db[2]; v[3] = db[3]; }
        if (k == K + 1) { v[0] = ds[0]; v[1] = ds[1]; v[2] = ds[2]; v[3] = ds[3]; }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) red[(sub * G + g) * 4 + c] = v[c];
        __syncthreads();
        if (sub == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float s = 0.f;
                for (int q = 0; q < nsub; ++q) s += red[(q * G + g) * 4 + c];
                ws[((long)blockIdx.x * Cout + g * 4 + c) * (K + 2) + k] = s;
            }
        }
    }
}

// One workgroup per output value (Cout * (9 Cin + 1) weight / bias entries + the PReLU slope): 256 threads take the partial sums of the
// INB_BLOCKS blocks in a fixed interleaved order, then a fixed-order tree - bitwise repeatable.  (Until round 4 one THREAD walked the 2048
// partials of an entry, a chain of dependent loads: 0.6 ms on three workgroups at the very end of the backward.)
__global__ void __launch_bounds__(256) inconv_bwd_reduce_kernel(const float *ws, int nblocks, float *dw, float *db, float *dslope, int Cin, int Cout,
                                                                int accumulate) {
    __shared__ float red[4];
    const int K = 9 * Cin;
    const int total = Cout * (K + 1), e = blockIdx.x;
    float v = 0.f;
    if (e < total) {
        const int co = e / (K + 1), k = e - co * (K + 1);
        for (int b = threadIdx.x; b < nblocks; b += blockDim.x) v += ws[((long)b * Cout + co) * (K + 2) + k];
        const float s = block_sum(v, red);
        if (threadIdx.x == 0) {
            if (k < K) {
                const int tap = k / Cin, ci = k - tap * Cin;
                float *o = dw + (co * Cin + ci) * 9 + tap;
                *o = accumulate ? *o + s : s;
            } else {
                db[co] = accumulate ? db[co] + s : s;
            }
        }
    } else {
        for (int i = threadIdx.x; i < nblocks * Cout; i += blockDim.x) v += ws[(long)i * (K + 2) + K + 1];
        const float s = block_sum(v, red);
        if (threadIdx.x == 0) dslope[0] = accumulate ? dslope[0] + s : s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// last conv of the upsampler / side path of refine conv1's odd channel: Cin -> Cout (Cout <= 8), HBM-bound
// ---------------------------------------------------------------------------------------------------------
constexpr int OC_MAX = 8;
constexpr int OT = 16;                  // 16x16 output pixels per block
constexpr int OROW = 20;                // 16 channels + 4 pad floats per halo pixel (conflict-free ds_read_b128)

// y[p][co] = bias[co] + sum_{ci,t} x[p + t][ci] * w[co * wco + ci * wci + t]  (FLIP: tap 8 - t - a data gradient read as a convolution).
// thread = one pixel of a 16 x 16 tile, all COUT outputs.  The 18 x 18 halo of a 16-channel chunk sits in LDS, double-buffered: the
// next chunk travels global -> registers under the FMAs of this one, one barrier per chunk.  The weights never touch LDS: their
// addresses are wave-uniform, so they arrive through the scalar cache as SGPR operands of the FMAs (the round-1 kernel read every
// weight as a broadcast float4 from LDS - five of its six LDS reads per 20 FMAs - and was LDS-bound at 1.5 TB/s of input).
// Output pixel stride ldy >= COUT + yzero; channels COUT .. COUT + yzero - 1 are written as zeros.
template <int COUT, bool FLIP>
__global__ void __launch_bounds__(256) outconv_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                          float *__restrict__ y, int B, int H, int W, int Cin, int TX, int TY, int wco, int wci,
                                                          int ldy, int yzero) {
    constexpr int HP = (OT + 2) * (OT + 2), NLD = (HP * 4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float halo[2][HP * OROW];
    const int tile = blockIdx.x;
    const int b = tile / (TX * TY), tr = tile - b * TX * TY, ty = tr / TX, tx = tr - ty * TX;
    const int y0 = ty * OT, x0 = tx * OT;
    const int ly = threadIdx.x / OT, lx = threadIdx.x % OT;
    const float *xb = x + (long)b * H * W * Cin;
    // staging items e = tid + 256 k: (halo pixel e >> 2, channel quad e & 3); offset inside the image or -1 (outside: zero padding)
    int soff[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e = threadIdx.x + 256 * k, q = e & 3, hp = e >> 2, hy = hp / (OT + 2), hx = hp - hy * (OT + 2);
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        soff[k] = (hp < HP && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? (gy * W + gx) * Cin + q * 4 : -1;
    }
    float4 pre[NLD];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int q = (threadIdx.x + 256 * k) & 3;
            pre[k] = (soff[k] >= 0 && c0 + q * 4 < Cin) ? rnh_ld4(xb + soff[k] + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = threadIdx.x + 256 * k;
            if (e < HP * 4) rnh_st4(&halo[buf][(e >> 2) * OROW + (e & 3) * 4], pre[k]);
        }
    };
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = bias ? bias[co] : 0.f;
    fetch(0);
    park(0);
    __syncthreads();
    int cur = 0;
    for (int c0 = 0; c0 < Cin; c0 += 16, cur ^= 1) {
        const bool more = c0 + 16 < Cin;
        if (more) fetch(c0 + 16);
        const float *hb = halo[cur] + (ly * (OT + 2) + lx) * OROW;
        const int nch = min(16, Cin - c0);
        // (channel by channel: the 9 x COUT weights of one input channel are all the SGPRs the loop keeps alive)
#pragma unroll 2
        for (int cc = 0; cc < nch; ++cc) {
            float xs[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) xs[t] = hb[((t / 3) * (OT + 2) + t % 3) * OROW + cc];
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float *wp = w + (long)co * wco + (long)(c0 + cc) * wci;              // wave-uniform: scalar loads
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[co] = __builtin_fmaf(xs[t], wp[FLIP ? 8 - t : t], acc[co]);
            }
        }
        if (more) park(cur ^ 1);
        __syncthreads();
    }
    const int gy = y0 + ly, gx = x0 + lx;
    if (gy < H && gx < W) {
        float *o = y + (((long)b * H + gy) * W + gx) * ldy;
#pragma unroll
        for (int co = 0; co < COUT; ++co) o[co] = acc[co];
        for (int z = 0; z < yzero; ++z) o[COUT + z] = 0.f;
    }
}

// dx[p][ci] = sum_{t,co} dy[p - t][co] * w[co][ci][t]; thread = (pixel, 4 input channels)
__global__ void __launch_bounds__(256) outconv_dgrad_kernel(const float *dy, const float *w, float *dx, int B, int H, int W,
                                                            int Cin, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float sw[];   // [co][tap][ci]
    for (int e = threadIdx.x; e < Cout * 9 * Cin; e += blockDim.x) {
        const int ci = e % Cin, r = e / Cin, tap = r % 9, co = r / 9;
        sw[e] = w[(co * Cin + ci) * 9 + tap];
    }
    __syncthreads();
    const int G = Cin >> 2;
    const long total = (long)B * H * W * G;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int g = (int)(e % G);
        const long p = e / G;
        const int xx = (int)(p % W);
        const int yy = (int)((p / W) % H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy_ = tap / 3 - 1, dx_ = tap % 3 - 1;
            // contribution of output pixel p - t
            if ((unsigned)(yy - dy_) >= (unsigned)H || (unsigned)(xx - dx_) >= (unsigned)W) continue;
            const float *gp = dy + (p - dy_ * W - dx_) * Cout;
            for (int co = 0; co < Cout; ++co) {
                const float gv = gp[co];
                const float4 wv = rnh_ld4(sw + (co * 9 + tap) * Cin + g * 4);
                acc.x += gv * wv.x; acc.y += gv * wv.y; acc.z += gv * wv.z; acc.w += gv * wv.w;
            }
        }
        rnh_st4(dx + p * Cin + g * 4, acc);
    }
}

constexpr int OW_BLOCKS = 2048;
// thread = (ci, sub): acc[co][t] += x[q][ci] * dy[q - t][co]; partial[block][co][ci][9] and db partial[block][co]
__global__ void __launch_bounds__(256) outconv_wgrad_kernel(const float *x, const float *dy, float *ws, int B, int H, int W,
                                                            int Cin, int Cout) {
    __shared__ float red[256];
    const int ci = threadIdx.x % Cin, sub = threadIdx.x / Cin, nsub = blockDim.x / Cin;
    float acc[OC_MAX][9];
    float dbp[OC_MAX];
#pragma unroll
    for (int co = 0; co < OC_MAX; ++co) {
        dbp[co] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[co][t] = 0.f;
    }
    const long M = (long)B * H * W;
    const long per = (M + gridDim.x - 1) / gridDim.x;
    const long q0 = (long)blockIdx.x * per;
    long q1 = q0 + per;
    if (q1 > M) q1 = M;
    if (sub < nsub) {
        for (long q = q0 + sub; q < q1; q += nsub) {
            const int xx = (int)(q % W);
            const int yy = (int)((q / W) % H);
            const float xv = x[q * Cin + ci];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy_ = t / 3 - 1, dx_ = t % 3 - 1;
                if ((unsigned)(yy - dy_) >= (unsigned)H || (unsigned)(xx - dx_) >= (unsigned)W) continue;
                const float *gp = dy + (q - dy_ * W - dx_) * Cout;
#pragma unroll
                for (int co = 0; co < OC_MAX; ++co)
                    if (co < Cout) acc[co][t] += xv * gp[co];
            }
            if (ci == 0) {
#pragma unroll
                for (int co = 0; co < OC_MAX; ++co)
                    if (co < Cout) dbp[co] += dy[q * Cout + co];
            }
        }
    }
    float *out = ws + (long)blockIdx.x * (Cout * Cin * 9 + Cout);
    for (int co = 0; co < Cout; ++co) {
        for (int t = 0; t < 10; ++t) {
            float v = 0.f;
#pragma unroll
            for (int c2 = 0; c2 < OC_MAX; ++c2)
#pragma unroll
                for (int t2 = 0; t2 < 9; ++t2)
                    if (c2 == co && t2 == t) v = acc[c2][t2];
            if (t == 9) {
#pragma unroll
                for (int c2 = 0; c2 < OC_MAX; ++c2)
                    if (c2 == co) v = dbp[c2];
            }
            __syncthreads();
            red[threadIdx.x] = (sub < nsub) ? v : 0.f;
            __syncthreads();
            if (sub == 0) {
                float s = 0.f;
                for (int k = 0; k < nsub; ++k) s += red[k * Cin + ci];
                if (t < 9) out[(co * Cin + ci) * 9 + t] = s;
                else if (ci == 0) out[Cout * Cin * 9 + co] = s;
            }
        }
    }
}

// Streaming version used when Cin % 4 == 0 and 256 % (Cin/4) == 0: thread = (4 input channels, pixel lane), four
// pixels in flight per iteration (x is read exactly once, 16 B per lane), the 9 taps of dy come from L1.
template <int COUT>
__global__ void __launch_bounds__(256) outconv_wgrad_stream_kernel(const float *x, const float *dy, float *ws, int B, int H,
                                                                   int W, int Cin) {
    __shared__ float4 red[256];
    constexpr int UNR = 4;
    const int G = Cin >> 2, g = threadIdx.x % G, pl = threadIdx.x / G, PL = 256 / G;
    float4 acc[COUT][9];
    float dbp[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        dbp[co] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[co][t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const long M = (long)B * H * W;
    const long per = (M + gridDim.x - 1) / gridDim.x;
    const long q0 = (long)blockIdx.x * per;
    long q1 = q0 + per;
    if (q1 > M) q1 = M;
    for (long qb = q0 + pl; qb < q1; qb += (long)PL * UNR) {
        float4 xv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long q = qb + (long)u * PL;
            xv[u] = q < q1 ? rnh_ld4(x + q * Cin + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long q = qb + (long)u * PL;
            if (q >= q1) continue;
            const int xx = (int)(q % W);
            const int yy = (int)((q / W) % H);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy_ = t / 3 - 1, dx_ = t % 3 - 1;
                if ((unsigned)(yy - dy_) >= (unsigned)H || (unsigned)(xx - dx_) >= (unsigned)W) continue;
                const float *gp = dy + (q - dy_ * W - dx_) * COUT;
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const float gv = gp[co];
                    acc[co][t].x += xv[u].x * gv; acc[co][t].y += xv[u].y * gv;
                    acc[co][t].z += xv[u].z * gv; acc[co][t].w += xv[u].w * gv;
                }
            }
            if (g == 0) {
#pragma unroll
                for (int co = 0; co < COUT; ++co) dbp[co] += dy[q * COUT + co];
            }
        }
    }
    // pixel lanes of one channel group sit G lanes apart: butterfly inside the wave, then the 4 waves through LDS
    float *out = ws + (long)blockIdx.x * (COUT * Cin * 9 + COUT);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, Gw = G < 64 ? G : 64;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            float4 v = t < 9 ? acc[co][t < 9 ? t : 0] : make_float4(dbp[co], 0.f, 0.f, 0.f);
            for (int m = Gw; m < 64; m <<= 1) {
                v.x += __shfl_xor(v.x, m, 64); v.y += __shfl_xor(v.y, m, 64);
                v.z += __shfl_xor(v.z, m, 64); v.w += __shfl_xor(v.w, m, 64);
            }
            __syncthreads();
            if (lane < Gw) red[wave * 64 + lane] = v;
            __syncthreads();
            if (threadIdx.x < G) {                       // G <= 64: wave 0
                float4 s;
                s = (red[g] + red[64 + g]) + (red[128 + g] + red[192 + g]);
                if (t < 9) {
                    out[(co * Cin + g * 4 + 0) * 9 + t] = s.x;
                    out[(co * Cin + g * 4 + 1) * 9 + t] = s.y;
                    out[(co * Cin + g * 4 + 2) * 9 + t] = s.z;
                    out[(co * Cin + g * 4 + 3) * 9 + t] = s.w;
                } else if (g == 0) {
                    out[COUT * Cin * 9 + co] = s.x;
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256) outconv_wgrad_reduce_kernel(const float *ws, int nblocks, float *dw, float *db, int Cin, int Cout,
                                                                   int accumulate) {
    __shared__ float part[256];
    const int nw = Cout * Cin * 9, total = nw + Cout;
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;          // 64 outputs x 4 slices of the block list
    const int per = (nblocks + 3) / 4, b0 = sl * per, b1 = min(nblocks, b0 + per);
    float s = 0.f;
    if (e < total)
        for (int b = b0; b < b1; ++b) s += ws[(long)b * total + e];
    part[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && e < total) {
        s = (part[threadIdx.x] + part[64 + threadIdx.x]) + (part[128 + threadIdx.x] + part[192 + threadIdx.x]);
        float *o = e < nw ? dw + e : db + (e - nw);
        *o = accumulate ? *o + s : s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// ConvLSTM gate backward
// ---------------------------------------------------------------------------------------------------------
__global__ void lstm_gates_bwd_kernel(const float *dh, const float *dh2, const float *dcn, const float *gates, const float *cprev,
                                      const float *cnext, float *dgates, float *dcprev, long npix, int hd) {
    const int G = hd >> 2;
    const long total = npix * G;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int g = (int)(e % G);
        const long p = e / G;
        const long o = p * hd + g * 4, og = p * 4 * hd + g * 4;
        float4 vdh = rnh_ld4(dh + o);
        if (dh2) {                                              // dh' arrives in two parts (layer above + next frame)
            const float4 v2 = rnh_ld4(dh2 + o);
            vdh.x += v2.x; vdh.y += v2.y; vdh.z += v2.z; vdh.w += v2.w;
        }
        const float4 vdc = dcn ? rnh_ld4(dcn + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 vcp = cprev ? rnh_ld4(cprev + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 vcn = rnh_ld4(cnext + o);
        const float4 gi = rnh_ld4(gates + og), gf = rnh_ld4(gates + og + hd), go = rnh_ld4(gates + og + 2 * hd),
                     gg = rnh_ld4(gates + og + 3 * hd);
        float4 di, df, dgo, dg, dcp;
#define RNH_GATE(c)                                               \
    {                                                             \
        const float th = tanhf(vcn.c);                            \
        const float d_o = vdh.c * th;                             \
        const float dct = vdc.c + vdh.c * go.c * (1.f - th * th); \
        di.c = dct * gg.c * gi.c * (1.f - gi.c);                  \
        df.c = dct * vcp.c * gf.c * (1.f - gf.c);                 \
        dgo.c = d_o * go.c * (1.f - go.c);                         \
        dg.c = dct * gi.c * (1.f - gg.c * gg.c);                  \
        dcp.c = dct * gf.c;                                       \
    }
        RNH_GATE(x) RNH_GATE(y) RNH_GATE(z) RNH_GATE(w)
#undef RNH_GATE
        rnh_st4(dgates + og, di);
        rnh_st4(dgates + og + hd, df);
        rnh_st4(dgates + og + 2 * hd, dgo);
        rnh_st4(dgates + og + 3 * hd, dg);
        if (dcprev) rnh_st4(dcprev + o, dcp);
    }
}

// ---------------------------------------------------------------------------------------------------------
// loss + gradient
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) loss_kernel(const float *o, const float *y, float *d_o, const float *gscale, float *ws,
                                                   int T, long per, int kind, float eps) {
    __shared__ float red[4];
    const int gi = blockIdx.y, g = gi / T, i = gi - g * T;
    const float *op = o + (long)gi * per, *yp = y + (long)i * per;
    float *dp = d_o ? d_o + (long)gi * per : nullptr;
    const float sc = (d_o && gscale) ? gscale[gi] / (float)per : 0.f;
    float part = 0.f;
    const long n4 = per >> 2;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        const float4 a = rnh_ld4(op + e * 4), b = rnh_ld4(yp + e * 4);
        float d[4] = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}, gr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (kind == RNH_LOSS_L1) {
                part += fabsf(d[k]);
                gr[k] = d[k] > 0.f ? sc : (d[k] < 0.f ? -sc : 0.f);
            } else {
                const float r = sqrtf(d[k] * d[k] + eps);
                part += r;
                gr[k] = sc * d[k] / r;
            }
        }
        if (dp) rnh_st4(dp + e * 4, make_float4(gr[0], gr[1], gr[2], gr[3]));
    }
    // tail (per not a multiple of 4)
    for (long e = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (long)gridDim.x * blockDim.x) {
        const float d = op[e] - yp[e];
        float gr;
        if (kind == RNH_LOSS_L1) {
            part += fabsf(d);
            gr = d > 0.f ? sc : (d < 0.f ? -sc : 0.f);
        } else {
            const float r = sqrtf(d * d + eps);
            part += r;
            gr = sc * d / r;
        }
        if (dp) dp[e] = gr;
    }
    const float s = block_sum(part, red);
    if (threadIdx.x == 0) ws[(long)gi * RNH_LOSS_BLOCKS + blockIdx.x] = s;
}

__global__ void loss_finalize_kernel(const float *ws, float *loss, int GT, long per) {
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= GT) return;
    float s = 0.f;
    for (int b = 0; b < RNH_LOSS_BLOCKS; ++b) s += ws[(long)gi * RNH_LOSS_BLOCKS + b];
    loss[gi] = s / (float)per;
}

// total = sum_g w[g] * mean_i loss[g*T + i]  (trainer :83-94: the discounted deep-supervision sum), or its gradient
// dloss[g*T + i] = dtotal * w[g] / T - fixed summation order, one wave
__global__ void loss_total_kernel(const float *in, const float *w, float *out, int G, int T, int backward) {
    if (backward) {
        const float g0 = in[0];
        for (int e = threadIdx.x; e < G * T; e += blockDim.x) out[e] = g0 * w[e / T] / (float)T;
        return;
    }
    if (threadIdx.x) return;
    float tot = 0.f;
    for (int g = 0; g < G; ++g) {
        float s = 0.f;
        for (int i = 0; i < T; ++i) s += in[g * T + i];
        tot += w[g] * (s / (float)T);
    }
    out[0] = tot;
}

// ---------------------------------------------------------------------------------------------------------
// element-wise
// ---------------------------------------------------------------------------------------------------------
__global__ void ew_add_kernel(float *out, const float *a, const float *b, const float *c, long n4, int accumulate) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        float4 v = rnh_ld4(a + e * 4);
        if (b) v = v + rnh_ld4(b + e * 4);
        if (c) v = v + rnh_ld4(c + e * 4);
        if (accumulate) v = v + rnh_ld4(out + e * 4);
        rnh_st4(out + e * 4, v);
    }
}

__global__ void phase_plane_kernel(const float *pos, float *out, int N, int F, long HW) {
    const long total = (long)N * F * HW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long img = e / HW;
        const int f = (int)(img / N), n = (int)(img - (long)f * N);
        rnh_st4(out + e * 4, make_float4(pos[n * F + f], 0.f, 0.f, 0.f));
    }
}

// ---------------------------------------------------------------------------------------------------------
// Side path of ONE output channel `co` of a convolution over J frame slots (refine conv1, channel 2*Cl: 129 = 4*32 + 1
// columns would cost a fifth 32-column MFMA tile).  With z_s[f][p][j] = conv3x3(source s of frame f; w[co][slot j]):
//   out[window i][p][co] = bias[co] + sum_j sum_s z_s[i + j][p][j]              (xcol_combine)
// and in the backward E[f][p][j] = dy[window f - j][p][co] (0 outside) turns the weight gradient of that channel into
// the J-output small-convolution gradient of each source (rnh_outconv_wgrad), every source frame read once.
// ---------------------------------------------------------------------------------------------------------
__global__ void xcol_pack_kernel(const float *w, float *out, int Cin, int co, int J, int cstride, int c0, int nch, int nvalid) {
    const int total = J * nch * 9;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int t = e % 9, c = (e / 9) % nch, j = e / (9 * nch);
        out[e] = c < nvalid ? w[((long)co * Cin + j * cstride + c0 + c) * 9 + t] : 0.f;
    }
}

__global__ void xcol_unpack_kernel(const float *dwx, const float *dbx, float *dw, float *db, int Cin, int co, int J, int cstride, int c0,
                                   int nch, int nvalid, int accumulate) {
    const int total = J * nvalid * 9;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total + 1; e += gridDim.x * blockDim.x) {
        if (e < total) {
            const int t = e % 9, c = (e / 9) % nvalid, j = e / (9 * nvalid);
            float *o = dw + ((long)co * Cin + j * cstride + c0 + c) * 9 + t;
            const float v = dwx[((long)j * nch + c) * 9 + t];
            *o = accumulate ? *o + v : v;
        } else if (dbx) {
            db[co] = accumulate ? db[co] + dbx[0] : dbx[0];
        }
    }
}

__global__ void xcol_combine_kernel(const float *z0, const float *z1, const float *z2, const float *bias, float *out, long npix, int N,
                                    int nwin, int J, int C, int c0) {
    const long total = (long)nwin * N * npix;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long img = e / npix, px = e - img * npix;
        float s = bias[0];
        for (int j = 0; j < J; ++j) {
            const long o = ((img + (long)j * N) * npix + px) * J + j;
            s += z0[o];
            if (z1) s += z1[o];
            if (z2) s += z2[o];
        }
        rnh_st4(out + e * C + c0, make_float4(s, 0.f, 0.f, 0.f));
    }
}

__global__ void xcol_gather_kernel(const float *dy, float *E, long npix, int N, int nwin, int J, int C, int c) {
    const long total = (long)(nwin + J - 1) * N * npix * J;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int j = (int)(e % J);
        const long q = e / J, img = q / npix, px = q - img * npix;
        const int f = (int)(img / N), n = (int)(img - (long)f * N), k = f - j;
        E[e] = (k >= 0 && k < nwin) ? dy[(((long)k * N + n) * npix + px) * C + c] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Phase planes of refine conv1 as a bias field.  The plane of frame slot j is the constant p[f + j][n] inside the
// image (zero padding outside), so its 3x3 convolution is p times the sum of the taps that stay inside the image -
// one of 16 border classes per pixel (bit 0: first row, 1: last row, 2: first column, 3: last column).
//   T[cls][j][co] = sum_{taps inside} w[co][j*cstride + c0][tap];   out[img][p][co] += sum_j p[(i + j)*N + n] * T[cls(p)][j][co]
// ---------------------------------------------------------------------------------------------------------
__global__ void phase_bias_table_kernel(const float *w, float *T, int Cin, int J, int cstride, int c0, int ncols) {
    const int total = 16 * J * ncols;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int co = e % ncols, j = (e / ncols) % J, cls = e / (ncols * J);
        const float *g = w + ((long)co * Cin + j * cstride + c0) * 9;
        float s = 0.f;
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const bool out = (dy < 0 && (cls & 1)) || (dy > 0 && (cls & 2)) || (dx < 0 && (cls & 4)) || (dx > 0 && (cls & 8));
            if (!out) s += g[t];
        }
        T[e] = s;
    }
}

__global__ void phase_bias_add_kernel(float *out, const float *planes, const float *T, int H, int W, int N, int nwin, int J, int C,
                                      int ncols) {
    const long npix = (long)H * W;
    const int G = ncols >> 2;
    const long total = (long)nwin * N * npix * G;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int g = (int)(e % G);
        const long q = e / G, img = q / npix, px = q - img * npix;
        const int y = (int)(px / W), x = (int)(px - (long)y * W);
        const int cls = (y == 0) | ((y == H - 1) << 1) | ((x == 0) << 2) | ((x == W - 1) << 3);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < J; ++j) {
            const float p = planes[(img + (long)j * N) * npix * 4];                 // channel 0 of pixel 0 of that frame's plane
            const float4 t = rnh_ld4(T + ((long)cls * J + j) * ncols + g * 4);
            acc.x += p * t.x; acc.y += p * t.y; acc.z += p * t.z; acc.w += p * t.w;
        }
        float *o = out + q * C + g * 4;
        const float4 v = rnh_ld4(o);
        rnh_st4(o, make_float4(v.x + acc.x, v.y + acc.y, v.z + acc.z, v.w + acc.w));
    }
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradient of the phase-plane input channels (round 3; the transpose of rnh_phase_bias_add): the plane of frame slot j is
// the constant p inside the image, so  dW[co][j*cstride + c0][t] = sum_{img = (i, n)} p[(i + j)*N + n] * S_t[img][co]  with
// S_t = the sum of dy[img][.][co] over the pixels at which tap t stays inside the image = total - border row - border column +
// corner: nine sums per (image, channel) - Q = {total, first row, last row, first column, last column, four corners} - instead of a
// pixel-contraction GEMM over five 4-channel sources (1.2 ms per stage at BASELINE config 2, 75 % padding).
// ---------------------------------------------------------------------------------------------------------
constexpr int PWG_ROWS = 8;           // image rows per block of the first stage

// block = (image, strip of PWG_ROWS rows); thread = (4 channels g, pixel lane s of 256 / G); partial Q per block
__global__ void __launch_bounds__(256) phase_wgrad_sum_kernel(const float *dy, float *part, int H, int W, int C, int ncols, int nstrips) {
    extern __shared__ __attribute__((aligned(16))) float red[];             // [lanes][9][ncols]
    const int img = blockIdx.x / nstrips, strip = blockIdx.x - img * nstrips;
    const int G = ncols >> 2, lanes = 256 / G, g = threadIdx.x % G, sl = threadIdx.x / G;
    float4 q[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int y0 = strip * PWG_ROWS, y1 = min(H, y0 + PWG_ROWS);
    if (sl < lanes)
        for (int p = y0 * W + sl; p < y1 * W; p += lanes) {
            const int y = p / W, x = p - y * W;
            const float4 v = rnh_ld4(dy + ((long)img * H * W + p) * C + g * 4);
            const bool r0 = y == 0, r1 = y == H - 1, c0 = x == 0, c1 = x == W - 1;
            const bool sel[9] = {true, r0, r1, c0, c1, r0 && c0, r0 && c1, r1 && c0, r1 && c1};
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (sel[k]) { q[k].x += v.x; q[k].y += v.y; q[k].z += v.z; q[k].w += v.w; }
        }
    if (sl < lanes)
#pragma unroll
        for (int k = 0; k < 9; ++k) rnh_st4(red + ((long)sl * 9 + k) * ncols + g * 4, q[k]);
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * ncols; e += 256) {                     // fixed order over the pixel lanes
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += red[(long)l * 9 * ncols + e];
        part[((long)img * nstrips + strip) * 9 * ncols + e] = s;
    }
}

// Q[img][k][co] = sum over strips (fixed order)
__global__ void phase_wgrad_q_kernel(const float *part, float *Q, int nimg, int nstrips, int ncols) {
    const long total = (long)nimg * 9 * ncols;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long img = e / (9 * ncols), r = e - img * 9 * ncols;
        float s = 0.f;
        for (int b = 0; b < nstrips; ++b) s += part[(img * nstrips + b) * 9 * ncols + r];
        Q[e] = s;
    }
}

// dw[co][j*cstride + c0][t] (+)= sum_img p[(i + j)*N + n] * S_t[img][co]
__global__ void phase_wgrad_finish_kernel(const float *Q, const float *planes, float *dw, long npix, int N, int nwin, int J, int Cin, int cstride,
                                          int c0, int ncols, int accumulate) {
    const int total = ncols * J * 9;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int t = e % 9, j = (e / 9) % J, co = e / (9 * J);
        const int dy = t / 3 - 1, dx = t % 3 - 1;
        const int kr = dy < 0 ? 1 : (dy > 0 ? 2 : -1), kc = dx < 0 ? 3 : (dx > 0 ? 4 : -1);          // excluded border row / column
        const int kk = (kr < 0 || kc < 0) ? -1 : 5 + (kr - 1) * 2 + (kc - 3);                        // their common corner
        float s = 0.f;
        for (int img = 0; img < nwin * N; ++img) {
            const float *q = Q + (long)img * 9 * ncols + co;
            float v = q[0];
            if (kr >= 0) v -= q[(long)kr * ncols];
            if (kc >= 0) v -= q[(long)kc * ncols];
            if (kk >= 0) v += q[(long)kk * ncols];
            s += planes[(img + (long)j * N) * npix * 4] * v;
        }
        float *o = dw + ((long)co * Cin + j * cstride + c0) * 9 + t;
        *o = accumulate ? *o + s : s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Data gradient of ONE output channel c of a J-slot convolution (refine conv1's channel 2*Cl, round 3): every frame collects, from the
// J windows that used it, a 45-tap stencil of that single gradient plane per input channel:
//   dX[(f, n)][p][ci] += sum_j sum_t g[(f + 2 hw - j, n)][p - off(t)][c] * w[c][j*cstride + ci][t],   ci < 2 Cl
// (g = the gradient planes with their window halo of zero frames; dX = two tensors of Cl channels each).  Replaces an implicit GEMM whose
// K is 5 x 4 channels (one real) padded to 16-channel steps: 1.9 ms per stage at 5 TFLOP/s.  Structure of uptail_dgrad_tile_kernel:
// thread = pixel of a 16 x 16 tile, the J gradient patches in LDS, weights wave-uniform (scalar loads), 64 channels per thread.
// ---------------------------------------------------------------------------------------------------------
__global__ void xdgrad_pack_kernel(const float *w, float *wx, int Cin, int c, int J, int cstride, int ncols) {
    const int total = J * 9 * ncols;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int ci = e % ncols, t = (e / ncols) % 9, j = e / (9 * ncols);
        wx[e] = w[((long)c * Cin + j * cstride + ci) * 9 + t];
    }
}

constexpr int XDT = 16, XDP = XDT + 2;
template <int CCH>
__global__ void __launch_bounds__(256) xdgrad_tile_kernel(const float *__restrict__ g, const float *__restrict__ wx, float *__restrict__ d0,
                                                          float *__restrict__ d1, int H, int W, int N, int J, int C, int c, int Cl, int ncols,
                                                          int TX, int TY, int hw2) {
    extern __shared__ __attribute__((aligned(16))) float patch[];          // [J][XDP][XDP]
    const int tb = blockIdx.x;
    const int b = tb / (TX * TY), trem = tb - b * TX * TY, tyb = trem / TX, txb = trem - tyb * TX;
    const int y0 = tyb * XDT, x0 = txb * XDT, ch0 = blockIdx.y * CCH;
    const int f = b / N, n = b - f * N;
    for (int e = threadIdx.x; e < J * XDP * XDP; e += 256) {
        const int j = e / (XDP * XDP), r = e - j * XDP * XDP, py = y0 - 1 + r / XDP, px = x0 - 1 + r % XDP;
        const long img = (long)(f + hw2 - j) * N + n;
        patch[e] = ((unsigned)py < (unsigned)H && (unsigned)px < (unsigned)W) ? g[((img * H + py) * W + px) * C + c] : 0.f;
    }
    __syncthreads();
    const int ly = threadIdx.x / XDT, lx = threadIdx.x % XDT, qy = y0 + ly, qx = x0 + lx;
    float acc[CCH];
#pragma unroll
    for (int k = 0; k < CCH; ++k) acc[k] = 0.f;
    for (int j = 0; j < J; ++j)
        for (int t = 0; t < 9; ++t) {
            const float d = patch[(j * XDP + ly + 1 - (t / 3 - 1)) * XDP + lx + 1 - (t % 3 - 1)];      // dY[p - off(t)]
            const float *k = wx + ((long)j * 9 + t) * ncols + ch0;
#pragma unroll
            for (int q = 0; q < CCH; ++q) acc[q] = fmaf(d, k[q], acc[q]);
        }
    if (qy >= H || qx >= W) return;
    float *o = (ch0 < Cl ? d0 + ch0 : d1 + (ch0 - Cl)) + (((long)b * H + qy) * W + qx) * Cl;
#pragma unroll
    for (int q = 0; q < CCH; q += 4) {
        const float4 v = rnh_ld4(o + q);
        rnh_st4(o + q, make_float4(v.x + acc[q], v.y + acc[q + 1], v.z + acc[q + 2], v.w + acc[q + 3]));
    }
}

inline int grid_for(long work_items, int block = 256, int cap = 8192) {
    long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------
extern "C" int rnh_pack_weights(const float *w, const float *bias, float *wp, float *biasp, const int32_t *kbase,
                                const int32_t *knv, const int32_t *ktap, const int32_t *kcoff, const int32_t *colmap, int nk,
                                int Npad, int Cout, int Cin, int ntaps, int kstride, int transposed, void *stream) {
    if (!w || !wp || !kbase || !knv || !ktap || !colmap || nk < 1 || Npad < 1 || Cout < 1 || Cin < 1 || kstride < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_pack_weights: bad arguments");
    if (ntaps != 9 && ntaps != 1) RNH_FAIL(RNH_E_RANGE, "rnh_pack_weights: ntaps %d", ntaps);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid_for((long)nk * Npad * 16)), dim3(256), 0, (hipStream_t)stream, w, bias, wp,
                       biasp, kbase, knv, ktap, kcoff, colmap, nk, Npad, Cout, Cin, ntaps, kstride, transposed);
    RNH_CHECK_LAUNCH("rnh_pack_weights");
    return 0;
}

extern "C" int rnh_inconv_prelu_fwd(const float *x, const float *w, const float *bias, const float *slope, float *y, int B, int H,
                                    int W, int Cin, int Cout, void *stream) {
    if (!x || !w || !bias || !slope || !y || B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_inconv_prelu_fwd: bad arguments");
    if (Cout & 3) RNH_FAIL(RNH_E_ALIGN, "rnh_inconv_prelu_fwd: Cout must be a multiple of 4");
    const size_t shm = (size_t)9 * Cin * Cout * sizeof(float);
    if (shm > 60000) RNH_FAIL(RNH_E_RANGE, "rnh_inconv_prelu_fwd: 9*Cin*Cout too large for LDS");
    hipLaunchKernelGGL(inconv_fwd_kernel, dim3(grid_for((long)B * H * W * (Cout / 4))), dim3(256), shm, (hipStream_t)stream, x, w,
                       bias, slope, y, B, H, W, Cin, Cout);
    RNH_CHECK_LAUNCH("rnh_inconv_prelu_fwd");
    return 0;
}

extern "C" int64_t rnh_inconv_bwd_ws_floats(int Cin, int Cout) { return (int64_t)INB_BLOCKS * Cout * (9 * Cin + 2); }

extern "C" int rnh_inconv_prelu_bwd(const float *x, const float *w, const float *bias, const float *slope, const float *dy,
                                    float *dw, float *db, float *dslope, float *ws, int B, int H, int W, int Cin, int Cout,
                                    int accumulate, void *stream) {
    if (!x || !w || !bias || !slope || !dy || !dw || !db || !dslope || !ws || B < 1 || H < 1 || W < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_inconv_prelu_bwd: bad arguments");
    if (9 * Cin > INB_MAXK) RNH_FAIL(RNH_E_RANGE, "rnh_inconv_prelu_bwd: Cin > 4 not supported");
    if (Cout < 4 || Cout > 256 || (Cout & 3) || 256 % (Cout / 4)) RNH_FAIL(RNH_E_RANGE, "rnh_inconv_prelu_bwd: Cout / 4 must divide 256");
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 1) hipLaunchKernelGGL(inconv_bwd_kernel<9>, dim3(INB_BLOCKS), dim3(256), 0, st, x, w, bias, slope, dy, ws, B, H, W, Cin, Cout);
    else hipLaunchKernelGGL(inconv_bwd_kernel<INB_MAXK>, dim3(INB_BLOCKS), dim3(256), 0, st, x, w, bias, slope, dy, ws, B, H, W, Cin, Cout);
    RNH_CHECK_LAUNCH("rnh_inconv_prelu_bwd");
    hipLaunchKernelGGL(inconv_bwd_reduce_kernel, dim3((unsigned)(Cout * (9 * Cin + 1) + 1)), dim3(256), 0, st, ws, INB_BLOCKS, dw, db,
                       dslope, Cin, Cout, accumulate);
    RNH_CHECK_LAUNCH("rnh_inconv_prelu_bwd(reduce)");
    return 0;
}

template <bool FLIP>
static int launch_outconv_fwd(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int Cin, int Cout, int wco,
                              int wci, int ldy, int yzero, hipStream_t st) {
    const int TX = (W + OT - 1) / OT, TY = (H + OT - 1) / OT;
    const dim3 grid((unsigned)(B * TX * TY)), block(256);
#define RNH_OC(n)                                                                                                                    \
    case n:                                                                                                                          \
        hipLaunchKernelGGL((outconv_fwd_kernel<n, FLIP>), grid, block, 0, st, x, w, bias, y, B, H, W, Cin, TX, TY, wco, wci, ldy, yzero); \
        break;
    switch (Cout) {
        RNH_OC(1) RNH_OC(2) RNH_OC(3) RNH_OC(4) RNH_OC(5) RNH_OC(6) RNH_OC(7) RNH_OC(8)
    }
#undef RNH_OC
    return 0;
}

static int outconv_fwd_checked(const char *who, const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int Cin,
                               int Cout, int wco, int wci, int flip, int ldy, int yzero, void *stream) {
    if (!x || !w || !y || B < 1 || H < 1 || W < 1 || Cin < 1) RNH_FAIL(RNH_E_ARG, "%s: bad arguments", who);
    if (Cin & 3) RNH_FAIL(RNH_E_ALIGN, "%s: Cin must be a multiple of 4", who);
    if (Cout < 1 || Cout > OC_MAX) RNH_FAIL(RNH_E_RANGE, "%s: Cout must be 1..%d", who, OC_MAX);
    if (yzero < 0 || ldy < Cout + yzero) RNH_FAIL(RNH_E_ARG, "%s: output pixel stride %d < %d channels", who, ldy, Cout + yzero);
    if ((long)H * W * Cin >= (1L << 31)) RNH_FAIL(RNH_E_RANGE, "%s: image too large for 32-bit offsets", who);
    if (flip) launch_outconv_fwd<true>(x, w, bias, y, B, H, W, Cin, Cout, wco, wci, ldy, yzero, (hipStream_t)stream);
    else launch_outconv_fwd<false>(x, w, bias, y, B, H, W, Cin, Cout, wco, wci, ldy, yzero, (hipStream_t)stream);
    RNH_CHECK_LAUNCH(who);
    return 0;
}

extern "C" int rnh_outconv_fwd(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int Cin, int Cout,
                               void *stream) {
    if (!bias) RNH_FAIL(RNH_E_ARG, "rnh_outconv_fwd: bad arguments");
    return outconv_fwd_checked("rnh_outconv_fwd", x, w, bias, y, B, H, W, Cin, Cout, Cin * 9, 9, 0, Cout, 0, stream);
}

extern "C" int rnh_outconv_fwd_ld(const float *x, const float *w, int64_t wco, int64_t wci, int flip, const float *bias, float *y, int ldy,
                                  int yzero, int B, int H, int W, int Cin, int Cout, void *stream) {
    if (wco < 0 || wci < 9 || wco * Cout >= (1L << 31) || wci * Cin >= (1L << 31)) RNH_FAIL(RNH_E_ARG, "rnh_outconv_fwd_ld: weight strides");
    return outconv_fwd_checked("rnh_outconv_fwd_ld", x, w, bias, y, B, H, W, Cin, Cout, (int)wco, (int)wci, flip, ldy, yzero, stream);
}

extern "C" int rnh_outconv_dgrad(const float *dy, const float *w, float *dx, int B, int H, int W, int Cin, int Cout, void *stream) {
    if (!dy || !w || !dx || B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) RNH_FAIL(RNH_E_ARG, "rnh_outconv_dgrad: bad arguments");
    if (Cin & 3) RNH_FAIL(RNH_E_ALIGN, "rnh_outconv_dgrad: Cin must be a multiple of 4");
    const size_t shm = (size_t)Cout * 9 * Cin * sizeof(float);
    if (shm > 60000) RNH_FAIL(RNH_E_RANGE, "rnh_outconv_dgrad: Cout*Cin too large");
    hipLaunchKernelGGL(outconv_dgrad_kernel, dim3(grid_for((long)B * H * W * (Cin / 4), 256, 16384)), dim3(256), shm,
                       (hipStream_t)stream, dy, w, dx, B, H, W, Cin, Cout);
    RNH_CHECK_LAUNCH("rnh_outconv_dgrad");
    return 0;
}

extern "C" int64_t rnh_outconv_wgrad_ws_floats(int Cin, int Cout) { return (int64_t)OW_BLOCKS * (Cout * Cin * 9 + Cout); }

extern "C" int rnh_outconv_wgrad(const float *x, const float *dy, float *dw, float *db, float *ws, int B, int H, int W, int Cin,
                                 int Cout, int accumulate, void *stream) {
    if (!x || !dy || !dw || !db || !ws || B < 1 || H < 1 || W < 1) RNH_FAIL(RNH_E_ARG, "rnh_outconv_wgrad: bad arguments");
    if (Cin < 1 || Cin > 256) RNH_FAIL(RNH_E_RANGE, "rnh_outconv_wgrad: Cin must be 1..256");
    if (Cout < 1 || Cout > OC_MAX) RNH_FAIL(RNH_E_RANGE, "rnh_outconv_wgrad: Cout must be 1..%d", OC_MAX);
    hipStream_t st = (hipStream_t)stream;
    const bool stream_ok = (Cin % 4 == 0) && (256 % (Cin / 4) == 0);
    // one round of blocks for small inputs, at most OW_BLOCKS (the workspace bound) for the high-resolution tensors
    const long M = (long)B * H * W;
    const int nb = (int)(M / 4096 < 512 ? 512 : (M / 4096 > OW_BLOCKS ? OW_BLOCKS : M / 4096));
    if (stream_ok && Cout == 1) hipLaunchKernelGGL(outconv_wgrad_stream_kernel<1>, dim3(nb), dim3(256), 0, st, x, dy, ws, B, H, W, Cin);
    else if (stream_ok && Cout == 2) hipLaunchKernelGGL(outconv_wgrad_stream_kernel<2>, dim3(nb), dim3(256), 0, st, x, dy, ws, B, H, W, Cin);
    else if (stream_ok && Cout == 5) hipLaunchKernelGGL(outconv_wgrad_stream_kernel<5>, dim3(nb), dim3(256), 0, st, x, dy, ws, B, H, W, Cin);
    else hipLaunchKernelGGL(outconv_wgrad_kernel, dim3(nb), dim3(256), 0, st, x, dy, ws, B, H, W, Cin, Cout);
    RNH_CHECK_LAUNCH("rnh_outconv_wgrad");
    hipLaunchKernelGGL(outconv_wgrad_reduce_kernel, dim3((Cout * Cin * 9 + Cout + 63) / 64), dim3(256), 0, st, ws, nb, dw, db, Cin, Cout,
                       accumulate);
    RNH_CHECK_LAUNCH("rnh_outconv_wgrad(reduce)");
    return 0;
}

extern "C" int rnh_lstm_gates_bwd(const float *dh, const float *dh2, const float *dc_next, const float *gates, const float *c_prev,
                                  const float *c_next, float *dgates, float *dc_prev, int64_t npix, int hd, void *stream) {
    if (!dh || !gates || !c_next || !dgates || npix < 1 || hd < 1) RNH_FAIL(RNH_E_ARG, "rnh_lstm_gates_bwd: bad arguments");
    if (hd & 3) RNH_FAIL(RNH_E_ALIGN, "rnh_lstm_gates_bwd: hd must be a multiple of 4");
    hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(grid_for(npix * (hd / 4))), dim3(256), 0, (hipStream_t)stream, dh, dh2, dc_next,
                       gates, c_prev, c_next, dgates, dc_prev, (long)npix, hd);
    RNH_CHECK_LAUNCH("rnh_lstm_gates_bwd");
    return 0;
}

extern "C" int rnh_loss_fwd_bwd(const float *o, const float *y, float *loss, float *d_o, const float *gscale, float *ws, int G,
                                int T, int64_t per, int kind, float eps, void *stream) {
    if (!o || !y || !loss || !ws || G < 1 || T < 1 || per < 1) RNH_FAIL(RNH_E_ARG, "rnh_loss_fwd_bwd: bad arguments");
    if (d_o && !gscale) RNH_FAIL(RNH_E_ARG, "rnh_loss_fwd_bwd: gradient requested without gscale");
    if (kind != RNH_LOSS_L1 && kind != RNH_LOSS_CHARBONNIER) RNH_FAIL(RNH_E_RANGE, "rnh_loss_fwd_bwd: kind %d", kind);
    if (per & 3) RNH_FAIL(RNH_E_ALIGN, "rnh_loss_fwd_bwd: per-image element count must be a multiple of 4");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_kernel, dim3(RNH_LOSS_BLOCKS, G * T), dim3(256), 0, st, o, y, d_o, gscale, ws, T, (long)per, kind, eps);
    RNH_CHECK_LAUNCH("rnh_loss_fwd_bwd");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3((G * T + 63) / 64), dim3(64), 0, st, ws, loss, G * T, (long)per);
    RNH_CHECK_LAUNCH("rnh_loss_fwd_bwd(finalize)");
    return 0;
}

extern "C" int rnh_loss_total(const float *in, const float *w, float *out, int G, int T, int backward, void *stream) {
    if (!in || !w || !out || G < 1 || T < 1) RNH_FAIL(RNH_E_ARG, "rnh_loss_total: bad arguments");
    hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, w, out, G, T, backward);
    RNH_CHECK_LAUNCH("rnh_loss_total");
    return 0;
}

extern "C" int rnh_ew_add(float *out, const float *a, const float *b, const float *c, int64_t n, int accumulate, void *stream) {
    if (!out || !a || n < 1) RNH_FAIL(RNH_E_ARG, "rnh_ew_add: bad arguments");
    if (n & 3) RNH_FAIL(RNH_E_ALIGN, "rnh_ew_add: n must be a multiple of 4");
    hipLaunchKernelGGL(ew_add_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, out, a, b, c, (long)(n / 4),
                       accumulate);
    RNH_CHECK_LAUNCH("rnh_ew_add");
    return 0;
}

extern "C" int rnh_phase_plane(const float *pos, float *out, int N, int F, int H, int W, void *stream) {
    if (!pos || !out || N < 1 || F < 1 || H < 1 || W < 1) RNH_FAIL(RNH_E_ARG, "rnh_phase_plane: bad arguments");
    hipLaunchKernelGGL(phase_plane_kernel, dim3(grid_for((long)N * F * H * W)), dim3(256), 0, (hipStream_t)stream, pos, out, N, F,
                       (long)H * W);
    RNH_CHECK_LAUNCH("rnh_phase_plane");
    return 0;
}

extern "C" int rnh_xcol_pack(const float *w, float *out, int Cin, int co, int J, int cstride, int c0, int nch, int nvalid, void *stream) {
    if (!w || !out || Cin < 1 || co < 0 || J < 1 || nch < 1 || nvalid < 1 || nvalid > nch || (J - 1) * cstride + c0 + nvalid > Cin)
        RNH_FAIL(RNH_E_ARG, "rnh_xcol_pack: bad arguments");
    hipLaunchKernelGGL(xcol_pack_kernel, dim3(grid_for((long)J * nch * 9)), dim3(256), 0, (hipStream_t)stream, w, out, Cin, co, J, cstride, c0,
                       nch, nvalid);
    RNH_CHECK_LAUNCH("rnh_xcol_pack");
    return 0;
}

extern "C" int rnh_xcol_unpack(const float *dwx, const float *dbx, float *dw, float *db, int Cin, int co, int J, int cstride, int c0, int nch,
                               int nvalid, int accumulate, void *stream) {
    if (!dwx || !dw || (dbx && !db) || Cin < 1 || co < 0 || J < 1 || nch < 1 || nvalid < 1 || nvalid > nch ||
        (J - 1) * cstride + c0 + nvalid > Cin)
        RNH_FAIL(RNH_E_ARG, "rnh_xcol_unpack: bad arguments");
    hipLaunchKernelGGL(xcol_unpack_kernel, dim3(grid_for((long)J * nvalid * 9 + 1)), dim3(256), 0, (hipStream_t)stream, dwx, dbx, dw, db, Cin,
                       co, J, cstride, c0, nch, nvalid, accumulate);
    RNH_CHECK_LAUNCH("rnh_xcol_unpack");
    return 0;
}

extern "C" int rnh_xcol_combine(const float *z0, const float *z1, const float *z2, const float *bias, float *out, int64_t npix, int N,
                                int nwin, int J, int C, int c0, void *stream) {
    if (!z0 || !bias || !out || npix < 1 || N < 1 || nwin < 1 || J < 1 || c0 < 0 || c0 + 4 > C)
        RNH_FAIL(RNH_E_ARG, "rnh_xcol_combine: bad arguments");
    if ((C & 3) || (c0 & 3)) RNH_FAIL(RNH_E_ALIGN, "rnh_xcol_combine: C and c0 must be multiples of 4");
    hipLaunchKernelGGL(xcol_combine_kernel, dim3(grid_for((long)nwin * N * npix)), dim3(256), 0, (hipStream_t)stream, z0, z1, z2, bias, out,
                       (long)npix, N, nwin, J, C, c0);
    RNH_CHECK_LAUNCH("rnh_xcol_combine");
    return 0;
}

extern "C" int rnh_xcol_gather(const float *dy, float *E, int64_t npix, int N, int nwin, int J, int C, int c, void *stream) {
    if (!dy || !E || npix < 1 || N < 1 || nwin < 1 || J < 1 || c < 0 || c >= C) RNH_FAIL(RNH_E_ARG, "rnh_xcol_gather: bad arguments");
    hipLaunchKernelGGL(xcol_gather_kernel, dim3(grid_for((long)(nwin + J - 1) * N * npix * J)), dim3(256), 0, (hipStream_t)stream, dy, E,
                       (long)npix, N, nwin, J, C, c);
    RNH_CHECK_LAUNCH("rnh_xcol_gather");
    return 0;
}

extern "C" int64_t rnh_phase_wgrad_ws_floats(int H, int N, int nwin, int ncols) {
    const int64_t nstrips = (H + PWG_ROWS - 1) / PWG_ROWS;
    return (int64_t)nwin * N * (nstrips + 1) * 9 * ncols + 64;
}

extern "C" int rnh_phase_wgrad(const float *dy, const float *planes, float *dw, float *ws, int H, int W, int N, int nwin, int J, int Cin,
                               int cstride, int c0, int C, int ncols, int accumulate, void *stream) {
    if (!dy || !planes || !dw || !ws || H < 1 || W < 1 || N < 1 || nwin < 1 || J < 1 || ncols < 4 || ncols > C || (J - 1) * cstride + c0 >= Cin)
        RNH_FAIL(RNH_E_ARG, "rnh_phase_wgrad: bad arguments");
    if ((ncols & 3) || (C & 3) || 256 % (ncols / 4)) RNH_FAIL(RNH_E_ALIGN, "rnh_phase_wgrad: ncols / 4 must divide 256, C a multiple of 4");
    hipStream_t st = (hipStream_t)stream;
    const int nstrips = (H + PWG_ROWS - 1) / PWG_ROWS, nimg = nwin * N, lanes = 256 / (ncols / 4);
    const size_t shm = (size_t)lanes * 9 * ncols * sizeof(float);
    if (shm > 160 * 1024) RNH_FAIL(RNH_E_RANGE, "rnh_phase_wgrad: ncols too large");
    float *part = ws, *Q = ws + (long)nimg * nstrips * 9 * ncols;
    hipLaunchKernelGGL(phase_wgrad_sum_kernel, dim3((unsigned)(nimg * nstrips)), dim3(256), shm, st, dy, part, H, W, C, ncols, nstrips);
    RNH_CHECK_LAUNCH("rnh_phase_wgrad(sum)");
    hipLaunchKernelGGL(phase_wgrad_q_kernel, dim3(grid_for((long)nimg * 9 * ncols)), dim3(256), 0, st, part, Q, nimg, nstrips, ncols);
    RNH_CHECK_LAUNCH("rnh_phase_wgrad(Q)");
    hipLaunchKernelGGL(phase_wgrad_finish_kernel, dim3(grid_for((long)ncols * J * 9)), dim3(256), 0, st, Q, planes, dw, (long)H * W, N, nwin, J,
                       Cin, cstride, c0, ncols, accumulate);
    RNH_CHECK_LAUNCH("rnh_phase_wgrad");
    return 0;
}

extern "C" int rnh_xcol_dgrad(const float *g, const float *w, float *dx0, float *dx1, float *ws, int H, int W, int N, int T, int J, int Cin,
                              int cstride, int c, int C, int Cl, void *stream) {
    if (!g || !w || !dx0 || !dx1 || !ws || H < 1 || W < 1 || N < 1 || T < 1 || J < 1 || !(J & 1) || c < 0 || c >= C || (J - 1) * cstride + 2 * Cl > Cin)
        RNH_FAIL(RNH_E_ARG, "rnh_xcol_dgrad: bad arguments");
    if (Cl % 64) RNH_FAIL(RNH_E_RANGE, "rnh_xcol_dgrad: built for hidden widths in multiples of 64");
    hipStream_t st = (hipStream_t)stream;
    const int ncols = 2 * Cl;
    hipLaunchKernelGGL(xdgrad_pack_kernel, dim3(grid_for((long)J * 9 * ncols)), dim3(256), 0, st, w, ws, Cin, c, J, cstride, ncols);
    RNH_CHECK_LAUNCH("rnh_xcol_dgrad(pack)");
    const int TX = (W + XDT - 1) / XDT, TY = (H + XDT - 1) / XDT;
    const size_t shm = (size_t)J * XDP * XDP * sizeof(float);
    hipLaunchKernelGGL(xdgrad_tile_kernel<64>, dim3((unsigned)(T * N * TX * TY), ncols / 64), dim3(256), shm, st, g, ws, dx0, dx1, H, W, N, J, C, c,
                       Cl, ncols, TX, TY, J - 1);
    RNH_CHECK_LAUNCH("rnh_xcol_dgrad");
    return 0;
}

extern "C" int rnh_phase_bias_add(float *out, const float *planes, const float *w, float *ws, int H, int W, int N, int nwin, int J, int Cin,
                                  int cstride, int c0, int C, int ncols, void *stream) {
    if (!out || !planes || !w || !ws || H < 1 || W < 1 || N < 1 || nwin < 1 || J < 1 || ncols < 4 || ncols > C ||
        (J - 1) * cstride + c0 >= Cin)
        RNH_FAIL(RNH_E_ARG, "rnh_phase_bias_add: bad arguments");
    if ((ncols & 3) || (C & 3)) RNH_FAIL(RNH_E_ALIGN, "rnh_phase_bias_add: ncols and C must be multiples of 4");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(phase_bias_table_kernel, dim3(grid_for(16L * J * ncols)), dim3(256), 0, st, w, ws, Cin, J, cstride, c0, ncols);
    RNH_CHECK_LAUNCH("rnh_phase_bias_add(table)");
    hipLaunchKernelGGL(phase_bias_add_kernel, dim3(grid_for((long)nwin * N * H * W * (ncols / 4), 256, 32768)), dim3(256), 0, st, out, planes,
                       ws, H, W, N, nwin, J, C, ncols);
    RNH_CHECK_LAUNCH("rnh_phase_bias_add");
    return 0;
}
