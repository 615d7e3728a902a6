// What follows the backward pass in a training step (gfx950), SURVEY.md section 8 rows f3 and f4:
//
//   rnh_adam_step           the Adam update on flat buffers (parameters, gradients and both moments each one
//                           contiguous fp32 range): replaces torch.optim.Adam.step over 25 tensors
//                           (reference src/main.py:76, acdc_vsr_refinenet_trainer.py:47, exp1_x4.yaml:56-60).
//   rnh_metrics_psnr_ssim   denormalize + PSNR + SSIM of all (output, target) image pairs of a step in one pass
//                           (reference acdc_vsr_refinenet_trainer.py:103-120, src/utils.py:1-20,
//                           src/model/metrics.py:20-36 and :86-113): the reference runs, per frame, 2 denormalisations
//                           (5 elementwise kernels each), an MSE, and five 11x11 depthwise convolutions plus a dozen
//                           elementwise kernels over 512x512 maps.
//
// Both are HBM-bound.  Adam: 28 B per parameter (p, g, m, v read; p, m, v written), 16-byte accesses.  Metrics: every
// pixel of both images is read once from HBM (8 B per pixel pair; the 10-pixel halo of a tile comes from L2), the
// Gaussian window is separable (11 + 11 taps instead of 121) and lives in scalar registers, the five windowed sums go
// through LDS once, nothing but two partial sums per tile is written.
#include "rnh_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// Adam
// ---------------------------------------------------------------------------------------------------------------
struct AdamK {
    float lerp_w;      // 1 - beta1
    float beta2, one_m_beta2;
    float bc2_sqrt;    // sqrt(1 - beta2^t)
    float eps, step_size /* lr / (1 - beta1^t) */, weight_decay;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamK &k) {
    if (k.weight_decay != 0.f) g = __fadd_rn(g, __fmul_rn(k.weight_decay, p));
    m = __fadd_rn(m, __fmul_rn(k.lerp_w, __fsub_rn(g, m)));                                   // lerp_(g, 1 - beta1)
    v = __fadd_rn(__fmul_rn(v, k.beta2), __fmul_rn(__fmul_rn(k.one_m_beta2, g), g));          // mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), k.bc2_sqrt), k.eps);
    p = __fadd_rn(p, __fmul_rn(-k.step_size, __fdiv_rn(m, denom)));                           // addcdiv_(m, denom, -step_size)
}

// head: the 0..3 elements before the first 16-byte boundary (the four buffers are congruent modulo 16 bytes: checked on the host)
__global__ void __launch_bounds__(256) adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, long n, int head, AdamK k) {
    const long n4 = (n - head) >> 2;
    float *pb = p + head, *mb = m + head, *vb = v + head;
    const float *gb = g + head;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 P = rnh_ld4(pb + 4 * i), G = rnh_ld4(gb + 4 * i), M = rnh_ld4(mb + 4 * i), V = rnh_ld4(vb + 4 * i);
        adam_one(P.x, G.x, M.x, V.x, k);
        adam_one(P.y, G.y, M.y, V.y, k);
        adam_one(P.z, G.z, M.z, V.z, k);
        adam_one(P.w, G.w, M.w, V.w, k);
        rnh_st4(pb + 4 * i, P);
        rnh_st4(mb + 4 * i, M);
        rnh_st4(vb + 4 * i, V);
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) {          // lanes 0..3: head elements, lanes 4..7: tail elements
        const int t = threadIdx.x;
        const long tail0 = head + (n4 << 2);
        const long i = t < 4 ? (long)t : tail0 + (t - 4);
        if (t < 4 ? t < head : i < n) {
            float P = p[i], M = m[i], V = v[i];
            adam_one(P, g[i], M, V, k);
            p[i] = P, m[i] = M, v[i] = V;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// PSNR + SSIM
// ---------------------------------------------------------------------------------------------------------------
constexpr int KW = 11;               // window
constexpr int TY = 16, TX = 64;      // tile of the SSIM map per workgroup (one wave per row: conflict-free LDS rows)
constexpr int PY = TY + KW - 1, PX = TX + KW - 1;   // 26 x 74 input patch

struct Gauss11 {
    float g[KW];
};

__device__ __forceinline__ float denorm1(float x, float mean, float stdv) {
    // (x * std + mean).round().clamp(0, 255): two roundings like the two ATen kernels, round half to even
    return fminf(fmaxf(rintf(__fadd_rn(__fmul_rn(x, stdv), mean)), 0.f), 255.f);
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__global__ void __launch_bounds__(256) metrics_tile_kernel(const float *__restrict__ o, const float *__restrict__ y, int H, int W, int tiles_y,
                                                           int tiles_x, int total, int denorm, int want_ssim, float mean, float stdv, float c1, float c2,
                                                           Gauss11 G, float *__restrict__ partial) {
    __shared__ float so[PY][PX];
    __shared__ float sy[PY][PX];
    __shared__ float hs[5][PY][TX];
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = rnh_xcd_remap(blockIdx.x, total);
    const int per = tiles_y * tiles_x;
    const int img = bid / per, t = bid - img * per;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y0 = ty * TY, x0 = tx * TX;
    const float *oi = o + (long)img * H * W, *yi = y + (long)img * H * W;

    // patch -> LDS (denormalised), squared error of the pixels this tile owns (its TY x TX corner of the patch)
    float se = 0.f;
    for (int i = tid; i < PY * PX; i += 256) {
        const int r = i / PX, c = i - r * PX;
        const int gy = y0 + r, gx = x0 + c;
        float a = 0.f, b = 0.f;
        if (gy < H && gx < W) {
            a = oi[(long)gy * W + gx];
            b = yi[(long)gy * W + gx];
            if (denorm) a = denorm1(a, mean, stdv), b = denorm1(b, mean, stdv);
            if (r < TY && c < TX) se += (a - b) * (a - b);
        }
        so[r][c] = a;
        sy[r][c] = b;
    }
    __syncthreads();
    // horizontal pass: one wave per patch row, a lane per output column
    for (int r = wave; want_ssim && r < PY; r += 4) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const float a = so[r][lane + k], b = sy[r][lane + k], w = G.g[k];
            const float wa = w * a, wb = w * b;
            s0 += wa, s1 += wb, s2 += w * (a * a), s3 += w * (b * b), s4 += w * (a * b);
        }
        hs[0][r][lane] = s0, hs[1][r][lane] = s1, hs[2][r][lane] = s2, hs[3][r][lane] = s3, hs[4][r][lane] = s4;
    }
    __syncthreads();
    // vertical pass + SSIM map, masked to the valid (H-10) x (W-10) range
    float ss = 0.f;
    for (int r = wave; want_ssim && r < TY; r += 4) {
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const float w = G.g[k];
            mu1 += w * hs[0][r + k][lane], mu2 += w * hs[1][r + k][lane];
            e11 += w * hs[2][r + k][lane], e22 += w * hs[3][r + k][lane], e12 += w * hs[4][r + k][lane];
        }
        if (y0 + r < H - (KW - 1) && x0 + lane < W - (KW - 1)) {
            const float m11 = mu1 * mu1, m22 = mu2 * mu2, m12 = mu1 * mu2;
            const float s1 = e11 - m11, s2 = e22 - m22, s12 = e12 - m12;
            ss += ((2.f * m12 + c1) * (2.f * s12 + c2)) / ((m11 + m22 + c1) * (s1 + s2 + c2));
        }
    }
    se = wave_sum(se), ss = wave_sum(ss);
    if (lane == 0) red[0][wave] = se, red[1][wave] = ss;
    __syncthreads();
    if (tid == 0) {
        partial[2 * (long)bid] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[2 * (long)bid + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__device__ __forceinline__ double wave_sum_d(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// One workgroup: sums the tiles of every image in fixed order (double), then the per-sample PSNR and the means.
// result: [0] mean PSNR over samples, [1] mean SSIM over images, [2 .. 2+S) PSNR per sample (S = P / cps),
//         [2+S .. 2+S+P) SSIM-map mean per image, [2+S+P .. 2+S+2P) MSE per image.
__global__ void __launch_bounds__(1024) metrics_finish_kernel(const float *__restrict__ partial, int P, int cps, int per, double inv_px, double inv_map,
                                                              double max_sq, float *__restrict__ result) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = P / cps;
    float *psnr_s = result + 2, *ssim_i = result + 2 + S, *mse_i = result + 2 + S + P;
    for (int img = wave; img < P; img += 16) {
        double a = 0.0, b = 0.0;
        for (int t = lane; t < per; t += 64) a += (double)partial[2 * ((long)img * per + t)], b += (double)partial[2 * ((long)img * per + t) + 1];
        a = wave_sum_d(a), b = wave_sum_d(b);
        if (lane == 0) mse_i[img] = (float)(a * inv_px), ssim_i[img] = (float)(b * inv_map);
    }
    __threadfence_block();
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += 1024) {
        double m = 0.0;
        for (int c = 0; c < cps; ++c) m += (double)mse_i[s * cps + c];
        const float mse = (float)(m / cps);
        psnr_s[s] = (float)(10.0 * log10(max_sq / ((double)mse + 1e-10)));
    }
    __threadfence_block();
    __syncthreads();
    if (wave == 0) {
        double a = 0.0, b = 0.0;
        for (int s = lane; s < S; s += 64) a += (double)psnr_s[s];
        for (int i = lane; i < P; i += 64) b += (double)ssim_i[i];
        a = wave_sum_d(a), b = wave_sum_d(b);
        if (lane == 0) result[0] = (float)(a / S), result[1] = (float)(b / P);
    }
}

}  // namespace

extern "C" int rnh_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, int32_t step, float lr,
                             float beta1, float beta2, float eps, float weight_decay, void *stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq) RNH_FAIL(RNH_E_ARG, "rnh_adam_step: null pointer");
    if (n <= 0 || step < 1) RNH_FAIL(RNH_E_ARG, "rnh_adam_step: n and step must be positive");
    const uintptr_t a = (uintptr_t)param & 15;
    if ((a & 3) || ((uintptr_t)grad & 15) != a || ((uintptr_t)exp_avg & 15) != a || ((uintptr_t)exp_avg_sq & 15) != a)
        RNH_FAIL(RNH_E_ALIGN, "rnh_adam_step: the four buffers must be 4-byte aligned and congruent modulo 16 bytes");
    int head = (int)(((16 - a) & 15) >> 2);
    if (head > n) head = (int)n;
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f)) RNH_FAIL(RNH_E_ARG, "rnh_adam_step: betas must lie in [0, 1)");
    // scalars in double like torch.optim's Python arithmetic, rounded once to fp32
    const double b1 = (double)beta1, b2 = (double)beta2;
    AdamK k;
    k.lerp_w = (float)(1.0 - b1);
    k.beta2 = beta2;
    k.one_m_beta2 = (float)(1.0 - b2);
    k.bc2_sqrt = (float)sqrt(1.0 - pow(b2, (double)step));
    k.eps = eps;
    k.step_size = (float)((double)lr / (1.0 - pow(b1, (double)step)));
    k.weight_decay = weight_decay;
    long grid = ((n >> 2) + 255) / 256;
    if (grid < 1) grid = 1;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (long)n, head, k);
    RNH_CHECK_LAUNCH("rnh_adam_step");
    return 0;
}

extern "C" int64_t rnh_metrics_ws_floats(int P, int H, int W) {
    if (P <= 0 || H <= 0 || W <= 0) return 0;
    return 2 * (int64_t)P * ((H + TY - 1) / TY) * ((W + TX - 1) / TX);
}

extern "C" int rnh_metrics_psnr_ssim(const float *out, const float *tgt, int P, int cps, int H, int W, int denorm, int want_ssim, float mean, float stdv,
                                     float max_value, float value_range, const float *window11_host, float *ws, float *result, void *stream) {
    if (!out || !tgt || !window11_host || !ws || !result) RNH_FAIL(RNH_E_ARG, "rnh_metrics_psnr_ssim: null pointer");
    if (P <= 0 || cps <= 0 || P % cps) RNH_FAIL(RNH_E_ARG, "rnh_metrics_psnr_ssim: P must be a positive multiple of cps");
    if (H <= 0 || W <= 0) RNH_FAIL(RNH_E_ARG, "rnh_metrics_psnr_ssim: bad image size");
    if (want_ssim && (H < KW || W < KW)) RNH_FAIL(RNH_E_ARG, "rnh_metrics_psnr_ssim: the image (%dx%d) is smaller than the 11x11 window", H, W);
    const int tiles_y = (H + TY - 1) / TY, tiles_x = (W + TX - 1) / TX;
    const long total = (long)P * tiles_y * tiles_x;
    if (total > 0x7fffffffL) RNH_FAIL(RNH_E_ARG, "rnh_metrics_psnr_ssim: too many tiles");
    Gauss11 G;
    for (int k = 0; k < KW; ++k) G.g[k] = window11_host[k];
    const float c1 = (0.01f * value_range) * (0.01f * value_range), c2 = (0.03f * value_range) * (0.03f * value_range);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(metrics_tile_kernel, dim3((unsigned)total), dim3(256), 0, st, out, tgt, H, W, tiles_y, tiles_x, (int)total, denorm, want_ssim, mean, stdv,
                       c1, c2, G, ws);
    RNH_CHECK_LAUNCH("rnh_metrics_psnr_ssim (tiles)");
    hipLaunchKernelGGL(metrics_finish_kernel, dim3(1), dim3(1024), 0, st, ws, P, cps, tiles_y * tiles_x, 1.0 / ((double)H * W),
                       want_ssim ? 1.0 / ((double)(H - KW + 1) * (W - KW + 1)) : 0.0, (double)max_value * (double)max_value, result);
    RNH_CHECK_LAUNCH("rnh_metrics_psnr_ssim (finish)");
    return 0;
}
