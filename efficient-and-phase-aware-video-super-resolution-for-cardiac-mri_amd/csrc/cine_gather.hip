// Input feeding for the RefineNet hot path (gfx950), SURVEY.md section 8 row f1: the cines stay resident in HBM
// (one pool of fp32 frames), and one launch per batch cuts the LR window / HR targets / phase codes of every sample,
// applies the flips and the crop of the reference's augmentation list and the normalisation, and writes the packed
// (F, N, h, w) / (T, N, sh, sw) / (N, F) buffers the engine consumes.  It replaces, per __getitem__, two nib.load of
// whole .nii.gz cines, np.flip, slicing, Normalize, ToTensor, and the default collate (reference
// src/data/datasets/acdc_vsr_refinenet_dataset.py:49-89, src/data/transforms.py:74-168,321-450).
//
// HBM-bound byte mover: one wave per output row (rows are 128 B .. 2 KB contiguous on both sides, a horizontal flip
// only reverses the lane order inside the row), the sample descriptor is wave-uniform and read through scalar loads.
// Algorithmic bytes: 8 B per output pixel (4 read + 4 written).
#include "rnh_common.h"
#include <string.h>

namespace {

// The sample descriptors travel BY VALUE in the kernel-argument segment (copied by the runtime when the launch is
// enqueued), RNH_CINE_CHUNK samples per launch.  An earlier version uploaded them with hipMemcpyAsync from the caller's
// pageable array, which the Python binding frees (and the next batch re-fills) as soon as rnh_cine_gather returns: correct
// only as long as the runtime stages pageable copies before returning.  ROCm 7.2 on MI355X does (tools/pageable_async_probe.hip:
// 0 of 400 copies saw later host writes, profiles/ARCHIVE/r02_a_pageable_probe.txt), so this was a latent dependence on
// unspecified behaviour, not an observed fault; by-value arguments remove it and the descriptor buffer.
constexpr int RNH_CINE_CHUNK = 32;
struct cine_chunk_t {
    rnh_cine_sample_t s[RNH_CINE_CHUNK];
};

__global__ void __launch_bounds__(256) cine_gather_kernel(const float *__restrict__ pool, const cine_chunk_t S, int n0, int cnt, int N, int F, int T,
                                                          int s, int h, int w, int normalize, float mean, float stdv,
                                                          float *__restrict__ inputs, float *__restrict__ targets, float *__restrict__ pos) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sh = s * h, sw = s * w;
    const int rows_lr = F * h, rows_per = rows_lr + T * sh;
    const long units = (long)cnt * rows_per;
    for (long u = (long)blockIdx.x * 4 + wave; u < units; u += (long)gridDim.x * 4) {
        const int nl = (int)(u / rows_per);
        const int n = n0 + nl;
        int r = (int)(u - (long)nl * rows_per);
        const rnh_cine_sample_t c = S.s[nl];                // wave-uniform index into the kernel arguments: scalar loads
        const bool lr = r < rows_lr;
        if (!lr) r -= rows_lr;
        const int rh = lr ? h : sh, rw = lr ? w : sw, k = lr ? 1 : s;
        const int slot = r / rh, y = r - slot * rh;
        const int Hc = lr ? c.Hl : c.Hh, Wc = lr ? c.Wl : c.Wh;
        const int frame = ((lr ? c.lr_start : c.hr_start) + slot) % c.Tc;
        int ys = c.y0 * k + y;
        if (c.vflip) ys = Hc - 1 - ys;
        const float *src = pool + (lr ? c.lr_off : c.hr_off) + ((long)frame * Hc + ys) * Wc;
        float *dst = (lr ? inputs + ((long)slot * N + n) * h * (long)w : targets + ((long)slot * N + n) * sh * (long)sw) + (long)y * rw;
        const int xb = c.x0 * k;
        for (int x = lane; x < rw; x += 64) {
            const int xs = c.hflip ? Wc - 1 - (xb + x) : xb + x;
            float v = src[xs];
            if (normalize) v = (v - mean) / stdv;            // IEEE division (hipcc's default): bit-exact with numpy
            dst[x] = v;
        }
    }
    // phase codes: pos[n][k] = code[(lr_start + k) mod Tc]  (not normalised: dataset :71)
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)cnt * F; e += (long)gridDim.x * blockDim.x) {
        const int nl = (int)(e / F), kf = (int)(e - (long)nl * F);
        pos[(long)n0 * F + e] = pool[S.s[nl].code_off + (S.s[nl].lr_start + kf) % S.s[nl].Tc];
    }
}

}  // namespace

extern "C" int rnh_cine_gather(const float *pool, int64_t pool_floats, const rnh_cine_sample_t *samples_host, rnh_cine_sample_t *samples_dev,
                               int N, int F, int T, int s, int h, int w, int normalize, float mean, float stdv, float *inputs,
                               float *targets, float *pos, void *stream) {
    (void)samples_dev;                                      // kept in the signature (ABI 1); no longer used
    if (!pool || !samples_host || !inputs || !targets || !pos) RNH_FAIL(RNH_E_ARG, "rnh_cine_gather: null pointer");
    if (N <= 0 || F <= 0 || T <= 0 || s <= 0 || h <= 0 || w <= 0) RNH_FAIL(RNH_E_ARG, "rnh_cine_gather: bad sizes");
    if (normalize && !(stdv != 0.f)) RNH_FAIL(RNH_E_ARG, "rnh_cine_gather: zero standard deviation");
    // every row the kernel will touch must lie inside the pool: checked here, on the host copy of the descriptors
    for (int n = 0; n < N; ++n) {
        const rnh_cine_sample_t &c = samples_host[n];
        if (c.Tc <= 0 || c.Hl <= 0 || c.Wl <= 0 || c.Hh <= 0 || c.Wh <= 0 || c.lr_start < 0 || c.hr_start < 0 || c.y0 < 0 || c.x0 < 0)
            RNH_FAIL(RNH_E_ARG, "rnh_cine_gather: sample %d: bad descriptor", n);
        if (c.y0 + h > c.Hl || c.x0 + w > c.Wl)
            RNH_FAIL(RNH_E_ARG, "rnh_cine_gather: sample %d: LR crop (%d..%d, %d..%d) outside the %dx%d image", n, c.y0, c.y0 + h, c.x0, c.x0 + w, c.Hl, c.Wl);
        if ((c.y0 + h) * s > c.Hh || (c.x0 + w) * s > c.Wh)
            RNH_FAIL(RNH_E_ARG, "rnh_cine_gather: sample %d: HR crop outside the %dx%d image", n, c.Hh, c.Wh);
        const int64_t lr_end = c.lr_off + (int64_t)c.Tc * c.Hl * c.Wl, hr_end = c.hr_off + (int64_t)c.Tc * c.Hh * c.Wh;
        if (c.lr_off < 0 || c.hr_off < 0 || c.code_off < 0 || lr_end > pool_floats || hr_end > pool_floats || c.code_off + c.Tc > pool_floats)
            RNH_FAIL(RNH_E_ARG, "rnh_cine_gather: sample %d: cine outside the pool", n);
    }
    hipStream_t st = (hipStream_t)stream;
    for (int n0 = 0; n0 < N; n0 += RNH_CINE_CHUNK) {
        const int cnt = N - n0 < RNH_CINE_CHUNK ? N - n0 : RNH_CINE_CHUNK;
        cine_chunk_t ch;
        memset(&ch, 0, sizeof(ch));
        memcpy(ch.s, samples_host + n0, (size_t)cnt * sizeof(rnh_cine_sample_t));      // read here, on the calling thread
        const long units = (long)cnt * ((long)F * h + (long)T * s * h);
        long grid = (units + 3) / 4;
        if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(cine_gather_kernel, dim3((unsigned)grid), dim3(256), 0, st, pool, ch, n0, cnt, N, F, T, s, h, w, normalize, mean, stdv,
                           inputs, targets, pos);
        RNH_CHECK_LAUNCH("rnh_cine_gather");
    }
    return 0;
}
