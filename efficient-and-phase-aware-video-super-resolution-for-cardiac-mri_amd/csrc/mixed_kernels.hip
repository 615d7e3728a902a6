// HBM-bound helpers of the bf16-storage path (gfx950): the element type of every operand is a run-time tag (RNH_DT_F32 /
// RNH_DT_BF16), arithmetic is fp32, every thread moves 8 consecutive elements with 16-byte accesses.
//   rnh_ew_add_m          residual adds / feature update / gradient sums (refine_net.py:102-133 and their backward)
//   rnh_lstm_gates_bwd_m  backward of the ConvLSTM gate math (refine_net.py:258-265), c and dc in fp32
//   rnh_cast              fp32 <-> bf16 (the input block's features and their gradient cross the precision boundary here)
//   rnh_phase_plane_m     phase plane with Cp channels (p, 0, ..., 0) of either type (refine_net.py:168)
#include "rnh_common.h"

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned mpk2(float a, float b) {
    const bf16x2 r = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void mload8(const void *p, int dt, long e, float *f) {
    if (dt == RNH_DT_BF16) {
        const uint4 u = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned short *>(p) + e);
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __builtin_bit_cast(float, w[i] << 16);
            f[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
        }
    } else {
        const float4 a = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + e);
        const float4 b = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + e + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
}
__device__ __forceinline__ void mstore8(void *p, int dt, long e, const float *f) {
    if (dt == RNH_DT_BF16) {
        *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned short *>(p) + e) =
            make_uint4(mpk2(f[0], f[1]), mpk2(f[2], f[3]), mpk2(f[4], f[5]), mpk2(f[6], f[7]));
    } else {
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p) + e) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p) + e + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
}

__global__ void __launch_bounds__(256) ew_add_m_kernel(void *out, int odt, const void *a, int adt, const void *b, int bdt, const void *c, int cdt,
                                                       long n8, int accumulate) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float s[8], t[8];
        mload8(a, adt, i * 8, s);
        if (b) {
            mload8(b, bdt, i * 8, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += t[e];
        }
        if (c) {
            mload8(c, cdt, i * 8, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += t[e];
        }
        if (accumulate) {
            mload8(out, odt, i * 8, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += t[e];
        }
        mstore8(out, odt, i * 8, s);
    }
}

__global__ void __launch_bounds__(256) gates_bwd_m_kernel(const void *dh, int hdt, const void *dh2, int h2dt, const float *dcn, const void *gates, int gdt,
                                                          const float *cprev, const float *cnext, void *dgates, int dgdt, float *dcprev, long npix,
                                                          int hd) {
    const int G = hd >> 3;
    const long total = npix * G;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % G);
        const long p = idx / G;
        const long o = p * hd + g * 8, og = p * 4 * hd + g * 8;
        float vdh[8], t[8], vdc[8], vcp[8], vcn[8], gi[8], gf[8], go[8], gg[8];
        mload8(dh, hdt, o, vdh);
        if (dh2) {
            mload8(dh2, h2dt, o, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) vdh[e] += t[e];
        }
        if (dcn) mload8(dcn, RNH_DT_F32, o, vdc);
        if (cprev) mload8(cprev, RNH_DT_F32, o, vcp);
        mload8(cnext, RNH_DT_F32, o, vcn);
        mload8(gates, gdt, og, gi);
        mload8(gates, gdt, og + hd, gf);
        mload8(gates, gdt, og + 2 * hd, go);
        mload8(gates, gdt, og + 3 * hd, gg);
        float di[8], df[8], dgo[8], dg[8], dcp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // bf16 gates = the bf16-storage path, whose forward cell computed h' = o tanh(c') with this form of tanh (conv_bf16.hip b_tanh_fast:
            // one exp, one rcp, absolute error <= 2 ulp of 1); the fused epilogue of conv_bf16d_kernel<LSTM_BWD> computes the same expression
            const float th = gdt == RNH_DT_BF16 ? __builtin_fmaf(2.f, __builtin_amdgcn_rcpf(1.f + __expf(-2.f * vcn[e])), -1.f) : tanhf(vcn[e]);
            const float d_o = vdh[e] * th;
            const float dct = (dcn ? vdc[e] : 0.f) + vdh[e] * go[e] * (1.f - th * th);
            di[e] = dct * gg[e] * gi[e] * (1.f - gi[e]);
            df[e] = dct * (cprev ? vcp[e] : 0.f) * gf[e] * (1.f - gf[e]);
            dgo[e] = d_o * go[e] * (1.f - go[e]);
            dg[e] = dct * gi[e] * (1.f - gg[e] * gg[e]);
            dcp[e] = dct * gf[e];
        }
        mstore8(dgates, dgdt, og, di);
        mstore8(dgates, dgdt, og + hd, df);
        mstore8(dgates, dgdt, og + 2 * hd, dgo);
        mstore8(dgates, dgdt, og + 3 * hd, dg);
        if (dcprev) mstore8(dcprev, RNH_DT_F32, o, dcp);
    }
}

__global__ void __launch_bounds__(256) cast_kernel(const void *src, int sdt, void *dst, int ddt, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float f[8];
        mload8(src, sdt, i * 8, f);
        mstore8(dst, ddt, i * 8, f);
    }
}

// out[(f*N + n)][pixel][0..Cp) = (pos[n*F + f], 0, ..., 0), Cp = 8
__global__ void __launch_bounds__(256) phase_plane_m_kernel(const float *pos, void *out, int odt, int N, int F, long hw) {
    const long total = (long)N * F * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long img = i / hw;
        const int f = (int)(img / N), n = (int)(img - (long)f * N);
        const float v[8] = {pos[(long)n * F + f], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        mstore8(out, odt, i * 8, v);
    }
}

// Frame-wise side path of ONE output channel of refine conv1 in the bf16-storage path (the channel c0 = 2 Cl that makes 129 out of
// 128 columns): z (F' N, H, W, 8) fp32 holds, per SOURCE frame, the J slot convolutions of that channel (one small rnh_conv_bf16 over
// the frames); a window's value is bias + sum_j z[frame i + j][slot j].  The 8 channels c0 .. c0 + 7 of out are written: (value, 0 x 7)
__global__ void __launch_bounds__(256) xcol_combine_m_kernel(const float *z, const float *bias, void *out, int odt, long npix, int N, int nwin,
                                                             int J, int C, int c0) {
    const long total = (long)nwin * N * npix;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long img = idx / npix, p = idx - img * npix;
        const int i = (int)(img / N), n = (int)(img - (long)i * N);
        float s = bias[c0];
        for (int j = 0; j < J; ++j) s += z[((((long)(i + j) * N + n) * npix) + p) * 8 + j];
        const float v[8] = {s, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        mstore8(out, odt, idx * C + c0, v);
    }
}

// ... and its backward: E (F' N, H, W, 8)[frame f][slot j] = dy[window f - j][channel c] (0 where f - j is no window; slots >= J: 0):
// the gradient operand of the per-frame convolution's weight gradient
__global__ void __launch_bounds__(256) xcol_gather_m_kernel(const void *dy, int ydt, void *E, int edt, long npix, int N, int nwin, int J, int C,
                                                            int c) {
    const long total = (long)(nwin + J - 1) * N * npix;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long img = idx / npix, p = idx - img * npix;
        const int f = (int)(img / N), n = (int)(img - (long)f * N);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int w = f - j;
            v[j] = 0.f;
            if (j < J && w >= 0 && w < nwin) {
                const long e = ((((long)w * N + n) * npix) + p) * C + c;
                v[j] = ydt == RNH_DT_BF16 ? __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short *>(dy)[e] << 16)
                                          : reinterpret_cast<const float *>(dy)[e];
            }
        }
        mstore8(E, edt, idx * 8, v);
    }
}

inline int mgrid(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}
inline bool bad_dt(int d) { return d != RNH_DT_F32 && d != RNH_DT_BF16; }

}  // namespace

extern "C" int rnh_ew_add_m(void *out, int out_dt, const void *a, int a_dt, const void *b, int b_dt, const void *c, int c_dt, int64_t n,
                            int accumulate, void *stream) {
    if (!out || !a || n < 1 || bad_dt(out_dt) || bad_dt(a_dt) || (b && bad_dt(b_dt)) || (c && bad_dt(c_dt))) RNH_FAIL(RNH_E_ARG, "rnh_ew_add_m: bad arguments");
    if (n & 7) RNH_FAIL(RNH_E_ALIGN, "rnh_ew_add_m: n must be a multiple of 8");
    hipLaunchKernelGGL(ew_add_m_kernel, dim3(mgrid(n / 8)), dim3(256), 0, (hipStream_t)stream, out, out_dt, a, a_dt, b, b_dt, c, c_dt, (long)(n / 8),
                       accumulate);
    RNH_CHECK_LAUNCH("rnh_ew_add_m");
    return 0;
}

extern "C" int rnh_lstm_gates_bwd_m(const void *dh, int dh_dt, const void *dh2, int dh2_dt, const float *dc_next, const void *gates, int g_dt, const float *c_prev,
                                    const float *c_next, void *dgates, int dg_dt, float *dc_prev, int64_t npix, int hd, void *stream) {
    if (!dh || !gates || !c_next || !dgates || npix < 1 || hd < 8 || bad_dt(dh_dt) || (dh2 && bad_dt(dh2_dt)) || bad_dt(g_dt) || bad_dt(dg_dt)) RNH_FAIL(RNH_E_ARG, "rnh_lstm_gates_bwd_m: bad arguments");
    if (hd & 7) RNH_FAIL(RNH_E_ALIGN, "rnh_lstm_gates_bwd_m: hd must be a multiple of 8");
    hipLaunchKernelGGL(gates_bwd_m_kernel, dim3(mgrid(npix * (hd / 8))), dim3(256), 0, (hipStream_t)stream, dh, dh_dt, dh2, dh2_dt, dc_next, gates, g_dt,
                       c_prev, c_next, dgates, dg_dt, dc_prev, (long)npix, hd);
    RNH_CHECK_LAUNCH("rnh_lstm_gates_bwd_m");
    return 0;
}

extern "C" int rnh_cast(const void *src, int src_dt, void *dst, int dst_dt, int64_t n, void *stream) {
    if (!src || !dst || n < 1 || bad_dt(src_dt) || bad_dt(dst_dt)) RNH_FAIL(RNH_E_ARG, "rnh_cast: bad arguments");
    if (n & 7) RNH_FAIL(RNH_E_ALIGN, "rnh_cast: n must be a multiple of 8");
    hipLaunchKernelGGL(cast_kernel, dim3(mgrid(n / 8)), dim3(256), 0, (hipStream_t)stream, src, src_dt, dst, dst_dt, (long)(n / 8));
    RNH_CHECK_LAUNCH("rnh_cast");
    return 0;
}

extern "C" int rnh_phase_plane_m(const float *pos, void *out, int out_dt, int N, int F, int H, int W, void *stream) {
    if (!pos || !out || N < 1 || F < 1 || H < 1 || W < 1 || bad_dt(out_dt)) RNH_FAIL(RNH_E_ARG, "rnh_phase_plane_m: bad arguments");
    hipLaunchKernelGGL(phase_plane_m_kernel, dim3(mgrid((long)N * F * H * W)), dim3(256), 0, (hipStream_t)stream, pos, out, out_dt, N, F, (long)H * W);
    RNH_CHECK_LAUNCH("rnh_phase_plane_m");
    return 0;
}

extern "C" int rnh_xcol_combine_m(const float *z, const float *bias, void *out, int out_dt, int64_t npix, int N, int nwin, int J, int C, int c0,
                                  void *stream) {
    if (!z || !bias || !out || npix < 1 || N < 1 || nwin < 1 || J < 1 || J > 8 || bad_dt(out_dt)) RNH_FAIL(RNH_E_ARG, "rnh_xcol_combine_m: bad arguments");
    if ((C & 7) || (c0 & 7) || c0 + 8 > C) RNH_FAIL(RNH_E_ALIGN, "rnh_xcol_combine_m: c0 .. c0 + 7 must be an aligned group of 8 channels of C");
    hipLaunchKernelGGL(xcol_combine_m_kernel, dim3(mgrid((long)nwin * N * npix)), dim3(256), 0, (hipStream_t)stream, z, bias, out, out_dt,
                       (long)npix, N, nwin, J, C, c0);
    RNH_CHECK_LAUNCH("rnh_xcol_combine_m");
    return 0;
}

extern "C" int rnh_xcol_gather_m(const void *dy, int dy_dt, void *E, int e_dt, int64_t npix, int N, int nwin, int J, int C, int c, void *stream) {
    if (!dy || !E || npix < 1 || N < 1 || nwin < 1 || J < 1 || J > 8 || c < 0 || c >= C || bad_dt(dy_dt) || bad_dt(e_dt))
        RNH_FAIL(RNH_E_ARG, "rnh_xcol_gather_m: bad arguments");
    hipLaunchKernelGGL(xcol_gather_m_kernel, dim3(mgrid((long)(nwin + J - 1) * N * npix)), dim3(256), 0, (hipStream_t)stream, dy, dy_dt, E, e_dt,
                       (long)npix, N, nwin, J, C, c);
    RNH_CHECK_LAUNCH("rnh_xcol_gather_m");
    return 0;
}

extern "C" void rnh_struct_sizes_bf16(int32_t out[4]) {
    out[0] = (int32_t)sizeof(rnh_msrc_t);
    out[1] = (int32_t)sizeof(rnh_mdst_t);
    out[2] = (int32_t)sizeof(rnh_conv_bf16_args_t);
    out[3] = (int32_t)sizeof(rnh_wgrad_bf16_args_t);
}
