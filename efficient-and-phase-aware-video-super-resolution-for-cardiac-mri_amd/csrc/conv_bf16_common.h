// Helpers shared by the bf16 MFMA convolution kernels (conv_bf16.hip, conv_bf16p.hip): packing / unpacking of 8-element groups, the gate
// activations, raw-buffer descriptors.  Internal to csrc/ (not part of the C ABI of include/refinenet_hip.h).
#pragma once
#include "rnh_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int TW = 32, HPW = TW + 2;                                         // tile width; halo width
constexpr int PITCH = 48;                                                    // bytes per halo pixel / weight column

__device__ __forceinline__ unsigned pk2(float a, float b) {
    const bf16x2 r = {(__bf16)a, (__bf16)b};                                 // v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ uint4 pack8(const float4 a, const float4 b) {
    return make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w));
}
// two / eight bf16 -> IEEE half (exact above 2^-14; v_cvt_pkrtz_f16_f32 has nothing to truncate there): the operands of the f16 MFMA forms
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned bf2h2(unsigned x) {
    const float lo = __builtin_bit_cast(float, x << 16), hi = __builtin_bit_cast(float, x & 0xffff0000u);
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi));
}
__device__ __forceinline__ uint4 bf2h8(uint4 v) { return make_uint4(bf2h2(v.x), bf2h2(v.y), bf2h2(v.z), bf2h2(v.w)); }
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

// tanh as 2 sigmoid(2x) - 1: one exp, one rcp, two FMAs; absolute error <= 2 ulp of 1 (as the candidate gate of
// conv_wino.hip) - in the bf16-storage path h' = o tanh(c') is rounded to 8 bits anyway
__device__ __forceinline__ float b_tanh_fast(float x) { return __builtin_fmaf(2.f, __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)), -1.f); }

// wave-uniform raw-buffer descriptor: base + 2 GiB window; an offset of 0xFFFFFFFF is out of range and reads as 0 -
// zero padding and absent channels cost no branch (the convention of conv_igemm.hip / conv_wino.hip)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bdesc(const void *p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(u & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint4 bld16(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

}  // namespace
