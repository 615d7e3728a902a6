// Shared helpers of the gfx950 kernels behind include/refinenet_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "refinenet_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

void rnh_set_error(const char *fmt, ...);

#define RNH_FAIL(code, ...)            \
    do {                               \
        rnh_set_error(__VA_ARGS__);    \
        return (code);                 \
    } while (0)

#define RNH_CHECK_LAUNCH(name)                                               \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            rnh_set_error("%s: %s", (name), hipGetErrorString(e__));         \
            return (int)e__;                                                 \
        }                                                                    \
    } while (0)

static inline int rnh_check_src(const rnh_src_t &s, const char *who) {
    if (!s.ptr) RNH_FAIL(RNH_E_ARG, "%s: null source pointer", who);
    if (s.C <= 0 || s.nch <= 0 || s.c0 < 0 || s.c0 + s.nch > s.C) RNH_FAIL(RNH_E_ARG, "%s: bad channel range", who);
    if ((s.C & 3) || (s.c0 & 3) || (s.nch & 3)) RNH_FAIL(RNH_E_ALIGN, "%s: channels must be multiples of 4", who);
    if (s.scale < 1 || s.sub_y < 0 || s.sub_x < 0 || s.sub_y >= s.scale || s.sub_x >= s.scale)
        RNH_FAIL(RNH_E_ARG, "%s: bad scale / sub-pixel", who);
    return 0;
}

// Blocks are dealt round-robin over the 8 XCDs; give every XCD one contiguous chunk of the tile list so
// that neighbouring tiles (which share halo rows and weight slabs) hit the same L2.  Bijective for any n.
__device__ __forceinline__ int rnh_xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ float4 rnh_ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void rnh_st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// csrc/uptail.hip internals shared with csrc/uptail_bf16.hip (C++ linkage; not part of the C ABI of include/refinenet_hip.h)
int rnh_uptail_fwd_compose_(const float *w2, const float *b2, const float *w3, float *ws, int C1, int Cq, int r, int Co, hipStream_t st);
int rnh_uptail_fwd_border_bf16_(const void *y1, const float *ws, float *out, int B, int Hm, int Wm, int C1, int r, int Co, hipStream_t st);
int rnh_uptail_dgrad_border_bf16_(const float *d_o, const float *G, void *dy1, int B, int Hm, int Wm, int C1, int r, hipStream_t st);
void rnh_uptail_xcorr_shape_(int B, int Hm, int Wm, int r, int *TX, int *TY, int *nblk, int *nchunk, int *NT);
int rnh_uptail_xcorr_finish_bf16_(const void *y1, const float *d_o, float *M, float *S, float *ws, int B, int Hm, int Wm, int C1, int r,
                                  hipStream_t st);
