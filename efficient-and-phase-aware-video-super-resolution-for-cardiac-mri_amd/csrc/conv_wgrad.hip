// Weight gradient of the 3x3 / 1x1 convolutions on the fp32 matrix cores.
//
//   dW[tap][i][j] = sum_p X[p + tap][i] * dY[p][j]        i: forward input channel, j: output channel
//
// A GEMM whose contraction index is the pixel: per workgroup one tap, one [rows x cols] tile of dW and one
// contiguous range of pixels (grid.x = nsplit ranges); the partial tiles go to a slab and
// rnh_wgrad_reduce sums them in a fixed order (bitwise reproducible, no float atomics).
// Both operands are gathered from lists of NHWC sources exactly like the forward kernel's A operand (X is
// tap-shifted; dY can be pixel-unshuffled), staged through LDS as [16 pixels][tile columns] and read with
// ds_read_b32 (lane = channel, so consecutive lanes hit consecutive banks).
// Column sums of dY (= the bias gradient) are taken from the staged dY tile by the tap-0 / row-tile-0 blocks.
#include "rnh_common.h"

namespace {

constexpr int PK = 16;   // pixels per K step

template <int WM, int WN, int MI, int NI>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(const rnh_wgrad_args_t P, const int XT, const int YT,
                                                         const int steps_per_split, const int nsteps) {
    constexpr int BX = WM * MI * 32, BY = WN * NI * 32;      // tile rows (X channels) / columns (dY channels)
    constexpr int XG = BX / 4, YG = BY / 4;                    // float4 groups per pixel row
    constexpr int XIT = (PK * XG + 255) / 256, YIT = (PK * YG + 255) / 256;
    constexpr int STAGE = PK * (BX + BY);                      // floats per stage
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int split = blockIdx.x;
    const int xt = blockIdx.y / YT, yt = blockIdx.y - xt * YT;
    const int tap = blockIdx.z;
    const int H = P.H, W = P.W, HW = H * W;
    const int Mtot = P.B * HW;
    int dy = 0, dx = 0;
    if (P.ntaps == 9) {
        dy = tap / 3 - 1;
        dx = tap - (dy + 1) * 3 - 1;
    }

    // ---- static per-thread column groups --------------------------------------------------------------
    // X: element e = tid + 256*it -> pixel pp = e / XG, group g = e % XG
    int xpp[XIT], xsrc[XIT], xch[XIT];
    int ypp[YIT], ysrc[YIT], ych[YIT];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int e = tid + 256 * it;
        xpp[it] = e / XG;
        const int g = e - xpp[it] * XG;
        const int code = (e < PK * XG) ? P.xgrp[xt * XG + g] : -1;
        xsrc[it] = code < 0 ? -1 : (code >> 16);
        xch[it] = code & 0xffff;
    }
#pragma unroll
    for (int it = 0; it < YIT; ++it) {
        const int e = tid + 256 * it;
        ypp[it] = e / YG;
        const int g = e - ypp[it] * YG;
        const int code = (e < PK * YG) ? P.ygrp[yt * YG + g] : -1;
        ysrc[it] = code < 0 ? -1 : (code >> 16);
        ych[it] = code & 0xffff;
    }

    // ---- per-element state: pre-offset source pointer, sc*C, and pixel coordinates advanced incrementally -----
    // source pixel of output pixel (b, y, x), tap (dy, dx):  const + sc * ((b*H + y) * W*sc + x)   (no division in the loop)
    const int s_begin = split * steps_per_split;
    int s_end = s_begin + steps_per_split;
    if (s_end > nsteps) s_end = nsteps;
    const int xsc = P.xs[0].scale, ysc = P.ys[0].scale;          // uniform per operand (checked on the host)
    const int xWs = W * xsc, yWs = W * ysc;
    const int RH = P.B * H;                                       // number of global rows
    const float *xptr[XIT], *yptr[YIT];
    int xscC[XIT], xr[XIT], xy[XIT], xx[XIT], yscC[YIT], yr[YIT], yy[YIT], yx[YIT];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int p = s_begin * PK + xpp[it];
        const int b = p / HW, rem = p - b * HW;
        xy[it] = rem / W;
        xx[it] = rem - xy[it] * W;
        xr[it] = b * H + xy[it];
        xptr[it] = nullptr;
        xscC[it] = 0;
        if (xsrc[it] >= 0) {
            const rnh_src_t &S = P.xs[xsrc[it]];
            const long cp = ((long)S.img_off * H * xsc + S.sub_y) * xWs + S.sub_x + (long)(dy * xWs + dx) * xsc;
            xptr[it] = S.ptr + cp * S.C + S.c0 + xch[it];
            xscC[it] = xsc * S.C;
        }
    }
#pragma unroll
    for (int it = 0; it < YIT; ++it) {
        const int p = s_begin * PK + ypp[it];
        const int b = p / HW, rem = p - b * HW;
        yy[it] = rem / W;
        yx[it] = rem - yy[it] * W;
        yr[it] = b * H + yy[it];
        yptr[it] = nullptr;
        yscC[it] = 0;
        if (ysrc[it] >= 0) {
            const rnh_src_t &S = P.ys[ysrc[it]];
            const long cp = ((long)S.img_off * H * ysc + S.sub_y) * yWs + S.sub_x;
            yptr[it] = S.ptr + cp * S.C + S.c0 + ych[it];
            yscC[it] = ysc * S.C;
        }
    }

    float4 rx[XIT], ry[YIT];
    auto load_stage = [&]() {          // loads the step the coordinates point at, then advances them by PK pixels
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xptr[it] && xr[it] < RH && (unsigned)(xy[it] + dy) < (unsigned)H && (unsigned)(xx[it] + dx) < (unsigned)W)
                v = rnh_ld4(xptr[it] + (long)(xr[it] * xWs + xx[it]) * xscC[it]);
            rx[it] = v;
            xx[it] += PK;
            while (xx[it] >= W) { xx[it] -= W; ++xr[it]; if (++xy[it] == H) xy[it] = 0; }
        }
#pragma unroll
        for (int it = 0; it < YIT; ++it) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yptr[it] && yr[it] < RH) v = rnh_ld4(yptr[it] + (long)(yr[it] * yWs + yx[it]) * yscC[it]);
            ry[it] = v;
            yx[it] += PK;
            while (yx[it] >= W) { yx[it] -= W; ++yr[it]; }
        }
    };
    auto store_stage = [&](int buf) {
        float *Xs = lds + buf * STAGE;
        float *Ys = Xs + PK * BX;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int e = tid + 256 * it;
            if ((PK * XG) % 256 == 0 || e < PK * XG) rnh_st4(Xs + e * 4, rx[it]);
        }
#pragma unroll
        for (int it = 0; it < YIT; ++it) {
            const int e = tid + 256 * it;
            if ((PK * YG) % 256 == 0 || e < PK * YG) rnh_st4(Ys + e * 4, ry[it]);
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool do_bias = P.bslab && tap == 0 && xt == 0 && tid < BY;
    float bsum = 0.f;

    auto compute = [&](int buf) {
        const float *Xs = lds + buf * STAGE;
        const float *Ys = Xs + PK * BX;
#pragma unroll
        for (int k = 0; k < PK / 2; ++k) {
            float a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = Xs[(2 * k + kh) * BX + (wm * MI + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = Ys[(2 * k + kh) * BY + (wn * NI + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (do_bias) {
#pragma unroll
            for (int k = 0; k < PK; ++k) bsum += Ys[k * BY + tid];
        }
    };

    if (s_begin < s_end) {
        load_stage();
        store_stage(0);
        __syncthreads();
        for (int st = s_begin; st < s_end; ++st) {
            const bool more = st + 1 < s_end;
            if (more) load_stage();
            compute((st - s_begin) & 1);
            if (more) store_stage((st - s_begin + 1) & 1);
            __syncthreads();
        }
    }

    // slab[((split*ntaps + tap) * xcols_pad + row) * ycols_pad + col]
    float *out = P.slab + ((long)(split * P.ntaps + tap) * P.xcols_pad + xt * BX) * P.ycols_pad + yt * BY;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int col = (wn * NI + j) * 32 + l31;
                out[(long)row * P.ycols_pad + col] = acc[i][j][r];
            }
    if (do_bias) P.bslab[(long)split * P.ycols_pad + yt * BY + tid] = bsum;
}

__global__ void wgrad_reduce_kernel(const float *slab, const float *bslab, int nsplit, int ntaps, int XP, int YP,
                                    const int *rowmap, const int *colmap, int Cin, float *dw, float *db, int accumulate) {
    const long total = (long)ntaps * XP * YP;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int j = (int)(e % YP);
        const long r = e / YP;
        const int i = (int)(r % XP), tap = (int)(r / XP);
        const int ci = rowmap[i], co = colmap[j];
        if (ci < 0 || co < 0) continue;
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += slab[(long)sp * total + e];
        const long o = ((long)co * Cin + ci) * ntaps + tap;
        dw[o] = accumulate ? dw[o] + s : s;
    }
    if (bslab && db) {
        for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < YP; j += (long)gridDim.x * blockDim.x) {
            const int co = colmap[j];
            if (co < 0) continue;
            float s = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) s += bslab[(long)sp * YP + j];
            db[co] = accumulate ? db[co] + s : s;
        }
    }
}

template <int WM, int WN, int MI, int NI>
int launch_wgrad(const rnh_wgrad_args_t &a, hipStream_t st) {
    constexpr int BX = WM * MI * 32, BY = WN * NI * 32;
    if (a.xcols_pad % BX || a.ycols_pad % BY) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wgrad: padded sizes do not match the tile");
    const long M = (long)a.B * a.H * a.W;
    const int nsteps = (int)((M + PK - 1) / PK);
    const int sps = (nsteps + a.nsplit - 1) / a.nsplit;
    const dim3 grid(a.nsplit, (a.xcols_pad / BX) * (a.ycols_pad / BY), a.ntaps), block(256);
    hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, MI, NI>), grid, block, 0, st, a, a.xcols_pad / BX, a.ycols_pad / BY, sps,
                       nsteps);
    RNH_CHECK_LAUNCH("rnh_conv_wgrad");
    return 0;
}

}  // namespace

extern "C" int rnh_conv_wgrad(const rnh_wgrad_args_t *args, void *stream) {
    if (!args) RNH_FAIL(RNH_E_ARG, "rnh_conv_wgrad: null args");
    const rnh_wgrad_args_t &a = *args;
    if (a.nxs < 1 || a.nxs > RNH_MAX_SRC || a.nys < 1 || a.nys > RNH_MAX_SRC) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wgrad: source count");
    for (int i = 0; i < a.nxs; ++i)
        if (int e = rnh_check_src(a.xs[i], "rnh_conv_wgrad(x)")) return e;
    for (int i = 0; i < a.nys; ++i)
        if (int e = rnh_check_src(a.ys[i], "rnh_conv_wgrad(dy)")) return e;
    for (int i = 0; i < a.nxs; ++i)
        if (a.xs[i].ptr2 || a.xs[i].scale != a.xs[0].scale) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wgrad: x sources must share one scale and have no ptr2");
    for (int i = 0; i < a.nys; ++i)
        if (a.ys[i].ptr2 || a.ys[i].scale != a.ys[0].scale) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wgrad: dy sources must share one scale and have no ptr2");
    if (a.B < 1 || a.H < 1 || a.W < 1 || (a.ntaps != 9 && a.ntaps != 1)) RNH_FAIL(RNH_E_ARG, "rnh_conv_wgrad: bad geometry");
    if ((long)a.B * a.H * a.W >= (1L << 31) / 16) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wgrad: too many pixels");
    if (!a.slab || !a.xgrp || !a.ygrp || a.nsplit < 1 || a.nsplit > 65535) RNH_FAIL(RNH_E_ARG, "rnh_conv_wgrad: workspace / maps / nsplit");
    hipStream_t st = (hipStream_t)stream;
    switch (a.tile) {
        case RNH_TILE_128x128: return launch_wgrad<2, 2, 2, 2>(a, st);
        case RNH_TILE_128x160: return launch_wgrad<4, 1, 1, 5>(a, st);
        case RNH_TILE_256x64:  return launch_wgrad<4, 1, 2, 2>(a, st);
        case RNH_TILE_64x128:  return launch_wgrad<2, 2, 1, 2>(a, st);
        case RNH_TILE_64x256:  return launch_wgrad<1, 4, 2, 2>(a, st);
        default: RNH_FAIL(RNH_E_RANGE, "rnh_conv_wgrad: unsupported tile %d", a.tile);
    }
}

extern "C" int rnh_wgrad_reduce(const float *slab, const float *bslab, int nsplit, int ntaps, int xcols_pad, int ycols_pad,
                                const int32_t *rowmap, const int32_t *colmap, int Cin, float *dw, float *db, int accumulate,
                                void *stream) {
    if (!slab || !rowmap || !colmap || !dw || nsplit < 1 || ntaps < 1 || xcols_pad < 1 || ycols_pad < 1 || Cin < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_wgrad_reduce: bad arguments");
    const long total = (long)ntaps * xcols_pad * ycols_pad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slab, bslab, nsplit, ntaps,
                       xcols_pad, ycols_pad, rowmap, colmap, Cin, dw, db, accumulate);
    RNH_CHECK_LAUNCH("rnh_wgrad_reduce");
    return 0;
}
