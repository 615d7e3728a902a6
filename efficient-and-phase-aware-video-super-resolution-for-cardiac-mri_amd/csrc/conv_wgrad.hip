// Weight gradient of the 3x3 / 1x1 convolutions on the fp32 matrix cores, without LDS and without barriers.
//
//   dW[tap][i][j] = sum_p X[p + tap][i] * dY[p][j]        i: forward input channel, j: output channel
//
// A GEMM whose contraction index is the pixel.  One WAVE owns one work item = (pixel range, tap, 32*MI rows,
// 32*NI columns): v_mfma_f32_32x32x2_f32 contracts two pixels per instruction, lane-half kh holding pixel
// 2s + kh.  The operands come straight from global memory in MFMA layout: lane l of a half loads the MI
// consecutive channels MI*l .. MI*l+MI-1 of its pixel as ONE 4*MI-byte load (the 32 lanes of a half read
// 128*MI contiguous bytes), and register e of that load is the A operand of row tile e - i.e. row i of row tile
// e is channel MI*i + e; the permutation is undone when the tile is written.  dY likewise with NI.  So one
// 16-byte and one 8-byte load feed 8 MFMAs (512 cycles), D pixel pairs are prefetched while the previous D
// are multiplied, and lanes whose pixel is outside the image (tap shift), outside the work item's range or in a
// padding channel read a zero page instead of branching.
// Partial tiles of the pixel ranges go to a slab; rnh_wgrad_reduce sums them in a fixed order (bitwise
// reproducible, no float atomics).  Column sums of dY (= the bias gradient) fall out of the B operands of the
// tap-0 / row-tile-0 work items.
#include <type_traits>
#include "rnh_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<2> { typedef f32x2 T; };
template <> struct Vec<4> { typedef f32x4 T; };

template <int N>
__device__ __forceinline__ void unpack(const typename Vec<N>::T &v, float (&o)[N]) {
    if constexpr (N == 1) o[0] = v;
    if constexpr (N == 2) { o[0] = v[0]; o[1] = v[1]; }
    if constexpr (N == 4) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
}

// Loads hidden from hipcc's scheduler (ROWS path).  hipcc sinks ordinary prefetch loads below the MFMAs that still
// read the previous contents of their registers, which leaves them a quarter of a set of lead time; volatile asm
// keeps program order, so the next set is in flight during the whole current one.  The matching wait_set() names
// every destination register "+v": no consumer can be scheduled above it, and it waits with a COUNTED vmcnt that
// leaves exactly the newer set's loads in flight.
template <int N>
__device__ __forceinline__ void gload(typename Vec<N>::T &dst, const float *p) {
    if constexpr (N == 1) asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
    if constexpr (N == 2) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
    if constexpr (N == 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}

// ROWS: W % (2*D) == 0 and chunk % W == 0 (host-checked).  Then every set of D pixel pairs lies inside one image
// row, the pixel coordinates are wave-uniform (the lane-half's +1 pixel is folded into the lane's base pointer) and
// live in scalar registers, and the D loads of a set differ by a constant stride: almost no vector ALU work is left
// beside the MFMAs.  Otherwise (odd / tiny images) the coordinates are tracked per lane.
template <int MI, int NI, int D, bool ROWS>
__global__ void __launch_bounds__(256, (MI * NI * 16 + 4 * D * (MI + NI) > 200 ? 1 : 2)) conv_wgrad_kernel(const rnh_wgrad_args_t P, const int RT, const int CT,
                                                           const int chunk /* pixels per split, multiple of 4*D */) {
    typedef typename Vec<MI>::T VA;
    typedef typename Vec<NI>::T VB;
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    // Work items of one pixel range (ntaps * RT * CT of them) read the same X / dY pixels, so a range should live on
    // ONE XCD (workgroups are dealt round-robin over the 8 XCDs: workgroup b and b + 8 share an L2) and stream from
    // HBM once.  Each XCD therefore owns nsplit/8 whole ranges, whose items are packed four to a workgroup without
    // regard to range boundaries (at most 3 idle waves per XCD).
    const int ips = P.ntaps * RT * CT;
    int split, item;
    if ((P.nsplit & 7) == 0) {
        const int xcd = blockIdx.x & 7, spx = P.nsplit >> 3;
        const int li = (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
        if (li >= spx * ips) return;                  // whole wave
        split = xcd * spx + li / ips;
        item = li % ips;
    } else {
        const int li = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (li >= P.nsplit * ips) return;
        split = li / ips;
        item = li % ips;
    }
    const int ct = item % CT;
    int r_ = item / CT;
    const int rt = r_ % RT;
    const int tap = r_ / RT;
    const int H = P.H, W = P.W, HW = H * W;
    const int Mtot = P.B * HW;
    int dy = 0, dx = 0;
    if (P.ntaps == 9) {
        dy = tap / 3 - 1;
        dx = tap - (dy + 1) * 3 - 1;
    }
    const int pb = split * chunk;
    int pe = pb + chunk;
    if (pe > Mtot) pe = Mtot;

    // ---- per-lane operand descriptors ---------------------------------------------------------------------
    // source pixel of output pixel (b, y, x) and tap (dy, dx):  const + sc * ((b*H + y) * W*sc + x)
    const float *zp = P.zero_page;
    const int xsc = P.xs[0].scale, ysc = P.ys[0].scale;           // uniform per operand (checked on the host)
    const int xWs = W * xsc * xsc, yWs = W * ysc * ysc;             // pixel-index stride of one output row
    const float *xbase = nullptr, *ybase = nullptr;
    long xC = 0, yC = 0;
    {
        const int code = P.xgrp[rt * 32 + l31];
        if (code >= 0) {
            const rnh_src_t &S = P.xs[code >> 16];
            const long cp = ((long)S.img_off * H * xsc + S.sub_y) * (W * xsc) + S.sub_x + (long)(dy * W * xsc + dx) * xsc;
            xbase = S.ptr + cp * S.C + S.c0 + (code & 0xffff);
            xC = S.C;
        }
    }
    {
        const int code = P.ygrp[ct * 32 + l31];
        if (code >= 0) {
            const rnh_src_t &S = P.ys[code >> 16];
            const long cp = ((long)S.img_off * H * ysc + S.sub_y) * (W * ysc) + S.sub_x;
            ybase = S.ptr + cp * S.C + S.c0 + (code & 0xffff);
            yC = S.C;
        }
    }
    struct Set {
        VA a[D];
        VB b[D];
    };
    // generic path state: coordinates of this lane-half's next pixel p = pb + kh, advanced by 2 per pixel pair
    int p = pb + kh;
    int cy, cx, cr;
    {
        const int b = p / HW, rem = p - b * HW;
        cy = rem / W;
        cx = rem - cy * W;
        cr = b * H + cy;
    }
    // ROWS path state (wave-uniform): global row, row within the image, first pixel of the next set
    int ur = pb / W, uy = ur % H, ux0 = 0;
    const int rows_end = pe / W;                                    // pe is a multiple of W in ROWS mode
    const float *xlane = xbase ? xbase + (long)kh * xsc * xC : nullptr;
    const float *ylane = ybase ? ybase + (long)kh * ysc * yC : nullptr;

    auto load_set = [&](Set &F) {
        if constexpr (ROWS) {
            const bool rowin = ur < rows_end;
            const bool xrow = rowin && xlane && (unsigned)(uy + dy) < (unsigned)H;
            const float *xb = xrow ? xlane + ((long)ur * xWs + ux0 * xsc) * xC : zp;
            const long xst = xrow ? 2 * xsc * xC : 0;
            const bool yrow = rowin && ylane;
            const float *yb = yrow ? ylane + ((long)ur * yWs + ux0 * ysc) * yC : zp;
            const long yst = yrow ? 2 * ysc * yC : 0;
            const bool cut_first = dx < 0 && ux0 == 0 && kh == 0;              // x - 1 < 0
            const bool cut_last = dx > 0 && ux0 + 2 * D == W && kh == 1;       // x + 1 >= W
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float *pa = xb + d * xst;
                if (d == 0) pa = cut_first ? zp : pa;
                if (d == D - 1) pa = cut_last ? zp : pa;
                gload<MI>(F.a[d], pa);
                gload<NI>(F.b[d], yb + d * yst);
            }
            ux0 += 2 * D;                                   // branch-free row advance
            const int wrap = ux0 == W;
            ux0 = wrap ? 0 : ux0;
            ur += wrap;
            uy += wrap;
            uy = uy == H ? 0 : uy;
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const bool in = p < pe;
                const bool xv = in && xbase && (unsigned)(cy + dy) < (unsigned)H && (unsigned)(cx + dx) < (unsigned)W;
                const float *pa = xv ? xbase + (long)(cr * xWs + cx * xsc) * xC : zp;
                F.a[d] = *reinterpret_cast<const VA *>(pa);
                const bool yv = in && ybase;
                const float *pb_ = yv ? ybase + (long)(cr * yWs + cx * ysc) * yC : zp;
                F.b[d] = *reinterpret_cast<const VB *>(pb_);
                p += 2;
                cx += 2;
                while (cx >= W) {
                    cx -= W;
                    ++cr;
                    if (++cy == H) cy = 0;
                }
            }
        }
    };

    // counted wait for the loads of set F: at most KEEP younger loads may still be in flight afterwards
    auto wait_set = [&](Set &F, auto keep) {
        constexpr int KEEP = decltype(keep)::value;
        static_assert(D == 4 || D == 8, "wait_set is written out for D = 4 and 8");
        (void)F;
        if constexpr (D == 4)
            asm volatile("s_waitcnt vmcnt(%c8)"
                         : "+v"(F.a[0]), "+v"(F.a[1]), "+v"(F.a[2]), "+v"(F.a[3]), "+v"(F.b[0]), "+v"(F.b[1]), "+v"(F.b[2]), "+v"(F.b[3])
                         : "i"(KEEP));
        else
            asm volatile("s_waitcnt vmcnt(%c16)"
                         : "+v"(F.a[0]), "+v"(F.a[1]), "+v"(F.a[2]), "+v"(F.a[3]), "+v"(F.a[4]), "+v"(F.a[5]), "+v"(F.a[6]), "+v"(F.a[7]),
                           "+v"(F.b[0]), "+v"(F.b[1]), "+v"(F.b[2]), "+v"(F.b[3]), "+v"(F.b[4]), "+v"(F.b[5]), "+v"(F.b[6]), "+v"(F.b[7])
                         : "i"(KEEP));
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool do_bias = P.bslab && tap == 0 && rt == 0;
    float bs[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) bs[j] = 0.f;

    auto compute_set = [&](const Set &F) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float a[MI], b[NI];
            unpack<MI>(F.a[d], a);
            unpack<NI>(F.b[d], b);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < NI; ++j) bs[j] += b[j];
            }
        }
    };

    // ---- main loop: two register sets, D pixel pairs each; loads past the range read the zero page ---------
    {
        Set F0, F1;
        const int nsets2 = (chunk / (2 * D) + 1) / 2;            // pairs of sets covering the whole chunk
        load_set(F0);
        if constexpr (ROWS) {
            // One wave's instruction stream is [2 loads + their few address instructions of the NEXT set's pair d]
            // [8 MFMAs of the CURRENT set's pair d], D times per set: the vector instructions issue in the shadow of
            // the wave's own MFMAs (64 cycles each).  Measured the other way round - a load-only phase next to a
            // partner wave that streams MFMAs - every vector instruction of the loader waited one MFMA slot (2800
            // cycles for ~45 instructions against 2040 for the partner's 32 MFMAs; 510 cycles with the SIMD to
            // itself): the matrix pipe idled a quarter of the time.  Each pair waits with vmcnt(2*D): exactly one
            // whole set of younger loads stays in flight, so every load has a full set of MFMAs as lead time.
            auto step = [&](Set &FL, Set &FC) {
                const bool rowin = ur < rows_end;
                const bool xrow = rowin && xlane && (unsigned)(uy + dy) < (unsigned)H;
                const float *xb = xrow ? xlane + ((long)ur * xWs + ux0 * xsc) * xC : zp;
                const long xst = xrow ? 2 * xsc * xC : 0;
                const bool yrow = rowin && ylane;
                const float *yb = yrow ? ylane + ((long)ur * yWs + ux0 * ysc) * yC : zp;
                const long yst = yrow ? 2 * ysc * yC : 0;
                const bool cut_first = dx < 0 && ux0 == 0 && kh == 0;              // x - 1 < 0
                const bool cut_last = dx > 0 && ux0 + 2 * D == W && kh == 1;       // x + 1 >= W
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float *pa = xb + d * xst;
                    if (d == 0) pa = cut_first ? zp : pa;
                    if (d == D - 1) pa = cut_last ? zp : pa;
                    gload<MI>(FL.a[d], pa);
                    gload<NI>(FL.b[d], yb + d * yst);
                    asm volatile("s_waitcnt vmcnt(%c2)" : "+v"(FC.a[d]), "+v"(FC.b[d]) : "i"(2 * D));
                    float a[MI], b[NI];
                    unpack<MI>(FC.a[d], a);
                    unpack<NI>(FC.b[d], b);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                    if (do_bias) {
#pragma unroll
                        for (int j = 0; j < NI; ++j) bs[j] += b[j];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                ux0 += 2 * D;                                   // branch-free row advance
                const int wrap = ux0 == W;
                ux0 = wrap ? 0 : ux0;
                ur += wrap;
                uy += wrap;
                uy = uy == H ? 0 : uy;
            };
            for (int it = 0; it < nsets2; ++it) {
                step(F1, F0);
                step(F0, F1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the last, unused prefetch
        } else {
            for (int it = 0; it < nsets2; ++it) {
                load_set(F1);
                compute_set(F0);
                load_set(F0);
                compute_set(F1);
            }
        }
    }

    // ---- write the partial tile: row = MI*i + em, column = NI*j + en -------------------------------------------
    float *out = P.slab + ((long)(split * P.ntaps + tap) * P.xcols_pad + rt * 32 * MI) * P.ycols_pad + ct * 32 * NI;
#pragma unroll
    for (int em = 0; em < MI; ++em)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * kh;
            float *o = out + (long)(MI * i + em) * P.ycols_pad + NI * l31;
#pragma unroll
            for (int en = 0; en < NI; ++en) o[en] = acc[em][en][r];
        }
    if (do_bias) {
#pragma unroll
        for (int en = 0; en < NI; ++en) {
            const float v = bs[en] + __shfl_xor(bs[en], 32, 64);
            if (kh == 0) P.bslab[(long)split * P.ycols_pad + ct * 32 * NI + NI * l31 + en] = v;
        }
    }
}

// One thread sums four consecutive columns of one (tap, row) over the splits: 16-byte loads, eight of them in flight, added in split
// order (the same sums, element by element, as a plain loop over the splits; the scalar version - one column per thread, one 4-byte load at
// a time - read the ConvLSTM weight gradient's 75 MB of partial sums at 0.6 TB/s: 120 us, 3.6 ms per bf16 step).  YP % 4 == 0.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *__restrict__ slab, const float *__restrict__ bslab, int nsplit, int ntaps,
                                                           int XP, int YP, const int *__restrict__ rowmap, const int *__restrict__ colmap, int Cin,
                                                           float *dw, float *db, int accumulate) {
    const long total = (long)ntaps * XP * YP, total4 = total >> 2;
    for (long e4 = (long)blockIdx.x * blockDim.x + threadIdx.x; e4 < total4; e4 += (long)gridDim.x * blockDim.x) {
        const long e = e4 << 2;
        const int j = (int)(e % YP);
        const long r = e / YP;
        const int i = (int)(r % XP), tap = (int)(r / XP);
        const int ci = rowmap[i];
        const int co[4] = {colmap[j], colmap[j + 1], colmap[j + 2], colmap[j + 3]};
        if (ci < 0 || (co[0] < 0 && co[1] < 0 && co[2] < 0 && co[3] < 0)) continue;
        const float *p = slab + e;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int sp = 0;
        for (; sp + 8 <= nsplit; sp += 8) {
            float4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = rnh_ld4(p + (long)(sp + q) * total);
#pragma unroll
            for (int q = 0; q < 8; ++q) s.x += v[q].x, s.y += v[q].y, s.z += v[q].z, s.w += v[q].w;
        }
        for (; sp < nsplit; ++sp) {
            const float4 v = rnh_ld4(p + (long)sp * total);
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
        const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (co[q] < 0) continue;
            const long o = ((long)co[q] * Cin + ci) * ntaps + tap;
            dw[o] = accumulate ? dw[o] + sv[q] : sv[q];
        }
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

template <int MI, int NI, int D>
int launch_wgrad(const rnh_wgrad_args_t &a, hipStream_t st) {
    if (a.xcols_pad % (32 * MI) || a.ycols_pad % (32 * NI)) RNH_FAIL(RNH_E_RANGE, "rnh_conv_wgrad: padded sizes do not match the tile");
    const long M = (long)a.B * a.H * a.W;
    const bool rows = a.W % (2 * D) == 0;
    const int q = rows ? 2 * a.W : 4 * D;           // rows: whole image rows per range (and 4*D divides 2*W)
    long chunk = (M + a.nsplit - 1) / a.nsplit;
    chunk = (chunk + q - 1) / q * q;
    const int RT = a.xcols_pad / (32 * MI), CT = a.ycols_pad / (32 * NI);
    const long ips = (long)a.ntaps * RT * CT;                      // work items (waves) per pixel range
    const long nblk = (a.nsplit & 7) == 0 ? 8 * ((a.nsplit / 8 * ips + 3) / 4) : (a.nsplit * ips + 3) / 4;
    const dim3 grid((unsigned)nblk), block(256);
    if (rows) hipLaunchKernelGGL((conv_wgrad_kernel<MI, NI, D, true>), grid, block, 0, st, a, RT, CT, (int)chunk);
    else hipLaunchKernelGGL((conv_wgrad_kernel<MI, NI, D, false>), grid, block, 0, st, a, RT, CT, (int)chunk);
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
    if (!a.slab || !a.xgrp || !a.ygrp || !a.zero_page || a.nsplit < 1 || a.nsplit > 65535)
        RNH_FAIL(RNH_E_ARG, "rnh_conv_wgrad: workspace / maps / zero page / nsplit");
    hipStream_t st = (hipStream_t)stream;
    switch (a.tile) {
        case RNH_WTILE_128x64: return launch_wgrad<4, 2, 4>(a, st);
        case RNH_WTILE_64x128: return launch_wgrad<2, 4, 4>(a, st);
        case RNH_WTILE_64x64:  return launch_wgrad<2, 2, 8>(a, st);
        case RNH_WTILE_128x32: return launch_wgrad<4, 1, 8>(a, st);
        case RNH_WTILE_128x128: return launch_wgrad<4, 4, 8>(a, st);
        case 0x824: return launch_wgrad<2, 4, 8>(a, st);          // experiment
        default: RNH_FAIL(RNH_E_RANGE, "rnh_conv_wgrad: unsupported tile %d", a.tile);
    }
}

extern "C" int rnh_wgrad_reduce(const float *slab, const float *bslab, int nsplit, int ntaps, int xcols_pad, int ycols_pad,
                                const int32_t *rowmap, const int32_t *colmap, int Cin, float *dw, float *db, int accumulate,
                                void *stream) {
    if (!slab || !rowmap || !colmap || !dw || nsplit < 1 || ntaps < 1 || xcols_pad < 1 || ycols_pad < 1 || Cin < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_wgrad_reduce: bad arguments");
    if (ycols_pad & 3) RNH_FAIL(RNH_E_ALIGN, "rnh_wgrad_reduce: ycols_pad must be a multiple of 4");
    const long total = (long)ntaps * xcols_pad * ycols_pad;
    int blocks = (int)((total / 4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slab, bslab, nsplit, ntaps,
                       xcols_pad, ycols_pad, rowmap, colmap, Cin, dw, db, accumulate);
    RNH_CHECK_LAUNCH("rnh_wgrad_reduce");
    return 0;
}
