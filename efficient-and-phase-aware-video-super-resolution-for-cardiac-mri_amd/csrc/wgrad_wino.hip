// Weight gradient of a 3x3 convolution in Winograd form F(3x3, 2x2) on fp32 MFMA - rnh_wino_wgrad.
//
//   dg = G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G        d: 4x4 input patch, dY: 2x2 output-gradient tile
//
// 16 GEMMs dU_xi[ci][co] = sum_tiles V_xi[tile][ci] * Z_xi[tile][co] (contraction over tiles, two per
// v_mfma_f32_32x32x2_f32) instead of 9 contractions over pixels: 4/9 of the multiplications of conv_wgrad.hip.  One wave
// = 32 input channels x 32 output channels x all 16 xi (256 accumulators) over a range of tile rows; a lane is one input
// channel (A operand) and one output channel (B operand) of the two tiles of a step and computes both transforms itself
// (its 16 + 4 operand loads are 128-byte rows of 32 consecutive channels).
//
// No border cases in the kernel: the inputs are first gathered into one zero-padded tensor xp (B, H+2, W+2, Cx)
// (rnh_wino_wgrad does that; it also removes the multi-source handling from the inner loop), so every load is in
// range - which the hand-counted waits need (a buffer load with all lanes out of range returns ahead of older loads).
// The loop body is branch-free, every register set has one asynchronous definition site and nothing is prefetched past
// the end (see conv_wino.hip for what goes wrong otherwise; tests/test_isa_guards.py checks the generated code).
// Partial sums per tile-row range go to a slab; rnh_wino_wgrad's reduction sums them in fixed order and applies G^T . G.
#include "rnh_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// one workgroup per padded image row: zero border, interior gathered from the sources (16 bytes per thread)
__global__ void __launch_bounds__(256) wino_pad_kernel(const rnh_wgrad_args_t P, float *xp, int Cx) {
    const int H = P.H, W = P.W, Hp = H + 2, Wp = W + 2, C4 = Cx >> 2;
    const int row = blockIdx.x, b = row / Hp, y = row - b * Hp - 1;
    float *dst = xp + (long)row * Wp * Cx;
    const bool inside = (unsigned)y < (unsigned)H;
    for (int e = threadIdx.x; e < Wp * C4; e += 256) {
        const int xq = e / C4, c = (e - xq * C4) * 4, x = xq - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inside && (unsigned)x < (unsigned)W) {
            int cc = c, s = 0;
            while (cc >= P.xs[s].nch) cc -= P.xs[s++].nch;
            const rnh_src_t &S = P.xs[s];
            v = rnh_ld4(S.ptr + ((((long)b + S.img_off) * H + y) * W + x) * S.C + S.c0 + cc);
        }
        rnh_st4(dst + (long)e * 4, v);
    }
}

__device__ __forceinline__ i32x4 wdesc64(const float *p) {
    const unsigned long long u = (unsigned long long)p;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    d[1] = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
    d[2] = 0x7fffffff;
    d[3] = 0x00020000;
    return d;
}

struct WSet {
    float xa[16], xb[16];     // 4x4 patches of the lane's input channel, tiles a and b
    float ya[4], yb[4];       // 2x2 output-gradient tiles of the lane's output channel
};

__global__ void __launch_bounds__(256, 1) wino_wgrad_kernel(const rnh_wgrad_args_t P, const float *xp, const int Cx, const int Cy,
                                                            const int KS, const int rows_per) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const int RT = Cx >> 5, CT = Cy >> 5;
    // (the wave index is wave-uniform, but only a readfirstlane tells the compiler so: everything derived from it -
    // tile-row ranges, descriptors - must live in scalar registers)
    // (contiguous block lists per XCD: the items of one tile-row range share their operands through one L2, see the LDS variant)
    const int item = rnh_xcd_remap(blockIdx.x, gridDim.x) * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (item >= KS * RT * CT) return;                         // whole waves only; no barrier in this kernel
    const int ks = item / (RT * CT), rc = item - ks * RT * CT, rt = rc / CT, ct = rc - rt * CT;
    const int H = P.H, W = P.W, TY = H >> 1, G = W >> 3, Hp = H + 2, Wp = W + 2;
    const int rows_total = P.B * TY;
    const int r0 = ks * rows_per, r1 = min(rows_total, r0 + rows_per);

    // ---- output-gradient source of this column tile -----------------------------------------------------------
    int ysrc = 0, cy0 = ct * 32;
    while (cy0 >= P.ys[ysrc].nch) cy0 -= P.ys[ysrc++].nch;
    const rnh_src_t &Y = P.ys[ysrc];
    const int sc = Y.scale, Hs = H * sc, Ws = W * sc;

    // per-lane constant offsets inside one group of 4 tiles (8 pixels): tile a = 4g + kh, tile b = 4g + 2 + kh
    int vx[2][4], vy[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) vx[t][j] = ((2 * (kh + 2 * t) + j) * Cx + rt * 32 + l31) * 4;
#pragma unroll
        for (int b = 0; b < 2; ++b) vy[t][b] = ((2 * (kh + 2 * t) + b) * sc * Y.C + l31) * 4;
    }
    // loader state: tile row r (= image * TY + ty) and group g; descriptors are re-based per tile row
    int lr = r0, lg = 0;
    i32x4 xdesc, ydesc;
    auto rebase = [&]() {
        const int img = lr / TY, ty = lr - img * TY;
        xdesc = wdesc64(xp + ((long)img * Hp + 2 * ty) * Wp * Cx);
        ydesc = wdesc64(Y.ptr + Y.c0 + cy0 + ((((long)img + Y.img_off) * Hs + (long)2 * ty * sc + Y.sub_y) * Ws + Y.sub_x) * Y.C);
    };
    rebase();
    const int xrow = Wp * Cx * 4, xgrp = 8 * Cx * 4, yrow = sc * Ws * Y.C * 4, ygrp = 8 * sc * Y.C * 4;

    auto ld8x = [&](float *da, float *db, int i, int soff) {      // patch row i of both tiles
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dword %0, %8, %16, %17 offen\n\t"
            "buffer_load_dword %1, %9, %16, %17 offen\n\t"
            "buffer_load_dword %2, %10, %16, %17 offen\n\t"
            "buffer_load_dword %3, %11, %16, %17 offen\n\t"
            "buffer_load_dword %4, %12, %16, %17 offen\n\t"
            "buffer_load_dword %5, %13, %16, %17 offen\n\t"
            "buffer_load_dword %6, %14, %16, %17 offen\n\t"
            "buffer_load_dword %7, %15, %16, %17 offen"
            : "=&v"(da[i * 4 + 0]), "=&v"(da[i * 4 + 1]), "=&v"(da[i * 4 + 2]), "=&v"(da[i * 4 + 3]), "=&v"(db[i * 4 + 0]),
              "=&v"(db[i * 4 + 1]), "=&v"(db[i * 4 + 2]), "=&v"(db[i * 4 + 3])
            : "v"(vx[0][0]), "v"(vx[0][1]), "v"(vx[0][2]), "v"(vx[0][3]), "v"(vx[1][0]), "v"(vx[1][1]), "v"(vx[1][2]), "v"(vx[1][3]),
              "s"(xdesc), "s"(soff)
            : "memory");
    };
    auto ld8y = [&](float *ya, float *yb, int s0, int s1) {       // both rows of both 2x2 tiles
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dword %0, %8, %12, %13 offen\n\t"
            "buffer_load_dword %1, %9, %12, %13 offen\n\t"
            "buffer_load_dword %2, %8, %12, %14 offen\n\t"
            "buffer_load_dword %3, %9, %12, %14 offen\n\t"
            "buffer_load_dword %4, %10, %12, %13 offen\n\t"
            "buffer_load_dword %5, %11, %12, %13 offen\n\t"
            "buffer_load_dword %6, %10, %12, %14 offen\n\t"
            "buffer_load_dword %7, %11, %12, %14 offen"
            : "=&v"(ya[0]), "=&v"(ya[1]), "=&v"(ya[2]), "=&v"(ya[3]), "=&v"(yb[0]), "=&v"(yb[1]), "=&v"(yb[2]), "=&v"(yb[3])
            : "v"(vy[0][0]), "v"(vy[0][1]), "v"(vy[1][0]), "v"(vy[1][1]), "s"(ydesc), "s"(s0), "s"(s1)
            : "memory");
    };
    auto issue = [&](WSet &S) {                                   // 40 loads of the loader's (row, group), then advance
        const int gx = __builtin_amdgcn_readfirstlane(lg * xgrp), gy = __builtin_amdgcn_readfirstlane(lg * ygrp);
#pragma unroll
        for (int i = 0; i < 4; ++i) ld8x(S.xa, S.xb, i, gx + i * xrow);
        ld8y(S.ya, S.yb, gy, gy + yrow);
        if (++lg == G) {
            lg = 0;
            if (++lr < r1) rebase();
        }
    };
    auto wait = [&](WSet &S, auto keep) {                          // the set's 40 loads have landed (in-order returns)
        asm volatile("s_waitcnt vmcnt(%c20)"
                     : "+v"(S.xa[0]), "+v"(S.xa[1]), "+v"(S.xa[2]), "+v"(S.xa[3]), "+v"(S.xa[4]), "+v"(S.xa[5]), "+v"(S.xa[6]), "+v"(S.xa[7]),
                       "+v"(S.xa[8]), "+v"(S.xa[9]), "+v"(S.xa[10]), "+v"(S.xa[11]), "+v"(S.xa[12]), "+v"(S.xa[13]), "+v"(S.xa[14]),
                       "+v"(S.xa[15]), "+v"(S.ya[0]), "+v"(S.ya[1]), "+v"(S.ya[2]), "+v"(S.ya[3])
                     : "i"(decltype(keep)::value));
        asm volatile(""
                     : "+v"(S.xb[0]), "+v"(S.xb[1]), "+v"(S.xb[2]), "+v"(S.xb[3]), "+v"(S.xb[4]), "+v"(S.xb[5]), "+v"(S.xb[6]), "+v"(S.xb[7]),
                       "+v"(S.xb[8]), "+v"(S.xb[9]), "+v"(S.xb[10]), "+v"(S.xb[11]), "+v"(S.xb[12]), "+v"(S.xb[13]), "+v"(S.xb[14]),
                       "+v"(S.xb[15]), "+v"(S.yb[0]), "+v"(S.yb[1]), "+v"(S.yb[2]), "+v"(S.yb[3]));
    };

    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[xi][v] = 0.f;
    float bsum = 0.f;

    auto tile_mfma = [&](const float *d, const float *y) {
        float tq[16], V[16], tz[8], Z[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                              // V = B^T d B
            tq[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
            tq[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
            tq[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
            tq[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            V[i * 4 + 0] = tq[i * 4 + 0] - tq[i * 4 + 2];
            V[i * 4 + 1] = tq[i * 4 + 1] + tq[i * 4 + 2];
            V[i * 4 + 2] = tq[i * 4 + 2] - tq[i * 4 + 1];
            V[i * 4 + 3] = tq[i * 4 + 1] - tq[i * 4 + 3];
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {                              // Z = A dY A^T,  A = [1 0; 1 1; 1 -1; 0 -1]
            tz[0 * 2 + b] = y[0 * 2 + b];
            tz[1 * 2 + b] = y[0 * 2 + b] + y[1 * 2 + b];
            tz[2 * 2 + b] = y[0 * 2 + b] - y[1 * 2 + b];
            tz[3 * 2 + b] = -y[1 * 2 + b];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Z[i * 4 + 0] = tz[i * 2 + 0];
            Z[i * 4 + 1] = tz[i * 2 + 0] + tz[i * 2 + 1];
            Z[i * 4 + 2] = tz[i * 2 + 0] - tz[i * 2 + 1];
            Z[i * 4 + 3] = -tz[i * 2 + 1];
        }
        bsum += tz[1 * 2 + 0] + tz[1 * 2 + 1];                     // (y0 + y2) + (y1 + y3): the column sums exist already
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[xi], Z[xi], acc[xi], 0, 0, 0);
    };
    auto compute = [&](const WSet &S) {
        tile_mfma(S.xa, S.ya);
        tile_mfma(S.xb, S.yb);
    };

    // ---- main loop: NIT = rows * G steps of 4 tiles each (NIT is even: the host requires an even G) ---------------
    // Two register sets: S1 takes the even steps, S0 the odd ones; the first compute(S0) runs on zeros.  Each set is
    // waited for at the END of the other set's MFMAs, so that nothing is in flight at the loop edge: hipcc may put
    // v_mov copies of loop-carried registers there, and a copy of a register whose load is still in flight reads stale
    // data (conv_wino.hip; tests/test_isa_guards.py checks the generated code).
    const int NIT = (r1 - r0) * G;
    using K0 = std::integral_constant<int, 0>;
    WSet S0, S1;
#pragma unroll
    for (int i = 0; i < 16; ++i) S0.xa[i] = S0.xb[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) S0.ya[i] = S0.yb[i] = 0.f;
    for (int it = 0; it < NIT; it += 2) {
        // (scheduling barriers: hipcc is free to move the MFMAs and the transform arithmetic across a volatile asm,
        // which would turn "wait at the end" into "wait right after the issue")
        issue(S1);                    // step it
        __builtin_amdgcn_sched_barrier(0);
        compute(S0);                  // step it - 1 (zeros the first time)
        __builtin_amdgcn_sched_barrier(0);
        wait(S1, K0());
        issue(S0);                    // step it + 1
        __builtin_amdgcn_sched_barrier(0);
        compute(S1);
        __builtin_amdgcn_sched_barrier(0);
        wait(S0, K0());
    }
    if (NIT > 0) compute(S0);

    // ---- partial sums to the slab: [ks][xi][row][col] ---------------------------------------------------------
    float *out = P.slab + (((long)ks * 16) * Cx + rt * 32) * Cy + ct * 32 + l31;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = (v & 3) + 8 * (v >> 2) + 4 * kh;
            out[((long)xi * Cx + row) * Cy] = acc[xi][v];
        }
    if (P.bslab && rt == 0) {
        const float b = bsum + __shfl_xor(bsum, 32, 64);
        if (kh == 0) P.bslab[(long)ks * Cy + ct * 32 + l31] = b;
    }
}

// ---- variant with the input transform shared through LDS -----------------------------------------------------------
// The four waves of a workgroup are four column tiles (ct) of the same (tile-row range, row tile): they need the SAME
// transformed input patches V = B^T d B.  Here each wave loads and transforms one of four consecutive groups (a "quad" =
// 16 tiles) and parks V in LDS; after a barrier every wave reads all four groups (four ds_read_b128 per tile; lane stride
// 20 floats: conflict-free) and contributes only its own output-gradient transform.  Per group and wave: 8 + 28 + 2
// (its quarter of the loads / transform arithmetic / LDS writes) + 8 LDS reads + 8 loads + 24 VALU instead of 32 loads +
// 112 + 24 VALU next to the 32 MFMAs.  Needs CT % 4 == 0 and G % 4 == 0 (W % 32 == 0); compiler-scheduled loads
// (raw buffer load builtins), phases pinned by scheduling barriers.
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wg_rsrc(const float *p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(u & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float wg_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

constexpr int WL_STRIDE = 36;                                   // floats per (group, lane) in LDS: 16 xi x (tile a, tile b) + 4 pad
constexpr int WL_BUF = 4 * 64 * WL_STRIDE;                      // floats per buffer (4 groups x 64 lanes)

__global__ void __launch_bounds__(256, 1) wino_wgrad_lds_kernel(const rnh_wgrad_args_t P, const float *xp, const int Cx, const int Cy,
                                                                const int KS, const int rows_per) {
    __shared__ __attribute__((aligned(16))) float sV[2 * WL_BUF];   // 72 KB
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int RT = Cx >> 5, CT = Cy >> 5;
    // The RT * CT / 4 workgroups of one tile-row range read the same input patches and output gradients: consecutive block
    // indices go round-robin over the 8 XCDs (8 L2s), so without the remap every L2 fetched every row range once - 5.7 GB
    // of HBM reads per ConvLSTM weight gradient for 1.4 GB of operands.  Contiguous block lists per XCD (rnh_xcd_remap) keep
    // the workgroups of a range on one L2.
    const int lb = rnh_xcd_remap(blockIdx.x, gridDim.x);
    const int item = lb * 4 + wave;                             // CT % 4 == 0: the four waves share (ks, rt)
    const int ks = item / (RT * CT), rc = item - ks * RT * CT, rt = rc / CT, ct = rc - rt * CT;
    const int H = P.H, W = P.W, TY = H >> 1, G = W >> 3, Q = G >> 2;
    const int rows_total = P.B * TY;
    const int r0 = ks * rows_per, r1 = min(rows_total, r0 + rows_per);

    int ysrc = 0, cy0 = ct * 32;
    while (cy0 >= P.ys[ysrc].nch) cy0 -= P.ys[ysrc++].nch;
    const rnh_src_t &Y = P.ys[ysrc];
    const int sc = Y.scale, Hs = H * sc, Ws = W * sc;
    // input source of this row tile (32 channels never straddle two sources: nch % 32 == 0)
    int xsrc = 0, cx0 = rt * 32;
    while (cx0 >= P.xs[xsrc].nch) cx0 -= P.xs[xsrc++].nch;
    const rnh_src_t &X = P.xs[xsrc];
    const int XC = X.C;
    // The patches come straight from the unpadded sources (no gathered copy): the descriptor base sits one pixel LEFT of
    // the image, so column j of tile t is at (2 (kh + 2t) + j) pixels; the one column left of the first group / right of
    // the last one gets offset 0xFFFFFFFF (hardware range check: zero, no access; the other lane half is in range, so
    // the load still goes to memory in order); rows above / below the image are clamped and zeroed after the load
    // (a load with ALL lanes out of range would return ahead of older loads).
    int vx[2][4], vy[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) vx[t][j] = ((2 * (kh + 2 * t) + j) * XC + l31) * 4;
#pragma unroll
        for (int b = 0; b < 2; ++b) vy[t][b] = ((2 * (kh + 2 * t) + b) * sc * Y.C + l31) * 4;
    }
    const int xrow = W * XC * 4, xgrp = 8 * XC * 4, yrow = sc * Ws * Y.C * 4, ygrp = 8 * sc * Y.C * 4;
    // Row pointers advance by constants (no multiplications in the loop): both tensors are dense.
    const long ximg = (long)H * W * XC, ystep = (long)2 * sc * Ws * Y.C;
    auto ximg_ptr = [&](int img) { return X.ptr + X.c0 + cx0 + ((long)img + X.img_off) * ximg - XC; };
    auto yrow_ptr = [&](int img, int ty) {
        return Y.ptr + Y.c0 + cy0 + ((((long)img + Y.img_off) * Hs + (long)2 * ty * sc + Y.sub_y) * Ws + Y.sub_x) * Y.C;
    };

    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[xi][v] = 0.f;
    float bsum = 0.f;

    // The output gradients of a lane's two tiles ride in the halves of packed registers (x = tile a, y = tile b): their
    // transform is an add / subtract network, one v_pk_add_f32 per pair (written as asm: hipcc scalarises packed adds; the
    // s_nop covers the VALU-write -> MFMA-read wait states, which hipcc does not add behind an asm statement).
    typedef float f32x2w __attribute__((ext_vector_type(2)));
    auto pk_add = [&](f32x2w p, f32x2w r) {
        f32x2w o;
        asm("v_pk_add_f32 %0, %1, %2\n\ts_nop 1" : "=v"(o) : "v"(p), "v"(r));
        return o;
    };
    auto pk_sub = [&](f32x2w p, f32x2w r) {
        f32x2w o;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 1" : "=v"(o) : "v"(p), "v"(r));
        return o;
    };
    // (results that only go to other VALU instructions or to LDS need no pad)
    auto pk_add0 = [&](f32x2w p, f32x2w r) {
        f32x2w o;
        asm("v_pk_add_f32 %0, %1, %2" : "=v"(o) : "v"(p), "v"(r));
        return o;
    };
    auto pk_sub0 = [&](f32x2w p, f32x2w r) {
        f32x2w o;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(p), "v"(r));
        return o;
    };
    // x patches of this wave's group (quad position q of row r): tiles a and b, 32 loads
    auto load_x = [&](f32x2w (&d)[16], const float *ximg_p, int ty, int q) {
        const __amdgpu_buffer_rsrc_t xd = wg_rsrc(ximg_p);
        const int gq = 4 * q + wave;
        const int gx = __builtin_amdgcn_readfirstlane(gq * xgrp);
        int vo[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) vo[t][j] = vx[t][j];
        if (gq == 0 && kh == 0) vo[0][0] = -1;                    // x = -1
        if (gq == G - 1 && kh == 1) vo[1][3] = -1;                // x = W
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yi = 2 * ty - 1 + i;
            const int yc = yi < 0 ? 0 : (yi >= H ? H - 1 : yi);
            const int so = __builtin_amdgcn_readfirstlane(gx + yc * xrow);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d[i * 4 + j].x = wg_ld(xd, vo[0][j], so);
                d[i * 4 + j].y = wg_ld(xd, vo[1][j], so);
            }
        }
        const bool top = ty == 0, bottom = ty == TY - 1;          // wave-uniform: selects
        const f32x2w zero = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d[0 * 4 + j] = top ? zero : d[0 * 4 + j];
            d[3 * 4 + j] = bottom ? zero : d[3 * 4 + j];
        }
    };
    // V = B^T d B of both tiles (packed) -> LDS buffer `buf`, slot of this wave's group: [group][lane][xi][tile]
    auto xform_store = [&](const f32x2w (&d)[16], int buf) {
        f32x2w tq[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tq[0 * 4 + j] = pk_sub0(d[0 * 4 + j], d[2 * 4 + j]);
            tq[1 * 4 + j] = pk_add0(d[1 * 4 + j], d[2 * 4 + j]);
            tq[2 * 4 + j] = pk_sub0(d[2 * 4 + j], d[1 * 4 + j]);
            tq[3 * 4 + j] = pk_sub0(d[1 * 4 + j], d[3 * 4 + j]);
        }
        float *o = sV + buf * WL_BUF + (wave * 64 + lane) * WL_STRIDE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2w v0 = pk_sub0(tq[i * 4 + 0], tq[i * 4 + 2]), v1 = pk_add0(tq[i * 4 + 1], tq[i * 4 + 2]);
            const f32x2w v2 = pk_sub0(tq[i * 4 + 2], tq[i * 4 + 1]), v3 = pk_sub0(tq[i * 4 + 1], tq[i * 4 + 3]);
            *reinterpret_cast<f32x4v *>(o + 8 * i) = f32x4v{v0.x, v0.y, v1.x, v1.y};
            *reinterpret_cast<f32x4v *>(o + 8 * i + 4) = f32x4v{v2.x, v2.y, v3.x, v3.y};
        }
    };
    auto read_v = [&](f32x4v (&V)[8], int buf, int j) {
        const float *o = sV + buf * WL_BUF + (j * 64 + lane) * WL_STRIDE;
#pragma unroll
        for (int i = 0; i < 8; ++i) V[i] = *reinterpret_cast<const f32x4v *>(o + 4 * i);
    };
    auto load_y = [&](f32x2w (&y)[4], __amdgpu_buffer_rsrc_t yd, int g) {
        const int gy = __builtin_amdgcn_readfirstlane(g * ygrp);
        y[0].x = wg_ld(yd, vy[0][0], gy);
        y[1].x = wg_ld(yd, vy[0][1], gy);
        y[2].x = wg_ld(yd, vy[0][0], gy + yrow);
        y[3].x = wg_ld(yd, vy[0][1], gy + yrow);
        y[0].y = wg_ld(yd, vy[1][0], gy);
        y[1].y = wg_ld(yd, vy[1][1], gy);
        y[2].y = wg_ld(yd, vy[1][0], gy + yrow);
        y[3].y = wg_ld(yd, vy[1][1], gy + yrow);
    };
    auto mfma_group = [&](const f32x4v (&V)[8], const f32x2w (&y)[4]) {
        f32x2w tz[8], Z[16];
#pragma unroll
        for (int b = 0; b < 2; ++b) {                              // Z' = A' dY A'^T,  A' = [1 0; 1 1; 1 -1; 0 1]
            tz[0 * 2 + b] = y[0 * 2 + b];
            tz[1 * 2 + b] = pk_add(y[0 * 2 + b], y[1 * 2 + b]);
            tz[2 * 2 + b] = pk_sub(y[0 * 2 + b], y[1 * 2 + b]);
            tz[3 * 2 + b] = y[1 * 2 + b];                          // sign folded into the reduction
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Z[i * 4 + 0] = tz[i * 2 + 0];
            Z[i * 4 + 1] = pk_add(tz[i * 2 + 0], tz[i * 2 + 1]);
            Z[i * 4 + 2] = pk_sub(tz[i * 2 + 0], tz[i * 2 + 1]);
            Z[i * 4 + 3] = tz[i * 2 + 1];
        }
        const f32x2w bs = pk_add(tz[1 * 2 + 0], tz[1 * 2 + 1]);    // (y0 + y2) + (y1 + y3) of both tiles
        bsum += bs.x;
        bsum += bs.y;
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[xi >> 1][(xi & 1) * 2], Z[xi].x, acc[xi], 0, 0, 0);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[xi >> 1][(xi & 1) * 2 + 1], Z[xi].y, acc[xi], 0, 0, 0);
    };

    // ---- quads: NQ = rows * Q, software-pipelined over the quads ---------------------------------------------------
    // Vector-memory loads return in order, so the ORDER of issue decides what a wait covers: the output gradients of a
    // group are issued one group ahead of their use and always BEFORE the 32 long-latency patch loads of the next quad
    // (which are needed only at the end of the iteration); the first group's gradients of quad n + 1 are issued in front
    // of the last MFMAs of quad n.  Only the LDS reads of a quad's first group (behind the barrier) are exposed.
    const int NQ = (r1 - r0) * Q;
    f32x2w xn[16];
    f32x4v Va[8], Vb[8];
    f32x2w ya[4], yb[4];
    int ty = r0 % TY, q = 0, n = 0;
    const float *px = ximg_ptr(r0 / TY), *py = yrow_ptr(r0 / TY, ty);          // current image of the input / tile row of the gradients
    if (NQ > 0) {
        load_x(xn, px, ty, 0);
        load_y(ya, wg_rsrc(py), 0);
        xform_store(xn, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (NQ > 0) read_v(Va, 0, 0);
    // The body is branch-free (the last quad is peeled off through the same lambda): with an `if (more)` inside, hipcc keeps
    // the transform of the next quad out of the MFMA stream and merges the wait counts of both paths conservatively.
    // The quad's barrier sits in front of its LAST group's MFMAs (as in conv_wino.hip): by then every wave has written the
    // next quad's V and issued its last reads of this one, so the next quad's first operands are fetched from LDS under
    // the cover of those MFMAs.  Va enters a quad holding its first group.
    auto quad = [&](auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
        const int buf = n & 1;
        int tyn = ty, qn = q + 1;
        const float *pxn = px, *pyn = py;
        if (qn == Q) {                                            // (wave-uniform: scalar selects, no branch)
            qn = 0;
            pyn += ystep;
            if (++tyn == TY) tyn = 0, pxn += ximg;
        }
        const __amdgpu_buffer_rsrc_t yd = wg_rsrc(py);
        read_v(Vb, buf, 1);
        load_y(yb, yd, 4 * q + 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (more) load_x(xn, pxn, tyn, qn);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(Va, ya);
        __builtin_amdgcn_sched_barrier(0);
        read_v(Va, buf, 2);
        load_y(ya, yd, 4 * q + 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(Vb, yb);
        __builtin_amdgcn_sched_barrier(0);
        read_v(Vb, buf, 3);
        load_y(yb, yd, 4 * q + 3);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (more) {                                     // the next quad's patches were loaded two groups ago
            xform_store(xn, buf ^ 1);
        }
        mfma_group(Va, ya);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if constexpr (more) {
            load_y(ya, wg_rsrc(pyn), 4 * qn);                     // first group of the next quad
            read_v(Va, buf ^ 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(Vb, yb);
        __builtin_amdgcn_sched_barrier(0);
        px = pxn, py = pyn, ty = tyn, q = qn, ++n;
    };
    while (n + 1 < NQ) quad(std::true_type());
    if (NQ > 0) quad(std::false_type());

    float *out = P.slab + (((long)ks * 16) * Cx + rt * 32) * Cy + ct * 32 + l31;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = (v & 3) + 8 * (v >> 2) + 4 * kh;
            out[((long)xi * Cx + row) * Cy] = acc[xi][v];
        }
    if (P.bslab && rt == 0) {
        const float b = bsum + __shfl_xor(bsum, 32, 64);
        if (kh == 0) P.bslab[(long)ks * Cy + ct * 32 + l31] = b;
    }
}

constexpr int WH_STRIDE = 20;                                   // floats per (group, lane) in LDS: 8 xi x (tile a, tile b) + 4 pad (b128 reads of 16 lanes: disjoint banks)
constexpr int WH_BUF = 4 * 64 * WH_STRIDE;                      // floats per buffer (4 groups x 64 lanes)

// One HALF of the transform domain per workgroup (HH: the positions xi = 4 i + 2 HH + jj, jj = 0, 1), two workgroups per CU: a wave keeps
// 128 accumulators instead of 256 and the two waves of a SIMD - one of each workgroup - fill each other's non-MFMA time, which the
// one-wave-per-SIMD kernel above cannot (0.67 of the fp32 MFMA rate: every load, transform add and LDS access of the only wave idles the
// matrix core).  Price: both workgroups load and column-transform the same patches (the row transform, the LDS traffic and the output-
// gradient transform are halved with the positions).  Same sums in the same order as the kernel above: bit-identical slabs.
__global__ void __launch_bounds__(256, 2) wino_wgrad_half_kernel(const rnh_wgrad_args_t P, const int Cx, const int Cy, const int KS, const int rows_per) {
    __shared__ __attribute__((aligned(16))) float sV[2 * WH_BUF];   // 40 KB: two workgroups per CU
    // consecutive blocks (the same XCD after the remap, dispatched together) = the two halves of one quad of items: they read the same
    // patches and output gradients within microseconds of each other
    const int lb2 = rnh_xcd_remap(blockIdx.x, gridDim.x);
    const int lb = lb2 >> 1;
    // (the half is a compile-time constant of the body: accumulator registers cannot be indexed at run time)
    auto body = [&](auto hh_tag) {
    constexpr int HH = decltype(hh_tag)::value;
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int RT = Cx >> 5, CT = Cy >> 5;
    const int item = lb * 4 + wave;                             // CT % 4 == 0: the four waves share (ks, rt)
    const int ks = item / (RT * CT), rc = item - ks * RT * CT, rt = rc / CT, ct = rc - rt * CT;
    const int H = P.H, W = P.W, TY = H >> 1, G = W >> 3, Q = G >> 2;
    const int rows_total = P.B * TY;
    const int r0 = ks * rows_per, r1 = min(rows_total, r0 + rows_per);

    int ysrc = 0, cy0 = ct * 32;
    while (cy0 >= P.ys[ysrc].nch) cy0 -= P.ys[ysrc++].nch;
    const rnh_src_t &Y = P.ys[ysrc];
    const int sc = Y.scale, Hs = H * sc, Ws = W * sc;
    // input source of this row tile (32 channels never straddle two sources: nch % 32 == 0)
    int xsrc = 0, cx0 = rt * 32;
    while (cx0 >= P.xs[xsrc].nch) cx0 -= P.xs[xsrc++].nch;
    const rnh_src_t &X = P.xs[xsrc];
    const int XC = X.C;
    // The patches come straight from the unpadded sources (no gathered copy): the descriptor base sits one pixel LEFT of
    // the image, so column j of tile t is at (2 (kh + 2t) + j) pixels; the one column left of the first group / right of
    // the last one gets offset 0xFFFFFFFF (hardware range check: zero, no access; the other lane half is in range, so
    // the load still goes to memory in order); rows above / below the image are clamped and zeroed after the load
    // (a load with ALL lanes out of range would return ahead of older loads).
    int vx[2][4], vy[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) vx[t][j] = ((2 * (kh + 2 * t) + j) * XC + l31) * 4;
#pragma unroll
        for (int b = 0; b < 2; ++b) vy[t][b] = ((2 * (kh + 2 * t) + b) * sc * Y.C + l31) * 4;
    }
    const int xrow = W * XC * 4, xgrp = 8 * XC * 4, yrow = sc * Ws * Y.C * 4, ygrp = 8 * sc * Y.C * 4;
    // Row pointers advance by constants (no multiplications in the loop): both tensors are dense.
    const long ximg = (long)H * W * XC, ystep = (long)2 * sc * Ws * Y.C;
    auto ximg_ptr = [&](int img) { return X.ptr + X.c0 + cx0 + ((long)img + X.img_off) * ximg - XC; };
    auto yrow_ptr = [&](int img, int ty) {
        return Y.ptr + Y.c0 + cy0 + ((((long)img + Y.img_off) * Hs + (long)2 * ty * sc + Y.sub_y) * Ws + Y.sub_x) * Y.C;
    };

    f32x16 acc[8];                                              // acc[2 i + jj] = position (row i, column 2 HH + jj)
#pragma unroll
    for (int xi = 0; xi < 8; ++xi)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[xi][v] = 0.f;
    float bsum = 0.f;

    // The output gradients of a lane's two tiles ride in the halves of packed registers (x = tile a, y = tile b): their
    // transform is an add / subtract network, one v_pk_add_f32 per pair (written as asm: hipcc scalarises packed adds; the
    // s_nop covers the VALU-write -> MFMA-read wait states, which hipcc does not add behind an asm statement).
    typedef float f32x2w __attribute__((ext_vector_type(2)));
    auto pk_add = [&](f32x2w p, f32x2w r) {
        f32x2w o;
        asm("v_pk_add_f32 %0, %1, %2\n\ts_nop 1" : "=v"(o) : "v"(p), "v"(r));
        return o;
    };
    auto pk_sub = [&](f32x2w p, f32x2w r) {
        f32x2w o;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 1" : "=v"(o) : "v"(p), "v"(r));
        return o;
    };
    // (results that only go to other VALU instructions or to LDS need no pad)
    auto pk_add0 = [&](f32x2w p, f32x2w r) {
        f32x2w o;
        asm("v_pk_add_f32 %0, %1, %2" : "=v"(o) : "v"(p), "v"(r));
        return o;
    };
    auto pk_sub0 = [&](f32x2w p, f32x2w r) {
        f32x2w o;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(p), "v"(r));
        return o;
    };
    // x patches of this wave's group (quad position q of row r): tiles a and b, 32 loads
    auto load_x = [&](f32x2w (&d)[16], const float *ximg_p, int ty, int q) {
        const __amdgpu_buffer_rsrc_t xd = wg_rsrc(ximg_p);
        const int gq = 4 * q + wave;
        const int gx = __builtin_amdgcn_readfirstlane(gq * xgrp);
        int vo[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) vo[t][j] = vx[t][j];
        if (gq == 0 && kh == 0) vo[0][0] = -1;                    // x = -1
        if (gq == G - 1 && kh == 1) vo[1][3] = -1;                // x = W
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yi = 2 * ty - 1 + i;
            const int yc = yi < 0 ? 0 : (yi >= H ? H - 1 : yi);
            const int so = __builtin_amdgcn_readfirstlane(gx + yc * xrow);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d[i * 4 + j].x = wg_ld(xd, vo[0][j], so);
                d[i * 4 + j].y = wg_ld(xd, vo[1][j], so);
            }
        }
        const bool top = ty == 0, bottom = ty == TY - 1;          // wave-uniform: selects
        const f32x2w zero = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d[0 * 4 + j] = top ? zero : d[0 * 4 + j];
            d[3 * 4 + j] = bottom ? zero : d[3 * 4 + j];
        }
    };
    // V = B^T d B of both tiles (packed) -> LDS buffer `buf`, slot of this wave's group: [group][lane][xi][tile]
    auto xform_store = [&](const f32x2w (&d)[16], int buf) {
        f32x2w tq[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tq[0 * 4 + j] = pk_sub0(d[0 * 4 + j], d[2 * 4 + j]);
            tq[1 * 4 + j] = pk_add0(d[1 * 4 + j], d[2 * 4 + j]);
            tq[2 * 4 + j] = pk_sub0(d[2 * 4 + j], d[1 * 4 + j]);
            tq[3 * 4 + j] = pk_sub0(d[1 * 4 + j], d[3 * 4 + j]);
        }
        float *o = sV + buf * WH_BUF + (wave * 64 + lane) * WH_STRIDE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                              // the two columns of this half only
            const f32x2w va = HH == 0 ? pk_sub0(tq[i * 4 + 0], tq[i * 4 + 2]) : pk_sub0(tq[i * 4 + 2], tq[i * 4 + 1]);
            const f32x2w vb = HH == 0 ? pk_add0(tq[i * 4 + 1], tq[i * 4 + 2]) : pk_sub0(tq[i * 4 + 1], tq[i * 4 + 3]);
            *reinterpret_cast<f32x4v *>(o + 4 * i) = f32x4v{va.x, va.y, vb.x, vb.y};
        }
    };
    auto read_v = [&](f32x4v (&V)[4], int buf, int j) {
        const float *o = sV + buf * WH_BUF + (j * 64 + lane) * WH_STRIDE;
#pragma unroll
        for (int i = 0; i < 4; ++i) V[i] = *reinterpret_cast<const f32x4v *>(o + 4 * i);
    };
    auto load_y = [&](f32x2w (&y)[4], __amdgpu_buffer_rsrc_t yd, int g) {
        const int gy = __builtin_amdgcn_readfirstlane(g * ygrp);
        y[0].x = wg_ld(yd, vy[0][0], gy);
        y[1].x = wg_ld(yd, vy[0][1], gy);
        y[2].x = wg_ld(yd, vy[0][0], gy + yrow);
        y[3].x = wg_ld(yd, vy[0][1], gy + yrow);
        y[0].y = wg_ld(yd, vy[1][0], gy);
        y[1].y = wg_ld(yd, vy[1][1], gy);
        y[2].y = wg_ld(yd, vy[1][0], gy + yrow);
        y[3].y = wg_ld(yd, vy[1][1], gy + yrow);
    };
    auto mfma_group = [&](const f32x4v (&V)[4], const f32x2w (&y)[4]) {
        f32x2w tz[8], Z[8];
#pragma unroll
        for (int b = 0; b < 2; ++b) {                              // Z' = A' dY A'^T,  A' = [1 0; 1 1; 1 -1; 0 1]
            tz[0 * 2 + b] = y[0 * 2 + b];
            tz[1 * 2 + b] = pk_add(y[0 * 2 + b], y[1 * 2 + b]);
            tz[2 * 2 + b] = pk_sub(y[0 * 2 + b], y[1 * 2 + b]);
            tz[3 * 2 + b] = y[1 * 2 + b];                          // sign folded into the reduction
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (HH == 0) {
                Z[i * 2 + 0] = tz[i * 2 + 0];
                Z[i * 2 + 1] = pk_add(tz[i * 2 + 0], tz[i * 2 + 1]);
            } else {
                Z[i * 2 + 0] = pk_sub(tz[i * 2 + 0], tz[i * 2 + 1]);
                Z[i * 2 + 1] = tz[i * 2 + 1];
            }
        }
        if constexpr (HH == 0) {                                   // the bias gradient rides with the first half
            const f32x2w bs = pk_add(tz[1 * 2 + 0], tz[1 * 2 + 1]);    // (y0 + y2) + (y1 + y3) of both tiles
            bsum += bs.x;
            bsum += bs.y;
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[a >> 1][(a & 1) * 2], Z[a].x, acc[a], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[a >> 1][(a & 1) * 2 + 1], Z[a].y, acc[a], 0, 0, 0);
    };

    // ---- quads: NQ = rows * Q, software-pipelined over the quads ---------------------------------------------------
    // Vector-memory loads return in order, so the ORDER of issue decides what a wait covers: the output gradients of a
    // group are issued one group ahead of their use and always BEFORE the 32 long-latency patch loads of the next quad
    // (which are needed only at the end of the iteration); the first group's gradients of quad n + 1 are issued in front
    // of the last MFMAs of quad n.  Only the LDS reads of a quad's first group (behind the barrier) are exposed.
    const int NQ = (r1 - r0) * Q;
    f32x2w xn[16];
    f32x4v Va[4], Vb[4];
    f32x2w ya[4], yb[4];
    int ty = r0 % TY, q = 0, n = 0;
    const float *px = ximg_ptr(r0 / TY), *py = yrow_ptr(r0 / TY, ty);          // current image of the input / tile row of the gradients
    if (NQ > 0) {
        load_x(xn, px, ty, 0);
        load_y(ya, wg_rsrc(py), 0);
        xform_store(xn, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (NQ > 0) read_v(Va, 0, 0);
    // The body is branch-free (the last quad is peeled off through the same lambda): with an `if (more)` inside, hipcc keeps
    // the transform of the next quad out of the MFMA stream and merges the wait counts of both paths conservatively.
    // The quad's barrier sits in front of its LAST group's MFMAs (as in conv_wino.hip): by then every wave has written the
    // next quad's V and issued its last reads of this one, so the next quad's first operands are fetched from LDS under
    // the cover of those MFMAs.  Va enters a quad holding its first group.
    auto quad = [&](auto more_tag) {
        constexpr bool more = decltype(more_tag)::value;
        const int buf = n & 1;
        int tyn = ty, qn = q + 1;
        const float *pxn = px, *pyn = py;
        if (qn == Q) {                                            // (wave-uniform: scalar selects, no branch)
            qn = 0;
            pyn += ystep;
            if (++tyn == TY) tyn = 0, pxn += ximg;
        }
        const __amdgpu_buffer_rsrc_t yd = wg_rsrc(py);
        read_v(Vb, buf, 1);
        load_y(yb, yd, 4 * q + 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (more) load_x(xn, pxn, tyn, qn);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(Va, ya);
        __builtin_amdgcn_sched_barrier(0);
        read_v(Va, buf, 2);
        load_y(ya, yd, 4 * q + 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(Vb, yb);
        __builtin_amdgcn_sched_barrier(0);
        read_v(Vb, buf, 3);
        load_y(yb, yd, 4 * q + 3);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (more) {                                     // the next quad's patches were loaded two groups ago
            xform_store(xn, buf ^ 1);
        }
        mfma_group(Va, ya);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if constexpr (more) {
            load_y(ya, wg_rsrc(pyn), 4 * qn);                     // first group of the next quad
            read_v(Va, buf ^ 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(Vb, yb);
        __builtin_amdgcn_sched_barrier(0);
        px = pxn, py = pyn, ty = tyn, q = qn, ++n;
    };
    while (n + 1 < NQ) quad(std::true_type());
    if (NQ > 0) quad(std::false_type());

    float *out = P.slab + (((long)ks * 16) * Cx + rt * 32) * Cy + ct * 32 + l31;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = (v & 3) + 8 * (v >> 2) + 4 * kh, xi = 4 * (a >> 1) + 2 * HH + (a & 1);
            out[((long)xi * Cx + row) * Cy] = acc[a][v];
        }
    if (HH == 0 && P.bslab && rt == 0) {
        const float b = bsum + __shfl_xor(bsum, 32, 64);
        if (kh == 0) P.bslab[(long)ks * Cy + ct * 32 + l31] = b;
    }
    };
    if (lb2 & 1) body(std::integral_constant<int, 1>());
    else body(std::integral_constant<int, 0>());
}

// stage 1: U[xi][ci][co] = sum over the tile-row ranges, in fixed order (one thread per element: coalesced, 16*Cx*Cy threads)
// (n is a multiple of 4: 16 * Cx * Cy.)  One float4 column per thread; the partial slabs are fetched eight at a time so
// that eight 16-byte loads are in flight per lane, and added in slab order - the same sums, element by element, as a
// plain loop over k.
__global__ void __launch_bounds__(256) wino_wgrad_sum_kernel(const float *__restrict__ slab, float *__restrict__ U, int KS, long n) {
    const long n4 = n >> 2;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        const float *p = slab + 4 * e;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 8 <= KS; k += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = rnh_ld4(p + (long)(k + j) * n);
#pragma unroll
            for (int j = 0; j < 8; ++j) s.x += v[j].x, s.y += v[j].y, s.z += v[j].z, s.w += v[j].w;
        }
        for (; k < KS; ++k) {
            const float4 v = rnh_ld4(p + (long)k * n);
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
        rnh_st4(U + 4 * e, s);
    }
}

// stage 2: dw[(colmap[co]*Cin + rowmap[ci])*9 + 3a + b] (+)= (G^T U G)[a][b];  db[colmap[co]] (+)= sum bslab
__global__ void wino_wgrad_reduce_kernel(const float *U, const float *bslab, int KS, int Cx, int Cy, const int *rowmap,
                                         const int *colmap, int Cin, float *dw, float *db, int accumulate, int folded) {
    const long total = (long)Cx * Cy;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total + Cy; e += (long)gridDim.x * blockDim.x) {
        if (e >= total) {
            const int co = colmap[e - total];
            if (bslab && db && co >= 0) {
                float s = 0.f;
                for (int k = 0; k < KS; ++k) s += bslab[(long)k * Cy + (e - total)];
                db[co] = accumulate ? db[co] + s : s;
            }
            continue;
        }
        const int j = (int)(e % Cy), i = (int)(e / Cy);
        const int ci = rowmap[i], co = colmap[j];
        if (ci < 0 || co < 0) continue;
        float Uv[16];
#pragma unroll
        // folded: the LDS kernel accumulates with A' = [1 0; 1 1; 1 -1; 0 1] (no negations next to the MFMAs):
        // Z = s_a s_b Z' with s = (1, 1, 1, -1), applied here, exactly
        for (int xi = 0; xi < 16; ++xi) {
            const float u = U[((long)xi * Cx + i) * Cy + j];
            Uv[xi] = (folded && (((xi >> 2) == 3) != ((xi & 3) == 3))) ? -u : u;
        }
        // G^T (4x4) G with G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
        float T[3][4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            T[0][x] = Uv[0 * 4 + x] + 0.5f * (Uv[1 * 4 + x] + Uv[2 * 4 + x]);
            T[1][x] = 0.5f * (Uv[1 * 4 + x] - Uv[2 * 4 + x]);
            T[2][x] = 0.5f * (Uv[1 * 4 + x] + Uv[2 * 4 + x]) + Uv[3 * 4 + x];
        }
        float *o = dw + ((long)co * Cin + ci) * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float g0 = T[a][0] + 0.5f * (T[a][1] + T[a][2]), g1 = 0.5f * (T[a][1] - T[a][2]),
                        g2 = 0.5f * (T[a][1] + T[a][2]) + T[a][3];
            o[a * 3 + 0] = accumulate ? o[a * 3 + 0] + g0 : g0;
            o[a * 3 + 1] = accumulate ? o[a * 3 + 1] + g1 : g1;
            o[a * 3 + 2] = accumulate ? o[a * 3 + 2] + g2 : g2;
        }
    }
}

inline int wg_grid_for(long n, int cap = 16384) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

struct WinoWgradShape {
    int Cx, Cy, KS, rows_per;
};

bool wino_wgrad_shape(const rnh_wgrad_args_t &a, WinoWgradShape *s) {
    if (a.nxs < 1 || a.nys < 1 || a.nxs > RNH_MAX_SRC || a.nys > RNH_MAX_SRC) return false;
    if ((a.H & 1) || (a.W & 15) || a.ntaps != 9) return false;          // W % 16: an even number of 4-tile groups per row
    int Cx = 0, Cy = 0;
    for (int i = 0; i < a.nxs; ++i) {
        if ((a.xs[i].nch & 31) || a.xs[i].scale != 1 || a.xs[i].ptr2) return false;
        Cx += a.xs[i].nch;
    }
    for (int i = 0; i < a.nys; ++i) {
        if ((a.ys[i].nch & 31) || a.ys[i].ptr2 || a.ys[i].scale != a.ys[0].scale) return false;
        Cy += a.ys[i].nch;
    }
    const long rows = (long)a.B * (a.H / 2);
    const int items = (Cx / 32) * (Cy / 32);
    long ks = 1024 / items;
    if (ks < 1) ks = 1;
    if (ks > rows) ks = rows;
    if (ks > 256) ks = 256;
    s->Cx = Cx;
    s->Cy = Cy;
    s->KS = (int)ks;
    s->rows_per = (int)((rows + ks - 1) / ks);
    s->KS = (int)((rows + s->rows_per - 1) / s->rows_per);
    return true;
}

// The LDS-sharing kernel needs whole quads of groups per tile row and four column tiles per workgroup.  RNH_WGRAD_LDS=0
// keeps the per-lane kernel for A/B measurements; both accumulate the tiles in the same order: bit-identical sums.
bool wino_wgrad_uses_lds(const rnh_wgrad_args_t &a, const WinoWgradShape &s) {
    const char *e = getenv("RNH_WGRAD_LDS");
    return !(e && e[0] == '0') && (s.Cy / 32) % 4 == 0 && (a.W % 32) == 0;
}

}  // namespace

extern "C" int rnh_wino_wgrad_supported(const rnh_wgrad_args_t *args) {
    WinoWgradShape s;
    return args && wino_wgrad_shape(*args, &s);
}

/* floats: [0] padded input copy, [1] slab, [2] bias slab */
extern "C" int rnh_wino_wgrad_ws_floats(const rnh_wgrad_args_t *args, int64_t *out3) {
    WinoWgradShape s;
    if (!args || !out3 || !wino_wgrad_shape(*args, &s)) RNH_FAIL(RNH_E_RANGE, "rnh_wino_wgrad_ws_floats: shape not supported");
    // the gathered, zero-padded input copy is only needed by the per-lane kernel
    out3[0] = wino_wgrad_uses_lds(*args, s) ? 4 : (int64_t)args->B * (args->H + 2) * (args->W + 2) * s.Cx;
    out3[1] = (int64_t)(s.KS + 1) * 16 * s.Cx * s.Cy;          // partial slabs + their sum
    out3[2] = (int64_t)s.KS * s.Cy;
    return 0;
}

extern "C" int rnh_wino_wgrad(const rnh_wgrad_args_t *args, float *xp, const int32_t *rowmap, const int32_t *colmap, int Cin, float *dw,
                              float *db, int accumulate, void *stream) {
    WinoWgradShape s;
    if (!args || !xp || !rowmap || !colmap || !dw) RNH_FAIL(RNH_E_ARG, "rnh_wino_wgrad: bad arguments");
    const rnh_wgrad_args_t &a = *args;
    if (!wino_wgrad_shape(a, &s)) RNH_FAIL(RNH_E_RANGE, "rnh_wino_wgrad: shape not supported (rnh_wino_wgrad_supported)");
    if (!a.slab || (db && !a.bslab)) RNH_FAIL(RNH_E_ARG, "rnh_wino_wgrad: workspaces missing");
    for (int i = 0; i < a.nxs; ++i)
        if (int rc = rnh_check_src(a.xs[i], "rnh_wino_wgrad")) return rc;
    for (int i = 0; i < a.nys; ++i)
        if (int rc = rnh_check_src(a.ys[i], "rnh_wino_wgrad")) return rc;
    if ((long)(a.H + 2) * (a.W + 2) * s.Cx * 4 * 4 >= (1L << 31)) RNH_FAIL(RNH_E_RANGE, "rnh_wino_wgrad: image too large for 32-bit row offsets");
    hipStream_t st = (hipStream_t)stream;

    rnh_wgrad_args_t b = a;
    if (!db) b.bslab = nullptr;
    const int items = s.KS * (s.Cx / 32) * (s.Cy / 32);
    const bool lds = wino_wgrad_uses_lds(a, s);
    if (!lds) {                                                  // the per-lane kernel reads the gathered, zero-padded copy
        hipLaunchKernelGGL(wino_pad_kernel, dim3((unsigned)(a.B * (a.H + 2))), dim3(256), 0, st, a, xp, s.Cx);
        RNH_CHECK_LAUNCH("rnh_wino_wgrad(pad)");
    }
    // RNH_WGRAD_HALF=0 keeps the one-workgroup-per-CU kernel for A/B measurements (bit-identical results)
    const char *eh = getenv("RNH_WGRAD_HALF");
    if (lds && !(eh && eh[0] == '0'))
        hipLaunchKernelGGL(wino_wgrad_half_kernel, dim3(items / 2), dim3(256), 0, st, b, s.Cx, s.Cy, s.KS, s.rows_per);
    else if (lds)
        hipLaunchKernelGGL(wino_wgrad_lds_kernel, dim3(items / 4), dim3(256), 0, st, b, xp, s.Cx, s.Cy, s.KS, s.rows_per);
    else
        hipLaunchKernelGGL(wino_wgrad_kernel, dim3((items + 3) / 4), dim3(256), 0, st, b, xp, s.Cx, s.Cy, s.KS, s.rows_per);
    RNH_CHECK_LAUNCH("rnh_wino_wgrad");
    // the sums go behind the partial slabs in the same workspace (rnh_wino_wgrad_ws_floats sizes it for KS + 1 slabs)
    const long nU = (long)16 * s.Cx * s.Cy;
    float *U = a.slab + (long)s.KS * nU;
    hipLaunchKernelGGL(wino_wgrad_sum_kernel, dim3(wg_grid_for(nU / 4)), dim3(256), 0, st, a.slab, U, s.KS, nU);
    RNH_CHECK_LAUNCH("rnh_wino_wgrad(sum)");
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(wg_grid_for((long)s.Cx * s.Cy + s.Cy)), dim3(256), 0, st, U, b.bslab, s.KS, s.Cx, s.Cy,
                       rowmap, colmap, Cin, dw, db, accumulate, (int)lds);
    RNH_CHECK_LAUNCH("rnh_wino_wgrad(reduce)");
    return 0;
}
