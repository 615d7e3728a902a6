// Weight gradient of a 3x3 / 1x1 convolution on bf16 MFMA for gfx950 - rnh_wgrad_bf16 (bf16-storage path; same call
// sites as rnh_conv_wgrad: the weight / bias part of aten::convolution_backward issued by loss.backward(), reference
// src/runner/trainers/acdc_vsr_refinenet_trainer.py:46).
//
//   dW[tap][ci][co] = sum_pixels X[p + off(tap)][ci] * dY[p][co]
//
// is a GEMM whose K dimension is the PIXEL index, while the tensors are NHWC (channel-fastest): both MFMA operands need
// 8 consecutive pixels of one channel per lane.  The transpose happens in the staging writes: every thread loads 8
// channels of one pixel (16 bytes) and writes them as eight 2-byte LDS stores into channel-major row images
// [channel][32 pixels] (80-byte pitch: the 16-byte fragment reads of 32 consecutive rows are bank-conflict free).
// Channel c of a 64-channel tile lives in LDS row (c & 7) * 8 + (c >> 3): the eight channel groups a wave writes with
// one ds_write_b16 then sit in consecutive rows, 20 dwords apart (banks 0, 20, 8, 28, 16, 4, 24, 12 - with the natural
// order they were 160 dwords apart, i.e. all in ONE bank: an 8-way conflict on every staging write, measured at 19 % of
// the bf16 MFMA peak); the MFMA rows / columns come out in that order and are un-permuted when the slab is written.
//
//   * work item = (image, 32-pixel column strip, range of RPI rows); per image row y one barrier-separated step;
//   * the x shift of the taps would make the fragment reads of X start at odd 2-byte offsets, so every input row is
//     written three times, pre-shifted by dx - 1 = -1, 0, +1; the y shift is a choice of row slot: input rows live in a
//     ring of 4 slots, row y + 2 is loaded while row y is being multiplied, so each input row is staged once per strip
//     and serves the three dy taps of three output rows;
//   * one workgroup = 64 input channels x 64 output channels x all taps (wave = 32 x 32 x 9 taps = 144 accumulator
//     registers), looping over its share of the work items (item = split, split + nsplit, ...); the partial sums go to
//     a slab [nsplit][ntaps][rows][cols] in the layout of rnh_conv_wgrad and are summed in fixed order by
//     rnh_wgrad_reduce (no atomics: bitwise repeatable);
//   * the bias gradient (column sums of dY) is taken from the staged dY rows by the row-tile-0 workgroups.
//
// MFMA operand maps: lane l, r = l & 31, h = l >> 5 holds A[row r][k = 8h + j], B[k = 8h + j][col r]; rows = input
// channels, columns = output channels, k = pixel inside the 16-pixel K step.
#include "rnh_common.h"

int rnh_check_msrc(const rnh_msrc_t &s, const char *who);      // conv_bf16.hip

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int WT = 32;                       // pixels per strip row: two K steps of 16
constexpr int RP = 80;                       // bytes per channel row of an LDS image: 64 B of pixels + 16 B pad
constexpr int XS_BYTES = 3 * 64 * RP;        // one input-row slot: three shifted copies of 64 channels
constexpr int YS_BYTES = 64 * RP;
constexpr int NXS = 4, NYS = 2;
constexpr int SMEM = NXS * XS_BYTES + NYS * YS_BYTES;          // 71 680 B
constexpr int XPIECES = (WT + 2) * 8;        // (pixel -1 .. 32) x 8 channel groups of 8

__device__ __forceinline__ unsigned wpk2(float a, float b) {
    const bf16x2 r = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, r);
}

struct Grp {                                 // where the 8 channels of this thread's channel group come from
    const void *ptr;
    int dtype, C, c0, img_off, sub_y, sub_x, ok;
};

// One piece in flight: 8 channels of a pixel as raw 16-byte loads (lo: 8 bf16 or 4 fp32; hi: the other 4 fp32).  The loads
// are global loads of a clamped address (never a branch around a load: hipcc waits vmcnt(0) at the join) and the result
// is masked / converted only when it is written to LDS, a whole row of MFMAs later.
struct Piece {
    uint4 lo, hi;
    int ok;
};
__device__ __forceinline__ Piece load_piece(const Grp &g, int b, int y, int x, int H, int W, int sc) {
    Piece p;
    p.ok = g.ok && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const int yy = p.ok ? y : 0, xx = p.ok ? x : 0;             // a valid address either way
    const long e = ((((long)(b + g.img_off) * H * sc + (long)yy * sc + g.sub_y) * ((long)W * sc)) + (long)xx * sc + g.sub_x) * g.C + g.c0;
    const char *q = reinterpret_cast<const char *>(g.ptr) + e * (g.dtype == RNH_DT_BF16 ? 2 : 4);
    p.lo = *reinterpret_cast<const uint4 *>(q);
    p.hi = *reinterpret_cast<const uint4 *>(q + (g.dtype == RNH_DT_BF16 ? 0 : 16));
    return p;
}
__device__ __forceinline__ uint4 piece_bf16(const Piece &p, int dtype) {
    uint4 v = p.lo;
    if (dtype != RNH_DT_BF16) {
        const float4 lo = __builtin_bit_cast(float4, p.lo), hi = __builtin_bit_cast(float4, p.hi);
        v = make_uint4(wpk2(lo.x, lo.y), wpk2(lo.z, lo.w), wpk2(hi.x, hi.y), wpk2(hi.z, hi.w));
    }
    return p.ok ? v : make_uint4(0u, 0u, 0u, 0u);
}

__device__ __forceinline__ Grp find_group(const rnh_msrc_t *srcs, int nsrc, int ch) {
    Grp g;
    g.ok = 0;
    g.ptr = srcs[0].ptr;                     // a padding group still issues its (masked) loads: they read the first bytes of source 0
    g.dtype = srcs[0].dtype;
    g.C = g.c0 = g.img_off = g.sub_y = g.sub_x = 0;
    int base = 0;
    for (int i = 0; i < nsrc; ++i) {
        if (!g.ok && ch >= base && ch < base + srcs[i].nch) {
            g.ok = 1;
            g.ptr = srcs[i].ptr;
            g.dtype = srcs[i].dtype;
            g.C = srcs[i].C;
            g.c0 = srcs[i].c0 + (ch - base);
            g.img_off = srcs[i].img_off;
            g.sub_y = srcs[i].sub_y;
            g.sub_x = srcs[i].sub_x;
        }
        base += srcs[i].nch;
    }
    return g;
}

template <int NTAPS>
__global__ void __launch_bounds__(256, 2) wgrad_bf16_kernel(const rnh_wgrad_bf16_args_t P, const int RT, const int CT, const int nseg, const int RPI,
                                                            const int nitems) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char *Xs = smem, *Ys = smem + NXS * XS_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5, wave = tid >> 6;
    const int rb = wave & 1, cb = wave >> 1;
    const int tiles = RT * CT;
    const int split = blockIdx.x / tiles, tile = blockIdx.x - split * tiles;
    const int rt = tile / CT, ct = tile - rt * CT;
    const int H = P.H, W = P.W;
    const int scx = P.xs[0].scale, scy = P.ys[0].scale;
    const int c8 = tid & 7, pxt = tid >> 3;                      // this thread's channel group and pixel inside a piece round

    const Grp gx = find_group(P.xs, P.nxs, rt * 64 + c8 * 8);
    const Grp gy = find_group(P.ys, P.nys, ct * 64 + c8 * 8);

    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    float bsum = 0.f;
    const bool want_bias = P.bslab != nullptr && rt == 0 && tid < 64;

    // one input row -> its slot: three shifted channel-major copies.  piece 0: pixel pxt - 1 (-1 .. 30), piece 1 (threads
    // with pxt < 2): pixel pxt + 31 (31, 32)
    auto write_x = [&](int slot, const uint4 v, int prel) {
        const unsigned short h[8] = {(unsigned short)(v.x & 0xffff), (unsigned short)(v.x >> 16), (unsigned short)(v.y & 0xffff), (unsigned short)(v.y >> 16),
                                     (unsigned short)(v.z & 0xffff), (unsigned short)(v.z >> 16), (unsigned short)(v.w & 0xffff), (unsigned short)(v.w >> 16)};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = prel + 1 - s;                          // copy s holds X[x0 + k + s - 1] at position k
            if (k >= 0 && k < WT) {
                unsigned char *base = Xs + slot * XS_BYTES + (s * 64 + c8) * RP + k * 2;        // channel c8*8 + e -> row e*8 + c8
#pragma unroll
                for (int e = 0; e < 8; ++e) *reinterpret_cast<unsigned short *>(base + e * 8 * RP) = h[e];
            }
        }
    };
    auto write_y = [&](int slot, const uint4 v) {
        const unsigned short h[8] = {(unsigned short)(v.x & 0xffff), (unsigned short)(v.x >> 16), (unsigned short)(v.y & 0xffff), (unsigned short)(v.y >> 16),
                                     (unsigned short)(v.z & 0xffff), (unsigned short)(v.z >> 16), (unsigned short)(v.w & 0xffff), (unsigned short)(v.w >> 16)};
        unsigned char *base = Ys + slot * YS_BYTES + c8 * RP + pxt * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) *reinterpret_cast<unsigned short *>(base + e * 8 * RP) = h[e];
    };

    for (int item = split; item < nitems; item += P.nsplit) {
        // item -> (image, strip, row range)
        const int nrg = (H + RPI - 1) / RPI;
        const int b = item / (nseg * nrg), irem = item - b * (nseg * nrg), seg = irem / nrg, rg = irem - seg * nrg;
        const int x0 = seg * WT, ya = rg * RPI, yb = ya + RPI < H ? ya + RPI : H;

        // prologue: input rows ya - 1, ya, ya + 1 and gradient row ya
        __syncthreads();                                         // the previous item's last step is done with every slot
#pragma unroll
        for (int r = -1; r <= 1; ++r) {
            const int y = ya + r, slot = (y + 4) & 3;
            write_x(slot, piece_bf16(load_piece(gx, b, y, x0 + pxt - 1, H, W, scx), gx.dtype), pxt - 1);
            if (pxt < 2) write_x(slot, piece_bf16(load_piece(gx, b, y, x0 + pxt + 31, H, W, scx), gx.dtype), pxt + 31);
        }
        write_y(ya & 1, piece_bf16(load_piece(gy, b, ya, x0 + pxt, H, W, scy), gy.dtype));
        __syncthreads();

        for (int y = ya; y < yb; ++y) {
            // requests for the next step: input row y + 2, gradient row y + 1
            const bool more = y + 1 < yb;
            // (issued unconditionally: rows beyond the range read as clamped, masked pieces that are never written)
            const Piece nx0 = load_piece(gx, b, y + 2, x0 + pxt - 1, H, W, scx);
            const Piece nx1 = load_piece(gx, b, y + 2, x0 + (pxt < 2 ? pxt + 31 : pxt - 1), H, W, scx);
            const Piece ny = load_piece(gy, b, more ? y + 1 : y, x0 + pxt, H, W, scy);
            // multiply: taps (dy, dx) read input row y + dy - 1, copy dx
            const unsigned char *Yb = Ys + (y & 1) * YS_BYTES + (cb * 32 + l31) * RP + kh * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 bfrag = *reinterpret_cast<const bf16x8 *>(Yb + ks * 32);
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    const int dy = NTAPS == 9 ? t / 3 : 1, dx = NTAPS == 9 ? t % 3 : 1;
                    const int slot = (y + dy - 1 + 4) & 3;
                    const bf16x8 afrag = *reinterpret_cast<const bf16x8 *>(Xs + slot * XS_BYTES + (dx * 64 + rb * 32 + l31) * RP + kh * 16 + ks * 32);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc[t], 0, 0, 0);
                }
            }
            if (want_bias) {                                     // column sums of dY: thread = output channel, 32 pixels of the row
                const unsigned char *yr = Ys + (y & 1) * YS_BYTES + tid * RP;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint4 u = *reinterpret_cast<const uint4 *>(yr + q * 16);
                    const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) bsum += __builtin_bit_cast(float, w4[e] << 16) + __builtin_bit_cast(float, w4[e] & 0xffff0000u);
                }
            }
            if (more) {
                const int slot = (y + 2 + 4) & 3;
                write_x(slot, piece_bf16(nx0, gx.dtype), pxt - 1);
                if (pxt < 2) write_x(slot, piece_bf16(nx1, gx.dtype), pxt + 31);
                write_y((y + 1) & 1, piece_bf16(ny, gy.dtype));
            }
            __syncthreads();
        }
    }

    // ---- partial sums -> slab[split][tap][row][col] ------------------------------------------------------------------
    const long xr = P.xrows_pad, yc = P.ycols_pad;
    float *sl = P.slab + (long)split * NTAPS * xr * yc;
    auto chan = [](int L) { return (L & 7) * 8 + (L >> 3); };    // LDS row -> channel of the 64-channel tile
    const int col = ct * 64 + chan(cb * 32 + l31);
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = rt * 64 + chan(rb * 32 + (v & 3) + 8 * (v >> 2) + 4 * kh);
            sl[((long)t * xr + row) * yc + col] = acc[t][v];
        }
    if (want_bias) P.bslab[(long)split * yc + ct * 64 + chan(tid)] = bsum;
}

}  // namespace

extern "C" int rnh_wgrad_bf16(const rnh_wgrad_bf16_args_t *args, void *stream) {
    if (!args) RNH_FAIL(RNH_E_ARG, "rnh_wgrad_bf16: null args");
    const rnh_wgrad_bf16_args_t &a = *args;
    if (a.nxs < 1 || a.nxs > RNH_MAX_SRC || a.nys < 1 || a.nys > RNH_MAX_SRC || a.B < 1 || a.H < 1 || a.W < 1 || !a.slab || a.nsplit < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_wgrad_bf16: bad arguments");
    if (a.ntaps != 9 && a.ntaps != 1) RNH_FAIL(RNH_E_RANGE, "rnh_wgrad_bf16: ntaps must be 9 or 1");
    int rows = 0, cols = 0;
    for (int i = 0; i < a.nxs; ++i) {
        if (int rc = rnh_check_msrc(a.xs[i], "rnh_wgrad_bf16")) return rc;
        if ((a.xs[i].nch & 7) || (a.xs[i].c0 & 7) || (a.xs[i].C & 7)) RNH_FAIL(RNH_E_ALIGN, "rnh_wgrad_bf16: channels must be multiples of 8");
        if (a.xs[i].scale != a.xs[0].scale) RNH_FAIL(RNH_E_RANGE, "rnh_wgrad_bf16: one scale for the x sources");
        rows += a.xs[i].nch;
    }
    for (int i = 0; i < a.nys; ++i) {
        if (int rc = rnh_check_msrc(a.ys[i], "rnh_wgrad_bf16")) return rc;
        if ((a.ys[i].nch & 7) || (a.ys[i].c0 & 7) || (a.ys[i].C & 7)) RNH_FAIL(RNH_E_ALIGN, "rnh_wgrad_bf16: channels must be multiples of 8");
        if (a.ys[i].scale != a.ys[0].scale) RNH_FAIL(RNH_E_RANGE, "rnh_wgrad_bf16: one scale for the dy sources");
        cols += a.ys[i].nch;
    }
    if (a.xrows_pad % 64 || a.ycols_pad % 64 || rows > a.xrows_pad || cols > a.ycols_pad) RNH_FAIL(RNH_E_RANGE, "rnh_wgrad_bf16: padded sizes");
    const int RT = a.xrows_pad / 64, CT = a.ycols_pad / 64, nseg = (a.W + WT - 1) / WT;
    // rows per work item: strips of up to 32 rows (each item re-stages 2 halo rows)
    const int RPI = a.H < 32 ? a.H : 32;
    const long nitems = (long)a.B * nseg * ((a.H + RPI - 1) / RPI);
    if (nitems >= (1L << 30) || (long)RT * CT * a.nsplit >= (1L << 30)) RNH_FAIL(RNH_E_RANGE, "rnh_wgrad_bf16: too large");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(RT * CT * a.nsplit)), block(256);
    if (a.ntaps == 9) hipLaunchKernelGGL((wgrad_bf16_kernel<9>), grid, block, 0, st, a, RT, CT, nseg, RPI, (int)nitems);
    else hipLaunchKernelGGL((wgrad_bf16_kernel<1>), grid, block, 0, st, a, RT, CT, nseg, RPI, (int)nitems);
    RNH_CHECK_LAUNCH("rnh_wgrad_bf16");
    return 0;
}
