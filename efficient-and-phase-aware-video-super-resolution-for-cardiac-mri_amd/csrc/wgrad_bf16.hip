// Weight gradient of a 3x3 / 1x1 convolution on bf16 MFMA for gfx950 - rnh_wgrad_bf16 (bf16-storage path; same call
// sites as rnh_conv_wgrad: the weight / bias part of aten::convolution_backward issued by loss.backward(), reference
// src/runner/trainers/acdc_vsr_refinenet_trainer.py:46).
//
//   dW[tap][ci][co] = sum_pixels X[p + off(tap)][ci] * dY[p][co]
//
// is a GEMM whose K dimension is the PIXEL index, while the tensors are NHWC (channel-fastest): both MFMA operands need
// 8 consecutive pixels of one channel per lane.  The transpose happens in the staging writes: every thread loads 8
// channels of one pixel (16 bytes) and writes them as eight 2-byte LDS stores into channel-major row images
// [channel][pixels] (80 / 112-byte pitch: the 16-byte fragment reads of 32 consecutive rows are bank-conflict free).
// Channel c of a 64-channel tile lives in LDS row (c & 7) * 8 + (c >> 3): the eight channel groups a wave writes with
// one ds_write_b16 then sit in consecutive rows, 20 (28) dwords apart (distinct banks - with the natural
// order they were 160 dwords apart, i.e. all in ONE bank: an 8-way conflict on every staging write, measured at 19 % of
// the bf16 MFMA peak); the MFMA rows / columns come out in that order and are un-permuted when the slab is written.
//
//   * work item = (image, 32-pixel column strip, range of RPI rows); per image row y one barrier-separated step;
//   * the x shift of the taps would make the fragment reads of X start at odd 2-byte offsets; instead every lane reads
//     the five ALIGNED 16-byte pieces around its pixels once per input row and builds the dx = -1 / +1 fragments with four
//     v_alignbit_b32 each (the first version wrote every row three times, pre-shifted: 3x the LDS writes and LDS space);
//     the y shift is a choice of row slot: a step takes TWO output rows (36 MFMAs per wave between two barriers), input
//     rows live in a ring of 6 slots, rows y + 3, y + 4 are loaded while rows y, y + 1 are being multiplied, so each
//     input row is staged once per strip and serves the three dy taps of three output rows;
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
constexpr int XR = 112;                      // bytes per channel row of an input-row image: pixels -8 .. 39 (96 B) + 16 B pad
constexpr int YR = 80;                       // bytes per channel row of a gradient-row image: 32 pixels + 16 B pad
constexpr int XS_BYTES = 64 * XR, YS_BYTES = 64 * YR;
constexpr int NXS = 6, NYS = 4;              // ring slots: input rows y-1 .. y+4, gradient rows y .. y+3
constexpr int SMEM = NXS * XS_BYTES + NYS * YS_BYTES;          // 63 488 B: two workgroups per CU

__device__ __forceinline__ unsigned wpk2(float a, float b) {
    const bf16x2 r = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, r);
}

struct Grp {                                 // where the 8 channels of this thread's channel group come from
    const char *ptr;                         // element 0 of the group's first channel in image 0 (bytes)
    long img_stride, row_stride;             // bytes per image / per output row (scale applied)
    int pix_stride;                          // bytes per output pixel step in x (scale applied)
    int img_off, ok;
};

template <bool F32>
struct Piece {                               // 8 channels of one pixel in flight: raw 16-byte loads, converted when written
    uint4 lo, hi;
    int ok;
};
template <>
struct Piece<false> {
    uint4 lo;
    int ok;
};

template <bool F32>
__device__ __forceinline__ Piece<F32> load_piece(const Grp &g, int b, int y, int x, int H, int W) {
    Piece<F32> p;
    p.ok = g.ok && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const int yy = p.ok ? y : 0, xx = p.ok ? x : 0;             // a valid address either way: never a branch around a load
    const char *q = g.ptr + (long)(b + g.img_off) * g.img_stride + (long)yy * g.row_stride + (long)xx * g.pix_stride;
    p.lo = *reinterpret_cast<const uint4 *>(q);
    if constexpr (F32) p.hi = *reinterpret_cast<const uint4 *>(q + 16);
    return p;
}
template <bool F32>
__device__ __forceinline__ uint4 piece_bf16(const Piece<F32> &p) {
    uint4 v = p.lo;
    if constexpr (F32) {
        const float4 lo = __builtin_bit_cast(float4, p.lo), hi = __builtin_bit_cast(float4, p.hi);
        v = make_uint4(wpk2(lo.x, lo.y), wpk2(lo.z, lo.w), wpk2(hi.x, hi.y), wpk2(hi.z, hi.w));
    }
    return p.ok ? v : make_uint4(0u, 0u, 0u, 0u);
}

__device__ __forceinline__ Grp find_group(const rnh_msrc_t *srcs, int nsrc, int ch, int H, int W) {
    Grp g;
    g.ok = 0;
    g.ptr = reinterpret_cast<const char *>(srcs[0].ptr);        // a padding group still issues its (masked) loads: first bytes of source 0
    g.img_stride = g.row_stride = 0;
    g.pix_stride = g.img_off = 0;
    int base = 0;
    for (int i = 0; i < nsrc; ++i) {
        if (!g.ok && ch >= base && ch < base + srcs[i].nch) {
            const rnh_msrc_t &S = srcs[i];
            const long es = S.dtype == RNH_DT_BF16 ? 2 : 4, sc = S.scale;
            g.ok = 1;
            g.ptr = reinterpret_cast<const char *>(S.ptr) + ((S.sub_y * (long)W * sc + S.sub_x) * S.C + S.c0 + (ch - base)) * es;
            g.img_stride = (long)H * sc * W * sc * S.C * es;
            g.row_stride = sc * (long)W * sc * S.C * es;
            g.pix_stride = (int)(sc * S.C * es);
            g.img_off = S.img_off;
        }
        base += srcs[i].nch;
    }
    return g;
}

// v_alignbit_b32: the 32 bits starting `sh` bits into the 64-bit value {hi, lo}
__device__ __forceinline__ unsigned alignbit(unsigned hi, unsigned lo, unsigned sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
// 8 pixels starting one pixel BEFORE piece c (needs the last pixel of the previous piece p)
__device__ __forceinline__ uint4 shift_m1(const uint4 p, const uint4 c) {
    return make_uint4(alignbit(c.x, p.w, 16), alignbit(c.y, c.x, 16), alignbit(c.z, c.y, 16), alignbit(c.w, c.z, 16));
}
// 8 pixels starting one pixel AFTER the start of piece c (needs the first pixel of the next piece n)
__device__ __forceinline__ uint4 shift_p1(const uint4 c, const uint4 n) {
    return make_uint4(alignbit(c.y, c.x, 16), alignbit(c.z, c.y, 16), alignbit(c.w, c.z, 16), alignbit(n.x, c.w, 16));
}

template <int NTAPS, bool XF32, bool YF32>
__global__ void __launch_bounds__(256, 2) wgrad_bf16_kernel(const rnh_wgrad_bf16_args_t P, const int RT, const int CT, const int nseg, const int RPI,
                                                            const int nitems) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char *Xs = smem, *Ys = smem + NXS * XS_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5, wave = tid >> 6;
    const int rb = wave & 1, cb = wave >> 1;
    const int tiles = RT * CT;
    // (contiguous block lists per XCD: the RT * CT tiles of one pixel range read the same rows and share them through one L2)
    const int lb = rnh_xcd_remap(blockIdx.x, gridDim.x);
    const int split = lb / tiles, tile = lb - split * tiles;
    const int rt = tile / CT, ct = tile - rt * CT;
    const int H = P.H, W = P.W;
    const int c8 = tid & 7, pxt = tid >> 3;                      // this thread's channel group and pixel inside a piece round

    const Grp gx = find_group(P.xs, P.nxs, rt * 64 + c8 * 8, H, W);
    const Grp gy = find_group(P.ys, P.nys, ct * 64 + c8 * 8, H, W);

    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    float bsum = 0.f;
    const bool want_bias = P.bslab != nullptr && rt == 0 && tid < 64;

    // channel c8 * 8 + e of the tile -> LDS row e * 8 + c8 (conflict-free transposing writes, see above); pixel j of the
    // strip (-8 .. 39) sits at byte (j + 8) * 2 of an input row, pixel j (0 .. 31) at byte 2 j of a gradient row
    auto write8 = [&](unsigned char *base, const uint4 v, int pitch) {
        const unsigned short h[8] = {(unsigned short)(v.x & 0xffff), (unsigned short)(v.x >> 16), (unsigned short)(v.y & 0xffff), (unsigned short)(v.y >> 16),
                                     (unsigned short)(v.z & 0xffff), (unsigned short)(v.z >> 16), (unsigned short)(v.w & 0xffff), (unsigned short)(v.w >> 16)};
#pragma unroll
        for (int e = 0; e < 8; ++e) *reinterpret_cast<unsigned short *>(base + e * 8 * pitch) = h[e];
    };
    auto xslot = [&](int r) { return Xs + ((r + 6) % 6) * XS_BYTES; };
    auto yslot = [&](int r) { return Ys + (r & 3) * YS_BYTES; };
    auto write_x = [&](int r, const Piece<XF32> &p0, const Piece<XF32> &p1) {      // input row r: pixels pxt - 1 and (pxt < 2) pxt + 31
        unsigned char *row = xslot(r) + c8 * XR;
        write8(row + (pxt + 7) * 2, piece_bf16<XF32>(p0), XR);
        if (pxt < 2) write8(row + (pxt + 39) * 2, piece_bf16<XF32>(p1), XR);
    };
    auto write_y = [&](int r, const Piece<YF32> &p) { write8(yslot(r) + c8 * YR + pxt * 2, piece_bf16<YF32>(p), YR); };

    const int nrg = (H + RPI - 1) / RPI;
    for (int item = split; item < nitems; item += P.nsplit) {
        const int b = item / (nseg * nrg), irem = item - b * (nseg * nrg), seg = irem / nrg, rg = irem - seg * nrg;
        const int x0 = seg * WT, ya = rg * RPI, yb = ya + RPI < H ? ya + RPI : H;
        const int xa = x0 + pxt - 1, xb = x0 + (pxt < 2 ? pxt + 31 : pxt - 1), xy = x0 + pxt;
        auto ldx = [&](int r, Piece<XF32> &p0, Piece<XF32> &p1) {
            p0 = load_piece<XF32>(gx, b, r, xa, H, W);
            p1 = load_piece<XF32>(gx, b, r, xb, H, W);
        };
        // gradient rows at or beyond yb belong to another item (or lie outside the image): they enter as zeros
        auto ldy = [&](int r) { return load_piece<YF32>(gy, b, r < yb ? r : -1, xy, H, W); };

        // prologue: input rows ya - 1 .. ya + 2, gradient rows ya, ya + 1
        __syncthreads();                                         // the previous item's last step is done with every slot
        for (int r = ya - 1; r <= ya + 2; ++r) {
            Piece<XF32> p0, p1;
            ldx(r, p0, p1);
            write_x(r, p0, p1);
        }
        write_y(ya, ldy(ya));
        write_y(ya + 1, ldy(ya + 1));
        __syncthreads();

        for (int y = ya; y < yb; y += 2) {
            // requests for the next step (two output rows further): input rows y + 3, y + 4, gradient rows y + 2, y + 3
            Piece<XF32> n0a, n0b, n1a, n1b;
            ldx(y + 3, n0a, n0b);
            ldx(y + 4, n1a, n1b);
            const Piece<YF32> m0 = ldy(y + 2), m1 = ldy(y + 3);

            // gradient fragments of the two output rows: [row][k step]
            bf16x8 bfr[2][2];
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    bfr[o][ks] = *reinterpret_cast<const bf16x8 *>(yslot(y + o) + (cb * 32 + l31) * YR + kh * 16 + ks * 32);
            // input rows y - 1 .. y + 2: five aligned pieces per lane, the x shifts of the taps by funnel shifts in registers
#pragma unroll
            for (int ri = 0; ri < (NTAPS == 9 ? 4 : 2); ++ri) {
                const int r = NTAPS == 9 ? y - 1 + ri : y + ri;
                const unsigned char *row = xslot(r) + (rb * 32 + l31) * XR + kh * 16;
                uint4 pc[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) pc[q] = *reinterpret_cast<const uint4 *>(row + q * 16);      // pieces kh .. kh + 4
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    // centre piece of (ks, kh) is piece 1 + 2 ks + kh = pc[1 + 2 ks]
                    const uint4 f0 = shift_m1(pc[2 * ks], pc[2 * ks + 1]), f1 = pc[2 * ks + 1], f2 = shift_p1(pc[2 * ks + 1], pc[2 * ks + 2]);
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        if constexpr (NTAPS == 9) {
                            const int dy = ri - o;                       // input row y - 1 + ri feeds output row y + o through tap row dy
                            if (dy < 0 || dy > 2) continue;
                            acc[dy * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f0), bfr[o][ks], acc[dy * 3 + 0], 0, 0, 0);
                            acc[dy * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f1), bfr[o][ks], acc[dy * 3 + 1], 0, 0, 0);
                            acc[dy * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f2), bfr[o][ks], acc[dy * 3 + 2], 0, 0, 0);
                        } else {
                            if (o != ri) continue;                       // 1x1: input row y + o feeds output row y + o
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f1), bfr[o][ks], acc[0], 0, 0, 0);
                        }
                    }
                }
            }
            if (want_bias) {                                     // column sums of dY: thread = LDS row, 2 x 32 pixels
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const unsigned char *yr = yslot(y + o) + tid * YR;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint4 u = *reinterpret_cast<const uint4 *>(yr + q * 16);
                        const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) bsum += __builtin_bit_cast(float, w4[e] << 16) + __builtin_bit_cast(float, w4[e] & 0xffff0000u);
                    }
                }
            }
            if (y + 2 < yb) {
                write_x(y + 3, n0a, n0b);
                write_x(y + 4, n1a, n1b);
                write_y(y + 2, m0);
                write_y(y + 3, m1);
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
        if (a.xs[i].scale != a.xs[0].scale || a.xs[i].dtype != a.xs[0].dtype) RNH_FAIL(RNH_E_RANGE, "rnh_wgrad_bf16: one scale and one element type for the x sources");
        rows += a.xs[i].nch;
    }
    for (int i = 0; i < a.nys; ++i) {
        if (int rc = rnh_check_msrc(a.ys[i], "rnh_wgrad_bf16")) return rc;
        if ((a.ys[i].nch & 7) || (a.ys[i].c0 & 7) || (a.ys[i].C & 7)) RNH_FAIL(RNH_E_ALIGN, "rnh_wgrad_bf16: channels must be multiples of 8");
        if (a.ys[i].scale != a.ys[0].scale || a.ys[i].dtype != a.ys[0].dtype) RNH_FAIL(RNH_E_RANGE, "rnh_wgrad_bf16: one scale and one element type for the dy sources");
        cols += a.ys[i].nch;
    }
    if (a.xrows_pad % 64 || a.ycols_pad % 64 || rows > a.xrows_pad || cols > a.ycols_pad) RNH_FAIL(RNH_E_RANGE, "rnh_wgrad_bf16: padded sizes");
    const int RT = a.xrows_pad / 64, CT = a.ycols_pad / 64, nseg = (a.W + WT - 1) / WT;
    // rows per work item: strips of up to 32 rows (each item re-stages 3 halo rows); the kernel takes two rows per step
    const int RPI = a.H < 32 ? a.H : 32;
    const long nitems = (long)a.B * nseg * ((a.H + RPI - 1) / RPI);
    if (nitems >= (1L << 30) || (long)RT * CT * a.nsplit >= (1L << 30)) RNH_FAIL(RNH_E_RANGE, "rnh_wgrad_bf16: too large");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(RT * CT * a.nsplit)), block(256);
    const bool xf = a.xs[0].dtype == RNH_DT_F32, yf = a.ys[0].dtype == RNH_DT_F32;
#define RNH_WG(NTP)                                                                                                                \
    do {                                                                                                                           \
        if (xf && yf) hipLaunchKernelGGL((wgrad_bf16_kernel<NTP, true, true>), grid, block, 0, st, a, RT, CT, nseg, RPI, (int)nitems);      \
        else if (xf) hipLaunchKernelGGL((wgrad_bf16_kernel<NTP, true, false>), grid, block, 0, st, a, RT, CT, nseg, RPI, (int)nitems);     \
        else if (yf) hipLaunchKernelGGL((wgrad_bf16_kernel<NTP, false, true>), grid, block, 0, st, a, RT, CT, nseg, RPI, (int)nitems);     \
        else hipLaunchKernelGGL((wgrad_bf16_kernel<NTP, false, false>), grid, block, 0, st, a, RT, CT, nseg, RPI, (int)nitems);            \
    } while (0)
    if (a.ntaps == 9) RNH_WG(9);
    else RNH_WG(1);
#undef RNH_WG
    RNH_CHECK_LAUNCH("rnh_wgrad_bf16");
    return 0;
}
