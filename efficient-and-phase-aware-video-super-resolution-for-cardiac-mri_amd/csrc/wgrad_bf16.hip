// Weight gradient of a 3x3 / 1x1 convolution on bf16 MFMA for gfx950 - rnh_wgrad_bf16 (bf16-storage path; same call
// sites as rnh_conv_wgrad: the weight / bias part of aten::convolution_backward issued by loss.backward(), reference
// src/runner/trainers/acdc_vsr_refinenet_trainer.py:46).
//
//   dW[tap][ci][co] = sum_pixels X[p + off(tap)][ci] * dY[p][co]
//
// is a GEMM whose K dimension is the PIXEL index, while the tensors are NHWC (channel-fastest): both MFMA operands need
// 8 consecutive pixels of one channel per lane.  Since round 3 the transpose happens in the LDS READ path: rows are staged
// as they lie in memory - one 16-byte ds_write_b128 per (pixel, 8 channels) into [pixel][64 channels] row images - and the
// fragments are fetched with ds_read_b64_tr_b16, which hands every lane 4 consecutive pixels of its channel (two reads per
// fragment).  (Rounds 1-2 transposed in the staging WRITES: eight ds_write_b16 per piece into [channel][pixels] images and
// 16-byte fragment reads; PMC of that version at the ConvLSTM shape: 38 % of its LDS cycles were bank conflicts, 4.8 vector
// instructions per MFMA, LDS busy ~80 % of the MFMA time - 0.31 of the bf16 peak.)
//
//   * work item = (image, 32-pixel column strip, range of RPI rows); per image row y one barrier-separated step;
//   * the x shift of a tap is a shift of the pixel ROW address of the transposed read - no alignment constraint (the register-staged
//     kernel reads one fragment per shift; the LDS-DMA kernel, which is LDS-bandwidth-bound, reads 12 pixels once and makes the three
//     shifted fragments in registers: 3 reads + 4 v_alignbit instead of 6 reads); the y shift is a choice of row slot: a step takes TWO output rows (36 MFMAs per wave between two
//     barriers), input rows live in a ring of 6 slots, rows y + 3, y + 4 are loaded while rows y, y + 1 are being multiplied, so
//     each input row is staged once per strip and serves the three dy taps of three output rows;
//   * image layout: pixel row pitch 128 B = four 32-byte chunks of 16 channels; chunk c of pixel row r sits at c ^ (r & 2): the
//     four pixel rows x two chunks one 32-lane half of a transposed read touches then cover all 64 banks exactly once, for
//     every x shift;
//   * one workgroup = 64 input channels x 64 output channels x all taps (wave = 32 x 32 x 9 taps = 144 accumulator
//     registers), looping over its share of the work items (item = split, split + nsplit, ...); the partial sums go to
//     a slab [nsplit][ntaps][rows][cols] in the layout of rnh_conv_wgrad and are summed in fixed order by
//     rnh_wgrad_reduce (no atomics: bitwise repeatable);
//   * the bias gradient (column sums of dY) comes off the matrix cores too: the row-tile-0 workgroups multiply the staged dY
//     fragments with an all-ones operand (4 extra MFMAs per step in two of the four waves).
//
// MFMA operand maps: lane l, r = l & 31, h = l >> 5 holds A[row r][k = 8h + j], B[k = 8h + j][col r]; rows = input
// channels, columns = output channels, k = pixel inside the 16-pixel K step.  ds_read_b64_tr_b16 (cdna_hip_programming.md
// T10): per 16-lane group a block of 4 rows x 16 columns of 16-bit elements; lane 4q + p of the group supplies the address of
// row q, columns 4p .. 4p + 3; lane i receives column i of the four rows.  The four 16-lane groups of a wave are (channels
// 0-15, h = 0), (16-31, h = 0), (0-15, h = 1), (16-31, h = 1): block rows = pixels 8h + 4s .. + 3 for the two reads s = 0, 1.
#include "rnh_common.h"
#include <stdlib.h>

int rnh_check_msrc(const rnh_msrc_t &s, const char *who);      // conv_bf16.hip

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int WT = 32;                       // pixels per strip row: two K steps of 16
constexpr int XPX = WT + 2;                  // pixels x0 - 1 .. x0 + 32 of an input row
constexpr int XS_BYTES = XPX * 128, YS_BYTES = WT * 128;       // [pixel][64 channels] bf16, 32-byte chunks swizzled by (pixel & 2)
constexpr int NXS = 6, NYS = 4;              // ring slots: input rows y-1 .. y+4, gradient rows y .. y+3
constexpr int SMEM = NXS * XS_BYTES + NYS * YS_BYTES;          // 42 496 B

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

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

// Diagnostic ablations (never defined in the product build): -DRNH_WEXP=<mask>; results are wrong, only the time is of interest.
// 1: no global loads in the row loop, 2: no staging writes in the row loop, 4: fragments read once per step
#ifndef RNH_WEXP
#define RNH_WEXP 0
#endif

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
    // bias gradient (column sums of dY), row-tile-0 workgroups only: every thread sums the 8 channels of the pieces it stages
    // (each gradient pixel is staged exactly once per item); the 32 pixel columns are added in fixed order at the end
    const bool want_bias = P.bslab != nullptr && rt == 0;
    float bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = 0.f;

    // staging: the 16-byte piece (pixel row prow, channels 8 c8 .. + 7) of a row image
    auto piece_off = [&](int prow) { return prow * 128 + 32 * ((c8 >> 1) ^ (prow & 2)) + 16 * (c8 & 1); };
    auto xslot = [&](int r) { return Xs + ((r + 6) % 6) * XS_BYTES; };
    auto yslot = [&](int r) { return Ys + (r & 3) * YS_BYTES; };
    auto write_x = [&](int r, const Piece<XF32> &p0, const Piece<XF32> &p1) {      // input row r: pixels x0 + pxt - 1 and (pxt < 2) x0 + pxt + 31
        unsigned char *img = xslot(r);
        *reinterpret_cast<uint4 *>(img + piece_off(pxt)) = piece_bf16<XF32>(p0);
        if (pxt < 2) *reinterpret_cast<uint4 *>(img + piece_off(pxt + 32)) = piece_bf16<XF32>(p1);
    };
    auto write_y = [&](int r, const Piece<YF32> &p) {
        const uint4 v = piece_bf16<YF32>(p);
        *reinterpret_cast<uint4 *>(yslot(r) + piece_off(pxt)) = v;
        if (want_bias) {
            const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bs[2 * e] += __builtin_bit_cast(float, w4[e] << 16);
                bs[2 * e + 1] += __builtin_bit_cast(float, w4[e] & 0xffff0000u);
            }
        }
    };

    // transposed fragment reads: this lane's 16-lane group covers channels 16 gq .. + 15 of the wave's 32 (gq = (lane >> 4) & 1) and
    // pixels 8 kh + 4 s + q (q = (lane >> 2) & 3), 8 bytes p = lane & 3 of the 32-byte chunk
    const int gq = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;
    auto frag = [&](const unsigned char *img, int blk, int prow0) {               // 32-channel block blk (0, 1), pixel rows prow0 + 8 kh + ...
        const int ch = 2 * blk + gq;
        uint2 v[2];
#pragma unroll
        for (int sr = 0; sr < 2; ++sr) {
            const int prow = prow0 + 8 * kh + 4 * sr + tq;
            const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(img + prow * 128 + 32 * (ch ^ (prow & 2)) + 8 * tp));
            v[sr] = __builtin_bit_cast(uint2, t);
        }
        return __builtin_bit_cast(bf16x8, make_uint4(v[0].x, v[0].y, v[1].x, v[1].y));
    };

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
            Piece<YF32> m0, m1;
            if constexpr (RNH_WEXP & 1) {
                n0a.lo = n0b.lo = n1a.lo = n1b.lo = m0.lo = m1.lo = make_uint4(y, y, y, y);
                n0a.ok = n0b.ok = n1a.ok = n1b.ok = m0.ok = m1.ok = 1;
            } else {
                ldx(y + 3, n0a, n0b);
                ldx(y + 4, n1a, n1b);
                m0 = ldy(y + 2), m1 = ldy(y + 3);
            }

            // gradient fragments of the two output rows: [row][k step]
            bf16x8 bfr[2][2];
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bfr[o][ks] = frag(yslot(y + o), cb, 16 * ks);
            // input rows y - 1 .. y + 2; the tap's x shift dx - 1 moves the pixel rows of the read (image row 0 is pixel x0 - 1)
#pragma unroll
            for (int ri = 0; ri < (NTAPS == 9 ? 4 : 2); ++ri) {
                const int r = NTAPS == 9 ? y - 1 + ri : y + ri;
                const unsigned char *img = xslot(r);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if constexpr (NTAPS == 9) {
                        const bf16x8 f0 = (RNH_WEXP & 4) ? bfr[0][ks] : frag(img, rb, 16 * ks), f1 = (RNH_WEXP & 4) ? bfr[1][ks] : frag(img, rb, 16 * ks + 1),
                                     f2 = (RNH_WEXP & 4) ? bfr[0][ks ^ 1] : frag(img, rb, 16 * ks + 2);
#pragma unroll
                        for (int o = 0; o < 2; ++o) {
                            const int dy = ri - o;                       // input row y - 1 + ri feeds output row y + o through tap row dy
                            if (dy < 0 || dy > 2) continue;
                            acc[dy * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, bfr[o][ks], acc[dy * 3 + 0], 0, 0, 0);
                            acc[dy * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, bfr[o][ks], acc[dy * 3 + 1], 0, 0, 0);
                            acc[dy * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2, bfr[o][ks], acc[dy * 3 + 2], 0, 0, 0);
                        }
                    } else {
                        const bf16x8 f1 = frag(img, rb, 16 * ks + 1);       // 1x1: input row y + ri feeds output row y + ri
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, bfr[ri][ks], acc[0], 0, 0, 0);
                    }
                }
            }
            if (y + 2 < yb && !(RNH_WEXP & 2)) {
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
    const int col = ct * 64 + cb * 32 + l31;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = rt * 64 + rb * 32 + (v & 3) + 8 * (v >> 2) + 4 * kh;
            sl[((long)t * xr + row) * yc + col] = acc[t][v];
        }
    if (want_bias) {                                             // (block-uniform)
        __syncthreads();
        float *red = reinterpret_cast<float *>(smem);            // [pixel column 32][channel 64]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[pxt * 64 + c8 * 8 + e] = bs[e];
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
            for (int q = 0; q < 32; ++q) t += red[q * 64 + tid];
            P.bslab[(long)split * yc + ct * 64 + tid] = t;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same kernel for bf16 sources with the rows brought in by LDS-DMA (global_load_lds_dwordx4), three steps ahead.
//
// Ablation of the register-staged kernel above at the ConvLSTM shape (tools/experiments/r03/r03_run8.sh): 715 us per launch, 467 us without its
// global loads, 430 us with MFMAs only - the pieces requested at the top of a step are needed at its end, one step (~1 us)
// later, which is less than the latency of an HBM / Infinity-Cache read under load: SQ_WAIT_ANY was 47 % of the wave cycles.
// Here nothing passes through registers: every thread's 16-byte piece (pixel tid >> 3, channels 8 (tid & 7)) goes straight
// from global memory to its place in the row image - the LDS destination of a wave is linear (base + 16 lane), so the image's
// chunk swizzle is applied to the SOURCE channel group instead - pixels outside the image and gradient rows of other items
// are fetched from a zero page (every wave issues the same number of DMAs whatever the position: the counted waits stay
// valid).  Rows for step s + 3 are requested at the top of step s (10 input-row and 8 gradient-row slots, 76 KB: two workgroups
// per CU); s_waitcnt vmcnt(N) with N = the requests of the two younger steps + a raw s_barrier publish a step's rows.  hipcc
// drains ALL outstanding LDS-DMA (vmcnt(0)) in front of any LDS read it knows of, so the transposed fragment reads are inline asm
// with hand-counted lgkmcnt waits (cdna_hip_programming.md 5.7 form (ii)): one group of six reads (three x-shifted fragments)
// stays in flight under the previous group's MFMAs.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4];      // zero-initialised: the source of every masked piece

// Diagnostic ablations of the LDS-DMA kernel (never defined in the product build; results are wrong, only the time is of interest):
// -DRNH_DEXP=<mask>  1: no row requests inside the step loop (and no waits for them), 2: fragments read once per run, 4: no barrier in the loop,
//                    8: the requests are issued but never waited for
#ifndef RNH_DEXP
#define RNH_DEXP 0
#endif
// x fragments of the three taps of a row: RNH_WXF 0 = one transposed read pair per shift (6 reads), 1 (product) = 12 pixels read once, the
// shifts in registers (3 reads + 8 VALU).  Same box, ConvLSTM launch: 563-647 us against 529-546; a 4-read form measured like the 3-read one.
#ifndef RNH_WXF
#define RNH_WXF 1
#endif
constexpr int DNX = 10, DNY = 8;
constexpr int DSMEM = DNX * XS_BYTES + DNY * YS_BYTES;                    // 76 288 B

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));        // (a native vector: HIP's uint2 is a struct and cannot be an asm operand on the host pass)

#define RNH_TR(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(imm))

__global__ void __launch_bounds__(256, 2) wgrad_bf16_dma_kernel(const rnh_wgrad_bf16_args_t P, const int RT, const int CT, const int nseg,
                                                                const int RPI, const int nitems) {
    constexpr int NTAPS = 9;                                     // (the 1x1 case stays on the register-staged kernel)
    __shared__ __attribute__((aligned(16))) unsigned char smem[DSMEM];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = wave & 1, cb = wave >> 1;
    const int tiles = RT * CT;
    const int lb = rnh_xcd_remap(blockIdx.x, gridDim.x);
    const int split = lb / tiles, tile = lb - split * tiles;
    const int rt = tile / CT, ct = tile - rt * CT;
    const int H = P.H, W = P.W;
    const int c8 = tid & 7, pxt = tid >> 3;
    const int c8s = c8 ^ ((pxt & 2) << 1);                       // the channel group that belongs at linear position c8 of pixel row pxt
    const Grp gx = find_group(P.xs, P.nxs, rt * 64 + c8s * 8, H, W);
    const Grp gy = find_group(P.ys, P.nys, ct * 64 + c8s * 8, H, W);

    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    f32x16 bacc;                                                 // bias gradient: ones x dY (all 32 rows equal)
#pragma unroll
    for (int v = 0; v < 16; ++v) bacc[v] = 0.f;
    const bool want_bias = P.bslab != nullptr && rt == 0 && rb == 0;
    const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));

    // ---- DMA: one piece per thread into the row image at byte 16 tid (+ pixels 32, 33 of an input row by the first 16 lanes of wave 0).
    // Address generation was the larger part of this kernel's instruction stream (round 4: 290 of 430 instructions per step, 64-bit
    // multiplies and a modulo per request): a run's requests go out in increasing row order, so every piece has a RUNNING row pointer
    // (image, x position and its validity folded in once per run), a request selects that pointer or the zero page and advances it by
    // the row stride; ring slots are running scalar counters.  ConvLSTM launch 580 -> 505 us.
    const unsigned long long zp = (unsigned long long)g_zero_page;
    // per run: the running pointer of a piece IS the zero page (and its row stride 0) where the piece's column is outside the image or the channel group is
    // padding; rows outside the image are a wave-uniform condition: a branch between two requests, no select
    unsigned long long xp = zp, xp2 = zp, yp = zp, xst = 0, xst2 = 0, yst = 0;
    int xrow = 0, yrow = 0, yb = 0;                              // the rows the running pointers stand at; end of the run's gradient rows
    auto set_run = [&](int b, int x0, int ya) {
        const int xc = x0 + pxt - 1, xc2 = x0 + 31 + pxt, yc = x0 + pxt;
        const bool xok = gx.ok && (unsigned)xc < (unsigned)W, xok2 = gx.ok && (unsigned)xc2 < (unsigned)W, yok = gy.ok && (unsigned)yc < (unsigned)W;
        const char *xi = gx.ptr + (long)(b + gx.img_off) * gx.img_stride + (long)(ya - 1) * gx.row_stride;
        xp = xok ? (unsigned long long)(xi + (long)xc * gx.pix_stride) : zp;
        xp2 = xok2 ? (unsigned long long)(xi + (long)xc2 * gx.pix_stride) : zp;
        yp = yok ? (unsigned long long)(gy.ptr + (long)(b + gy.img_off) * gy.img_stride + (long)ya * gy.row_stride + (long)yc * gy.pix_stride) : zp;
        xst = xok ? (unsigned long long)gx.row_stride : 0;
        xst2 = xok2 ? (unsigned long long)gx.row_stride : 0;
        yst = yok ? (unsigned long long)gy.row_stride : 0;
        xrow = ya - 1;
        yrow = ya;
    };
    // slot of input row r0 + j given the slot s0 of row r0 (j < DNX)
    auto wrap = [&](int s0, int j) { const int t = s0 + j; return t >= DNX ? t - DNX : t; };
    auto dma_x = [&](int slot) {                                 // the next input row (xrow) of the run
        const bool rin = (unsigned)xrow < (unsigned)H;           // wave-uniform
        lptr_t l0 = (lptr_t)(smem + slot * XS_BYTES + wave * 1024), l1 = (lptr_t)(smem + slot * XS_BYTES + 32 * 128);
        if (rin) {
            __builtin_amdgcn_global_load_lds((gptr_t)xp, l0, 16, 0, 0);
            if (wave == 0) {
                if (lane < 16) __builtin_amdgcn_global_load_lds((gptr_t)xp2, l1, 16, 0, 0);
            }
        } else {
            __builtin_amdgcn_global_load_lds((gptr_t)zp, l0, 16, 0, 0);
            if (wave == 0) {
                if (lane < 16) __builtin_amdgcn_global_load_lds((gptr_t)zp, l1, 16, 0, 0);
            }
        }
        if (wave == 0) xp2 += xst2;
        xp += xst;
        ++xrow;
    };
    auto ys_off = [&](int r) { return DNX * XS_BYTES + (r & (DNY - 1)) * YS_BYTES; };
    auto dma_y = [&]() {                                         // the next gradient row (yrow); rows of other items (>= yb) come from the zero page
        const bool rin = (unsigned)yrow < (unsigned)yb;          // wave-uniform
        lptr_t l0 = (lptr_t)(smem + ys_off(yrow) + wave * 1024);
        if (rin) __builtin_amdgcn_global_load_lds((gptr_t)yp, l0, 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gptr_t)zp, l0, 16, 0, 0);
        yp += yst;
        ++yrow;
    };
    // what step y needs beyond step y - 2 (6 requests in wave 0, 4 elsewhere): input rows y + 1, y + 2 into the slots s, s + 1; gradient rows y, y + 1
    auto group = [&](int s) {
        dma_x(s);
        dma_x(wrap(s, 1));
        dma_y();
        dma_y();
    };

    // ---- transposed fragment reads (see the register-staged kernel): per-lane byte offsets inside a row image for the three x shifts
    const int gq = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;
    int xo[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int prow = 8 * kh + tq + dx;                       // (+ 16 ks + 4 sr: multiples of 4 leave bit 1 alone)
        xo[dx] = prow * 128 + 32 * ((2 * rb + gq) ^ (prow & 2)) + 8 * tp;
    }
    const int yo = (8 * kh + tq) * 128 + 32 * ((2 * cb + gq) ^ (tq & 2)) + 8 * tp;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)smem;

    // Work of this workgroup: a CONTIGUOUS range of steps (= pairs of output rows) in the order (image, strip, row pair), cut into
    // runs that stay inside one strip: a run needs one prologue, and consecutive row pairs of a strip continue the pipeline - with
    // 64 splits the ConvLSTM call has ~5 runs per workgroup instead of 14 items of 32 rows, each of which exposed a memory round trip
    // (RPI / nitems, the register-staged kernel's partition, are not used here)
    (void)RPI;
    (void)nitems;
    const int spr = (H + 1) >> 1;                                // steps per strip
    const long total = (long)P.B * nseg * spr;
    const long s0 = total * split / P.nsplit, s1 = total * (split + 1) / P.nsplit;
    for (long sc = s0; sc < s1;) {
        const long strip = sc / spr;
        const int st0 = (int)(sc - strip * spr);
        const int nst = (int)((s1 - sc) < (long)(spr - st0) ? (s1 - sc) : (long)(spr - st0));
        sc += nst;
        const int b = (int)(strip / nseg), x0 = (int)(strip - (long)b * nseg) * WT;
        const int ya = 2 * st0;
        yb = ya + 2 * nst < H ? ya + 2 * nst : H;
        set_run(b, x0, ya);
        // everything the previous item left in flight has landed and has been read before its slots are requested again
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int sm1 = 0;                                             // slot of input row y - 1 (every run starts its ring at slot 0)
        dma_x(0);                                                // rows ya - 1, ya
        dma_x(1);
        group(2);                                                // steps ya, ya + 2, ya + 4
        group(4);
        group(6);
        u32x2 yb4[2][2][2];
        u32x2 xf[2][RNH_WXF ? 3 : 6];                            // RNH_WXF 1: pixels 8 kh + 0..3, 4..7, 8..11 of the lane's channel
        for (int y = ya; y < yb; y += 2) {
            if constexpr (!(RNH_DEXP & 1) && !(RNH_DEXP & 8)) {                      // (8: requests go out, nobody waits for them)
                if (wave == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
            if constexpr (!(RNH_DEXP & 4)) __builtin_amdgcn_s_barrier();      // step y's rows are in LDS for every wave; step y - 2 has been read by all
            if constexpr (!(RNH_DEXP & 1)) group(wrap(sm1, 8));  // step y + 6's rows into the slots of input rows y - 3, y - 2 / gradient rows y - 2, y - 1

            // gradient fragments of the two output rows [row][k step]: 8 reads
            const bool rd = !(RNH_DEXP & 2) || y == ya;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                if (!rd) break;
                const unsigned ya_ = lds0 + ys_off(y + o) + yo;
                RNH_TR(yb4[o][0][0], ya_, 0);
                RNH_TR(yb4[o][0][1], ya_, 512);
                RNH_TR(yb4[o][1][0], ya_, 2048);
                RNH_TR(yb4[o][1][1], ya_, 2560);
            }
            // input fragments: group g = (input row ri, k step ks), three x shifts x two reads; group g + 1 is requested before
            // group g is multiplied
            auto xreads = [&](int set, int ri, int ks) {
                if (!rd) return;
                const unsigned base = lds0 + wrap(sm1, ri) * XS_BYTES;
#if RNH_WXF == 0
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const unsigned a_ = base + xo[dx];
                    if (ks == 0) {
                        RNH_TR(xf[set][2 * dx], a_, 0);
                        RNH_TR(xf[set][2 * dx + 1], a_, 512);
                    } else {
                        RNH_TR(xf[set][2 * dx], a_, 2048);
                        RNH_TR(xf[set][2 * dx + 1], a_, 2560);
                    }
                }
#else
                const unsigned a_ = base + xo[0];
                if (ks == 0) {
                    RNH_TR(xf[set][0], a_, 0);
                    RNH_TR(xf[set][1], a_, 512);
                    RNH_TR(xf[set][2], a_, 1024);
                } else {
                    RNH_TR(xf[set][0], a_, 2048);
                    RNH_TR(xf[set][1], a_, 2560);
                    RNH_TR(xf[set][2], a_, 3072);
                }
#endif
            };
            xreads(0, 0, 0);
            bf16x8 bfr[2][2];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int ri = g >> 1, ks = g & 1, set = g & 1;
                if (g + 1 < 8) xreads(set ^ 1, (g + 1) >> 1, (g + 1) & 1);
#if RNH_WXF == 0
                if (g + 1 < 8) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(xf[set][0]), "+v"(xf[set][1]), "+v"(xf[set][2]), "+v"(xf[set][3]), "+v"(xf[set][4]), "+v"(xf[set][5]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[set][0]), "+v"(xf[set][1]), "+v"(xf[set][2]), "+v"(xf[set][3]), "+v"(xf[set][4]), "+v"(xf[set][5]));
#else
                if (g + 1 < 8) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(xf[set][0]), "+v"(xf[set][1]), "+v"(xf[set][2]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[set][0]), "+v"(xf[set][1]), "+v"(xf[set][2]));
#endif
                if (g == 0) {                                    // (the gradient reads are older than group 0's: they have landed too)
#pragma unroll
                    for (int o = 0; o < 2; ++o)
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2) {
                            asm volatile("" : "+v"(yb4[o][k2][0]), "+v"(yb4[o][k2][1]));
                            bfr[o][k2] = __builtin_bit_cast(bf16x8, make_uint4(yb4[o][k2][0].x, yb4[o][k2][0].y, yb4[o][k2][1].x, yb4[o][k2][1].y));
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (g < 2 && want_bias) {
#pragma unroll
                    for (int o = 0; o < 2; ++o) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bfr[o][g], bacc, 0, 0, 0);
                }
                bf16x8 f[3];
#if RNH_WXF == 0
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    f[dx] = __builtin_bit_cast(bf16x8, make_uint4(xf[set][2 * dx].x, xf[set][2 * dx].y, xf[set][2 * dx + 1].x, xf[set][2 * dx + 1].y));
#else
                {
                    // the three x shifts of the 12 pixels in registers: dword i = pixels (2i, 2i + 1); shift 1 = four funnel shifts
                    const unsigned d0 = xf[set][0].x, d1 = xf[set][0].y, d2 = xf[set][1].x, d3 = xf[set][1].y, d4 = xf[set][2].x;
                    f[0] = __builtin_bit_cast(bf16x8, make_uint4(d0, d1, d2, d3));
                    f[1] = __builtin_bit_cast(bf16x8, make_uint4(__builtin_amdgcn_alignbit(d1, d0, 16), __builtin_amdgcn_alignbit(d2, d1, 16),
                                                                 __builtin_amdgcn_alignbit(d3, d2, 16), __builtin_amdgcn_alignbit(d4, d3, 16)));
                    f[2] = __builtin_bit_cast(bf16x8, make_uint4(d1, d2, d3, d4));
                }
#endif
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int dy = ri - o;                       // input row y - 1 + ri feeds output row y + o through tap row dy
                    if (dy < 0 || dy > 2) continue;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[dx], bfr[o][ks], acc[dy * 3 + dx], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            sm1 = wrap(sm1, 2);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // nothing may land in LDS after the workgroup has gone

    // ---- partial sums -> slab[split][tap][row][col] ------------------------------------------------------------------
    const long xr = P.xrows_pad, yc = P.ycols_pad;
    float *sl = P.slab + (long)split * NTAPS * xr * yc;
    const int col = ct * 64 + cb * 32 + l31;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = rt * 64 + rb * 32 + (v & 3) + 8 * (v >> 2) + 4 * kh;
            sl[((long)t * xr + row) * yc + col] = acc[t][v];
        }
    if (want_bias && kh == 0) P.bslab[(long)split * yc + col] = bacc[0];       // row 0 of the ones product: lanes 0 .. 31
}
#undef RNH_TR

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
    // RNH_WGRAD_DMA=0 keeps the register-staged kernel for A/B measurements; read per call like the other switches, so that one process
    // can compare the two (tests/test_bf16_path.py)
    const char *ed = getenv("RNH_WGRAD_DMA");
    const bool use_dma = !(ed && ed[0] == '0');
    if (a.ntaps == 9 && !xf && !yf && use_dma)
        hipLaunchKernelGGL(wgrad_bf16_dma_kernel, grid, block, 0, st, a, RT, CT, nseg, RPI, (int)nitems);
    else if (a.ntaps == 9) RNH_WG(9);
    else RNH_WG(1);
#undef RNH_WG
    RNH_CHECK_LAUNCH("rnh_wgrad_bf16");
    return 0;
}
