// Weight gradient of a 3x3 convolution in Winograd form F(3x3, 4x4) over 4x4 output tiles on fp32 MFMA, BOTH transforms fused - rnh_wino44f_wgrad (round 6).
//
//   dg = G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G        d: 6x6 input patch, dY: 4x4 output-gradient tile, G, B, A of F(4x4, 3x3) (conv_wino44.hip)
//
// 36 GEMMs dU_xi[ci][co] = sum_tiles V_xi[tile][ci] * Z_xi[tile][co] (contraction over tiles, two per v_mfma_f32_32x32x2_f32): 2.25 multiplications per
// (pixel, ci, co) where the F(2x2)-tile form (wgrad_wino.hip) needs 4 and the pixel contraction (conv_wgrad.hip) 9 - for the ConvLSTM cell's weight gradient
// (autograd of reference src/model/nets/refine_net.py:234-239, :256; 128 x 256 channels over T N H W pixels) 135 instead of 240 GFLOP per launch, the largest
// single item of the fp32 step (profiles/r05_zv_*).  Round 5 measured the matrix part alone at 0.92 of the fp32 MFMA peak on operands a separate launch had
// transformed (tools/probes/wino44_wgrad_gemm.hip) and found the materialised transforms - 2.25 x the bytes of both operands through HBM, beside kernels that
// want the same bandwidth - to cost what the form saves (csrc/wgrad_wino44.hip, opt-in).  Here nothing transformed ever leaves the CU:
//
//   workgroup = 12 waves = one 32 (ci) x 64 (co) block of all 36 dU_xi, over a range of tile QUADS (4 tiles side by side = 4 x 16 pixels; K split over
//     workgroups, partial sums summed in fixed order by the finish kernel);
//   8 CONSUMER waves (2 per SIMD): wave = (9 positions of the 6x6 transform domain) x (column half): 144 accumulator registers; per quad and position one
//     ds_read2_b32 per operand and two MFMAs - nothing else;
//   4 PRODUCER waves (1 per SIMD), each with TWO register sets of raw values: waves 8 / 11 compute V rows 0-2 / 3-5 of the quad's four tiles (lane = 2 input
//     channels x tile, so that every transform instruction is a packed v_pk_*_f32 on two channels: 72 for 18 positions), waves 9 / 10 compute Z of tiles 0-1 /
//     2-3 (lane = 2 output channels x tile: 90 for 36 positions), straight from the raw tensors (buffer_load_dwordx2: 128-byte rows of 32 channels per tile;
//     scalar offsets, no address arithmetic) into the LDS image of the NEXT quad - [xi][tile][32 channels], the layout in which both the producers' 8-byte writes
//     and the consumers' reads are conflict-free.  Iteration it requests quad it + 2 into the set that held quad it, then transforms quad it + 1 from the other:
//     every request is a whole iteration old at its first use.  ONE barrier per quad (LDS counter only: the requests stay in flight across it).
//   What the form costs on this chip (profiles/r06_n_*): a non-MFMA vector instruction takes ~9 cycles of matrix-core time from the SIMD it is issued on,
//   WHICHEVER wave issues it - hence the even split of the transforms over the four SIMDs; earlier versions (16 waves, one register set) lost a memory latency
//   per quad or carried a whole quad's transform on one SIMD.  1.91 ms for the ConvLSTM cell's problem at BASELINE config 2 against 2.37 of the F(2x2)-tile
//   kernel (matrix instructions alone: 0.95).  Where the forward's transformed images of the x operand are still alive, wf12v_wgrad_kernel (below) copies V from
//   them by LDS-DMA instead of computing it: 1.54 ms.
//
// The borders of the 6x6 patches cost no copy and no branch: a V producer reads its 32-channel block straight from the source tensor that holds it (a block
// never straddles two sources), a patch row above / below the image and the column left / right of it are requested at offset -1 - outside the buffer range,
// which reads as zero (the convention of conv_wino44.hip; the first version gathered the inputs into a zero-padded copy first: 0.17 ms and 0.25-2.7 GB of
// workspace per launch).  The bias gradient is the tile sum of dY = Z at position (1, 1), accumulated by the Z producers.  rnh_wino44f_wgrad_supported: 3x3, H % 4 == 0, W % 16 == 0, x sources of scale 1 in 32-channel
// multiples, dy sources of one common scale (a scale r gathers the sub-pixel planes of an r x larger tensor: the PixelShuffle convolutions) in 64-channel multiples.
#include "rnh_common.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

// raw-buffer descriptor over 2 GiB from a wave-uniform pointer (the convention of conv_wino44.hip: the quad's base goes here, every further offset is a scalar)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wf_desc(const float *p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(u & 0xffffffffu));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f2 wf_ld2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

constexpr int WF_V = 36 * 4 * 32;                 // floats of a stage's V image  [xi][tile 4][ci 32]
constexpr int WF_Z = 36 * 2 * 4 * 32;             // ... and of its Z image       [xi][column half 2][tile 4][co 32]
constexpr int WF_STAGE = WF_V + WF_Z;             // 13 824 floats = 55 296 bytes; two stages

// B^T of F(4x4, 3x3) applied to six values (conv_wino44.hip's bt6)
__device__ __forceinline__ void wf_bt6(const f2 d0, const f2 d1, const f2 d2, const f2 d3, const f2 d4, const f2 d5, f2 *r) {
    const f2 a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = d3 - d1;
    r[0] = 4.f * d0 - 5.f * d2 + d4;
    r[1] = a + b;
    r[2] = a - b;
    r[3] = c + 2.f * e;
    r[4] = c - 2.f * e;
    r[5] = 4.f * d1 - 5.f * d3 + d5;
}
// A (6x4: the adjoint of the output transform A^T) applied to four values
__device__ __forceinline__ void wf_a4(const f2 v0, const f2 v1, const f2 v2, const f2 v3, f2 *r) {
    const f2 s = v0 + v2, u = v1 + v3, c = v0 + 4.f * v2, d = 2.f * (v1 + 4.f * v3);
    r[0] = v0;
    r[1] = s + u;
    r[2] = s - u;
    r[3] = c + d;
    r[4] = c - d;
    r[5] = v3;
}

// The workgroup barrier of the quad loop: LDS traffic drained, s_barrier - and NOT the vmcnt(0) a __syncthreads() carries: the producers' global loads
// for the quad after next stay in flight across it (with __syncthreads() every iteration waited out a full HBM latency: 2.4 instead of 1.2 us per quad)
#define WF_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// diagnostic builds (-DWF_EXP=<mask>; results WRONG, only the time is of interest): 1: the producers load their first quad only, 2: the consumers skip their MFMAs,
// 4: the producers skip the transform arithmetic and the LDS writes
#ifndef WF_EXP
#define WF_EXP 0
#endif

// y: the dy tensor at (image offset, first channel); Yc its channels per pixel.  part [S][36][Cx][Cy], bpart [S][Cy / 64][4][64].
// 3 waves per SIMD = 168 registers per lane: 144 accumulators + 12 operands in a consumer, two sets of raw values in a producer.
struct wf_xsrc {                           // the x source of one 32-channel row block: tensor (at its first image), channels per pixel, first channel of the block
    const float *ptr;
    int C, c0;
};
struct wf_xsrcs {
    wf_xsrc blk[RNH_MAX_SRC * 8];          // (up to 16 sources x 256 channels)
};
struct wf_ysrc {                           // the dy source of one 64-channel column block: the tensor at (first image, sub-pixel, first channel of the block) and its
    const float *ptr;                      // strides in floats - a source of scale r gathers every r-th pixel of an r x larger tensor (the pixel-unshuffle of a
    int pix, row;                          // PixelShuffle convolution's output gradient, reference refine_net.py:199-200)
    long img;
};
struct wf_ysrcs {
    wf_ysrc blk[RNH_MAX_SRC * 4];
};

__global__ void __launch_bounds__(768) wf12_wgrad_kernel(const wf_xsrcs XS, const int Cx, const wf_ysrcs YS, const int Cy,
                                                         const int H, const int W, const int nquads, const int nper, float *__restrict__ part,
                                                         float *__restrict__ bpart) {
    __shared__ __attribute__((aligned(16))) float sm[2 * WF_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RT = Cx >> 5, CT = Cy >> 6;
    const int bid = rnh_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int s = bid / (RT * CT), rc = bid - s * RT * CT, rt = rc / CT, ct = rc - rt * CT;
    const int q0 = s * nper, n = max(0, min(nquads, q0 + nper) - q0);
    const int QX = W >> 4, TY = H >> 2;

    if (wave < 8) {
        // ---------------- consumers: positions 9 pg .. 9 pg + 8, column half ch ----------------
        const int pg = wave & 3, ch = wave >> 2, l31 = lane & 31, kh = lane >> 5;
        f16v acc[9];
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[p][v] = 0.f;
        const int voff = (9 * pg * 4 + kh) * 32 + l31, zoff = WF_V + ((9 * pg * 2 + ch) * 4 + kh) * 32 + l31;
        WF_BARRIER();
        for (int it = 0; it < n; ++it) {
            const float *st = sm + (it & 1) * WF_STAGE;
            // three positions at a time: 12 operand registers beside the 144 accumulators
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float a0[3], a1[3], b0[3], b1[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a0[p] = st[voff + (3 * g + p) * 128];
                    a1[p] = st[voff + (3 * g + p) * 128 + 64];
                    b0[p] = st[zoff + (3 * g + p) * 256];
                    b1[p] = st[zoff + (3 * g + p) * 256 + 64];
                }
                if (!(WF_EXP & 2)) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) acc[3 * g + p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[p], b0[p], acc[3 * g + p], 0, 0, 0);
#pragma unroll
                    for (int p = 0; p < 3; ++p) acc[3 * g + p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[p], b1[p], acc[3 * g + p], 0, 0, 0);
                } else {
#pragma unroll
                    for (int p = 0; p < 3; ++p) acc[3 * g + p][0] += a0[p] * b0[p] + a1[p] * b1[p];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            WF_BARRIER();
        }
        float *o = part + ((long)s * 36 * Cx + rt * 32) * Cy + ct * 64 + ch * 32 + l31;
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[((long)(9 * pg + p) * Cx + (v & 3) + 8 * (v >> 2) + 4 * kh) * Cy] = acc[p][v];
        return;
    }

    const int pr = wave - 8;
    // the quad a request is for: a cursor (tx4, ty, img) advanced by one per request - no division in the loop - and held at the range's last quad
    int ctx = q0 % QX, cty = (q0 / QX) % TY, cimg = q0 / (QX * TY), cleft = n;      // (cleft: quads not yet requested, this one included)
    auto advance = [&]() {
        if (cleft > 1) {
            --cleft;
            if (++ctx == QX) {
                ctx = 0;
                if (++cty == TY) {
                    cty = 0;
                    ++cimg;
                }
            }
        }
    };
    if (pr == 0 || pr == 3) {
        // ---- V rows 0-2 (patch rows 0..4) / rows 3-5 (patch rows 1..5): lane = (channel pair cp of the block's 32, tile t) ----
        const int cp = lane & 15, t = lane >> 4, r0 = pr == 3 ? 1 : 0;
        const wf_xsrc &X = XS.blk[rt];                                              // (an offset into the kernel-argument segment)
        const float *xb = X.ptr + X.c0;
        const int XC = X.C, xlane = ((4 * t) * XC + 2 * cp) * 4;
        f2 dA[5][6], dB[5][6];
        auto load = [&](f2 (&d)[5][6]) {
            const int tx4 = ctx, ty = cty, img = cimg;
            advance();
            // the descriptor starts at the patch's first pixel (4 ty - 1 + r0, 16 tx4 - 1) - possibly in front of the tensor: only pixels inside the image are
            // ever requested through it; the others at offset -1 (out of range -> zero)
            const __amdgpu_buffer_rsrc_t rs = wf_desc(xb + (((long)img * H + 4 * ty - 1 + r0) * W + 16 * tx4 - 1) * XC);
            const int vl = (t == 0 && tx4 == 0) ? -1 : xlane, vr = (t == 3 && tx4 == QX - 1) ? -1 : xlane;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const bool out = (ty == 0 && r0 + i == 0) || (ty == TY - 1 && r0 + i == 5);     // (wave-uniform)
#pragma unroll
                for (int j = 0; j < 6; ++j) d[i][j] = wf_ld2(rs, out ? -1 : (j == 0 ? vl : (j == 5 ? vr : xlane)), (i * W + j) * XC * 4);
            }
        };
        auto transform = [&](f2 (&d)[5][6], float *st) {
            // three rows of B^T d, column by column, in place (d[0..2][j])
            if (pr == 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const f2 a = d[4][j] - 4.f * d[2][j], b = d[3][j] - 4.f * d[1][j];
                    d[0][j] = 4.f * d[0][j] - 5.f * d[2][j] + d[4][j];
                    d[1][j] = a + b;
                    d[2][j] = a - b;
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const f2 c = d[3][j] - d[1][j], e = d[2][j] - d[0][j], r5 = 4.f * d[0][j] - 5.f * d[2][j] + d[4][j];
                    d[0][j] = c + 2.f * e;
                    d[1][j] = c - 2.f * e;
                    d[2][j] = r5;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            float *o = st + (((pr == 3 ? 18 : 0) * 4 + t) * 32) + 2 * cp;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f2 r[6];
                wf_bt6(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5], r);
#pragma unroll
                for (int j = 0; j < 6; ++j) *reinterpret_cast<f2 *>(o + (6 * i + j) * 128) = r[j];
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (n <= 0) {
            WF_BARRIER();
            return;
        }
        load(dA);
        load(dB);
        transform(dA, sm);
        WF_BARRIER();                                                               // B_0
        for (int it = 0; it < n; it += 2) {
            if (!(WF_EXP & 1)) load(dA);                                            // iteration it (even): request quad it + 2, transform quad it + 1 (set B)
            if (it + 1 < n && !(WF_EXP & 4)) transform(dB, sm + WF_STAGE);
            WF_BARRIER();                                                           // B_(it+1)
            if (it + 1 >= n) break;
            if (!(WF_EXP & 1)) load(dB);                                            // iteration it + 1: request quad it + 3, transform quad it + 2 (set A)
            if (it + 2 < n && !(WF_EXP & 4)) transform(dA, sm);
            WF_BARRIER();                                                           // B_(it+2)
        }
        return;
    }
    // ---- Z of tiles 0-1 (wave 9) / 2-3 (wave 10): lane = (channel pair cp of the block's 64, tile of the pair) ----
    const int cp = lane & 31, tl = lane >> 5, t = 2 * (pr - 1) + tl;
    const wf_ysrc &Y = YS.blk[ct];
    const int ypix = Y.pix, yrow = Y.row, ylane = ((4 * t) * ypix + 2 * cp) * 4;
    f2 yA[4][4], yB[4][4];
    f2 bsum = {0.f, 0.f};
    auto load = [&](f2 (&d)[4][4]) {
        const int tx4 = ctx, ty = cty, img = cimg;
        advance();
        const __amdgpu_buffer_rsrc_t rs = wf_desc(Y.ptr + (long)img * Y.img + (long)(4 * ty) * yrow + (long)(16 * tx4) * ypix);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = wf_ld2(rs, ylane, (i * yrow + j * ypix) * 4);
    };
    auto transform = [&](f2 (&in)[4][4], float *st) {
        f2 M[6][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f2 r[6];
            wf_a4(in[0][j], in[1][j], in[2][j], in[3][j], r);
#pragma unroll
            for (int i = 0; i < 6; ++i) M[i][j] = r[i];
            __builtin_amdgcn_sched_barrier(0);
        }
        float *o = st + WF_V + (((cp >> 4) * 4 + t) * 32) + 2 * (cp & 15);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            f2 r[6];
            wf_a4(M[i][0], M[i][1], M[i][2], M[i][3], r);
            if (i == 1) bsum += r[1];
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<f2 *>(o + (6 * i + j) * 256) = r[j];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (n <= 0) {
        WF_BARRIER();
    } else {
        load(yA);
        load(yB);
        transform(yA, sm);
        WF_BARRIER();
        for (int it = 0; it < n; it += 2) {
            if (!(WF_EXP & 1)) load(yA);
            if (it + 1 < n && !(WF_EXP & 4)) transform(yB, sm + WF_STAGE);
            WF_BARRIER();
            if (it + 1 >= n) break;
            if (!(WF_EXP & 1)) load(yB);
            if (it + 2 < n && !(WF_EXP & 4)) transform(yA, sm);
            WF_BARRIER();
        }
    }
    if (rt == 0 && bpart) {                                                         // bias partial sums: [s][ct][tile of the quad][64]
        float *o = bpart + (((long)s * CT + ct) * 4 + t) * 64 + 2 * cp;
        *reinterpret_cast<f2 *>(o) = bsum;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// The same weight gradient with the V operand taken from the TRANSFORMED images the forward pass already holds (rnh_wino44f_wgrad_v, ABI 7): the F(4x4)
// cells and refine conv1 read their inputs in transform-domain form V = B^T d B (rnh_wino44_transform: [tile block][16-channel chunk][xi][tile 32][16]),
// and where the engine keeps those images until the backward, the V producers' whole job - 30 requests, 72 packed transform instructions, 18 LDS writes per
// quad and wave, all of them matrix-core time on this chip (see the header) - becomes 18 LDS-DMA requests per quad and workgroup: lane = (position of a
// pair, tile of the quad, chunk of the block's two, 4-channel piece) lands 16 bytes at LDS offset 16 lane, which IS the consumers' [xi][tile][32 ci] image
// (the image's piece swizzle is undone in the source offset: one v_xor per quad).  All four producer waves now transform Z: wave = (tile pair) x (rows 0-2 /
// 3-5 of A dY A^T), ~45 packed instructions + 18 LDS writes each instead of 90 + 36 on two SIMDs.  Three V stages (a DMA for quad it + 2 is issued in
// iteration it, behind the transform, and waited for - by count - in front of the barrier of iteration it + 1), two Z stages: 129 KB of LDS.
constexpr int WF_W4_BUF = 36 * 32 * 16;          // floats of one (tile block, chunk) of a transformed image (conv_wino44.hip: W4_BUF)
constexpr int WFV_Z0 = 3 * WF_V;                 // float offset of the first Z stage

struct wf_vsrc {                                 // the transformed image of one 32-channel row block: frame 0 at its first chunk, floats between frames (signed),
    const float *v;                              // 16-channel chunks of the image's tensor
    long frame;
    int nchunks, pad;
};
struct wf_vsrcs {
    wf_vsrc blk[RNH_MAX_SRC * 8];
};

__global__ void __launch_bounds__(768) wf12v_wgrad_kernel(const wf_vsrcs VS, const int Cx, const wf_ysrcs YS, const int Cy, const int H, const int W,
                                                          const int NF, const int nquads, const int nper, float *__restrict__ part,
                                                          float *__restrict__ bpart) {
    __shared__ __attribute__((aligned(16))) float sm[3 * WF_V + 2 * WF_Z];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RT = Cx >> 5, CT = Cy >> 6;
    const int bid = rnh_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int s = bid / (RT * CT), rc = bid - s * RT * CT, rt = rc / CT, ct = rc - rt * CT;
    const int q0 = s * nper, n = max(0, min(nquads, q0 + nper) - q0);
    const int QX = W >> 4, TY = H >> 2;

    if (wave < 8) {
        // ---------------- consumers (as wf12_wgrad_kernel's; V stage it % 3, Z stage it & 1) ----------------
        const int pg = wave & 3, ch = wave >> 2, l31 = lane & 31, kh = lane >> 5;
        f16v acc[9];
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[p][v] = 0.f;
        const int voff = (9 * pg * 4 + kh) * 32 + l31, zoff = WFV_Z0 + ((9 * pg * 2 + ch) * 4 + kh) * 32 + l31;
        WF_BARRIER();
        int vs = 0;
        for (int it = 0; it < n; ++it) {
            const float *sv = sm + vs * WF_V, *sz = sm + (it & 1) * WF_Z;
            vs = vs == 2 ? 0 : vs + 1;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float a0[3], a1[3], b0[3], b1[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a0[p] = sv[voff + (3 * g + p) * 128];
                    a1[p] = sv[voff + (3 * g + p) * 128 + 64];
                    b0[p] = sz[zoff + (3 * g + p) * 256];
                    b1[p] = sz[zoff + (3 * g + p) * 256 + 64];
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) acc[3 * g + p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[p], b0[p], acc[3 * g + p], 0, 0, 0);
#pragma unroll
                for (int p = 0; p < 3; ++p) acc[3 * g + p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[p], b1[p], acc[3 * g + p], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            WF_BARRIER();
        }
        float *o = part + ((long)s * 36 * Cx + rt * 32) * Cy + ct * 64 + ch * 32 + l31;
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int v = 0; v < 16; ++v) o[((long)(9 * pg + p) * Cx + (v & 3) + 8 * (v >> 2) + 4 * kh) * Cy] = acc[p][v];
        return;
    }

    // ---------------- producers: wave 8 + pr = (tile pair tp of the quad) x (row half rh of Z) + a share of the quad's 18 V requests ----------------
    const int pr = wave - 8, tp = pr & 1, rh = pr >> 1;
    // the quad a request is for: cursor (tx4, ty, image of the frame, frame), advanced by one per request round, held at the range's last quad
    int ctx = q0 % QX, cty = (q0 / QX) % TY, cimg = q0 / (QX * TY), cleft = n;
    int cfn = cimg % NF, cfr = cimg / NF;
    auto advance = [&]() {
        if (cleft > 1) {
            --cleft;
            if (++ctx == QX) {
                ctx = 0;
                if (++cty == TY) {
                    cty = 0;
                    ++cimg;
                    if (++cfn == NF) {
                        cfn = 0;
                        ++cfr;
                    }
                }
            }
        }
    };
    const int cp = lane & 31, tl = lane >> 5, t = 2 * tp + tl;
    const wf_ysrc &Y = YS.blk[ct];
    const int ypix = Y.pix, yrow = Y.row, ylane = ((4 * t) * ypix + 2 * cp) * 4;
    const wf_vsrc &VB = VS.blk[rt];
    const float *vbase = VB.v;
    const long vframe = VB.frame;
    const int vnch = VB.nchunks;
    const bool blocked = !((W >> 2) & 7) && !(TY & 3);
    // V request lane = (position of the pair, tile, chunk, piece): byte offset inside the (tile block, chunk 0) image of the quad's first tile
    const int vlane0 = ((lane >> 2) & 1) * (WF_W4_BUF * 4) + (lane >> 5) * 2048 + ((lane >> 3) & 3) * 64 + (lane & 3) * 16;
    const unsigned sm_lds = (unsigned)(unsigned long long)(&sm[0]);
    f2 yA[4][4], yB[4][4];
    f2 bsum = {0.f, 0.f};
    int qtx = 0, qty = 0, qfn = 0, qfr = 0;                                          // the quad of the current request round
    // The dy requests are inline asm and so are their waits: hipcc counts only the loads it knows, and with LDS-DMA requests it does not know in the same
    // queue its own s_waitcnt would sit out requests issued moments ago (vector-memory requests return in order).  Every wait here is the same count: the
    // 16 dy requests + 5 V requests of the newest request round stay in flight, everything older has landed.
    typedef int i32x4q __attribute__((ext_vector_type(4)));
    auto desc = [&](const float *b) {
        const unsigned long long u = (unsigned long long)b;
        i32x4q rs;
        rs[0] = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
        rs[1] = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
        rs[2] = 0x7fffffff;
        rs[3] = 0x00020000;
        return rs;
    };
    auto load = [&](f2 (&d)[4][4]) {
        qtx = ctx, qty = cty, qfn = cfn, qfr = cfr;
        const int img = cimg;
        advance();
        const i32x4q rs = desc(Y.ptr + (long)img * Y.img + (long)(4 * qty) * yrow + (long)(16 * qtx) * ypix);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int so = (i * yrow + j * ypix) * 4;
                asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(d[i][j]) : "v"(ylane), "s"(rs), "s"(so) : "memory");
            }
    };
    // (every producer wave issues FIVE V requests per quad - waves 2 and 3 request the last position pair once more - so that every wait is the same
    // instruction: with two counts behind an if, hipcc merged the tied registers of the two asm statements through copies placed in FRONT of one of them)
    auto landed = [&](f2 (&d)[4][4]) {                                              // the set's values are there (the wait is tied to the registers it guards)
        asm volatile("s_waitcnt vmcnt(21)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[0][2]), "+v"(d[0][3]), "+v"(d[1][0]), "+v"(d[1][1]), "+v"(d[1][2]), "+v"(d[1][3])::"memory");
        asm volatile("" : "+v"(d[2][0]), "+v"(d[2][1]), "+v"(d[2][2]), "+v"(d[2][3]), "+v"(d[3][0]), "+v"(d[3][1]), "+v"(d[3][2]), "+v"(d[3][3])::"memory");
    };
    auto older_landed = [&]() { asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); };
    // the V requests of the quad `load` was last called for, into V stage vst: this wave's position pairs pr, pr + 4, ...
    auto dma = [&](int vst) {
        const int TXt = W >> 2;
        int tq;                                                                     // tile index (in its frame) of the quad's first tile
        if (blocked) {
            const int bpr = TXt >> 3, tx = 4 * qtx;
            tq = qfn * TXt * TY + (((qty >> 2) * bpr + (tx >> 3)) << 5) + ((qty & 3) << 3) + (tx & 7);
        } else {
            tq = (qfn * TY + qty) * TXt + 4 * qtx;
        }
        const i32x4q rs = desc(vbase + (long)qfr * vframe + (long)(tq >> 5) * vnch * WF_W4_BUF + (tq & 31) * 16);
        const int voff = vlane0 ^ ((((tq & 31) >> 2) & 3) << 4);
        const unsigned l0 = sm_lds + vst * (WF_V * 4);
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            const int pp = min(pr + 4 * p, 17);                                     // (wave-uniform; waves 2, 3: pair 17 again - the same bytes to the same place)
            const unsigned ld = __builtin_amdgcn_readfirstlane(l0 + pp * 1024);
            const int so = __builtin_amdgcn_readfirstlane(pp * 4096);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ld), "v"(voff), "s"(rs), "s"(so) : "memory");
        }
    };
    auto transform = [&](f2 (&in)[4][4], float *sz) {
        f2 M[3][4];
        if (rh == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f2 sa = in[0][j] + in[2][j], ua = in[1][j] + in[3][j];
                M[0][j] = in[0][j];
                M[1][j] = sa + ua;
                M[2][j] = sa - ua;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f2 c = in[0][j] + 4.f * in[2][j], dd = 2.f * (in[1][j] + 4.f * in[3][j]);
                M[0][j] = c + dd;
                M[1][j] = c - dd;
                M[2][j] = in[3][j];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float *o = sz + (((cp >> 4) * 4 + t) * 32) + 2 * (cp & 15) + 18 * rh * 256;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            f2 r[6];
            wf_a4(M[i][0], M[i][1], M[i][2], M[i][3], r);
            if (i == 1 && rh == 0) bsum += r[1];
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<f2 *>(o + (6 * i + j) * 256) = r[j];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    float *const Z0 = sm + WFV_Z0, *const Z1 = Z0 + WF_Z;
    if (n <= 0) {
        WF_BARRIER();
    } else {
        load(yA);                                                                   // quad 0
        dma(0);
        load(yB);                                                                   // quad 1
        dma(1);
        landed(yA);                                                                 // (and quad 0's V requests with it)
        transform(yA, Z0);
        WF_BARRIER();                                                               // B_0
        int vs = 2;                                                                 // V stage of the next request round (quad it + 2)
        for (int it = 0; it < n; it += 2) {
            // iteration it (even): request dy of quad it + 2 (set A), transform quad it + 1 (set B), request V of quad it + 2; in front of the barrier that
            // publishes quad it + 1 its V requests - issued an iteration ago - are waited for
            if (!(WF_EXP & 1)) load(yA);                                            // (diagnostic masks: 1 no dy requests, 4 no Z transform, 8 no V requests)
            landed(yB);
            if (it + 1 < n && !(WF_EXP & 4)) transform(yB, Z1);
            if (!(WF_EXP & 8)) dma(vs);
            vs = vs == 2 ? 0 : vs + 1;
            older_landed();
            WF_BARRIER();
            if (it + 1 >= n) break;
            if (!(WF_EXP & 1)) load(yB);
            landed(yA);
            if (it + 2 < n && !(WF_EXP & 4)) transform(yA, Z0);
            if (!(WF_EXP & 8)) dma(vs);
            vs = vs == 2 ? 0 : vs + 1;
            older_landed();
            WF_BARRIER();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // (no LDS-DMA request outlives the workgroup)
    }
    if (rt == 0 && bpart && rh == 0) {                                              // bias partial sums: [s][ct][tile of the quad][64]
        float *o = bpart + (((long)s * CT + ct) * 4 + t) * 64 + 2 * cp;
        *reinterpret_cast<f2 *>(o) = bsum;
    }
}

// (Measured and not kept, profiles/r06_y_*: the same kernel with 64 (ci) x 32 (co) blocks - Z of 32 columns is half the work, shared by the four producer waves as
// the four quadrants of A dY A^T, ~30 packed instructions each instead of ~61 - runs in the same time, 1.549 against 1.540 ms: its V image is twice the bytes
// (36 requests per quad) and all four waves request the whole dy tile.  Diagnostic builds of THIS kernel (-DWF_EXP, masks 1 / 4 / 8 = no dy requests / no Z
// transform / no V requests): 1.73 -> 1.53 / 1.45 / 1.50 ms, all three off 1.18 - the three cost about the same, and what they cost follows the bytes they bring
// into the CU (20 KB of V, 32 KB of dy per quad) as much as the instructions they issue.)

// dw[(colmap[j] Cin + rowmap[i]) 9 + 3 p + q] (+)= sum_{a, b} G[a][p] G[b][q] sum_s part[s][6 a + b][i][j];  db[colmap[j]] (+)= sum_s sum_k bpart[s][j / 64][k][j % 64].
// A workgroup = 64 consecutive (i, j) entries x 4 groups of 9 positions: the K-split sums of a position group by one wave (coalesced 256-byte rows), the four
// groups meet in LDS, then G^T . G per entry.  Fixed order: deterministic.
__global__ void __launch_bounds__(256) wf_finish_kernel(const float *__restrict__ part, const float *__restrict__ bpart, const int S, const int Cx, const int Cy,
                                                        const int *__restrict__ rowmap, const int *__restrict__ colmap, const int Cin, float *dw, float *db,
                                                        const int accumulate) {
    __shared__ float us[36][65];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + lane;                                    // (Cx Cy is a multiple of 64)
    const int j = (int)(e % Cy), i = (int)(e / Cy);
    for (int k = 0; k < 9; ++k) {
        const int xi = 9 * grp + k;
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += part[(((long)s * 36 + xi) * Cx + i) * Cy + j];
        us[xi][lane] = a;
    }
    __syncthreads();
    if (grp) return;
    const int co = colmap[j], ci = rowmap[i];
    if (i == 0 && db && co >= 0) {
        float b = 0.f;
        for (int s = 0; s < S; ++s)
            for (int k = 0; k < 4; ++k) b += bpart[(((long)s * (Cy >> 6) + (j >> 6)) * 4 + k) * 64 + (j & 63)];
        db[co] = accumulate ? db[co] + b : b;
    }
    if (co < 0 || ci < 0) return;
    const float G[6][3] = {{0.25f, 0.f, 0.f},          {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    float u[36];
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) u[xi] = us[xi][lane];
    float *o = dw + ((long)co * Cin + ci) * 9;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float g = 0.f;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                float row = 0.f;
#pragma unroll
                for (int b = 0; b < 6; ++b) row += u[6 * a + b] * G[b][q];
                g += G[a][p] * row;
            }
            o[3 * p + q] = accumulate ? o[3 * p + q] + g : g;
        }
}

struct WfGeo {
    int Cx, Cy, S, nquads, nper;
};

int wf_check(const rnh_wgrad_args_t &a, WfGeo &g, bool quiet) {
#define WF_NO(msg)                                        \
    do {                                                  \
        if (quiet) return 1;                              \
        RNH_FAIL(RNH_E_RANGE, "rnh_wino44f_wgrad: " msg); \
    } while (0)
    if (a.ntaps != 9 || a.B < 1 || a.H < 4 || a.W < 16 || (a.H & 3) || (a.W & 15)) WF_NO("3x3, H % 4 == 0, W % 16 == 0");
    if (a.nxs < 1 || a.nxs > RNH_MAX_SRC || a.nys < 1 || a.nys > RNH_MAX_SRC) WF_NO("1..16 x sources, 1..16 dy sources");
    int Cx = 0;
    for (int i = 0; i < a.nxs; ++i) {
        const rnh_src_t &s = a.xs[i];
        if (!s.ptr || s.ptr2 || s.scale != 1 || s.nch < 32 || (s.nch & 31) || (s.C & 1) || (s.c0 & 1)) WF_NO("x sources: scale 1, no second operand, 32-channel multiples");
        Cx += s.nch;
    }
    int Cy = 0;
    for (int i = 0; i < a.nys; ++i) {
        const rnh_src_t &y = a.ys[i];
        if (!y.ptr || y.ptr2 || y.scale < 1 || y.scale != a.ys[0].scale || y.sub_y < 0 || y.sub_x < 0 || y.sub_y >= y.scale || y.sub_x >= y.scale || y.nch < 64 || (y.nch & 63) ||
            (y.C & 1) || (y.c0 & 1))
            WF_NO("dy sources: one common scale, no second operand, 64-channel multiples");
        Cy += y.nch;
    }
    if (Cy > RNH_MAX_SRC * 4 * 64) WF_NO("too many output channels");
    if (Cx & 31) WF_NO("input channels in multiples of 32");
    if ((long)a.B * a.H * a.W >= (1L << 27)) WF_NO("too many pixels");
    g.Cx = Cx, g.Cy = Cy;
    g.nquads = a.B * (a.H >> 2) * (a.W >> 4);
    const int blocks = (Cx >> 5) * (g.Cy >> 6);
    int S = blocks >= 256 ? 1 : 256 / blocks;                     // ONE round of workgroups on the 256 CUs (12 waves, 108 KB of LDS: one per CU); at least two quads each
    if (S > g.nquads / 2) S = g.nquads / 2 > 0 ? g.nquads / 2 : 1;
    if (S > 64) S = 64;
    g.nper = (g.nquads + S - 1) / S;
    g.S = (g.nquads + g.nper - 1) / g.nper;
    return 0;
#undef WF_NO
}

void wf_fill_ysrcs(const rnh_wgrad_args_t &a, wf_ysrcs &ys) {
    int nb = 0;
    for (int i = 0; i < a.nys; ++i) {
        const rnh_src_t &y = a.ys[i];
        const long Hs = (long)a.H * y.scale, Ws = (long)a.W * y.scale;
        for (int c = 0; c < y.nch; c += 64) {
            ys.blk[nb].ptr = y.ptr + ((long)y.img_off * Hs + y.sub_y) * Ws * y.C + (long)y.sub_x * y.C + y.c0 + c;
            ys.blk[nb].pix = y.scale * y.C;
            ys.blk[nb].row = (int)(y.scale * Ws * y.C);
            ys.blk[nb++].img = Hs * Ws * y.C;
        }
    }
}

// the V-operand form: every x source comes with the transformed image of its tensor (all its channels, NF images per frame); source channels in chunk
// multiples, whole frames
int wfv_check(const rnh_wgrad_args_t &a, const rnh_wino44_vsrc_t *vs, int NF, bool quiet) {
#define WFV_NO(msg)                                         \
    do {                                                    \
        if (quiet) return 1;                                \
        RNH_FAIL(RNH_E_RANGE, "rnh_wino44f_wgrad_v: " msg); \
    } while (0)
    if (!vs || NF < 1 || a.B % NF) WFV_NO("whole frames of NF images");
    for (int i = 0; i < a.nxs; ++i) {
        const int rel = a.xs[i].c0 - vs[i].c_first;
        if (!vs[i].v || vs[i].nchunks < 2 || rel < 0 || (rel & 15) || rel + a.xs[i].nch > vs[i].nchunks * 16 || a.xs[i].img_off)
            WFV_NO("x sources: channels [c0, c0 + nch) inside the image's [c_first, c_first + 16 nchunks), whole chunks, img_off 0 (the frame pointer carries it)");
        if ((vs[i].frame_stride < 0 ? -vs[i].frame_stride : vs[i].frame_stride) >= (1L << 40)) WFV_NO("frame stride");
    }
    return 0;
#undef WFV_NO
}

}  // namespace

extern "C" int rnh_wino44f_wgrad_supported(const rnh_wgrad_args_t *args) {
    WfGeo g;
    return args && wf_check(*args, g, true) == 0;
}

extern "C" int rnh_wino44f_wgrad_ws_floats(const rnh_wgrad_args_t *args, int64_t *out3) {
    if (!args || !out3) RNH_FAIL(RNH_E_ARG, "rnh_wino44f_wgrad_ws_floats: null argument");
    WfGeo g;
    if (int rc = wf_check(*args, g, false)) return rc;
    out3[0] = 4;                                                  // (no padded copy of the inputs since the producers read the sources themselves)
    out3[1] = (int64_t)g.S * 36 * g.Cx * g.Cy;
    out3[2] = (int64_t)g.S * (g.Cy >> 6) * 4 * 64;
    return 0;
}

extern "C" int rnh_wino44f_wgrad(const rnh_wgrad_args_t *args, float *xp, float *part, float *bpart, const int32_t *rowmap, const int32_t *colmap, int Cin,
                                 float *dw, float *db, int accumulate, void *stream) {
    (void)xp;                                                     // (kept in the signature: ABI 6 was published with it; nothing is written there)
    if (!args || !part || !rowmap || !colmap || !dw || Cin < 1 || (db && !bpart)) RNH_FAIL(RNH_E_ARG, "rnh_wino44f_wgrad: bad arguments");
    const rnh_wgrad_args_t &a = *args;
    WfGeo g;
    if (int rc = wf_check(a, g, false)) return rc;
    hipStream_t st = (hipStream_t)stream;
    wf_xsrcs xs;
    int nb = 0;
    for (int i = 0; i < a.nxs; ++i)
        for (int c = 0; c < a.xs[i].nch; c += 32) {
            if (nb >= RNH_MAX_SRC * 8) RNH_FAIL(RNH_E_RANGE, "rnh_wino44f_wgrad: more than %d row blocks", RNH_MAX_SRC * 8);
            xs.blk[nb].ptr = a.xs[i].ptr + (long)a.xs[i].img_off * a.H * a.W * a.xs[i].C;
            xs.blk[nb].C = a.xs[i].C;
            xs.blk[nb++].c0 = a.xs[i].c0 + c;
        }
    wf_ysrcs ys;
    wf_fill_ysrcs(a, ys);
    const int blocks = (g.Cx >> 5) * (g.Cy >> 6) * g.S;
    hipLaunchKernelGGL(wf12_wgrad_kernel, dim3((unsigned)blocks), dim3(768), 0, st, xs, g.Cx, ys, g.Cy, a.H, a.W, g.nquads, g.nper, part, db ? bpart : nullptr);
    RNH_CHECK_LAUNCH("rnh_wino44f_wgrad");
    hipLaunchKernelGGL(wf_finish_kernel, dim3((unsigned)((long)g.Cx * g.Cy / 64)), dim3(256), 0, st, part, bpart, g.S, g.Cx, g.Cy, rowmap, colmap, Cin, dw,
                       db, accumulate);
    RNH_CHECK_LAUNCH("rnh_wino44f_wgrad(finish)");
    return 0;
}

extern "C" int rnh_wino44f_wgrad_v_supported(const rnh_wgrad_args_t *args, const rnh_wino44_vsrc_t *vsrcs, int images_per_frame) {
    WfGeo g;
    return args && wf_check(*args, g, true) == 0 && wfv_check(*args, vsrcs, images_per_frame, true) == 0;
}

extern "C" int rnh_wino44f_wgrad_v(const rnh_wgrad_args_t *args, const rnh_wino44_vsrc_t *vsrcs, int images_per_frame, float *part, float *bpart,
                                   const int32_t *rowmap, const int32_t *colmap, int Cin, float *dw, float *db, int accumulate, void *stream) {
    if (!args || !vsrcs || !part || !rowmap || !colmap || !dw || Cin < 1 || (db && !bpart)) RNH_FAIL(RNH_E_ARG, "rnh_wino44f_wgrad_v: bad arguments");
    const rnh_wgrad_args_t &a = *args;
    WfGeo g;
    if (int rc = wf_check(a, g, false)) return rc;
    if (int rc = wfv_check(a, vsrcs, images_per_frame, false)) return rc;
    hipStream_t st = (hipStream_t)stream;
    wf_vsrcs vs;
    int nb = 0;
    for (int i = 0; i < a.nxs; ++i)
        for (int c = 0; c < a.xs[i].nch; c += 32) {
            if (nb >= RNH_MAX_SRC * 8) RNH_FAIL(RNH_E_RANGE, "rnh_wino44f_wgrad_v: more than %d row blocks", RNH_MAX_SRC * 8);
            vs.blk[nb].v = vsrcs[i].v + (long)((a.xs[i].c0 - vsrcs[i].c_first + c) >> 4) * WF_W4_BUF;
            vs.blk[nb].frame = vsrcs[i].frame_stride;
            vs.blk[nb].nchunks = vsrcs[i].nchunks;
            vs.blk[nb++].pad = 0;
        }
    wf_ysrcs ys;
    wf_fill_ysrcs(a, ys);
    const int blocks = (g.Cx >> 5) * (g.Cy >> 6) * g.S;
    hipLaunchKernelGGL(wf12v_wgrad_kernel, dim3((unsigned)blocks), dim3(768), 0, st, vs, g.Cx, ys, g.Cy, a.H, a.W, images_per_frame, g.nquads, g.nper, part,
                       db ? bpart : nullptr);
    RNH_CHECK_LAUNCH("rnh_wino44f_wgrad_v");
    hipLaunchKernelGGL(wf_finish_kernel, dim3((unsigned)((long)g.Cx * g.Cy / 64)), dim3(256), 0, st, part, bpart, g.S, g.Cx, g.Cy, rowmap, colmap, Cin, dw,
                       db, accumulate);
    RNH_CHECK_LAUNCH("rnh_wino44f_wgrad_v(finish)");
    return 0;
}
