// 3x3 convolutions (padding 1) in Winograd form F(4x4, 3x3) on fp32 MFMA for gfx950: the ConvLSTM cell (rnh_wino44_cell) and, with a plain-store or
// PixelShuffle epilogue, refine conv1's forward and data gradient and the upsampler's first convolution (rnh_wino44_conv); rnh_wino44_transform.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        per 4x4 output tile, 6x6 input patch d, 3x3 filter g
//
// 36 GEMMs (one per position xi = 6 i + j of the 6x6 transform domain) of [tiles x C] x [C x N]: 2.25 multiplications per output where
// F(2x2, 3x3) (csrc/conv_wino.hip) needs 4 and the direct form 9.  True fp32 arithmetic throughout (transforms, v_mfma_f32_32x32x2_f32,
// gate math); results differ from the other forms by the rounding of the transforms only (tools/wino43_study.py: 0.2 % of every parity bar).
//
// Unlike conv_wino.hip the input transform is NOT fused into the matrix kernel: with 36 positions a wave's share of the transform domain
// is 9 positions x 32 tiles x 32 columns = 144 accumulator registers, and the 60 staging registers + 144 transform instructions per chunk
// that a fused B^T d B needs do not fit beside them (docs/HISTORY.md, round 5).  Instead
//   rnh_wino44_transform  writes V = B^T d B of an NHWC tensor to HBM (2.25 x its bytes; memory-bound, ~20 us per 64-channel tensor of config 2)
//                         in exactly the order the matrix kernel wants it in LDS; every cell output h is transformed ONCE and read by both of
//                         its consumers (the same layer's next frame, the next layer's same frame);
//   rnh_wino44_cell       brings V in by LDS-DMA (buffer_load ... lds: no staging registers, no transform arithmetic) and is, in its main loop,
//                         nothing but operand reads and MFMAs: 0.85 of the fp32 MFMA peak against 0.70 of conv_wino.hip's (tools/probes/
//                         wino44_gemm.hip).  Workgroup = 8 waves = 4 position groups (the 3x3 quarters of the 6x6 domain) x 2 column groups
//                         of 32: 32 tiles (512 pixels) x 64 columns = the four gates of 16 hidden channels (plans.lstm_colmap64).  Weights
//                         U = G g G^T stream from L2 through a register ring nine requests deep (section 8, hazard 2: the DMA requests share
//                         the weights' in-order counter).  Epilogue: every wave keeps a quarter of the (tile, column) entries and receives the
//                         other 27 positions of those from its three partners through LDS, computes A^T M A, activates its gates; the gates
//                         meet in LDS and the 512 threads finish (tile, pixel, 4 channels) items with 16-byte accesses - in two passes of
//                         half the entries, because the exchange of all of them (221 KB) does not fit the LDS.
//   rnh_wino44_conv       the same matrix kernel (template argument EPI) with a store epilogue: bias, then columns -> up to four destination segments
//                         (accumulating or not) or, PixelShuffle fused, -> the (B, rH, rW, cq) tensor; up to 16 transformed sources, each a number
//                         of tile blocks into a tensor that holds several frames (a window's frames are whole tile blocks apart); transposed-
//                         packed weights make it a data gradient.  Refine conv1's forward reads the transformed h' the top layer's cells wrote
//                         anyway - no transform of its own.
// Replaces src/model/nets/refine_net.py:245-265 (ConvLSTMCell.forward: cat, conv, split, sigmoid / tanh, state update), :149 + :170-181 (refine
// conv1 over the window's hidden states) and its autograd, :199-200 (upsampler conv + PixelShuffle) where the plans select it (hipvsr/plans.py,
// HipOps.wino44_ok: H, W multiples of 4, channel counts multiples of 16, an even number of 16-channel chunks).
#include "rnh_common.h"
#include <type_traits>
#include <utility>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4w __attribute__((ext_vector_type(4)));

#define W4_INL __attribute__((always_inline))

template <int... I, class F>
__device__ __forceinline__ void w4_sfor_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>()), ...);
}
template <int N, class F>
__device__ __forceinline__ void w4_sfor(F &&f) {
    w4_sfor_impl(std::make_integer_sequence<int, N>(), f);
}

constexpr int W4_TILES = 32, W4_CH = 16, W4_BUF = 36 * W4_TILES * W4_CH;      // floats of one staged 16-channel chunk of a tile block: 73 728 bytes
constexpr int W4_RB = 9, W4_RA = 3, W4_NQ = 18;                               // weight ring, LDS operand ring, (position, 8-channel block) pairs per chunk
constexpr int W4_TS = 16 * 16 + 16, W4_GS = 16 * W4_TS + 32;                  // gate exchange [gate][tile slot 16][pixel 16][channel 16]: strides that spread a wave's lanes over all banks

__device__ __forceinline__ float w4_tanh(float x) {                           // (as conv_wino.hip's h_tanh: v_exp_f32 / v_rcp_f32, no branch)
    const float ax = fabsf(x);
    const float big = __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * ax)), 1.f);
    const float small = ax * __builtin_fmaf(-0.33333334f * ax, ax, 1.f);
    return copysignf(ax < 0.04f ? small : big, x);
}

__device__ __forceinline__ i32x4 w4_desc(const void *p) {
    const unsigned long long u = (unsigned long long)p;
    i32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu));
    d[1] = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffu));
    d[2] = 0x7fffffff;
    d[3] = 0x00020000;
    return d;
}

// tile t of the list -> (image, tile row, tile column); 8 x 4 blocks of tiles where the grid allows it (a tile block = 32 x 16 pixels)
__device__ __forceinline__ void w4_tile_xy(int t, int TX, int TY, int &img, int &ty, int &tx) {
    img = t / (TY * TX);
    const int rem = t - img * TY * TX;
    if (!(TX & 7) && !(TY & 3)) {
        const int bi = rem >> 5, wi = rem & 31, bpr = TX >> 3, by = bi / bpr, bx = bi - by * bpr;
        ty = by * 4 + (wi >> 3);
        tx = bx * 8 + (wi & 7);
    } else {
        ty = rem / TX;
        tx = rem - ty * TX;
    }
}

// U[s8][xi][n][kh][m] = (G g G^T)[xi] of (output column colmap[n], input channel kch[8 s8 + 4 kh + m]); transposed: of the data gradient's filter
__global__ void wino44_pack_kernel(const float *w, const float *bias, float *wp, float *biasp, const int *kch, const int *kcoff, const int *colmap, int K, int Npad,
                                   int Cin, int transposed) {
    const long total = (long)(K / 8) * 36 * Npad * 8;
    const float G[6][3] = {{0.25f, 0.f, 0.f},          {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total + Npad; e += (long)gridDim.x * blockDim.x) {
        if (e >= total) {
            const int n = (int)(e - total);
            if (biasp) biasp[n] = (bias && !transposed && colmap[n] >= 0) ? bias[colmap[n]] : 0.f;
            continue;
        }
        const int km = (int)(e & 7), n = (int)((e >> 3) % Npad), xi = (int)((e / (8L * Npad)) % 36), s8 = (int)(e / (8L * Npad * 36));
        // (kcoff: per K slot, added to the column's channel - rnh_pack_weights' kcoff: the window slot of refine conv1's data gradient)
        const int col = colmap[n] < 0 ? -1 : colmap[n] + (kcoff ? kcoff[8 * s8 + km] : 0), c = kch[8 * s8 + km];
        float v = 0.f;
        if (col >= 0 && c >= 0) {
            // (transposed: the data gradient - the K slot is the forward weight's OUTPUT channel, the column its input channel, the taps flipped)
            const float *g = transposed ? w + ((long)c * Cin + col) * 9 : w + ((long)col * Cin + c) * 9;
            const int i = xi / 6, j = xi % 6;
            // (double: the products of sixths and twenty-fourths are rounded once)
            double a = 0.0;
            for (int p = 0; p < 3; ++p)
                for (int q = 0; q < 3; ++q) a += (double)G[i][p] * (double)G[j][q] * (double)g[transposed ? 8 - (p * 3 + q) : p * 3 + q];
            v = (float)a;
        }
        wp[e] = v;
    }
}

// V = B^T d B: thread = (tile, 4 consecutive channels); a wave = 16 tiles x the 4 pieces of one 16-channel chunk, so that every one of its 36
// stores is a contiguous kilobyte of the image [xi][tile][piece ^ swizzle][4] (the swizzle keeps the matrix kernel's ds_read_b128 conflict-free)
__global__ void __launch_bounds__(256) wino44_transform_kernel(const float *x, const int C, const int c0, const int nchunks, const int B, const int H, const int W,
                                                               const int TX, const int TY, const int ntiles, const int MT, float *V) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int half = gw & 1, rest = gw >> 1, chunk = rest % nchunks, mt = rest / nchunks;
    if (mt >= MT) return;
    const int tile = half * 16 + (lane >> 2), piece = lane & 3;
    const int t = mt * W4_TILES + tile;
    float *o = V + ((long)mt * nchunks + chunk) * W4_BUF + tile * 16 + ((piece ^ ((tile >> 2) & 3)) * 4);
    if (t >= ntiles) {
        for (int xi = 0; xi < 36; ++xi) *reinterpret_cast<f32x4w *>(o + xi * (W4_TILES * 16)) = f32x4w{0.f, 0.f, 0.f, 0.f};
        return;
    }
    int img, ty, tx;
    w4_tile_xy(t, TX, TY, img, ty, tx);
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    const float *xp = x + c0 + chunk * 16 + piece * 4;
    f32x4w d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int y = y0 + i, xx = x0 + j;
            const bool in = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
            const long pix = ((long)img * H + (in ? y : 0)) * W + (in ? xx : 0);
            const f32x4w v = *reinterpret_cast<const f32x4w *>(xp + pix * C);
            d[i][j] = in ? v : f32x4w{0.f, 0.f, 0.f, 0.f};
        }
    // B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
    auto bt6 = [](const f32x4w d0, const f32x4w d1, const f32x4w d2, const f32x4w d3, const f32x4w d4, const f32x4w d5, f32x4w *r) W4_INL {
        const f32x4w a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = d3 - d1;
        r[0] = 4.f * d0 - 5.f * d2 + d4;
        r[1] = a + b;
        r[2] = a - b;
        r[3] = c + 2.f * e;
        r[4] = c - 2.f * e;
        r[5] = 4.f * d1 - 5.f * d3 + d5;
    };
    f32x4w tq[6][6];                                                            // T = B^T d: column by column
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x4w r[6];
        bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], r);
#pragma unroll
        for (int i = 0; i < 6; ++i) tq[i][j] = r[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f32x4w r[6];
        bt6(tq[i][0], tq[i][1], tq[i][2], tq[i][3], tq[i][4], tq[i][5], r);
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4w *>(o + (6 * i + j) * (W4_TILES * 16)) = r[j];
    }
}

// The gate backward of a ConvLSTM cell (rnh_lstm_gates_bwd: autograd of reference src/model/nets/refine_net.py:258-265) AND V = B^T dG B of the gate
// gradients it produces, in one launch (round 6): the cell's data gradient in F(4x4, 3x3) form (rnh_wino44_conv on transposed weights) reads its 4 hd input
// channels in transform-domain form, and a rnh_wino44_transform launch of their own re-read all of them from HBM (107 us at BASELINE config 2, which ate
// the 126 us the form saves).  Here a workgroup owns HALF a tile block - 16 tiles = 8 x 32 pixels of one image - and 16 hidden channels: phase 1 computes
// the four gate gradients of its 10 x 34 HALO'd pixels (1.33 x the element-wise work: the 6x6 patches of neighbouring tiles overlap; zeros outside the
// image = the convolution's padding) into LDS, and stores those of its own 8 x 32 pixels (and dc_prev) to the tensors the weight gradient and the chain's
// next frame read; phase 2 = wino44_transform_kernel's arithmetic on the LDS image: wave = gate, lane = (tile, 4-channel piece), every store a contiguous
// kilobyte of the image rnh_wino44_conv copies to LDS.  The blocked tile geometry only (W % 32 == 0, H % 16 == 0: rnh_wino44_gates_bwd_supported).
constexpr int W4G_PITCH = 68, W4G_COLS = 34, W4G_ROWS = 10, W4G_PX = W4G_COLS * W4G_ROWS;       // floats per LDS pixel: 4 gates x 16 channels + 4 (conflict-free 16-byte reads)

__global__ void __launch_bounds__(256) wino44_gates_bwd_kernel(const float *__restrict__ dh, const float *__restrict__ dh2, const float *__restrict__ dcn,
                                                               const float *__restrict__ gates, const float *__restrict__ cprev, const float *__restrict__ cnext,
                                                               float *__restrict__ dgates, float *__restrict__ dcprev, float *__restrict__ V, const int H, const int W,
                                                               const int hd, const int TX, const int TY, const int nblocks) {
    __shared__ __attribute__((aligned(16))) float sm[W4G_PX * W4G_PITCH];         // 92 480 bytes
    const int tid = threadIdx.x, ncg = hd >> 4;
    const int bid = rnh_xcd_remap((int)blockIdx.x, nblocks);                        // (the channel groups of a half block on one XCD: they share every 256-byte pixel row)
    const int cg = bid % ncg, half = (bid / ncg) & 1, mt = bid / (2 * ncg);
    const int t0 = mt * W4_TILES, img = t0 / (TX * TY), rem = t0 - img * TX * TY, bi = rem >> 5, bpr = TX >> 3, by = bi / bpr, bx = bi - by * bpr;
    const int y0 = (by * 4 + 2 * half) * 4, x0 = bx * 32;                           // this half block's 8 x 32 pixels
    // ---- phase 1: item = (halo pixel, 4-channel piece), 1360 of them: up to six per thread.  ALL of a thread's loads are requested before the first
    // value is used (one workgroup of four waves per CU - the LDS image is 92 KB - hides no latency by occupancy: 54 requests of 16 bytes in flight per lane do)
    constexpr int NIT = (W4G_PX * 4 + 255) / 256;
    f32x4w vdh[NIT], vd2[NIT], vdc[NIT], vcp[NIT], vcn[NIT], gi[NIT], gf[NIT], go[NIT], gg[NIT];
    long oo[NIT];
    int pp[NIT];
    const f32x4w z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NIT; ++r) {
        const int it = tid + 256 * r, q = it & 3, p = it >> 2, pr = p / W4G_COLS, pc = p - pr * W4G_COLS;
        const int y = y0 - 1 + pr, x = x0 - 1 + pc;
        const bool in = it < W4G_PX * 4 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const long pix = ((long)img * H + (in ? y : 0)) * W + (in ? x : 0);
        const long o = pix * hd + cg * 16 + q * 4, og = pix * 4 * hd + cg * 16 + q * 4;
        oo[r] = in ? o : -1;
        pp[r] = it < W4G_PX * 4 ? (p | ((pr >= 1 && pr <= 8 && pc >= 1 && pc <= 32) ? 0x10000 : 0)) : -1;      // halo pixel | "one of this workgroup's own"
        vdh[r] = vd2[r] = vdc[r] = vcp[r] = vcn[r] = gi[r] = gf[r] = go[r] = gg[r] = z4;
        if (in) {
            vdh[r] = *reinterpret_cast<const f32x4w *>(dh + o);
            vcn[r] = *reinterpret_cast<const f32x4w *>(cnext + o);
            gi[r] = *reinterpret_cast<const f32x4w *>(gates + og);
            gf[r] = *reinterpret_cast<const f32x4w *>(gates + og + hd);
            go[r] = *reinterpret_cast<const f32x4w *>(gates + og + 2 * hd);
            gg[r] = *reinterpret_cast<const f32x4w *>(gates + og + 3 * hd);
            if (dh2) vd2[r] = *reinterpret_cast<const f32x4w *>(dh2 + o);
            if (dcn) vdc[r] = *reinterpret_cast<const f32x4w *>(dcn + o);
            if (cprev) vcp[r] = *reinterpret_cast<const f32x4w *>(cprev + o);
        }
    }
#pragma unroll
    for (int r = 0; r < NIT; ++r) {
        if (pp[r] < 0) continue;
        const int p = pp[r] & 0xffff, q = (tid + 256 * r) & 3;
        f32x4w di, df, dgo, dg, dcp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {                                               // (the expressions of gates_bwd_m_kernel, mixed_kernels.hip; zeros outside the image)
            const float d = vdh[r][e] + vd2[r][e];
            const float th = tanhf(vcn[r][e]);
            const float d_o = d * th;
            const float dct = vdc[r][e] + d * go[r][e] * (1.f - th * th);
            di[e] = dct * gg[r][e] * gi[r][e] * (1.f - gi[r][e]);
            df[e] = dct * vcp[r][e] * gf[r][e] * (1.f - gf[r][e]);
            dgo[e] = d_o * go[r][e] * (1.f - go[r][e]);
            dg[e] = dct * gi[r][e] * (1.f - gg[r][e] * gg[r][e]);
            dcp[e] = dct * gf[r][e];
        }
        if ((pp[r] & 0x10000) && oo[r] >= 0) {                                       // this workgroup's own pixels
            const long o = oo[r], og = (o - (cg * 16 + q * 4)) * 4 + cg * 16 + q * 4;
            *reinterpret_cast<f32x4w *>(dgates + og) = di;
            *reinterpret_cast<f32x4w *>(dgates + og + hd) = df;
            *reinterpret_cast<f32x4w *>(dgates + og + 2 * hd) = dgo;
            *reinterpret_cast<f32x4w *>(dgates + og + 3 * hd) = dg;
            if (dcprev) *reinterpret_cast<f32x4w *>(dcprev + o) = dcp;
        }
        float *l = sm + p * W4G_PITCH + q * 4;
        *reinterpret_cast<f32x4w *>(l) = di;
        *reinterpret_cast<f32x4w *>(l + 16) = df;
        *reinterpret_cast<f32x4w *>(l + 32) = dgo;
        *reinterpret_cast<f32x4w *>(l + 48) = dg;
    }
    __syncthreads();
    // ---- phase 2: wave = gate, lane = (tile of the half block, piece) ----
    const int gate = tid >> 6, lane = tid & 63, tile = lane >> 2, piece = lane & 3, tr = tile >> 3, tc = tile & 7;
    const float *lp = sm + ((4 * tr) * W4G_COLS + 4 * tc) * W4G_PITCH + gate * 16 + piece * 4;
    f32x4w d[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) d[i][j] = *reinterpret_cast<const f32x4w *>(lp + (i * W4G_COLS + j) * W4G_PITCH);
    auto bt6 = [](const f32x4w d0, const f32x4w d1, const f32x4w d2, const f32x4w d3, const f32x4w d4, const f32x4w d5, f32x4w *r) W4_INL {
        const f32x4w a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = d3 - d1;
        r[0] = 4.f * d0 - 5.f * d2 + d4;
        r[1] = a + b;
        r[2] = a - b;
        r[3] = c + 2.f * e;
        r[4] = c - 2.f * e;
        r[5] = 4.f * d1 - 5.f * d3 + d5;
    };
    f32x4w tq[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x4w r[6];
        bt6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], r);
#pragma unroll
        for (int i = 0; i < 6; ++i) tq[i][j] = r[i];
    }
    const int tb = half * 16 + tile, nchunks = 4 * ncg, chunk = gate * ncg + cg;
    float *o = V + ((long)mt * nchunks + chunk) * W4_BUF + tb * 16 + ((piece ^ ((tb >> 2) & 3)) * 4);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f32x4w r[6];
        bt6(tq[i][0], tq[i][1], tq[i][2], tq[i][3], tq[i][4], tq[i][5], r);
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4w *>(o + (6 * i + j) * (W4_TILES * 16)) = r[j];
    }
}

constexpr int W4_MAX_SRC = 16;
constexpr int W4_EPI_LSTM = 0, W4_EPI_STORE = 1;

struct w4_args {                          // the device view of rnh_wino44_cell_args_t / rnh_wino44_conv_args_t
    const float *v[W4_MAX_SRC];           // transformed sources in K order, each already advanced by its tile-block offset
    int vchunks[W4_MAX_SRC];
    int nsrc, nchunks, B, H, W;
    const float *wp, *bias;
    int Npad, hd;
    const float *c_prev;                  // LSTM epilogue
    float *h_out, *c_out, *gates_out;
    struct {                              // STORE epilogue: consecutive column ranges -> channels [c0, c0 + ncols) of NHWC tensors of C channels
        float *ptr;
        int C, c0, ncols, accumulate;
    } dst[RNH_MAX_DST];
    int ndst;
    int ps_r, ps_cq;                      // > 0: nn.PixelShuffle(ps_r) fused into the store - column n = (i r + j) cq + c -> pixel (r y + i, r x + j), channel c of dst[0]
};

// PP, nA: ONE launch may serve TWO calls of equal geometry (rnh_wino44_cell_pair: the cells of the two directions at small images, where a call alone
// leaves half the chip idle): workgroups [0, nA) belong to PP.call[0], the rest to PP.call[1] (an offset into the kernel-argument segment, no copy)
struct w4_pair {
    w4_args call[2];
};

template <int EPI>
__global__ void __launch_bounds__(512, 1) wino44_kernel(const w4_pair PP, const int nA, const int MT, const int NT, const int TX, const int TY) {
    const int second = (int)blockIdx.x >= nA;
    const w4_args &P = PP.call[second];
    __shared__ __attribute__((aligned(16))) float stage[2 * W4_BUF];          // 147 456 bytes: two staged chunks; the epilogue's exchange areas afterwards
    __shared__ int tpix[W4_TILES];                                            // top-left output pixel of the block's tiles, -1: no such tile
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3, cg = wave >> 2;                                  // quarter of the transform domain, column group
#ifdef W4_STAGGER
    // experiment (round 6; never defined in the product build): every other workgroup of the FIRST round starts W4_STAGGER x 64 x 100 cycles late, so that
    // afterwards half the CUs run their epilogue (stores) while the other half run their main loop
    if (blockIdx.x < 256 && (blockIdx.x & 8)) {
        for (int i = 0; i < W4_STAGGER; ++i) __builtin_amdgcn_s_sleep(100);
    }
#endif
    const int bid = rnh_xcd_remap(second ? (int)blockIdx.x - nA : (int)blockIdx.x, MT * NT);                  // (column block fastest: the NT workgroups of a tile block share an L2)
    const int mt = bid / NT, nt = bid - mt * NT;
    const int H = P.H, W = P.W, ntiles = P.B * TY * TX, m0 = mt * W4_TILES, nchunks = P.nchunks;
    if (tid < W4_TILES) {
        int img, ty, tx;
        const int t = m0 + tid;
        w4_tile_xy(t < ntiles ? t : 0, TX, TY, img, ty, tx);
        tpix[tid] = t < ntiles ? (img * H + 4 * ty) * W + 4 * tx : -1;
    }

    const unsigned lds0 = (unsigned)(size_t)stage;
    // ---- V: LDS-DMA, nine 16-byte pieces per thread and chunk (lane slot = M0 + 16 lane) ------------------------------------------------------
    const int vvoff = tid * 16;
    auto dma = [&, &vvoff = vvoff, &lds0 = lds0, &wave = wave](int buf, const i32x4 &vd, int blockoff, auto d_tag) W4_INL {
        constexpr int d = decltype(d_tag)::value;
        const unsigned ld = __builtin_amdgcn_readfirstlane(lds0 + buf * W4_BUF * 4 + (d * 512 + wave * 64) * 16);
        const int soff = __builtin_amdgcn_readfirstlane(blockoff + d * 8192);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ld), "v"(vvoff), "s"(vd), "s"(soff) : "memory");
    };
    // the chunks of the K dimension run through the sources in order: (source, chunk of the source) of the chunk whose image is requested next
    int src_i = 0, src_c = 0;
    auto chunk_src = [&](i32x4 &vd, int &blockoff) W4_INL {                    // descriptor and byte offset of that chunk's image; advances
        const int nl = P.vchunks[src_i];
        vd = w4_desc(P.v[src_i] + (long)mt * nl * W4_BUF);                   // (the tile block's images: 64-bit, so a transformed tensor may be of any size)
        blockoff = src_c * (W4_BUF * 4);
        if (++src_c == nl) {
            src_c = 0;
            src_i = src_i + 1 < P.nsrc ? src_i + 1 : src_i;                    // (past the end: never requested)
        }
    };
    // ---- A operand: lane (tile l31, k half kh), position p of this wave's nine, 8-channel block kb: 16 bytes = channels 8 kb + 4 kh + m -------
    const int I0 = 3 * (pg >> 1), J0 = 3 * (pg & 1);
    const int sw = (l31 >> 2) & 3;
    unsigned avl[2][2];                                                       // [buffer][kb]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) avl[b][kb] = lds0 + b * W4_BUF * 4 + (6 * I0 + J0) * 2048 + l31 * 64 + (((kb * 2 + kh) ^ sw) * 16);
    f32x4w aq[W4_RA];
    auto loada = [&](int buf, auto g_tag) W4_INL {                             // g = pair of the chunk (0 .. 17)
        constexpr int g = decltype(g_tag)::value, p = g % 9, kb = g / 9;
        constexpr int off = ((p / 3) * 6 + (p % 3)) * 2048;
        auto &aqr = aq;                                                       // (a generic lambda captures only what a non-dependent expression names)
        auto &avlr = avl;
        asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(aqr[g % W4_RA]) : "v"(avlr[buf][kb]), "i"(off) : "memory");
    };
    // ---- B operand: U[s8][xi][n][kh][4] -----------------------------------------------------------------------------------------------------
    const i32x4 udesc = w4_desc(P.wp);
    const int pstride = P.Npad * 32;                                          // bytes of one (8-channel block, position) slab
    int bvoff[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) bvoff[p] = (6 * (I0 + p / 3) + J0 + p % 3) * pstride + ((nt * 64 + cg * 32 + l31) * 2 + kh) * 16;
    const int s8max = 2 * nchunks - 1;
    f32x4w bq[W4_RB];
    auto loadb = [&, &udesc = udesc, &pstride = pstride, &s8max = s8max](int chunk, auto g_tag) W4_INL {   // g relative to the chunk's first pair (0 .. 25)
        constexpr int g = decltype(g_tag)::value, p = g % 9, kbr = g / 9;
        const int s8 = min(2 * chunk + kbr, s8max);                            // (past the end: a valid request nobody uses - the counts stay static)
        const int soff = __builtin_amdgcn_readfirstlane(s8 * 36 * pstride);
        auto &bqr = bq;
        auto &bvr = bvoff;
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(bqr[g % W4_RB]) : "v"(bvr[p]), "s"(udesc), "s"(soff) : "memory");
    };
    f32x16 acc[9];

    // ---- prologue ------------------------------------------------------------------------------------------------------------------------
    {
        i32x4 vd;
        int bo;
        chunk_src(vd, bo);
        w4_sfor<9>([&](auto d) W4_INL { dma(1, vd, bo, d); });
    }
    w4_sfor<8>([&](auto g) W4_INL { loadb(0, g); });
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the first chunk has landed (the 8 weight requests are younger), tpix is written
    loada(1, std::integral_constant<int, 0>());
    loada(1, std::integral_constant<int, 1>());
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[p][v] = 0.f;

    // ---- what the epilogue needs from memory: the bias of the lane's column and the previous cell state of the thread's four items; requested
    // behind the last chunk's last weight request (in-order completion: any earlier and the counted waits would sit them out) -----------------
    const int ncol = nt * 64 + cg * 32 + l31;
    const int hd = P.hd;
    // Both arrive by LDS-DMA like V, in slots of the staging buffer the last chunk does not read (buffer 1, behind the epilogue's exchange area):
    // no register is the target of a request in flight here.  (The first version requested them into registers with asm loads, as conv_wino.hip
    // does: at 251 registers hipcc spilled one target right behind its request and, in another build, copied the targets in front of their wait -
    // stale data or a wild address; tests/test_isa_guards.py.)
    constexpr int W4_EXCH = 8 * 3 * 5 * 1024;                                 // bytes of the exchange area
    constexpr int W4_LBIAS = W4_EXCH, W4_LSTATE = 2 * W4_BUF * 4 - 2 * 512 * 16;   // bias: a 256-byte slot per wave; state: [item q][thread] 16 bytes
    static_assert(W4_LBIAS + 8 * 256 <= W4_LSTATE && W4_LSTATE >= W4_BUF * 4, "the epilogue's requests live in buffer 1 behind the exchange area");
    int item_o[4];                                                             // pixel index of the items (pass s, q): it = q * 512 + tid
    auto item_geo = [&](int s, int q, int &tslot, int &px, int &c4, int &tile) W4_INL {
        const int it = q * 512 + tid;
        tslot = it >> 6, px = (it >> 2) & 15, c4 = it & 3;
        tile = 8 * (tslot >> 2) + 4 * (tslot & 1) + 2 * s + ((tslot >> 1) & 1);   // tile slot = 4 pg + 2 (entry & 1) + kh
    };
    const i32x4 cdesc = w4_desc(EPI == W4_EPI_LSTM ? (P.c_prev ? P.c_prev : P.c_out) : P.wp);   // (no previous state: a valid address, zeros behind the wait)
    auto state_request = [&](int s) W4_INL {                                   // two requests; the thread's own slots (no barrier between request and use)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int tslot, px, c4, tile;
            item_geo(s, q, tslot, px, c4, tile);
            const int tp = tpix[tile];
            item_o[2 * s + q] = (tp < 0 ? 0 : tp) + (px >> 2) * W + (px & 3);
            const int voff = (item_o[2 * s + q] * hd + nt * 16 + c4 * 4) * 4;
            const unsigned ld = __builtin_amdgcn_readfirstlane(lds0 + W4_LSTATE + (q * 512 + wave * 64) * 16);
            // (lgkmcnt(0): the thread's reads of these slots for the previous pass are back)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(ld), "v"(voff), "s"(cdesc) : "memory");
        }
    };
    const i32x4 biasdesc = w4_desc(P.bias ? P.bias : P.wp);                    // (no bias: a valid address, zero behind the wait)
    constexpr int NEPI = EPI == W4_EPI_LSTM ? 3 : 1;                           // requests of the epilogue in the last chunk
    auto epi_request = [&]() W4_INL {
        const unsigned ld = __builtin_amdgcn_readfirstlane(lds0 + W4_LBIAS + wave * 256);
        const int voff = (nt * 64 + lane) * 4;                                 // the 64 columns of the workgroup, once per wave (lane slot = M0 + 4 lane)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(ld), "v"(voff), "s"(biasdesc) : "memory");
        if constexpr (EPI == W4_EPI_LSTM) state_request(0);
    };

    auto chunk_body = [&](const int c, auto buf_tag, auto more_tag) W4_INL {
        constexpr bool more = decltype(more_tag)::value;
        constexpr int buf = decltype(buf_tag)::value;
        i32x4 vd;
        int bo = 0;
        if constexpr (more) chunk_src(vd, bo);
        w4_sfor<W4_NQ>([&](auto q_tag) W4_INL {
            constexpr int q = decltype(q_tag)::value;
            constexpr int p = q % 9;
            if constexpr (more || q + 8 < W4_NQ) loadb(c, std::integral_constant<int, q + 8>());
            // outstanding requests younger than this step's weights (requested 8 steps ago): the weight requests since (8; in the last chunk
            // they stop at its end), the DMA requests of steps 0 .. 8 of a chunk that has a successor, and in the last chunk the epilogue's
            // NEPI (issued at its step 9, behind the last weight request)
            constexpr int lo = q - 8 > 0 ? q - 8 : 0, hi = q - 1 < 8 ? q - 1 : 8;
            constexpr int nd = more && hi >= lo ? hi - lo + 1 : 0;
            constexpr int nw = more ? 8 : (W4_NQ - 1 - q < 8 ? W4_NQ - 1 - q : 8);
            constexpr int ne = !more && q >= 9 ? NEPI : 0;
            if constexpr (!more && q == 9) epi_request();
            if constexpr (q == 16) {
                // every DMA of the next chunk has landed (its youngest is older than the weight request of step 9), all LDS reads of this buffer
                // are back; behind the barrier the other buffer is complete and this one free
                asm volatile("s_waitcnt vmcnt(%c3) lgkmcnt(0)\n\ts_barrier" : "+v"(bq[q % W4_RB]), "+v"(aq[q % W4_RA]), "+v"(aq[(q + 1) % W4_RA]) : "i"(nw + ne) : "memory");
                if constexpr (more) loada(buf ^ 1, std::integral_constant<int, 0>());
            } else if constexpr (q == 17) {
                if constexpr (more) loada(buf ^ 1, std::integral_constant<int, 1>());
                if constexpr (more) asm volatile("s_waitcnt vmcnt(%c2) lgkmcnt(2)" : "+v"(bq[q % W4_RB]), "+v"(aq[q % W4_RA]) : "i"(nw + nd + ne) : "memory");
                else asm volatile("s_waitcnt vmcnt(%c2) lgkmcnt(0)" : "+v"(bq[q % W4_RB]), "+v"(aq[q % W4_RA]) : "i"(nw + nd + ne) : "memory");
            } else {
                loada(buf, std::integral_constant<int, q + 2>());
                asm volatile("s_waitcnt vmcnt(%c2) lgkmcnt(2)" : "+v"(bq[q % W4_RB]), "+v"(aq[q % W4_RA]) : "i"(nw + nd + ne) : "memory");
            }
            // (asm: hipcc's scheduler otherwise moves the MFMAs across the requests and waits and copies ring registers to do so)
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, %3, %4, %0\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, %5, %6, %0\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, %7, %8, %0"
                         : "+v"(acc[p])
                         : "v"(aq[q % W4_RA].x), "v"(bq[q % W4_RB].x), "v"(aq[q % W4_RA].y), "v"(bq[q % W4_RB].y), "v"(aq[q % W4_RA].z), "v"(bq[q % W4_RB].z),
                           "v"(aq[q % W4_RA].w), "v"(bq[q % W4_RB].w));
            if constexpr (more && q < 9) dma(buf ^ 1, vd, bo, std::integral_constant<int, q>());
        });
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    for (int c = 0; c + 2 < nchunks; c += 2) {                                 // (nchunks is even: the buffer of a chunk is a compile-time constant; even chunks: buffer 1)
        chunk_body(c, B1(), std::true_type());
        chunk_body(c + 1, B0(), std::true_type());
    }
    chunk_body(nchunks - 2, B1(), std::true_type());
    chunk_body(nchunks - 1, B0(), std::false_type());                         // (the last chunk runs out of buffer 0: buffer 1 takes the epilogue's requests)

    // ---- epilogue, two passes s over the entry pairs e = 2 s + e2: wave pg owns the entries v = 4 pg + e (tiles 8 pg + 4 kh + e) ---------------
    asm volatile("s_barrier" ::: "memory");                                   // (nobody reads the staging buffers any more)
    f32x4w *px4 = reinterpret_cast<f32x4w *>(stage);
    float *xg = stage;
    const int gate = 2 * cg + (l31 >> 4), gch = l31 & 15;
    const float gm = gate == 3 ? 2.f : 1.f, gb = gate == 3 ? -1.f : 0.f;      // sigmoid, and tanh as 2 sigmoid(2x) - 1: m rcp(1 + exp(-m x)) + b
    // A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
    auto at6 = [](const float m0_, const float m1, const float m2, const float m3, const float m4, const float m5, float *y) W4_INL {
        const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
        y[0] = m0_ + s1 + s2;
        y[1] = __builtin_fmaf(2.f, d2, d1);
        y[2] = __builtin_fmaf(4.f, s2, s1);
        y[3] = __builtin_fmaf(8.f, d2, d1) + m5;
    };
    auto epilogue = [&](auto pg_tag) W4_INL {
        constexpr int PG = decltype(pg_tag)::value;
        constexpr int PI0 = 3 * (PG >> 1), PJ0 = 3 * (PG & 1);
        auto pass = [&](auto s_tag) W4_INL {
            constexpr int S = decltype(s_tag)::value;
            // -- exchange: to partner r the 18 values (9 positions x 2 entries) of its entries: four 16-byte units and one of 8
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r == PG) continue;
                const int sidx = PG < r ? PG : PG - 1;                         // this wave's slot among r's three senders
                const int base = (((cg * 4 + r) * 3 + sidx) * 5) * 64 + lane;
                const int v0 = 4 * r + 2 * S;
#pragma unroll
                for (int u = 0; u < 4; ++u) px4[base + u * 64] = f32x4w{acc[2 * u][v0], acc[2 * u][v0 + 1], acc[2 * u + 1][v0], acc[2 * u + 1][v0 + 1]};
                *reinterpret_cast<f32x2 *>(&px4[base + 4 * 64]) = f32x2{acc[8][v0], acc[8][v0 + 1]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            float M[2][36];
#pragma unroll
            for (int p = 0; p < 9; ++p)
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) M[e2][6 * (PI0 + p / 3) + PJ0 + p % 3] = acc[p][4 * PG + 2 * S + e2];
#pragma unroll
            for (int sd = 0; sd < 4; ++sd) {
                if (sd == PG) continue;
                const int sidx = sd < PG ? sd : sd - 1;
                const int base = (((cg * 4 + PG) * 3 + sidx) * 5) * 64 + lane;
                const int SI0 = 3 * (sd >> 1), SJ0 = 3 * (sd & 1);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4w v4 = px4[base + u * 64];
                    const int pa = 6 * (SI0 + (2 * u) / 3) + SJ0 + (2 * u) % 3, pb = 6 * (SI0 + (2 * u + 1) / 3) + SJ0 + (2 * u + 1) % 3;
                    M[0][pa] = v4.x; M[1][pa] = v4.y; M[0][pb] = v4.z; M[1][pb] = v4.w;
                }
                const f32x2 v2 = *reinterpret_cast<const f32x2 *>(&px4[base + 4 * 64]);
                M[0][6 * (SI0 + 2) + SJ0 + 2] = v2.x; M[1][6 * (SI0 + 2) + SJ0 + 2] = v2.y;
            }
            // -- Y = A^T M A, bias, activation
            float Y[2][16];
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                float R[4][6];                                                  // R[a][j] = sum_i At[a][i] M[i][j]
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    float y[4];
                    at6(M[e2][j], M[e2][6 + j], M[e2][12 + j], M[e2][18 + j], M[e2][24 + j], M[e2][30 + j], y);
#pragma unroll
                    for (int a = 0; a < 4; ++a) R[a][j] = y[a];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) at6(R[a][0], R[a][1], R[a][2], R[a][3], R[a][4], R[a][5], &Y[e2][4 * a]);
            }
            float bv;
            {
                if constexpr (S == 0) asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(NEPI - 1) : "memory");     // the bias has landed, the state may still be on its way
                bv = P.bias ? stage[W4_LBIAS / 4 + wave * 64 + cg * 32 + l31] : 0.f;
            }
            if constexpr (EPI == W4_EPI_STORE) {
                // plain store of the lane's column: 32 pixels of 2 tiles per pass (a wave's store covers 32 consecutive channels of 2 pixels)
                if (P.ps_r) {
                    const int r = P.ps_r, cq = P.ps_cq;
                    if (ncol < cq * r * r) {
                        const int sub = ncol / cq, c = ncol - sub * cq, pi = sub / r, pj = sub - pi * r;
                        float *dp = P.dst[0].ptr + c;
                        const long Wr = (long)W * r;
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            const int tp = tpix[8 * PG + 4 * kh + 2 * S + e2];
                            if (tp < 0) continue;
                            const int row = tp / W, x0 = tp - row * W;      // row = image * H + y: the images' rows follow each other in both tensors
#pragma unroll
                            for (int k = 0; k < 16; ++k)
                                dp[((long)(row + (k >> 2)) * r + pi) * Wr * cq + ((long)(x0 + (k & 3)) * r + pj) * cq] = Y[e2][k] + bv;
                        }
                    }
                    if constexpr (S == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    return;
                }
                int seg = -1, cbase = 0;                                    // destination segment of this lane's column
                for (int dd = 0; dd < P.ndst; ++dd) {
                    if (seg < 0 && ncol < cbase + P.dst[dd].ncols) seg = dd;
                    if (seg < 0) cbase += P.dst[dd].ncols;
                }
                if (seg >= 0) {
                    const int dC = P.dst[seg].C;
                    const bool accum = P.dst[seg].accumulate != 0;
                    float *dp = P.dst[seg].ptr + P.dst[seg].c0 + (ncol - cbase);
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const int tp = tpix[8 * PG + 4 * kh + 2 * S + e2];
                        if (tp < 0) continue;
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            float *o = dp + (long)(tp + (k >> 2) * W + (k & 3)) * dC;
                            *o = accum ? *o + Y[e2][k] + bv : Y[e2][k] + bv;
                        }
                    }
                }
                // (all reads of the exchange area are back: the next pass writes over it)
                if constexpr (S == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                return;
            }
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2)
#pragma unroll
                for (int k = 0; k < 16; ++k) Y[e2][k] = __builtin_fmaf(gm, __builtin_amdgcn_rcpf(1.f + __expf(-gm * (Y[e2][k] + bv))), gb);
            // (all reads of the exchange area are back: the gates go over it)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                float *xw = xg + gate * W4_GS + (PG * 4 + e2 * 2 + kh) * W4_TS + gch;
#pragma unroll
                for (int k = 0; k < 16; ++k) xw[k * 16] = Y[e2][k];
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // -- items (tile, pixel, 4 hidden channels): c' = f c + i g, h' = o tanh c'
            // (pass 0: requested in the last chunk; pass 1: requested below, a whole pass ago.  Also a wait for this thread's stores of pass 0.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            f32x4w cpq[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) cpq[q] = *reinterpret_cast<const f32x4w *>(stage + W4_LSTATE / 4 + (q * 512 + tid) * 4);
            if (!P.c_prev) cpq[0] = cpq[1] = f32x4w{0.f, 0.f, 0.f, 0.f};
            if constexpr (S == 0) state_request(1);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                int tslot, px, c4, tile;
                item_geo(S, q, tslot, px, c4, tile);
                if (tpix[tile] < 0) continue;
                const float *xi = xg + tslot * W4_TS + px * 16 + c4 * 4;
                const f32x4w gi = *reinterpret_cast<const f32x4w *>(xi), gf = *reinterpret_cast<const f32x4w *>(xi + W4_GS);
                const f32x4w go = *reinterpret_cast<const f32x4w *>(xi + 2 * W4_GS), gg = *reinterpret_cast<const f32x4w *>(xi + 3 * W4_GS);
                const long po = item_o[2 * S + q];
#ifdef W4_DIAG_NOSTORE                                                         // (diagnostic build: how long is the epilogue without its HBM stores?)
                if (P.gates_out == P.c_out) {
#else
                if (P.gates_out) {
#endif
                    f32x4w *gp = reinterpret_cast<f32x4w *>(P.gates_out + po * 4 * hd + nt * 16 + c4 * 4);
#ifdef W4_GATES_NT          // experiment: the gates past the caches (nobody reads them before the backward)
                    __builtin_nontemporal_store(gi, gp); __builtin_nontemporal_store(gf, gp + hd / 4);
                    __builtin_nontemporal_store(go, gp + 2 * (hd / 4)); __builtin_nontemporal_store(gg, gp + 3 * (hd / 4));
#else
                    gp[0] = gi; gp[hd / 4] = gf; gp[2 * (hd / 4)] = go; gp[3 * (hd / 4)] = gg;
#endif
                }
                f32x4w cn, hn;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cn[j] = gf[j] * cpq[q][j] + gi[j] * gg[j];
                    hn[j] = go[j] * w4_tanh(cn[j]);
                }
                const long o = po * hd + nt * 16 + c4 * 4;
#ifdef W4_DIAG_NOSTORE
                if (cn[0] + hn[1] == 1.2345f)
#endif
                {
                    *reinterpret_cast<f32x4w *>(P.c_out + o) = cn;
                    *reinterpret_cast<f32x4w *>(P.h_out + o) = hn;
                }
            }
            if constexpr (S == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (the gates are read: the next exchange goes over them)
        };
        pass(std::integral_constant<int, 0>());
        pass(std::integral_constant<int, 1>());
    };
    if (pg == 0) epilogue(std::integral_constant<int, 0>());
    else if (pg == 1) epilogue(std::integral_constant<int, 1>());
    else if (pg == 2) epilogue(std::integral_constant<int, 2>());
    else epilogue(std::integral_constant<int, 3>());
}

inline int w4_grid(long n, int cap = 8192) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

static int w4_geometry(int B, int H, int W, const char *who, int &TX, int &TY, long &ntiles, int &MT) {
    if (B < 1 || H < 4 || W < 4) RNH_FAIL(RNH_E_ARG, "%s: bad geometry", who);
    if ((H & 3) || (W & 3)) RNH_FAIL(RNH_E_RANGE, "%s: H and W must be multiples of 4 (whole 4x4 tiles)", who);
    TX = W / 4, TY = H / 4;
    ntiles = (long)B * TY * TX;
    if ((long)B * H * W >= (1L << 27)) RNH_FAIL(RNH_E_RANGE, "%s: too many pixels for 32-bit offsets", who);
    MT = (int)((ntiles + W4_TILES - 1) / W4_TILES);
    return 0;
}

}  // namespace

extern "C" int64_t rnh_wino44_v_floats(int B, int H, int W, int nch) {
    if (B < 1 || H < 1 || W < 1 || nch < 1) return 0;
    const long ntiles = (long)B * ((H + 3) / 4) * ((W + 3) / 4);
    return ((ntiles + W4_TILES - 1) / W4_TILES) * ((nch + 15) / 16) * (int64_t)W4_BUF;
}

extern "C" int rnh_wino44_pack_weights(const float *w, const float *bias, float *wp, float *biasp, const int32_t *kch, const int32_t *kcoff, const int32_t *colmap, int K,
                                       int Npad, int Cout, int Cin, int transposed, void *stream) {
    if (!w || !wp || !kch || !colmap || K < 16 || Npad < 64 || Cout < 1 || Cin < 1) RNH_FAIL(RNH_E_ARG, "rnh_wino44_pack_weights: bad arguments");
    if (K % 32) RNH_FAIL(RNH_E_RANGE, "rnh_wino44_pack_weights: K must be a multiple of 32 (an even number of 16-channel chunks)");
    if (Npad % 64) RNH_FAIL(RNH_E_RANGE, "rnh_wino44_pack_weights: Npad must be a multiple of 64");
    hipLaunchKernelGGL(wino44_pack_kernel, dim3(w4_grid((long)(K / 8) * 36 * Npad * 8 + Npad)), dim3(256), 0, (hipStream_t)stream, w, bias, wp, biasp, kch, kcoff, colmap,
                       K, Npad, Cin, transposed);
    RNH_CHECK_LAUNCH("rnh_wino44_pack_weights");
    return 0;
}

extern "C" int rnh_wino44_transform(const float *x, int C, int c0, int nch, int B, int H, int W, float *v, void *stream) {
    if (!x || !v || C < 1 || c0 < 0 || nch < 1 || c0 + nch > C) RNH_FAIL(RNH_E_ARG, "rnh_wino44_transform: bad arguments");
    if ((C & 3) || (c0 & 3) || (nch & 15)) RNH_FAIL(RNH_E_ALIGN, "rnh_wino44_transform: C, c0 multiples of 4, nch a multiple of 16");
    int TX, TY, MT;
    long ntiles;
    if (int rc = w4_geometry(B, H, W, "rnh_wino44_transform", TX, TY, ntiles, MT)) return rc;
    const int nchunks = nch / 16;
    const long waves = (long)MT * nchunks * 2;
    hipLaunchKernelGGL(wino44_transform_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, C, c0, nchunks, B, H, W, TX, TY, (int)ntiles, MT, v);
    RNH_CHECK_LAUNCH("rnh_wino44_transform");
    return 0;
}

static int w4_launch(const w4_args &p, int epi, int MT, int TX, int TY, hipStream_t st, const char *who, const w4_args *second = nullptr) {
    const int NT = p.Npad / 64, nA = MT * NT;
    w4_pair pp;
    pp.call[0] = p;
    pp.call[1] = second ? *second : p;
    const dim3 grid((unsigned)(second ? 2 * nA : nA));
    if (epi == W4_EPI_LSTM) hipLaunchKernelGGL((wino44_kernel<W4_EPI_LSTM>), grid, dim3(512), 0, st, pp, nA, MT, NT, TX, TY);
    else hipLaunchKernelGGL((wino44_kernel<W4_EPI_STORE>), grid, dim3(512), 0, st, pp, nA, MT, NT, TX, TY);
    RNH_CHECK_LAUNCH(who);
    return 0;
}

static int w4_cell_fill(const rnh_wino44_cell_args_t &a, w4_args &p, int &TX, int &TY, int &MT, const char *who) {
    if (a.nsrc < 1 || a.nsrc > 2 || !a.v[0] || (a.nsrc == 2 && !a.v[1]) || !a.wp || !a.bias || !a.h_out || !a.c_out) RNH_FAIL(RNH_E_ARG, "%s: bad arguments", who);
    long ntiles;
    if (int rc = w4_geometry(a.B, a.H, a.W, who, TX, TY, ntiles, MT)) return rc;
    const int nchunks = a.vchunks[0] + (a.nsrc == 2 ? a.vchunks[1] : 0);
    if (a.vchunks[0] < 1 || (a.nsrc == 2 && a.vchunks[1] < 1) || (nchunks & 1)) RNH_FAIL(RNH_E_RANGE, "%s: an even number of 16-channel chunks", who);
    if (a.hd < 16 || (a.hd & 15) || a.Npad != 4 * a.hd) RNH_FAIL(RNH_E_RANGE, "%s: hidden channels in multiples of 16, Npad = 4 hd (plans.lstm_colmap64)", who);
    if ((long)a.B * a.H * a.W * a.hd * 4 >= (1L << 31)) RNH_FAIL(RNH_E_RANGE, "%s: a cell state of at most 2 GiB", who);
    p = w4_args{};
    for (int i = 0; i < a.nsrc; ++i) p.v[i] = a.v[i], p.vchunks[i] = a.vchunks[i];
    p.nsrc = a.nsrc, p.nchunks = nchunks, p.B = a.B, p.H = a.H, p.W = a.W;
    p.wp = a.wp, p.bias = a.bias, p.Npad = a.Npad, p.hd = a.hd;
    p.c_prev = a.c_prev, p.h_out = a.h_out, p.c_out = a.c_out, p.gates_out = a.gates_out;
    return 0;
}

extern "C" int rnh_wino44_gates_bwd_supported(int H, int W, int hd) { return H > 0 && W > 0 && hd >= 16 && !(H & 15) && !(W & 31) && !(hd & 15); }

extern "C" int rnh_wino44_gates_bwd(const float *dh, const float *dh2, const float *dc_next, const float *gates, const float *c_prev, const float *c_next,
                                    float *dgates, float *dc_prev, float *v, int B, int H, int W, int hd, void *stream) {
    if (!dh || !gates || !c_next || !dgates || !v || B < 1) RNH_FAIL(RNH_E_ARG, "rnh_wino44_gates_bwd: bad arguments");
    if (!rnh_wino44_gates_bwd_supported(H, W, hd)) RNH_FAIL(RNH_E_RANGE, "rnh_wino44_gates_bwd: H %% 16 == 0, W %% 32 == 0, hd %% 16 == 0 (whole 8 x 4 blocks of 4x4 tiles)");
    int TX, TY, MT;
    long ntiles;
    if (int rc = w4_geometry(B, H, W, "rnh_wino44_gates_bwd", TX, TY, ntiles, MT)) return rc;
    const long nblocks = (long)MT * 2 * (hd / 16);
    if (nblocks >= (1L << 30)) RNH_FAIL(RNH_E_RANGE, "rnh_wino44_gates_bwd: grid too large");
    hipLaunchKernelGGL(wino44_gates_bwd_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, dh, dh2, dc_next, gates, c_prev, c_next, dgates, dc_prev, v,
                       H, W, hd, TX, TY, (int)nblocks);
    RNH_CHECK_LAUNCH("rnh_wino44_gates_bwd");
    return 0;
}

extern "C" int rnh_wino44_cell(const rnh_wino44_cell_args_t *args, void *stream) {
    if (!args) RNH_FAIL(RNH_E_ARG, "rnh_wino44_cell: null args");
    w4_args p;
    int TX, TY, MT;
    if (int rc = w4_cell_fill(*args, p, TX, TY, MT, "rnh_wino44_cell")) return rc;
    return w4_launch(p, W4_EPI_LSTM, MT, TX, TY, (hipStream_t)stream, "rnh_wino44_cell");
}

extern "C" int rnh_wino44_cell_pair(const rnh_wino44_cell_args_t *args_a, const rnh_wino44_cell_args_t *args_b, void *stream) {
    if (!args_a || !args_b) RNH_FAIL(RNH_E_ARG, "rnh_wino44_cell_pair: null args");
    w4_args pa, pb;
    int TX, TY, MT, TXb, TYb, MTb;
    if (int rc = w4_cell_fill(*args_a, pa, TX, TY, MT, "rnh_wino44_cell_pair (first call)")) return rc;
    if (int rc = w4_cell_fill(*args_b, pb, TXb, TYb, MTb, "rnh_wino44_cell_pair (second call)")) return rc;
    if (pa.B != pb.B || pa.H != pb.H || pa.W != pb.W || pa.Npad != pb.Npad)
        RNH_FAIL(RNH_E_ARG, "rnh_wino44_cell_pair: the two calls must agree in B, H, W and Npad");
    return w4_launch(pa, W4_EPI_LSTM, MT, TX, TY, (hipStream_t)stream, "rnh_wino44_cell_pair", &pb);
}

extern "C" int rnh_wino44_conv(const rnh_wino44_conv_args_t *args, void *stream) {
    if (!args) RNH_FAIL(RNH_E_ARG, "rnh_wino44_conv: null args");
    const rnh_wino44_conv_args_t &a = *args;
    if (a.nsrc < 1 || a.nsrc > W4_MAX_SRC || !a.wp) RNH_FAIL(RNH_E_ARG, "rnh_wino44_conv: bad arguments");
    int TX, TY, MT;
    long ntiles;
    if (int rc = w4_geometry(a.B, a.H, a.W, "rnh_wino44_conv", TX, TY, ntiles, MT)) return rc;
    if (a.Npad < 64 || a.Npad % 64) RNH_FAIL(RNH_E_RANGE, "rnh_wino44_conv: Npad must be a multiple of 64");
    if (a.ndst < 1 || a.ndst > RNH_MAX_DST) RNH_FAIL(RNH_E_ARG, "rnh_wino44_conv: bad destination count");
    if (a.ps_r < 0 || (a.ps_r > 0 && (a.ndst != 1 || a.ps_cq < 1 || a.ps_cq * a.ps_r * a.ps_r > a.Npad || a.dst[0].img_off != 0)))
        RNH_FAIL(RNH_E_ARG, "rnh_wino44_conv: bad pixel-shuffle destination");
    int cols = 0;
    for (int d = 0; d < a.ndst; ++d) {
        const rnh_dst_t &D = a.dst[d];
        if (!D.ptr || D.ncols < 1 || D.c0 < 0 || D.c0 + D.ncols > D.C || D.img_off < 0) RNH_FAIL(RNH_E_ARG, "rnh_wino44_conv: bad destination %d", d);
        if (!a.ps_r && (long)(a.B + D.img_off) * a.H * a.W * D.C >= (1L << 31)) RNH_FAIL(RNH_E_RANGE, "rnh_wino44_conv: a destination of at most 2^31 elements");
        cols += D.ncols;
    }
    if (cols > a.Npad) RNH_FAIL(RNH_E_ARG, "rnh_wino44_conv: destination columns exceed Npad");
    w4_args p = {};
    int nchunks = 0;
    for (int i = 0; i < a.nsrc; ++i) {
        if (!a.v[i] || a.vchunks[i] < 1 || a.vblock_off[i] < 0) RNH_FAIL(RNH_E_ARG, "rnh_wino44_conv: bad source %d", i);
        p.v[i] = a.v[i] + (long)a.vblock_off[i] * a.vchunks[i] * W4_BUF;        // (a transformed tensor may hold several frames: the launch starts at this tile block)
        p.vchunks[i] = a.vchunks[i];
        nchunks += a.vchunks[i];
    }
    if (nchunks & 1) RNH_FAIL(RNH_E_RANGE, "rnh_wino44_conv: an even number of 16-channel chunks");
    p.nsrc = a.nsrc, p.nchunks = nchunks, p.B = a.B, p.H = a.H, p.W = a.W;
    p.wp = a.wp, p.bias = a.bias, p.Npad = a.Npad;
    p.ndst = a.ndst, p.ps_r = a.ps_r, p.ps_cq = a.ps_cq;
    for (int d = 0; d < a.ndst; ++d) {
        const rnh_dst_t &D = a.dst[d];
        p.dst[d].ptr = D.ptr + (long)D.img_off * a.H * a.W * D.C, p.dst[d].C = D.C, p.dst[d].c0 = D.c0, p.dst[d].ncols = D.ncols, p.dst[d].accumulate = D.accumulate;
    }
    return w4_launch(p, W4_EPI_STORE, MT, TX, TY, (hipStream_t)stream, "rnh_wino44_conv");
}
