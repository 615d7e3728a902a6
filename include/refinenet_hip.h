/*
 * refinenet_hip.h - C ABI of librefinenet_hip.so: the MI355X (gfx950) kernels of the RefineNet
 * forward/backward hot path.
 *
 * Boundary.  The reference implements this path as stock PyTorch operators called from Python
 * (reference src/model/nets/refine_net.py:61-344, src/runner/trainers/acdc_vsr_refinenet_trainer.py:41-47,
 * 76-101).  Each entry point below replaces the ATen operator sequence of one reference call site; the
 * call site it replaces is cited on the declaration.  INTEGRATION.md shows the ctypes binding a reference
 * maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless the parameter comment says "host" (the bf16-storage entry
 *     points at the end carry an explicit element type per tensor);
 *   - activations are NHWC: tensor[b][y][x][c], b = frame * N + n ("image" index), c contiguous;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (no allocation, no
 *     synchronisation, graph-capture safe); workspaces are provided by the caller;
 *   - return value: 0 on success, a negative RNH_E_* code on invalid arguments, a positive hipError_t if
 *     the launch failed.  rnh_last_error() returns a static string for the last failure on this thread.
 *   - channel counts, channel offsets and channel strides must be multiples of 4 (16-byte vector access).
 */
#ifndef REFINENET_HIP_H
#define REFINENET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RNH_ABI_VERSION 7       /* 2 (round 4): the argument constraints of rnh_inconv_prelu_bwd / rnh_wgrad_reduce (narrowed in round 3) are part of the contract; 3: + rnh_outconv_fwd_ld; 4 (round 5): + rnh_conv_bf16_pair, rnh_conv_wino_pair; 5: + rnh_wino44_*; 6 (round 6): + rnh_pack_weights_f16, rnh_conv_bf16_args_t.wp_f16 (was padding), rnh_uptail_fwd_bf16 contracts in f16, rnh_wino44_gates_bwd[_supported], rnh_wino44f_wgrad*; 7: + rnh_wino44f_wgrad_v[_supported] */

#define RNH_E_ARG      (-1)   /* null pointer / non-positive size                                  */
#define RNH_E_ALIGN    (-2)   /* channel count / offset / stride not a multiple of 4               */
#define RNH_E_RANGE    (-3)   /* too many sources / destinations, unsupported tile or mode         */
#define RNH_E_WORKSPACE (-4)  /* workspace too small                                               */

#define RNH_MAX_SRC 16
#define RNH_MAX_DST 4

/* One input operand of an implicit-GEMM convolution: `nch` channels [c0, c0+nch) of an NHWC tensor.
 * The K dimension of the GEMM is the concatenation of all sources (this replaces torch.cat at
 * refine_net.py:170,177,253).  Source image index = output image index + img_off.  With scale = r > 1 the
 * source has r x the output resolution and pixel (y, x) reads (r*y + sub_y, r*x + sub_x): the inverse of
 * nn.PixelShuffle(r), used by the backward of the upsampler (refine_net.py:200,204). */
typedef struct rnh_src {
    const float *ptr;    /* tensor base                                                           */
    const float *ptr2;   /* optional second tensor of the same geometry, added element-wise; or 0 */
    int32_t C;           /* channels per pixel of the tensor (stride)                             */
    int32_t c0;          /* first channel used                                                    */
    int32_t nch;         /* channels used                                                         */
    int32_t img_off;     /* image offset relative to the output image index                       */
    int32_t scale;       /* 1, or r for pixel-unshuffle gathering                                  */
    int32_t sub_y, sub_x;
    int32_t _pad;
} rnh_src_t;

/* One output segment: `ncols` consecutive GEMM columns go to channels [c0, c0+ncols) of an NHWC tensor. */
typedef struct rnh_dst {
    float *ptr;
    int32_t C;           /* channel stride of the destination tensor                              */
    int32_t c0;
    int32_t ncols;
    int32_t accumulate;  /* 0: store, 1: add to what is there                                     */
    int32_t img_off;     /* destination image index = output image index + img_off               */
    int32_t _pad;
} rnh_dst_t;

/* Epilogues of rnh_conv_igemm */
#define RNH_EPI_STORE 0   /* bias add, store / accumulate into the destination segments           */
#define RNH_EPI_PS    1   /* bias add, nn.PixelShuffle(ps_r) fused into the store (dst[0])        */
#define RNH_EPI_LSTM  2   /* bias add, ConvLSTM gate math, writes h, c (and the gates for backward)*/
#define RNH_EPI_LSTM_BWD 3 /* rnh_conv_bf16 only: data gradient of a ConvLSTM cell at frame t + the gate backward of frame t', the frame
                            * back-propagation through time visits NEXT (one time step earlier in the cell's own direction): see
                            * rnh_conv_bf16_args_t.  One term everywhere: t' = "the frame the chain processes next" */

/* Output-tile shapes (rows x columns of one workgroup) */
#define RNH_TILE_128x128 0   /* 2x2 waves of 64x64          */
#define RNH_TILE_128x128_G 1 /* 4x1 waves of 32x128 (LSTM: one wave holds the 4 gates of 32 channels) */
#define RNH_TILE_256x64  2   /* 4x1 waves of 64x64          */
#define RNH_TILE_128x160 3   /* 4x1 waves of 32x160         */
#define RNH_TILE_256x128 6   /* 4x1 waves of 64x128 (rnh_conv_igemm, DIRECT variant only; LSTM-capable) */
#define RNH_WINO_COLS64  64  /* rnh_conv_wino: workgroups of 32 tiles x 64 columns, two per CU (any other value of `tile` means this) */
#define RNH_WINO_COLS128 128 /* rnh_conv_wino: workgroups of 32 tiles x 128 columns (8 waves, 32-channel chunks), one per CU */
#define RNH_TILE_DIRECT  16  /* OR-ed into `tile` for rnh_conv_igemm: fragments straight from global memory, no LDS,
                                no barrier (same results bit for bit as the LDS-staged variant)                  */

typedef struct rnh_conv_args {
    rnh_src_t src[RNH_MAX_SRC];
    int32_t nsrc;
    int32_t B, H, W;          /* output geometry: images, rows, columns                            */
    int32_t ntaps;            /* 9 (3x3, padding 1) or 1 (1x1)                                    */
    int32_t nk;               /* number of 16-channel K steps = sum_src ceil(nch/16) * ntaps       */
    const float *wp;          /* packed weights [nk][Npad][16] from rnh_pack_weights               */
    const float *bias;        /* packed bias [Npad] or 0                                           */
    int32_t Npad;             /* padded column count, a multiple of the tile's column count        */
    int32_t epilogue;         /* RNH_EPI_*                                                         */
    int32_t tile;             /* RNH_TILE_*                                                        */
    int32_t ndst;
    rnh_dst_t dst[RNH_MAX_DST];
    int32_t ps_r;             /* RNH_EPI_PS: upscale factor r; column n = (i*r+j)*ps_cq + c         */
    int32_t ps_cq;            /* RNH_EPI_PS: channels after the shuffle                            */
    /* RNH_EPI_LSTM: column n = tile*128 + gate*32 + j <-> hidden channel tile*32 + j, gate order i,f,o,g */
    int32_t hd;               /* hidden channels                                                   */
    int32_t _pad;
    const float *c_prev;      /* [B][H][W][hd] or 0 (zero state)                                   */
    float *h_out;             /* [B][H][W][hd]                                                     */
    float *c_out;             /* [B][H][W][hd]                                                     */
    float *gates_out;         /* [B][H][W][4*hd] post-activation i,f,o,g (channel = gate*hd + ch) or 0 */
} rnh_conv_args_t;

/* Implicit-GEMM 3x3 / 1x1 convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces nn.Conv2d at refine_net.py:149,151,154 (refine block), :199,203 (upsampler convs followed by
 * PixelShuffle :200,204), :235-241 + :253-265 (ConvLSTM cell: cat, conv, split, sigmoid/tanh, state
 * update) and, with transposed/flipped packed weights, their data gradients (aten::convolution_backward
 * issued by loss.backward(), acdc_vsr_refinenet_trainer.py:46). */
int rnh_conv_igemm(const rnh_conv_args_t *args /* host */, void *stream);

/* Weight re-layout for rnh_conv_igemm: wp[ks][n][kk] = W[o][i][tap] (OIHW, reference state_dict layout)
 *   not transposed: o = colmap[n] + kcoff[ks],     i = kbase[ks] + kk*kstride, tap = ktap[ks]
 *   transposed    : o = kbase[ks] + kk*kstride,    i = colmap[n] + kcoff[ks],  tap = ntaps-1-ktap[ks]
 * zero where colmap[n] < 0 or kk >= knv[ks].  biasp[n] = bias[colmap[n]] (only if bias != 0, not transposed).
 * kbase/knv/ktap/kcoff/colmap are DEVICE int32 arrays (kcoff may be 0 = all zero). */
int rnh_pack_weights(const float *w, const float *bias, float *wp, float *biasp,
                     const int32_t *kbase, const int32_t *knv, const int32_t *ktap, const int32_t *kcoff,
                     const int32_t *colmap, int nk, int Npad, int Cout, int Cin, int ntaps, int kstride,
                     int transposed, void *stream);

/* The same 3x3 convolution in Winograd form F(2x2, 3x3) (csrc/conv_wino.hip): 16 GEMMs in the transform domain, 2.25x
 * fewer MFMA passes; input and output transforms fused (nothing of the transform domain reaches memory).  A wave owns half
 * of the transform domain of 32 tiles x 32 columns (128 accumulators), a workgroup computes 32 tiles x 64 columns and two
 * workgroups share a CU (tile = RNH_WINO_COLS64, or any other value but the next); with tile = RNH_WINO_COLS128 a workgroup is
 * 8 waves = 32 tiles x 128 columns, one per CU, and every nch % 32 == 0, Npad a multiple of 128, RNH_EPI_LSTM in the column
 * order plans.lstm_colmap (blocks of 128 = the gates i, f, o, g of 32 hidden channels).  Same rnh_conv_args_t, with: ntaps = 9;
 * one scale for all sources, ptr2 = 0, every nch % 16 == 0; nk = number of 4-channel steps (sum_src nch/4); wp from
 * rnh_wino_pack_weights; Npad a multiple of 64;
 * source images of at most 2^22 pixels (scale^2 * H * W); epilogue RNH_EPI_STORE, RNH_EPI_PS or RNH_EPI_LSTM.
 * RNH_EPI_LSTM expects the column order of plans.lstm_colmap64 (a block of 64 columns = gates i, f | o, g of 16 hidden
 * channels: column block * 64 + (gate >> 1) * 32 + (gate & 1) * 16 + channel) and Npad == 64 * ceil(hd / 16).
 * Results differ from rnh_conv_igemm by fp32 rounding of the transforms only.  (Replaces the same reference code as
 * rnh_conv_igemm: src/model/nets/refine_net.py:245-265 ConvLSTMCell.forward and the 3x3 convolutions of :102-133.) */
int rnh_conv_wino(const rnh_conv_args_t *args /* host */, void *stream);
/* TWO calls of rnh_conv_wino in ONE launch (ABI 4; the fp32 twin of rnh_conv_bf16_pair, same results as the two calls bit for bit): the cells of
 * the forward- and the backward-direction ConvLSTM of a layer at the same wavefront slot (refine_net.py:82-93), and the data gradients of the two
 * directions, where one call alone leaves much of the chip idle.  The two calls must agree in B, H, W, Npad, epilogue, tile and source scale. */
int rnh_conv_wino_pair(const rnh_conv_args_t *args_a /* host */, const rnh_conv_args_t *args_b /* host */, void *stream);
/* wp[s][xi][n][q] = (G g G^T)[xi], g = the 3x3 filter of (column n, input channel kbase[s] + q*kstride) - the mapping
 * conventions of rnh_pack_weights with 4-channel steps (q >= knv[s]: zero); biasp[n] as there. */
int rnh_wino_pack_weights(const float *w, const float *bias, float *wp, float *biasp, const int32_t *kbase,
                          const int32_t *knv, const int32_t *kcoff, const int32_t *colmap, int ns, int Npad, int Cout,
                          int Cin, int kstride, int transposed, void *stream);

/* Work-item shapes of rnh_conv_wgrad: one wave computes 32*MI rows x 32*NI columns of dW (code = MI << 4 | NI) */
#define RNH_WTILE_128x64 0x42
#define RNH_WTILE_64x128 0x24
#define RNH_WTILE_64x64  0x22
#define RNH_WTILE_128x32 0x41
#define RNH_WTILE_128x128 0x44  /* one wave per SIMD: 256 accumulator registers, two 16-byte loads per 16 MFMAs */

typedef struct rnh_wgrad_args {
    rnh_src_t xs[RNH_MAX_SRC];   /* forward input operand (rows of dW), tap-shifted; one common `scale`, no ptr2 */
    int32_t nxs;
    int32_t xcols_pad;           /* padded row count, multiple of 32*MI                              */
    rnh_src_t ys[RNH_MAX_SRC];   /* output-gradient operand (columns of dW); one common `scale`, no ptr2 */
    int32_t nys;
    int32_t ycols_pad;           /* padded column count, multiple of 32*NI                           */
    const int32_t *xgrp;         /* device [xcols_pad/MI]: lane slot -> (src << 16) | first channel of its MI
                                    consecutive channels, or -1 (zero rows)                          */
    const int32_t *ygrp;         /* device [ycols_pad/NI], likewise with NI                          */
    int32_t B, H, W, ntaps;
    int32_t tile;                /* RNH_WTILE_*                                                      */
    int32_t nsplit;              /* number of pixel ranges                                           */
    float *slab;                 /* workspace [nsplit][ntaps][xcols_pad][ycols_pad]                  */
    float *bslab;                /* workspace [nsplit][ycols_pad] (column sums = bias gradient) or 0 */
    const float *zero_page;      /* device, >= 16 bytes of zeros, 16-byte aligned: what masked lanes read */
} rnh_wgrad_args_t;

/* Weight gradient dW[tap][ci][co] = sum_pixels X[p+tap][ci] * dY[p][co] as an MFMA GEMM whose K dimension
 * is the pixel index, split over `nsplit` pixel ranges into partial slabs (deterministic, no atomics).
 * Replaces the weight/bias part of aten::convolution_backward (loss.backward(), trainer :46). */
int rnh_conv_wgrad(const rnh_wgrad_args_t *args /* host */, void *stream);

/* The same weight gradient in Winograd form F(3x3, 2x2) (csrc/wgrad_wino.hip): dg = G^T [sum_tiles (B^T d B) .* (A dY A^T)] G,
 * 16 GEMMs over tiles, 4/9 of the multiplications.  Takes the xs / ys / B / H / W / ntaps / slab / bslab fields of
 * rnh_wgrad_args_t (rows and columns in the natural order of the sources; xgrp, ygrp, tile, nsplit, zero_page unused).
 * Supported (rnh_wino_wgrad_supported): 3x3, H even, W % 16 == 0, every source channel count a multiple of 32, x sources
 * of scale 1, one common scale for the dy sources.  Workspaces: rnh_wino_wgrad_ws_floats -> {xp (zero-padded gathered
 * copy of the inputs; 4 floats when the kernel that shares the input transform through LDS applies - W % 32 == 0, output
 * channels in multiples of 128 - and reads the sources itself), slab, bslab} in floats.  The reduction (fixed order) scatters like rnh_wgrad_reduce:
 * dw[(colmap[j]*Cin + rowmap[i])*9 + tap], db[colmap[j]]. */
int rnh_wino_wgrad_supported(const rnh_wgrad_args_t *args /* host */);
int rnh_wino_wgrad_ws_floats(const rnh_wgrad_args_t *args /* host */, int64_t *out3 /* host: xp, slab, bslab */);
int rnh_wino_wgrad(const rnh_wgrad_args_t *args /* host */, float *xp, const int32_t *rowmap, const int32_t *colmap, int Cin,
                   float *dw, float *db, int accumulate, void *stream);

/* The same weight gradient in Winograd form F(3x3, 4x4) over 4x4 output tiles with BOTH transforms fused (ABI 6; csrc/wgrad_wino44f.hip): 36 GEMMs over
 * the tiles, 2.25 multiplications per (pixel, ci, co) instead of the 4 of rnh_wino_wgrad - the ConvLSTM cell's weight gradient (autograd of reference
 * src/model/nets/refine_net.py:234-239, :256), refine conv1's and conv2's (:149-151), the PixelShuffle convolutions' (:199-204).  Supported: 3x3, H % 4 == 0, W % 16 == 0, x sources of scale 1 whose
 * channel counts are multiples of 32, dy sources of one common scale with channel counts in multiples of 64 (scale r: the pixel-unshuffle gather of a PixelShuffle
 * convolution's output gradient).  Workspaces (floats): xp (4: unused - the sources are read in
 * place, zero padding by out-of-range buffer offsets), part (K-split partial sums [S][36][Cx][Cy]), bpart (bias partial sums).  Scatter as rnh_wgrad_reduce:
 * dw[(colmap[j] * Cin + rowmap[i]) * 9 + tap], db[colmap[j]]; fixed summation order (deterministic). */
int rnh_wino44f_wgrad_supported(const rnh_wgrad_args_t *args /* host */);
int rnh_wino44f_wgrad_ws_floats(const rnh_wgrad_args_t *args /* host */, int64_t *out3 /* host: xp, part, bpart */);
int rnh_wino44f_wgrad(const rnh_wgrad_args_t *args /* host */, float *xp, float *part, float *bpart, const int32_t *rowmap, const int32_t *colmap, int Cin,
                      float *dw, float *db, int accumulate, void *stream);
/* The same launch with its x operand taken from TRANSFORMED images that already exist (ABI 7): the forward's F(4x4) cells and refine conv1 read their
 * inputs as V = B^T d B (rnh_wino44_transform), and where the caller has kept those images the kernel copies them to LDS (LDS-DMA) instead of transforming the
 * raw tensor again - on this chip every vector instruction of a transform is matrix-core time.  vsrcs[i] belongs to args->xs[i]: `v` = the transformed image of
 * channels [c_first, c_first + 16 nchunks) of the tensor, of the frame that holds the launch's first `images_per_frame` images, frame f (images f *
 * images_per_frame ...) at v + f * frame_stride floats (signed: the backward direction's frames sit in descending slots); xs[i].c0 (c0 - c_first a multiple
 * of 16) and nch select channels inside the image, xs[i].img_off must be 0, xs[i].ptr is not read (non-null).  args->B a multiple of images_per_frame; the
 * images were transformed as rnh_wino44_transform(x, C, c_first, 16 nchunks, images_per_frame, H, W, ...).  Everything else as rnh_wino44f_wgrad (same
 * workspaces part / bpart, same finish). */
typedef struct rnh_wino44_vsrc {
    const float *v;
    int64_t frame_stride;
    int32_t nchunks;
    int32_t c_first;
} rnh_wino44_vsrc_t;
int rnh_wino44f_wgrad_v_supported(const rnh_wgrad_args_t *args /* host */, const rnh_wino44_vsrc_t *vsrcs /* host, args->nxs entries */, int images_per_frame);
int rnh_wino44f_wgrad_v(const rnh_wgrad_args_t *args /* host */, const rnh_wino44_vsrc_t *vsrcs /* host */, int images_per_frame, float *part, float *bpart,
                        const int32_t *rowmap, const int32_t *colmap, int Cin, float *dw, float *db, int accumulate, void *stream);

/* Sum the partial slabs and scatter into the reference-layout gradient:
 *   dw[(colmap[j]*Cin + rowmap[i])*ntaps + tap] (+)= sum_s slab[s][tap][i][j]   (rowmap/colmap < 0: skipped)
 *   db[colmap[j]] (+)= sum_s bslab[s][j]                                         (if bslab and db)
 * ycols_pad % 4 == 0 (since ABI 2: the slabs are summed with 16-byte loads; RNH_E_ALIGN otherwise - every producer of slabs pads its
 * columns to a multiple of 32). */
int rnh_wgrad_reduce(const float *slab, const float *bslab, int nsplit, int ntaps, int xcols_pad, int ycols_pad,
                     const int32_t *rowmap, const int32_t *colmap, int Cin, float *dw, float *db, int accumulate,
                     void *stream);

/* _InBlock forward (refine_net.py:188-192): y = PReLU_a(conv3x3(x) + b), x NHWC [B][H][W][Cin] (any Cin),
 * w OIHW [Cout][Cin][3][3], y NHWC [B][H][W][Cout], Cout % 4 == 0. */
int rnh_inconv_prelu_fwd(const float *x, const float *w, const float *bias, const float *slope, float *y,
                         int B, int H, int W, int Cin, int Cout, void *stream);

/* _InBlock backward: recomputes the pre-activation, returns dW, db, dslope (stored, or accumulated).
 * ws: workspace of rnh_inconv_bwd_ws_floats(Cin, Cout) floats.  4 <= Cout <= 256, Cout % 4 == 0 and Cout / 4 a divisor of 256
 * (since ABI 2: four output channels per thread; RNH_E_RANGE otherwise - num_features[0] of every reference YAML is 64; the
 * engine (hipvsr/plans.py) refuses other widths when the net is planned). */
int rnh_inconv_prelu_bwd(const float *x, const float *w, const float *bias, const float *slope, const float *dy,
                         float *dw, float *db, float *dslope, float *ws, int B, int H, int W, int Cin, int Cout,
                         int accumulate, void *stream);
int64_t rnh_inconv_bwd_ws_floats(int Cin, int Cout);

/* Last convolution of _OutBlock (refine_net.py:201,205), Cout = out_channels (1..8): HBM-bound direct
 * convolution, x NHWC [B][H][W][Cin], y NHWC [B][H][W][Cout]. */
int rnh_outconv_fwd(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int Cin,
                    int Cout, void *stream);
/* The same kernel with explicit weight and output strides (since ABI 3): weight element (co, ci, tap) = w[co*wco + ci*wci + tap]
 * (flip != 0: tap 8 - tap, i.e. a data gradient read as a convolution), bias may be null, output pixel stride ldy floats with
 * the Cout results at y[p*ldy .. ] and `yzero` zero channels behind them.  Replaces the 64-column GEMM launch that produced the one
 * real column of conv2's data gradient w.r.t. conv1's channel 128 (autograd of refine_net.py:152; w = conv2.weight + 128*9,
 * wci = 129*9, flip = 1, Cout = 1). */
int rnh_outconv_fwd_ld(const float *x, const float *w, int64_t wco, int64_t wci, int flip, const float *bias, float *y, int ldy,
                       int yzero, int B, int H, int W, int Cin, int Cout, void *stream);
/* Its data gradient dx[p][ci] = sum_{t,co} dy[p-t][co] w[co][ci][t] ... */
int rnh_outconv_dgrad(const float *dy, const float *w, float *dx, int B, int H, int W, int Cin, int Cout,
                      void *stream);
/* ... and weight / bias gradient (ws: rnh_outconv_wgrad_ws_floats floats). */
int rnh_outconv_wgrad(const float *x, const float *dy, float *dw, float *db, float *ws, int B, int H, int W, int Cin,
                      int Cout, int accumulate, void *stream);
int64_t rnh_outconv_wgrad_ws_floats(int Cin, int Cout);

/* Side path of ONE output channel `co` of a convolution whose input is J frame slots of cstride channels each
 * (_RefineBlock conv1, refine_net.py:150 / :176-182: 645 -> 129 channels, co = 128; 129 = 4*32 + 1 columns would cost
 * a fifth 32-column MFMA tile).  z_s = rnh_outconv_fwd(source s over ALL frames, rnh_xcol_pack(w, co, slot weights)):
 *   rnh_xcol_pack   : out (J, nch, 3, 3)[j][c][t] = w[co][j*cstride + c0 + c][t]  (0 for c >= nvalid)
 *   rnh_xcol_combine: out[(i*N + n)][p][c0..c0+3] = (bias[0] + sum_j sum_s z_s[((i + j)*N + n)][p][j], 0, 0, 0), nwin windows
 *   rnh_xcol_gather : E[((f*N + n)][p][j] = dy[((f - j)*N + n)][p][c] for 0 <= f - j < nwin, else 0; f < nwin + J - 1
 *                     (then rnh_outconv_wgrad(source frames, E) is the channel's weight gradient per slot)
 *   rnh_xcol_unpack : dw[co][j*cstride + c0 + c][t] (+)= dwx[j][c][t];  db[co] (+)= dbx[0] when dbx != 0 */
int rnh_xcol_pack(const float *w, float *out, int Cin, int co, int J, int cstride, int c0, int nch, int nvalid, void *stream);
int rnh_xcol_unpack(const float *dwx, const float *dbx, float *dw, float *db, int Cin, int co, int J, int cstride, int c0,
                    int nch, int nvalid, int accumulate, void *stream);
int rnh_xcol_combine(const float *z0, const float *z1, const float *z2, const float *bias, float *out, int64_t npix, int N,
                     int nwin, int J, int C, int c0, void *stream);
int rnh_xcol_gather(const float *dy, float *E, int64_t npix, int N, int nwin, int J, int C, int c, void *stream);
/* The same side path in the bf16-storage path (csrc/mixed_kernels.hip), where the J slot convolutions of channel c0 run as ONE small
 * rnh_conv_bf16 over the source frames against the view w[c0].view(J, cstride, 3, 3) of conv1's weight (and their weight gradient as
 * one rnh_wgrad_bf16 into the same view of the gradient): z (F' N, H, W, 8) fp32, F' = nwin + J - 1.
 *   rnh_xcol_combine_m: out[(i*N + n)][p][c0 .. c0+7] = (bias[c0] + sum_j z[((i + j)*N + n)][p][j], 0 x 7); out fp32 or bf16, C channels
 *   rnh_xcol_gather_m : E[((f*N + n)][p][j] = dy[((f - j)*N + n)][p][c] for 0 <= f - j < nwin, else 0; 8 channels, slots >= J zero */
int rnh_xcol_combine_m(const float *z, const float *bias, void *out, int out_dt, int64_t npix, int N, int nwin, int J, int C, int c0,
                       void *stream);
int rnh_xcol_gather_m(const void *dy, int dy_dt, void *E, int e_dt, int64_t npix, int N, int nwin, int J, int C, int c, void *stream);

/* Phase planes of _RefineBlock conv1 (refine_net.py:168-177: pos_codes repeated over H x W, concatenated as channel
 * 2*Cl of every frame slot) as a bias field: inside the image the plane of slot j is the constant p, so its 3x3
 * convolution is p times the sum of the taps that stay inside - 16 border classes per pixel.
 *   out (nwin*N, H, W, C)[img = i*N + n][p][0..ncols) += sum_j planes[((i + j)*N + n)][0][0][0] * T[cls(p)][j][.]
 * planes: (F*N, H, W, 4) from rnh_phase_plane; w: OIHW (., Cin, 3, 3), slot j's plane is input channel j*cstride + c0;
 * ws: 16*J*ncols floats (T, rebuilt on every call). */
int rnh_phase_bias_add(float *out, const float *planes, const float *w, float *ws, int H, int W, int N, int nwin, int J,
                       int Cin, int cstride, int c0, int C, int ncols, void *stream);
/* ... and its transpose, the weight gradient of those input channels (autograd of refine_net.py:149 w.r.t. the 129th, 258th, ... input
 * channel of conv1): dw[co][j*cstride + c0][t] (+)= sum_{img = (i, n)} planes[((i + j)*N + n)][0][0][0] * S_t[img][co], S_t = the sum of
 * dy (nwin*N, H, W, C)[img][.][co] over the pixels at which tap t stays inside the image (total - border row - border column + corner),
 * co < ncols.  ws: rnh_phase_wgrad_ws_floats(H, N, nwin, ncols) floats.  Fixed summation order. */
int rnh_phase_wgrad(const float *dy, const float *planes, float *dw, float *ws, int H, int W, int N, int nwin, int J, int Cin,
                    int cstride, int c0, int C, int ncols, int accumulate, void *stream);
int64_t rnh_phase_wgrad_ws_floats(int H, int N, int nwin, int ncols);
/* Data gradient of ONE output channel c of the J-slot convolution (conv1's channel 2*Cl) as a 45-tap stencil (autograd of
 * refine_net.py:149, :176-183 w.r.t. the hidden states, restricted to that output channel):
 *   dx[(f, n)][p][ci] += sum_j sum_t g[((f + J - 1 - j), n)][p - off(t)][c] * w[c][j*cstride + ci][t],  ci < 2*Cl: dx0 takes ci < Cl, dx1 the rest
 * g: ((T + J - 1)*N, H, W, C) gradient planes with (J - 1)/2 zero frames on both sides; dx0 / dx1: (T*N, H, W, Cl), accumulated into.
 * ws: J*9*2*Cl floats.  Cl % 64 == 0. */
int rnh_xcol_dgrad(const float *g, const float *w, float *dx0, float *dx1, float *ws, int H, int W, int N, int T, int J, int Cin,
                   int cstride, int c, int C, int Cl, void *stream);

/* Backward of the upsampler's tail = [last conv + PixelShuffle(r)] -> [final conv C -> out_channels]
 * (refine_net.py:199-205), collapsed algebraically because the tail is affine with out_channels (= 1) outputs
 * (derivation in csrc/uptail.hip).  w2: OIHW (Cq*r*r, C1, 3, 3) weight of the last PixelShuffle conv, w3: OIHW
 * (Co, Cq, 3, 3) weight of the final conv, d_o: gradient of the outputs, NHWC (B, r*Hm, r*Wm, Co); ND = r + 2.
 *   rnh_uptail_compose : G[co][t2][delta][c1] (Co*9*ND*ND*C1 floats) from w2 and w3 and, for Co == 1, behind it the
 *                        merged-offset kernel Kd[(3r+2)^2][C1]; G holds rnh_uptail_g_floats(C1, r, Co) floats
 *   rnh_uptail_dgrad   : dY1 (B, Hm, Wm, C1) = gradient w.r.t. the INPUT of the last PixelShuffle conv, from d_o and G
 *   rnh_uptail_expand  : D (B, Hm, Wm, Dc), D[q][co*ND*ND + delta] = d_o[r*q + delta - 1][co] (0 outside; Dc >= Co*ND*ND,
 *                        Dc % 4 == 0): the column operand of an rnh_conv_wgrad against the conv's input, which yields
 *                        M (Co*ND*ND, C1, 3, 3) and, as its bias output, S = column sums of D
 *   rnh_uptail_xcorr   : M and S directly from the conv's input y1 (B, Hm, Wm, C1) and d_o, without D: a 64-channel x
 *                        (3r+2)^2-offset cross-correlation on the matrix cores + a border term.  Only where
 *                        rnh_uptail_xcorr_supported(C1, r, Co) (Co == 1, r in {2, 3}, C1 % 64 == 0); ws:
 *                        rnh_uptail_xcorr_ws_floats(B, Hm, Wm, C1, r) floats
 *   rnh_uptail_wcontract: dW2, db2, dW3, db3 (stored or accumulated) from M, S and the weights
 * Replaces the aten::convolution_backward calls of those two convolutions and aten::pixel_unshuffle between them. */
/* Forward of the same tail: out (B, r*Hm, r*Wm, Co) = final_conv(PixelShuffle_r(conv(y1; w2, b2)); w3, b3) computed as one
 * composed 5x5 convolution of y1 (B, Hm, Wm, C1) per output sub-position (the paths that leave the image are subtracted on its 1-pixel border);
 * replaces refine_net.py:199-201 / :203-205 for the last PixelShuffle stage.  r in {2, 3}, Co == 1.
 * ws: rnh_uptail_fwd_ws_floats(C1, Cq, r, Co) floats (the composed weights are rebuilt on every call). */
int rnh_uptail_fwd(const float *y1, const float *w2, const float *b2, const float *w3, const float *b3, float *out,
                   float *ws, int B, int Hm, int Wm, int C1, int Cq, int r, int Co, void *stream);
int64_t rnh_uptail_fwd_ws_floats(int C1, int Cq, int r, int Co);
int rnh_uptail_compose(const float *w2, const float *w3, float *G, int C1, int Cq, int r, int Co, void *stream);
int64_t rnh_uptail_g_floats(int C1, int r, int Co);
int rnh_uptail_xcorr_supported(int C1, int r, int Co);
int64_t rnh_uptail_xcorr_ws_floats(int B, int Hm, int Wm, int C1, int r);
int rnh_uptail_xcorr(const float *y1, const float *d_o, float *M, float *S, float *ws, int B, int Hm, int Wm, int C1, int r,
                     void *stream);
int rnh_uptail_dgrad(const float *d_o, const float *G, float *dy1, int B, int Hm, int Wm, int C1, int Co, int r, void *stream);
int rnh_uptail_expand(const float *d_o, float *D, int B, int Hm, int Wm, int Co, int r, int Dc, void *stream);
int rnh_uptail_wcontract(const float *M, const float *S, const float *w2, const float *b2, const float *w3, float *dw2,
                         float *db2, float *dw3, float *db3, int C1, int Cq, int r, int Co, int accumulate2,
                         int accumulate3, void *stream);

/* The same tail in the bf16-storage path (BASELINE.json configs[2]; csrc/uptail_bf16.hip): the tail's input y1 (B, Hm, Wm, C1)
 * and the gradient dy1 that leaves it are bf16 in HBM, the contractions run on v_mfma_f32_16x16x32_bf16 with fp32 accumulators
 * (d_o enters as a bf16 hi + lo pair, the composed weights as bf16); out, d_o, M, S stay fp32.  Same call sites as rnh_uptail_fwd /
 * rnh_uptail_dgrad / rnh_uptail_xcorr (refine_net.py:199-205 and its backward).  Built for r == 2, C1 == 64, Co == 1
 * (rnh_uptail_bf16_supported); G from rnh_uptail_compose.  ws: rnh_uptail_fwd_bf16_ws_floats(C1, Cq, r, Co) /
 * rnh_uptail_dgrad_bf16_ws_floats() / rnh_uptail_xcorr_ws_floats(B, Hm, Wm, C1, r) floats. */
int rnh_uptail_bf16_supported(int C1, int r, int Co);
int64_t rnh_uptail_fwd_bf16_ws_floats(int C1, int Cq, int r, int Co);
int rnh_uptail_fwd_bf16(const void *y1, const float *w2, const float *b2, const float *w3, const float *b3, float *out, float *ws,
                        int B, int Hm, int Wm, int C1, int Cq, int r, int Co, void *stream);
int64_t rnh_uptail_dgrad_bf16_ws_floats(void);
int rnh_uptail_dgrad_bf16(const float *d_o, const float *G, void *dy1, float *ws, int B, int Hm, int Wm, int C1, int Co, int r,
                          void *stream);
int rnh_uptail_xcorr_bf16(const void *y1, const float *d_o, float *M, float *S, float *ws, int B, int Hm, int Wm, int C1, int r,
                          void *stream);

/* Backward of the ConvLSTM gate math (refine_net.py:258-265): from dh', dc' and the saved post-activation
 * gates, c_prev and c_next produce the pre-activation gate gradients [..][4*hd] (channel = gate*hd + ch,
 * order i,f,o,g) and dc_prev.  dc_next / c_prev may be 0 (zero).  n = B*H*W pixels. */
int rnh_lstm_gates_bwd(const float *dh, const float *dh2 /* 0 or a second summand of dh' */, const float *dc_next, const float *gates, const float *c_prev,
                       const float *c_next, float *dgates, float *dc_prev, int64_t npix, int hd, void *stream);

/* Loss + gradient for all output groups at once.  Replaces the 3*S*T loss_fn(output, target) calls of
 * AcdcVSRRefineNetTrainer._compute_losses (trainer :83-100) with torch.nn.L1Loss (kind 0) or
 * CharbonnierLoss (kind 1, src/model/losses.py:32-34).
 *   o: [G][T][per] outputs, y: [T][per] targets, per = N*C*sH*sW
 *   loss[g*T + i] = mean(l(o - y));  do (optional) = gscale[g*T + i] * l'(o - y) / per
 * (gscale = d(total loss)/d(loss[g*T+i]), e.g. discount_g / T).  ws: G*T*RNH_LOSS_BLOCKS floats. */
#define RNH_LOSS_L1 0
#define RNH_LOSS_CHARBONNIER 1
#define RNH_LOSS_BLOCKS 64
int rnh_loss_fwd_bwd(const float *o, const float *y, float *loss, float *d_o, const float *gscale /* device [G*T] */,
                     float *ws, int G, int T, int64_t per, int kind, float eps, void *stream);
/* The discounted deep-supervision sum of the trainer (acdc_vsr_refinenet_trainer.py:83-94: per group g the mean over the T
 * frames of loss * 0.5^(S-1-g/3), summed over the groups) as one launch, and its gradient as one more:
 *   backward = 0: out[0] = sum_g w[g] * mean_i in[g*T + i]          (in = loss[G*T] of rnh_loss_fwd_bwd, w = the discounts)
 *   backward = 1: out[g*T + i] = in[0] * w[g] / T                   (in = d(total); out = the gscale of rnh_loss_fwd_bwd)
 * All pointers device memory. */
int rnh_loss_total(const float *in, const float *w, float *out, int G, int T, int backward, void *stream);

/* out = (accumulate ? out : 0) + a (+ b) (+ c); n floats, n % 4 == 0.  b, c may be 0.
 * Replaces the residual adds of refine_net.py:102,107,112 and the feature update :118-133. */
int rnh_ew_add(float *out, const float *a, const float *b, const float *c, int64_t n, int accumulate, void *stream);

/* Phase-code plane for the refine block (replaces pos_codes.repeat(...).permute(...), refine_net.py:168):
 * out[(f*N + n)][y][x][0..3] = (pos[n*F + f], 0, 0, 0). */
int rnh_phase_plane(const float *pos /* [N][F] */, float *out, int N, int F, int H, int W, void *stream);

/* Input feeding.  Replaces, for one batch, the N calls of
 * AcdcVSRRefineNetDataset.__getitem__ (src/data/datasets/acdc_vsr_refinenet_dataset.py:49-89: nib.load of the LR and
 * the HR cine, RandomHorizontalFlip / RandomVerticalFlip / RandomCropPatch src/data/transforms.py:321-450, Normalize
 * :100-168, ToTensor :74-97, the tripled-cycle window :75-89) and the default collate, with the cines resident in
 * HBM.  pool: fp32, per cine the LR frames (Tc, Hl, Wl), the HR frames (Tc, Hh, Wh) and the phase code (Tc).
 *   inputs  (F, N, h, w)    [k][n] = norm(flip(LR frame (lr_start + k) mod Tc))[y0 : y0+h, x0 : x0+w]
 *   targets (T, N, s*h, s*w) [i][n] = norm(flip(HR frame (hr_start + i) mod Tc))[s*y0 : s*(y0+h), s*x0 : s*(x0+w)]
 *   pos     (N, F)           [n][k] = code[(lr_start + k) mod Tc]
 * flip: np.flip(img, 1) if hflip, np.flip(img, 0) if vflip, BEFORE the crop (the augments' order in exp1_x4.yaml:17-23);
 * norm(x) = (x - mean) / stdv in fp32 with IEEE division (stdv = float32(std + 1e-10), transforms.py:166), skipped if
 * normalize == 0.  Train sample with target frame t: lr_start = t + Tc - T + 1 - U, hr_start = t + Tc - T + 1,
 * F = T + 2U; whole-cycle (valid / test): lr_start = Tc - U, hr_start = 0, F = Tc + 2U, T = Tc.
 * samples_host: N descriptors in host memory, validated against pool_floats and READ COMPLETELY before the call
 * returns (they travel to the kernel by value, 32 samples per launch), so the caller may free or re-fill the array at
 * once.  samples_dev is unused (kept for ABI version 1; may be null). */
typedef struct rnh_cine_sample {
    int64_t lr_off, hr_off, code_off; /* float offsets into the pool */
    int32_t Tc, Hl, Wl, Hh, Wh;       /* frames per cycle, LR and HR frame size */
    int32_t lr_start, hr_start;       /* first LR / HR frame, taken modulo Tc */
    int32_t y0, x0;                   /* LR crop origin, in the flipped image's coordinates */
    int32_t hflip, vflip;
    int32_t reserved[3];
} rnh_cine_sample_t;
int rnh_cine_gather(const float *pool, int64_t pool_floats, const rnh_cine_sample_t *samples_host, rnh_cine_sample_t *samples_dev,
                    int N, int F, int T, int s, int h, int w, int normalize, float mean, float stdv, float *inputs,
                    float *targets, float *pos, void *stream);

/* Optimizer step on flat buffers.  Replaces torch.optim.Adam.step
 * (src/main.py:76 instantiates it from exp1_x4.yaml:56-60, acdc_vsr_refinenet_trainer.py:47 steps it) over the 25
 * parameter tensors that receive a gradient with ONE launch over contiguous ranges:
 *   g += weight_decay * p;  m = m + (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g g;
 *   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * step is the 1-based count of this update; the four buffers hold n floats each and are congruent modulo 16 bytes
 * (the same sub-range of four equally laid out flat buffers). */
int rnh_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, int32_t step, float lr,
                  float beta1, float beta2, float eps, float weight_decay, void *stream);

/* Per-step metrics.  Replaces AcdcVSRRefineNetTrainer._compute_metrics
 * (acdc_vsr_refinenet_trainer.py:103-120) / the predictor's (acdc_vsr_refinenet_predictor.py:140-160) with
 * metric_fns = [PSNR, SSIM]: denormalize (src/utils.py:1-20: clamp(round(x * stdv + mean), 0, 255), applied when
 * denorm != 0), PSNR (src/model/metrics.py:20-36) and SSIM (:86-113, 11x11 window, valid convolution) of P image
 * pairs of H x W in one pass.  out / tgt: [P][H][W] fp32; cps = planes per sample (the channel count: PSNR averages
 * the MSE over the planes of a sample, P % cps == 0).  window11_host: the 11 normalised 1-D window weights, in HOST
 * memory (the reference's 2-D window is their outer product).  c1, c2 = (0.01, 0.03 * value_range)^2.  want_ssim == 0
 * skips the windowed sums (PSNR only; images smaller than the window are then allowed, the SSIM slots read 0).
 * ws: rnh_metrics_ws_floats(P, H, W) floats.  result (device, 2 + P/cps + 2P floats): [0] mean over samples of the
 * PSNR, [1] mean over images of the SSIM-map mean, then PSNR per sample, SSIM per image, MSE per image. */
int64_t rnh_metrics_ws_floats(int P, int H, int W);
int rnh_metrics_psnr_ssim(const float *out, const float *tgt, int P, int cps, int H, int W, int denorm, int want_ssim, float mean, float stdv,
                          float max_value, float value_range, const float *window11_host, float *ws, float *result, void *stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * bf16-storage path (BASELINE.json configs[2]: "bf16, 8xMI355X").  The reference is fp32
 * throughout (refine_net.py:234-241 are plain nn.Conv2d); this is the build's own mixed-precision form of the same
 * call sites: activations of the recurrent / refine / upsampler-input tensors in HBM as bf16, weights packed to bf16 per
 * step from the fp32 state_dict, v_mfma_f32_32x32x16_bf16 with fp32 accumulators, fp32 cell state c, fp32 bias add and
 * gate math, fp32 weight gradients, fp32 final convolution and loss.  Tensors carry their element type explicitly. */
#define RNH_DT_F32  0
#define RNH_DT_BF16 1

typedef struct rnh_msrc {       /* rnh_src_t with an element type; bf16: C, c0, nch multiples of 8; f32: of 4 */
    const void *ptr;
    int32_t dtype;              /* RNH_DT_*                                                              */
    int32_t C, c0, nch, img_off;
    int32_t scale, sub_y, sub_x; /* one common scale for all sources of a call                           */
} rnh_msrc_t;

typedef struct rnh_mdst {       /* rnh_dst_t with an element type; C, c0, ncols multiples of 8           */
    void *ptr;
    int32_t dtype;
    int32_t C, c0, ncols, accumulate, img_off;
    int32_t _pad;
} rnh_mdst_t;

typedef struct rnh_conv_bf16_args {
    rnh_msrc_t src[RNH_MAX_SRC];
    int32_t nsrc;
    int32_t B, H, W;
    int32_t ntaps;              /* 9 or 1                                                                */
    int32_t nchunks;            /* sum_src ceil(nch / 16): 16-channel K chunks                           */
    const void *wp;             /* bf16 [nchunks * ntaps][Npad][16] from rnh_pack_weights_bf16           */
    const float *bias;          /* packed fp32 bias [Npad] or 0                                          */
    int32_t Npad;               /* multiple of 64; column tiles of 128 if Npad % 128 == 0, else of 64    */
    int32_t epilogue;           /* RNH_EPI_*                                                             */
    int32_t ndst;
    int32_t ps_r, ps_cq;        /* RNH_EPI_PS: dst[0] is the (B, rH, rW, cq) tensor; cq % 8 == 0         */
    int32_t hd;                 /* RNH_EPI_LSTM: hidden channels (multiple of 8), columns as rnh_conv_args_t */
    rnh_mdst_t dst[RNH_MAX_DST];
    const float *c_prev;        /* fp32 [B][H][W][hd] or 0                                               */
    float *c_out;               /* fp32 [B][H][W][hd]                                                    */
    void *h_out;                /* [B][H][W][hd] of h_dtype                                              */
    void *gates_out;            /* [B][H][W][4*hd] of gates_dtype or 0                                   */
    int32_t h_dtype, gates_dtype;
    /* RNH_EPI_LSTM_BWD - the data gradient of ConvLSTM cell (layer l, frame t) fused with the gate backward of frame t' = the frame
     * the chain processes next (autograd of refine_net.py:258-265; replaces one rnh_lstm_gates_bwd_m launch and the round trip of
     * the recurrent state gradient through HBM).  Npad == the column tile (all columns in one workgroup); columns [0, dst[0].ncols)
     * = the input gradient, stored to dst[0] as RNH_EPI_STORE does; the next hd columns = dh_rec, the gradient w.r.t. h_{t'}: rounded to
     * bw_rec_dtype (the element type the unfused path stores it in) and consumed in place:  dh = bw_dh + dh_rec, then exactly rnh_lstm_gates_bwd_m
     * (dh, dc_next = bw_dc_next, gates = bw_gates, c_prev = bw_c_prev, c_next = bw_c_next) -> bw_dgates, bw_dc_prev. */
    const void *bw_dh;          /* [B][H][W][hd] of bw_dh_dtype: the summand of dh that does not come from this convolution */
    const float *bw_dc_next;    /* fp32 [B][H][W][hd] or 0                                                */
    const void *bw_gates;       /* [B][H][W][4*hd] of gates_dtype: the gates of frame t'                  */
    const float *bw_c_prev;     /* fp32: the cell state before frame t' (0: zeros)                        */
    const float *bw_c_next;     /* fp32: the cell state AFTER frame t' itself, c_{t'} (the argument rnh_lstm_gates_bwd[_m] calls
                                 * c_next: "next" relative to c_prev, not the next frame; hipvsr/engine.py passes it as c_next too) */
    void *bw_dgates;            /* out [B][H][W][4*hd] of bw_dgates_dtype                                 */
    float *bw_dc_prev;          /* out fp32 [B][H][W][hd] or 0                                            */
    int32_t bw_dh_dtype, bw_dgates_dtype, bw_rec_dtype;
    int32_t wp_f16;             /* 1: wp is IEEE half from rnh_pack_weights_f16 and the contraction runs on v_mfma_f32_32x32x16_f16 (the bf16 inputs
                                 * convert exactly; values beyond 65504 would saturate to inf): RNH_EPI_PS over bf16 sources of 32-channel multiples
                                 * only (ABI 6; the field was padding until then, 0 = the bf16 form) */
} rnh_conv_bf16_args_t;

/* Implicit-GEMM 3x3 / 1x1 convolution on bf16 MFMA: one workgroup = 8 x 32 output pixels x 128 (64) columns; per
 * 16- / 32-channel chunk the 10 x 34 pixel halo of the inputs (converted to bf16 if the source is fp32) goes through LDS once
 * and serves all 9 taps, the packed weights stream from L2 into registers; the epilogue parks the accumulators in LDS so that
 * every global access is a contiguous run of a pixel's channels.  Same call sites as rnh_conv_igemm. */
int rnh_conv_bf16(const rnh_conv_bf16_args_t *args /* host */, void *stream);
/* TWO calls of rnh_conv_bf16 in ONE launch (ABI 4): workgroups [0, n) compute args_a, [n, 2n) args_b - the same results as the two calls,
 * bit for bit.  For call pairs that are independent and small: the cells of the forward- and the backward-direction ConvLSTM of one layer at the
 * same wavefront slot (reference refine_net.py:82-93 runs the two directions one after the other, 2 x F x L cell calls per stage) at the reference's
 * own training shape (16 crops of 32 x 32, configs/train/refine_net/exp1_x4.yaml:20-33) are 128 workgroups each on a 256-CU chip.  The two calls must
 * agree in B, H, W, Npad, ntaps, epilogue, source scale and chunk size (16 / 32 channels); everything else - sources, weights, destinations - is per call. */
int rnh_conv_bf16_pair(const rnh_conv_bf16_args_t *args_a /* host */, const rnh_conv_bf16_args_t *args_b /* host */, void *stream);
/* wp[ks][n][kk] (bf16, kk = 0..15 in natural order) with the index conventions of rnh_pack_weights; biasp fp32. */
int rnh_pack_weights_bf16(const float *w, const float *bias, void *wp, float *biasp, const int32_t *kbase,
                          const int32_t *knv, const int32_t *ktap, const int32_t *kcoff, const int32_t *colmap, int nk,
                          int Npad, int Cout, int Cin, int ntaps, int kstride, int transposed, void *stream);

/* The same slabs in IEEE half for calls with wp_f16 = 1 (ABI 6): the upsampler's PixelShuffle convolutions in the forward (reference
 * src/model/nets/refine_net.py:199-204) - the network's outputs are two linear maps away from them, and at trained weights the 8-bit weights of
 * exactly these layers moved the PSNR by more than the contract's 0.01 dB (profiles/r06_a_*, r06_b_*); 3x3 only. */
int rnh_pack_weights_f16(const float *w, const float *bias, void *wp, float *biasp, const int32_t *kbase,
                         const int32_t *knv, const int32_t *ktap, const int32_t *kcoff, const int32_t *colmap, int nk,
                         int Npad, int Cout, int Cin, int ntaps, int kstride, int transposed, void *stream);

typedef struct rnh_wgrad_bf16_args {
    rnh_msrc_t xs[RNH_MAX_SRC]; /* forward inputs (rows of dW): scale 1                                  */
    int32_t nxs;
    int32_t xrows_pad;          /* padded row count (sum of nch, rounded up to 64)                       */
    rnh_msrc_t ys[RNH_MAX_SRC]; /* output gradients (columns of dW); one common scale                    */
    int32_t nys;
    int32_t ycols_pad;          /* padded column count (rounded up to 64)                                */
    int32_t B, H, W, ntaps;
    int32_t nsplit;             /* pixel-row ranges; slab [nsplit][ntaps][xrows_pad][ycols_pad] fp32     */
    float *slab;
    float *bslab;               /* [nsplit][ycols_pad] or 0                                              */
} rnh_wgrad_bf16_args_t;
/* Weight gradient on bf16 MFMA with the pixel index as the K dimension: image rows are staged channel-major in LDS
 * (the transpose happens in the staging writes), 3 input rows serve the 9 taps.  Partial slabs per row range in the
 * layout of rnh_conv_wgrad, to be summed by rnh_wgrad_reduce (fixed order, no atomics). */
int rnh_wgrad_bf16(const rnh_wgrad_bf16_args_t *args /* host */, void *stream);

/* Mixed-type versions of the small HBM-bound kernels (element types per operand, RNH_DT_*). */
int rnh_ew_add_m(void *out, int out_dt, const void *a, int a_dt, const void *b, int b_dt, const void *c, int c_dt,
                 int64_t n, int accumulate, void *stream);                     /* n % 8 == 0 */
int rnh_lstm_gates_bwd_m(const void *dh, int dh_dt, const void *dh2, int dh2_dt, const float *dc_next, const void *gates, int g_dt,
                         const float *c_prev, const float *c_next, void *dgates, int dg_dt, float *dc_prev, int64_t npix,
                         int hd, void *stream);                                /* hd % 8 == 0 */
int rnh_cast(const void *src, int src_dt, void *dst, int dst_dt, int64_t n, void *stream);   /* n % 8 == 0 */
/* out[(f*N + n)][y][x][0..8) = (pos[n*F + f], 0, ..., 0): the phase plane as an 8-channel source of either type */
int rnh_phase_plane_m(const float *pos /* [N][F] */, void *out, int out_dt, int N, int F, int H, int W, void *stream);

/* ---- The ConvLSTM cell in Winograd form F(4x4, 3x3) (csrc/conv_wino44.hip; ABI 5) -------------------------------------------------------
 * 2.25 multiplications per output (F(2x2, 3x3): 4, direct: 9), true fp32 arithmetic.  The input transform is a kernel of its own:
 * rnh_wino44_transform writes V = B^T d B of `nch` channels [c0, c0 + nch) of an NHWC tensor x [B][H][W][C] (H, W multiples of 4; nch a multiple
 * of 16; zero padding of 1 as nn.Conv2d(padding=1)) into v, rnh_wino44_v_floats(B, H, W, nch) floats, laid out
 *   [tile block of 32 tiles][16-channel chunk][position xi = 6 i + j of the 6x6 domain][tile][16 channels, 16-byte pieces swizzled by tile]
 * = the image rnh_wino44_cell copies to LDS as it is.  A cell output h is transformed once and serves both cells that read it (reference
 * refine_net.py:88-93: the next frame of its layer, the same frame of the next layer). */
int64_t rnh_wino44_v_floats(int B, int H, int W, int nch);
int rnh_wino44_transform(const float *x, int C, int c0, int nch, int B, int H, int W, float *v, void *stream);
/* rnh_lstm_gates_bwd (autograd of reference src/model/nets/refine_net.py:258-265) AND rnh_wino44_transform of the 4 hd gate gradients it writes, in ONE
 * launch (ABI 6): dgates / dc_prev exactly as rnh_lstm_gates_bwd's (all tensors fp32), v = B^T dgates B as rnh_wino44_transform(dgates, 4 hd, 0, 4 hd, ...)
 * would write it - the input of the cell's data gradient in F(4x4, 3x3) form (rnh_wino44_conv on transposed weights), without the second pass over the
 * gate gradients.  H % 16 == 0, W % 32 == 0, hd % 16 == 0 (rnh_wino44_gates_bwd_supported); dh2, dc_next, c_prev, dc_prev may be 0 as there. */
int rnh_wino44_gates_bwd_supported(int H, int W, int hd);
int rnh_wino44_gates_bwd(const float *dh, const float *dh2, const float *dc_next, const float *gates, const float *c_prev, const float *c_next,
                         float *dgates, float *dc_prev, float *v, int B, int H, int W, int hd, void *stream);
/* wp[s8][xi][n][kh][m] = (G g G^T)[xi], g = the 3x3 filter w[colmap[n]][kch[8 s8 + 4 kh + m]] (w OIHW [Cout][Cin][3][3]; kch [K] = the
 * weight's input channel of every K slot in the order the transformed sources are passed to rnh_wino44_cell, colmap [Npad]; device int32
 * arrays, negative = zero); wp: K / 8 * 36 * Npad * 8 floats; biasp[n] = bias[colmap[n]].  K a multiple of 32, Npad of 64.
 * kcoff [K] or 0: per K slot, added to colmap[n] (rnh_pack_weights' kcoff).  transposed (the data gradient, as rnh_pack_weights):
 * g = w[kch[..]][colmap[n] + kcoff[..]] with the taps flipped, no bias. */
int rnh_wino44_pack_weights(const float *w, const float *bias, float *wp, float *biasp, const int32_t *kch, const int32_t *kcoff, const int32_t *colmap, int K,
                            int Npad, int Cout, int Cin, int transposed, void *stream);
typedef struct rnh_wino44_cell_args {
    const float *v[2];          /* transformed sources (rnh_wino44_transform) in K order: the cell's input x_t, its previous output h_{t-1} */
    int32_t vchunks[2];         /* their 16-channel chunks; the sum must be even                                                       */
    int32_t nsrc;               /* 1 (zero state: no h) or 2                                                                            */
    int32_t B, H, W;            /* output geometry; H, W multiples of 4                                                                 */
    int32_t Npad, hd;           /* columns = 4 hd in the order of plans.lstm_colmap64 (blocks of 64 = gates i, f | o, g of 16 channels); hd % 16 == 0 */
    int32_t _pad;
    const float *wp;            /* rnh_wino44_pack_weights                                                                              */
    const float *bias;          /* packed bias [Npad]                                                                                   */
    const float *c_prev;        /* [B][H][W][hd] or 0 (zero state)                                                                      */
    float *h_out, *c_out;       /* [B][H][W][hd]                                                                                        */
    float *gates_out;           /* [B][H][W][4 hd] post-activation i, f, o, g (channel = gate * hd + ch) or 0                           */
} rnh_wino44_cell_args_t;
/* One ConvLSTM cell (reference src/model/nets/refine_net.py:245-265 ConvLSTMCell.forward: cat, 3x3 conv, split, sigmoid / tanh, c' = f c + i g,
 * h' = o tanh c') on its transformed inputs: 36 GEMMs on v_mfma_f32_32x32x2_f32 with V through LDS-DMA, output transform and gate math in the
 * epilogue.  Same results as rnh_conv_wino / rnh_conv_igemm with RNH_EPI_LSTM up to the rounding of the transforms. */
int rnh_wino44_cell(const rnh_wino44_cell_args_t *args /* host */, void *stream);
/* TWO cells of equal geometry in ONE launch (the cells of the forward- and the backward-direction ConvLSTM of a layer at the same wavefront slot, as
 * rnh_conv_wino_pair): the same results as the two calls, bit for bit; the calls must agree in B, H, W and Npad (the number of chunks may differ). */
int rnh_wino44_cell_pair(const rnh_wino44_cell_args_t *args_a /* host */, const rnh_wino44_cell_args_t *args_b /* host */, void *stream);

/* A 3x3 convolution (padding 1) with a plain-store epilogue in the same F(4x4, 3x3) form, on transformed sources that already exist: refine
 * conv1's forward over the hidden states of the top ConvLSTM layer (reference refine_net.py:149, :170-181: torch.cat of the window's frames,
 * conv) reads the transformed h' the cells of that layer wrote for their own successors.  The K dimension is the list of sources
 * (wp from rnh_wino44_pack_weights with the matching kch); a transformed tensor may hold several frames of B images each - `vblock_off[i]` is the
 * tile block (32 tiles; B * H/4 * W/4 must then be a multiple of 32) the launch starts at in source i.  Destination segments as for
 * rnh_conv_igemm (RNH_EPI_STORE); bias packed [Npad] or 0.  With transposed-packed weights: the ConvLSTM cell's data gradient (autograd of
 * refine_net.py:253-257) on the transformed gate gradients. */
typedef struct rnh_wino44_conv_args {
    const float *v[16];
    int32_t vchunks[16];        /* 16-channel chunks of each source; the sum must be even                                             */
    int32_t vblock_off[16];
    int32_t nsrc;
    int32_t B, H, W;
    int32_t Npad;               /* padded column count, a multiple of 64                                                              */
    int32_t _pad[3];
    const float *wp;
    const float *bias;
    int32_t ndst;               /* destination segments as for rnh_conv_igemm's RNH_EPI_STORE: consecutive column ranges                 */
    int32_t ps_r;               /* > 0: nn.PixelShuffle(ps_r) fused into the store as RNH_EPI_PS does (column n = (i*r+j)*ps_cq + c); ndst = 1,
                                   dst[0].ptr = the (B, r H, r W, ps_cq) tensor                                                        */
    int32_t ps_cq;
    int32_t _pad2;
    rnh_dst_t dst[RNH_MAX_DST];
} rnh_wino44_conv_args_t;
int rnh_wino44_conv(const rnh_wino44_conv_args_t *args /* host */, void *stream);

/* ---- Weight gradient in Winograd form F(4x4, 3x3) over 4x4 output tiles (csrc/wgrad_wino44.hip; ABI 5) ------------------------------------
 * dg = G^T [ sum_tiles (B^T d B) .* (A dY A^T) ] G: 36 GEMMs whose K dimension is the tile index, 2.25 multiplications per (pixel, ci, co).
 * rnh_wino44_tmajor writes the transformed inputs (mode 0: B^T d B of the 6x6 patches of `nch` channels [c0, c0 + nch) of x [B][H][W][C], zero
 * padding 1) or output gradients (mode 1: A dY A^T of the 4x4 tiles) tile-major, [36][nch / 32][K8][32][8] with K8 = B * H/4 * W/4 / 8
 * (rnh_wino44_tmajor_floats floats; H, W multiples of 4, nch of 32, the tile count of 8).  A tensor may hold all frames of a source: a window
 * slot that pairs frame f + j with the gradient of window f starts j * (tiles per frame) / 8 rows into it. */
int64_t rnh_wino44_tmajor_floats(int B, int H, int W, int nch);
int rnh_wino44_tmajor(const float *x, int C, int c0, int nch, int B, int H, int W, int mode, float *out, void *stream);
typedef struct rnh_wino44_wgrad_args {
    const float *a[8][2];       /* per problem (a block of 128 input channels against all CO outputs): its two 64-channel halves, tmajor tensors */
    int64_t a_k8[8][2];         /* K8 of those tensors                                                                                          */
    int64_t a_k80[8][2];        /* the problem's first k8 row in them                                                                            */
    const float *z;             /* tmajor tensor (mode 1) of the CO gradient channels, K8 = z_k8 >= T8                                           */
    int64_t z_k8;
    float *part;                /* workspace [S][nprob][36][128][CO] floats: partial sums per K split                                            */
    int32_t nprob, CO;          /* CO a multiple of 128                                                                                          */
    int32_t T8, S;              /* k8 rows of the problem, K splits; T8 % (2 S) == 0                                                             */
} rnh_wino44_wgrad_args_t;
/* The 36 GEMMs on v_mfma_f32_32x32x2_f32 (a workgroup = four positions x a problem's two halves, 128 columns, one K split; operands straight from
 * L2 into a register ring), then: K splits summed in order, dg = G^T dU G, dw[co][rowbase[p] + ci][3][3] (+)= dg for co < ncol (dw OIHW with Cin
 * input channels), db[co] (+)= the tile sum of Z at position (1, 1) (zpart: nchunk * CO floats; z_k8 must equal T8).  Replaces the weight part of
 * aten::convolution_backward for refine conv1 over the hidden states (reference refine_net.py:149; loss.backward(), trainer :46). */
int rnh_wino44_wgrad_gemm(const rnh_wino44_wgrad_args_t *args /* host */, void *stream);
int rnh_wino44_wgrad_finish(const rnh_wino44_wgrad_args_t *args /* host */, const int32_t *rowbase /* device [nprob] */, int ncol, int Cin, float *dw,
                            float *db /* or 0 */, float *zpart, int nchunk, int accumulate, void *stream);

const char *rnh_last_error(void);
int rnh_abi_version(void);
/* sizeof(rnh_src_t), sizeof(rnh_dst_t), sizeof(rnh_conv_args_t), sizeof(rnh_wgrad_args_t): lets a binding
 * check its struct mirrors. */
void rnh_struct_sizes(int32_t out[4]);
/* sizeof(rnh_msrc_t), sizeof(rnh_mdst_t), sizeof(rnh_conv_bf16_args_t), sizeof(rnh_wgrad_bf16_args_t) */
void rnh_struct_sizes_bf16(int32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* REFINENET_HIP_H */
