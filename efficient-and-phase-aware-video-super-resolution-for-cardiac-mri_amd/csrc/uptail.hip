// Backward of the upsampler's tail - the last PixelShuffle convolution followed by the final C -> out_channels
// convolution (reference src/model/nets/refine_net.py:199-205) - collapsed algebraically.
//
// The tail is affine and ends in out_channels (= 1) channels, so the gradient that reaches the last PixelShuffle
// convolution, dZ[q][(c2, ij)] = sum_{co, t3} dO[r*q + ij - t3][co] * w3[co][c2][t3], has rank <= 9*out_channels per
// pixel although it is stored as r*r*C channels.  Substituting it (delta = ij - t3 in [-1, r]^2, ND = r + 2):
//
//   dY1[q][c1]  = sum_{t2, delta, co} dO[r*(q - t2) + delta][co] * G[co][t2][delta][c1],
//                 G = sum_{c2, (ij,t3): ij - t3 = delta} W2[c2*r*r + ij][c1][t2] * w3[co][c2][t3]       (rnh_uptail_dgrad)
//   M[(co,delta)][c1][t2] = sum_q Y1[q + t2][c1] * D[q][(co,delta)],  D[q][(co,delta)] = dO[r*q + delta][co]
//                 (an ordinary rnh_conv_wgrad with 16*out_channels columns instead of r*r*C)
//   dW2[(c2,ij)][c1][t2] = sum_{co,t3} w3[co][c2][t3] * M[(co, ij - t3)][c1][t2]
//   dW3[co][c2][t3]      = sum_{ij} ( sum_{c1,t2} W2[(c2,ij)][c1][t2] * M[(co, ij - t3)][c1][t2] + b2[(c2,ij)] * S[(co, ij - t3)] )
//   db2[(c2,ij)] = sum_{co,t3} w3[co][c2][t3] * S[(co, ij - t3)],   db3[co] = sum_{ij} S[(co, ij)],   S = column sums of D
//
// exactly (zero padding included: a path is dropped precisely when its intermediate pixel lies outside the image,
// which is a property of q - t2 and of r*(q - t2) + delta only).  Instead of 2 * 9*C*r*r*C MACs per pixel on the
// matrix cores plus three passes over r*r*C-channel tensors this costs 9*ND*ND*C MACs on the vector ALU and a
// wgrad GEMM with ND*ND*out_channels columns; the 64-channel high-resolution gradient tensor is never materialised.
#include "rnh_common.h"

namespace {

// 4 consecutive elements of a tensor stored as fp32 or bf16 (the bf16-storage path keeps the tail's input in bf16:
// csrc/uptail_bf16.hip; the border kernels below are shared)
__device__ __forceinline__ float4 ld4t(const float *p) { return rnh_ld4(p); }
__device__ __forceinline__ float4 ld4t(const unsigned short *p) {
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                       __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
}
__device__ __forceinline__ float ld1t(const float *p) { return *p; }
__device__ __forceinline__ float ld1t(const unsigned short *p) { return __builtin_bit_cast(float, (unsigned)*p << 16); }
__device__ __forceinline__ void st1t(float *p, float v) { *p = v; }
__device__ __forceinline__ void st1t(unsigned short *p, float v) {
    const __bf16 b = (__bf16)v;
    *p = __builtin_bit_cast(unsigned short, b);
}

// G[co][t2][dl][c1], dl = (dy+1)*ND + (dx+1)
__global__ void uptail_compose_kernel(const float *w2, const float *w3, float *G, int C1, int Cq, int r, int Co) {
    const int ND = r + 2, r2 = r * r;
    const int total = Co * 9 * ND * ND * C1;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int c1 = e % C1;
        int q = e / C1;
        const int dl = q % (ND * ND);
        q /= ND * ND;
        const int t2 = q % 9, co = q / 9;
        const int ddy = dl / ND - 1, ddx = dl % ND - 1;
        float s = 0.f;
        for (int ty = -1; ty <= 1; ++ty) {
            const int i = ddy + ty;                        // delta = ij - t3
            if (i < 0 || i >= r) continue;
            for (int tx = -1; tx <= 1; ++tx) {
                const int j = ddx + tx;
                if (j < 0 || j >= r) continue;
                const int ij = i * r + j, t3 = (ty + 1) * 3 + tx + 1;
                for (int c2 = 0; c2 < Cq; ++c2)
                    s += w2[((long)(c2 * r2 + ij) * C1 + c1) * 9 + t2] * w3[((long)co * Cq + c2) * 9 + t3];
            }
        }
        G[e] = s;
    }
}

// thread = (mid-resolution pixel q, 4 channels of dY1); G in LDS
__global__ void __launch_bounds__(256) uptail_dgrad_kernel(const float *dO, const float *G, float *dY1, int B, int Hm, int Wm, int C1,
                                                           int Co, int r) {
    extern __shared__ __attribute__((aligned(16))) float sG[];
    const int ND = r + 2;
    const int ng = Co * 9 * ND * ND * C1;
    for (int e = threadIdx.x; e < ng; e += blockDim.x) sG[e] = G[e];
    __syncthreads();
    const int Gc = C1 >> 2, Hh = Hm * r, Wh = Wm * r;
    const long total = (long)B * Hm * Wm * Gc;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int g = (int)(e % Gc);
        const long q = e / Gc;
        const int qx = (int)(q % Wm);
        const int qy = (int)((q / Wm) % Hm);
        const long b = q / ((long)Hm * Wm);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t2 = 0; t2 < 9; ++t2) {
            const int sy = qy - (t2 / 3 - 1), sx = qx - (t2 % 3 - 1);          // q' = q - t2
            if ((unsigned)sy >= (unsigned)Hm || (unsigned)sx >= (unsigned)Wm) continue;
            for (int dyi = 0; dyi < ND; ++dyi) {
                const int py = sy * r + dyi - 1;
                if ((unsigned)py >= (unsigned)Hh) continue;
                const float *row = dO + ((b * Hh + py) * Wh) * Co;
                for (int dxi = 0; dxi < ND; ++dxi) {
                    const int px = sx * r + dxi - 1;
                    if ((unsigned)px >= (unsigned)Wh) continue;
                    for (int co = 0; co < Co; ++co) {
                        const float d = row[(long)px * Co + co];
                        const float4 gv = rnh_ld4(sG + (((co * 9 + t2) * ND + dyi) * ND + dxi) * C1 + g * 4);
                        acc.x += d * gv.x; acc.y += d * gv.y; acc.z += d * gv.z; acc.w += d * gv.w;
                    }
                }
            }
        }
        rnh_st4(dY1 + q * C1 + g * 4, acc);
    }
}

// D[q][co*ND*ND + dl] = dO[r*q + delta][co] (0 outside); channels padded to Dc
__global__ void uptail_expand_kernel(const float *dO, float *D, int B, int Hm, int Wm, int Co, int r, int Dc) {
    const int ND = r + 2, Hh = Hm * r, Wh = Wm * r;
    const long total = (long)B * Hm * Wm * Dc;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(e % Dc);
        const long q = e / Dc;
        float v = 0.f;
        if (ch < Co * ND * ND) {
            const int co = ch / (ND * ND), dl = ch % (ND * ND);
            const int qx = (int)(q % Wm), qy = (int)((q / Wm) % Hm);
            const long b = q / ((long)Hm * Wm);
            const int py = qy * r + dl / ND - 1, px = qx * r + dl % ND - 1;
            if ((unsigned)py < (unsigned)Hh && (unsigned)px < (unsigned)Wh) v = dO[((b * Hh + py) * Wh + px) * Co + co];
        }
        D[e] = v;
    }
}

// contraction of M (and S) with the weights into dW2, db2, dW3, db3
__global__ void uptail_wcontract_kernel(const float *M, const float *S, const float *w2, const float *b2, const float *w3, float *dw2,
                                        float *db2, float *dw3, float *db3, int C1, int Cq, int r, int Co, int acc2, int acc3) {
    const int ND = r + 2, r2 = r * r;
    const int n2 = Cq * r2 * C1 * 9, nb2 = Cq * r2, n3 = Co * Cq * 9, nb3 = Co;
    const int total = n2 + nb2 + n3 + nb3;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        if (e < n2 + nb2) {                                   // dW2[(c2,ij)][c1][t2]  /  db2[(c2,ij)]
            const bool isb = e >= n2;
            const int idx = isb ? e - n2 : e;
            const int t2 = isb ? 0 : idx % 9, c1 = isb ? 0 : (idx / 9) % C1, k = isb ? idx : idx / (9 * C1);
            const int c2 = k / r2, ij = k % r2, i = ij / r, j = ij % r;
            float s = 0.f;
            for (int co = 0; co < Co; ++co)
                for (int t3 = 0; t3 < 9; ++t3) {
                    const int dl = (i - (t3 / 3 - 1) + 1) * ND + (j - (t3 % 3 - 1) + 1);
                    const float wv = w3[((long)co * Cq + c2) * 9 + t3];
                    s += wv * (isb ? S[co * ND * ND + dl] : M[((long)(co * ND * ND + dl) * C1 + c1) * 9 + t2]);
                }
            float *o = isb ? db2 + idx : dw2 + idx;
            *o = acc2 ? *o + s : s;
        } else if (e < n2 + nb2 + n3) {                       // dW3[co][c2][t3]: uptail_wcontract3_kernel (one wave per entry)
        } else {                                              // db3[co] = sum over the r*r sub-positions of S[(co, delta = ij)]
            const int co = e - n2 - nb2 - n3;
            float s = 0.f;
            for (int ij = 0; ij < r2; ++ij) s += S[co * ND * ND + (ij / r + 1) * ND + ij % r + 1];
            db3[co] = acc3 ? db3[co] + s : s;
        }
    }
}

// dW3[co][c2][t3] = sum_ij ( <W2[(c2,ij)], M[(co, ij - t3)]> + b2[(c2,ij)] S[(co, ij - t3)] ): r*r dot products of C1*9 terms each.
// One wave per entry, lanes over the C1*9 terms, fixed-order butterfly reduction (in uptail_wcontract_kernel a single thread
// walked all r*r*C1*9 = 2304 terms of an entry: 0.41 ms per call at C1 = 64, r = 2 for 576 busy threads).
__global__ void __launch_bounds__(256) uptail_wcontract3_kernel(const float *M, const float *S, const float *w2, const float *b2, float *dw3,
                                                                int C1, int Cq, int r, int Co, int acc3) {
    const int ND = r + 2, r2 = r * r, n3 = Co * Cq * 9;
    const int lane = threadIdx.x & 63, idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= n3) return;
    const int t3 = idx % 9, c2 = (idx / 9) % Cq, co = idx / (9 * Cq);
    float s = 0.f;
    for (int ij = 0; ij < r2; ++ij) {
        const int dl = (ij / r - (t3 / 3 - 1) + 1) * ND + (ij % r - (t3 % 3 - 1) + 1);
        const float *Mp = M + (long)(co * ND * ND + dl) * C1 * 9;
        const float *Wp = w2 + (long)(c2 * r2 + ij) * C1 * 9;
        for (int x = lane; x < C1 * 9; x += 64) s += Wp[x] * Mp[x];
        if (lane == 0) s += b2[c2 * r2 + ij] * S[co * ND * ND + dl];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) dw3[idx] = acc3 ? dw3[idx] + s : s;
}

// ---------------------------------------------------------------------------------------------------------
// forward of the tail:  O[p][co] = b3[co] + sum_{t3: p+t3 inside} ( beta[co][ij][t3] + sum_{t2: q'+t2 inside} <Y1[q'+t2], Gf[co][t2][ij][t3]> )
// with (q', ij) = divmod(p + t3, r).  Away from the border every path exists and the double sum is ONE 5x5
// convolution of Y1 per output sub-position (Kf), 25*C MACs per output instead of 9*C*r*r*C per pixel through the
// r*r*C-channel intermediate, which is never formed.  Output pixels ON the border lose the paths through p + t3 outside
// the image: uptail_border_kernel subtracts them afterwards.
// ---------------------------------------------------------------------------------------------------------
__global__ void uptail_compose_fwd1_kernel(const float *w2, const float *b2, const float *w3, float *Gf, float *beta, int C1, int Cq,
                                           int r, int Co) {
    const int r2 = r * r;
    const int ng = Co * 9 * r2 * 9 * C1, nb = Co * r2 * 9;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < ng + nb; e += gridDim.x * blockDim.x) {
        if (e < ng) {                                        // Gf[co][t2][ij][t3][c1]
            const int c1 = e % C1;
            int q = e / C1;
            const int t3 = q % 9;
            q /= 9;
            const int ij = q % r2;
            q /= r2;
            const int t2 = q % 9, co = q / 9;
            float s = 0.f;
            for (int c2 = 0; c2 < Cq; ++c2) s += w3[((long)co * Cq + c2) * 9 + t3] * w2[((long)(c2 * r2 + ij) * C1 + c1) * 9 + t2];
            Gf[e] = s;
        } else {                                             // beta[co][ij][t3]
            const int idx = e - ng;
            const int t3 = idx % 9, ij = (idx / 9) % r2, co = idx / (9 * r2);
            float s = 0.f;
            for (int c2 = 0; c2 < Cq; ++c2) s += w3[((long)co * Cq + c2) * 9 + t3] * b2[c2 * r2 + ij];
            beta[idx] = s;
        }
    }
}

__device__ __forceinline__ int floordiv(int a, int r) { return (a >= 0) ? a / r : -((-a + r - 1) / r); }

// Kf[u][c1 (padded to a multiple of 16 with zeros)][o (sub-position, padded to an even count)], bsum[o]; Co == 1
__global__ void uptail_compose_fwd2_kernel(const float *Gf, const float *beta, float *Kf, float *bsum, int C1, int C1p, int r, int NOP) {
    const int r2 = r * r;
    const int nk = 25 * C1p * NOP;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nk + r2; e += gridDim.x * blockDim.x) {
        if (e < nk) {
            const int ij0 = e % NOP, c1 = (e / NOP) % C1p, u = e / (NOP * C1p);
            float s = 0.f;
            if (ij0 < r2 && c1 < C1) {
                const int i0 = ij0 / r, j0 = ij0 % r, uy = u / 5 - 2, ux = u % 5 - 2;
                for (int t3 = 0; t3 < 9; ++t3) {
                    const int ay = i0 + t3 / 3 - 1, ax = j0 + t3 % 3 - 1;
                    const int fy = floordiv(ay, r), fx = floordiv(ax, r);
                    const int ij = (ay - fy * r) * r + (ax - fx * r);
                    const int t2y = uy - fy, t2x = ux - fx;
                    if (t2y < -1 || t2y > 1 || t2x < -1 || t2x > 1) continue;
                    const int t2 = (t2y + 1) * 3 + t2x + 1;
                    s += Gf[(((long)t2 * r2 + ij) * 9 + t3) * C1 + c1];
                }
            }
            Kf[e] = s;
        } else {
            const int ij0 = e - nk, i0 = ij0 / r, j0 = ij0 % r;
            float s = 0.f;
            for (int t3 = 0; t3 < 9; ++t3) {
                const int ay = i0 + t3 / 3 - 1, ax = j0 + t3 % 3 - 1;
                const int fy = floordiv(ay, r), fx = floordiv(ax, r);
                s += beta[((ay - fy * r) * r + (ax - fx * r)) * 9 + t3];
            }
            bsum[ij0] = s;
        }
    }
}

constexpr int UT = 16;            // 16x16 mid-resolution pixels per block
constexpr int UROW = 20;          // 16 channels + 4 pad floats per halo pixel
constexpr int UH = UT + 4;        // halo tile edge

// thread = one mid-resolution pixel, all r*r sub-positions; the 16-channel chunk of its 5x5 neighbourhood comes from
// the LDS tile (4 ds_read_b128 per tap), the composed weights are wave-uniform and arrive through scalar loads
template <int R>
__global__ void __launch_bounds__(256) uptail_fwd_kernel(const float *__restrict__ y1, const float *__restrict__ Kf,
                                                         const float *__restrict__ bsum, const float *__restrict__ b3,
                                                         float *__restrict__ out, int B, int Hm, int Wm, int C1, int C1p, int TX, int TY) {
    constexpr int R2 = R * R, NOP = (R2 + 1) & ~1;
    extern __shared__ __attribute__((aligned(16))) float tile[];           // [UH*UH][UROW]
    const int tb = blockIdx.x;
    const int b = tb / (TX * TY), trem = tb - b * TX * TY, tyb = trem / TX, txb = trem - tyb * TX;
    const int y0 = tyb * UT, x0 = txb * UT;
    const int ly = threadIdx.x / UT, lx = threadIdx.x % UT;
    const int qy = y0 + ly, qx = x0 + lx;
    float acc[NOP];
#pragma unroll
    for (int o = 0; o < NOP; ++o) acc[o] = 0.f;
    for (int c0 = 0; c0 < C1; c0 += 16) {
        __syncthreads();
        for (int e = threadIdx.x; e < UH * UH * 4; e += 256) {
            const int q4 = e & 3, hp = e >> 2, hy = hp / UH, hx = hp - hy * UH;
            const int gy = y0 + hy - 2, gx = x0 + hx - 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)gy < (unsigned)Hm && (unsigned)gx < (unsigned)Wm && c0 + q4 * 4 < C1)
                v = rnh_ld4(y1 + (((long)b * Hm + gy) * Wm + gx) * C1 + c0 + q4 * 4);
            rnh_st4(tile + hp * UROW + q4 * 4, v);
        }
        __syncthreads();
        for (int uy = 0; uy < 5; ++uy)
            for (int ux = 0; ux < 5; ++ux) {
                const float *tp = tile + ((ly + uy) * UH + lx + ux) * UROW;
                const float4 x0v = rnh_ld4(tp), x1v = rnh_ld4(tp + 4), x2v = rnh_ld4(tp + 8), x3v = rnh_ld4(tp + 12);
                const float xs[16] = {x0v.x, x0v.y, x0v.z, x0v.w, x1v.x, x1v.y, x1v.z, x1v.w,
                                      x2v.x, x2v.y, x2v.z, x2v.w, x3v.x, x3v.y, x3v.z, x3v.w};
                const float *k = Kf + ((long)(uy * 5 + ux) * C1p + c0) * NOP;
#pragma unroll
                for (int c = 0; c < 16; ++c)
#pragma unroll
                    for (int o = 0; o < NOP; ++o) acc[o] = fmaf(xs[c], k[c * NOP + o], acc[o]);
            }
    }
    if (qy >= Hm || qx >= Wm) return;
    const int Hh = Hm * R, Wh = Wm * R;
#pragma unroll
    for (int o = 0; o < R2; ++o)
        out[((long)b * Hh + qy * R + o / R) * Wh + qx * R + o % R] = acc[o] + bsum[o] + b3[0];
}

// The composed 5x5 kernel sums every (t3, t2) path; for an output pixel ON the border of the r-times larger image the
// paths whose intermediate pixel p + t3 lies outside it do not exist (zero padding of the PixelShuffle output, not of
// y1).  One wave per border pixel subtracts exactly those paths: lanes split (t3, t2, 4-channel group), wave-reduce.
template <typename T>
__global__ void __launch_bounds__(256) uptail_border_kernel(const T *y1, const float *Gf, const float *beta, float *out, int B, int Hm,
                                                            int Wm, int C1, int r, int Co) {
    const int Hh = Hm * r, Wh = Wm * r, r2 = r * r, C4 = C1 >> 2;
    const int nper = 2 * Wh + 2 * (Hh - 2);
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)B * nper) return;
    const int b = (int)(item / nper), k = (int)(item - (long)b * nper);
    int py, px;
    if (k < Wh) { py = 0; px = k; }
    else if (k < 2 * Wh) { py = Hh - 1; px = k - Wh; }
    else { const int k2 = k - 2 * Wh; py = 1 + (k2 >> 1); px = (k2 & 1) ? Wh - 1 : 0; }
    for (int co = 0; co < Co; ++co) {
        float s = 0.f;
        // the missing t3 (3 on an edge, 5 in a corner) in a wave-uniform loop; lanes over (t2, 4-channel group) of one t3
        for (int t3 = 0; t3 < 9; ++t3) {
            const int ppy = py + t3 / 3 - 1, ppx = px + t3 % 3 - 1;
            if ((unsigned)ppy < (unsigned)Hh && (unsigned)ppx < (unsigned)Wh) continue;      // this path exists
            const int sy = floordiv(ppy, r), sx = floordiv(ppx, r), ij = (ppy - sy * r) * r + (ppx - sx * r);
            for (int idx = lane; idx < 9 * C4; idx += 64) {
                const int c4 = idx % C4, t2 = idx / C4;
                const int yy = sy + t2 / 3 - 1, xx = sx + t2 % 3 - 1;
                if ((unsigned)yy >= (unsigned)Hm || (unsigned)xx >= (unsigned)Wm) continue;
                const float4 a = ld4t(y1 + (((long)b * Hm + yy) * Wm + xx) * C1 + c4 * 4);
                const float4 g = rnh_ld4(Gf + ((((long)co * 9 + t2) * r2 + ij) * 9 + t3) * C1 + c4 * 4);
                s += a.x * g.x + a.y * g.y + a.z * g.z + a.w * g.w;
            }
            if (lane == 0) s += beta[(co * r2 + ij) * 9 + t3];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        if (lane == 0) out[(((long)b * Hh + py) * Wh + px) * Co + co] -= s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Merged-offset form of the backward for out_channels == 1.  In dY1 and in M the pair (t2, delta) only enters
// through the offset  off = delta - r*t2  in [-r-1, 2r]^2  (NOFF = 3r + 2 per axis) between the mid-resolution
// pixel and the dO pixel it meets:
//   dY1[q][c1]      = sum_off dO[r*q + off] * Kd[off][c1]            - (paths with q - t2 outside, border pixels only)
//   M[dl][c1][t2]   = X[dl - r*t2][c1]                               - (terms with q' - t2 outside, border pixels only)
//   X[off][c1]      = sum_q' Y1[q'][c1] * dO[r*q' + off],   S[dl] = SX[dl] = sum_q' dO[r*q' + dl]
// 64 (r = 2) / 121 (r = 3) offsets instead of 9*ND*ND = 144 / 225 (t2, delta) pairs, and Y1 is read once, unshifted.
// ---------------------------------------------------------------------------------------------------------
__global__ void uptail_compose_kd_kernel(const float *G, float *Kd, int C1, int r) {
    const int ND = r + 2, NOFF = 3 * r + 2;
    const int total = NOFF * NOFF * C1;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int c1 = e % C1, off = e / C1;
        const int offy = off / NOFF - (r + 1), offx = off % NOFF - (r + 1);
        float s = 0.f;
        for (int ty = -1; ty <= 1; ++ty) {
            const int ddy = offy + r * ty;
            if (ddy < -1 || ddy > r) continue;
            for (int tx = -1; tx <= 1; ++tx) {
                const int ddx = offx + r * tx;
                if (ddx < -1 || ddx > r) continue;
                s += G[((long)((ty + 1) * 3 + tx + 1) * ND * ND + (ddy + 1) * ND + ddx + 1) * C1 + c1];
            }
        }
        Kd[e] = s;
    }
}

// the k-th candidate of the 1-pixel border of an Hm x Wm image (rows first, then the columns without their ends)
__device__ __forceinline__ bool border_pixel(int k, int Hm, int Wm, int &y, int &x) {
    if (k < Wm) { y = 0; x = k; return true; }
    k -= Wm;
    if (k < Wm) { y = Hm - 1; x = k; return Hm > 1; }
    k -= Wm;
    if (k < Hm) { y = k; x = 0; return k > 0 && k < Hm - 1; }
    k -= Hm;
    y = k; x = Wm - 1;
    return k > 0 && k < Hm - 1 && Wm > 1;
}

constexpr int DT = 16;            // 16x16 mid-resolution pixels per block, one pixel x CCH channels per thread

// dO patch of the tile in LDS; Kd is wave-uniform (scalar loads), so the inner loop is 1 LDS read + CCH FMAs
template <int CCH>
__global__ void __launch_bounds__(256) uptail_dgrad_tile_kernel(const float *__restrict__ dO, const float *__restrict__ Kd,
                                                                float *__restrict__ dY1, int Hm, int Wm, int C1, int r, int TX, int TY) {
    extern __shared__ __attribute__((aligned(16))) float patch[];
    const int NOFF = 3 * r + 2, PW = DT * r + 2 * r + 2;
    const int tb = blockIdx.x;
    const int b = tb / (TX * TY), trem = tb - b * TX * TY, tyb = trem / TX, txb = trem - tyb * TX;
    const int y0 = tyb * DT, x0 = txb * DT, c0 = blockIdx.y * CCH;
    const int Hh = Hm * r, Wh = Wm * r, hy0 = r * y0 - (r + 1), hx0 = r * x0 - (r + 1);
    for (int e = threadIdx.x; e < PW * PW; e += 256) {
        const int py = hy0 + e / PW, px = hx0 + e % PW;
        patch[e] = ((unsigned)py < (unsigned)Hh && (unsigned)px < (unsigned)Wh) ? dO[((long)b * Hh + py) * Wh + px] : 0.f;
    }
    __syncthreads();
    const int ly = threadIdx.x / DT, lx = threadIdx.x % DT, qy = y0 + ly, qx = x0 + lx;
    float acc[CCH];
#pragma unroll
    for (int c = 0; c < CCH; ++c) acc[c] = 0.f;
    const float *pp = patch + (r * ly) * PW + r * lx;
    const float *kp = Kd + c0;
    for (int oy = 0; oy < NOFF; ++oy)
        for (int ox = 0; ox < NOFF; ++ox) {
            const float d = pp[oy * PW + ox];
            const float *k = kp + (long)(oy * NOFF + ox) * C1;
#pragma unroll
            for (int c = 0; c < CCH; ++c) acc[c] = fmaf(d, k[c], acc[c]);
        }
    if (qy >= Hm || qx >= Wm) return;
    float *o = dY1 + (((long)b * Hm + qy) * Wm + qx) * C1 + c0;
#pragma unroll
    for (int c = 0; c < CCH; c += 4) rnh_st4(o + c, make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]));
}

// one wave per border pixel q: subtract the (t2, delta) paths whose q - t2 lies outside the image
template <typename T>
__global__ void __launch_bounds__(256) uptail_dgrad_border_kernel(const float *dO, const float *G, T *dY1, int B, int Hm, int Wm,
                                                                  int C1, int r) {
    const int ND = r + 2, Hh = Hm * r, Wh = Wm * r, nper = 2 * (Hm + Wm);
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (long)B * nper) return;
    const int b = (int)(item / nper);
    int qy, qx;
    if (!border_pixel((int)(item - (long)b * nper), Hm, Wm, qy, qx)) return;
    for (int c1 = lane; c1 < C1; c1 += 64) {
        float s = 0.f;
        for (int t2 = 0; t2 < 9; ++t2) {
            const int sy = qy - (t2 / 3 - 1), sx = qx - (t2 % 3 - 1);
            if ((unsigned)sy < (unsigned)Hm && (unsigned)sx < (unsigned)Wm) continue;       // this path exists
            for (int dl = 0; dl < ND * ND; ++dl) {
                const int py = sy * r + dl / ND - 1, px = sx * r + dl % ND - 1;
                if ((unsigned)py >= (unsigned)Hh || (unsigned)px >= (unsigned)Wh) continue;
                s += dO[((long)b * Hh + py) * Wh + px] * G[((long)t2 * ND * ND + dl) * C1 + c1];
            }
        }
        T *o = dY1 + (((long)b * Hm + qy) * Wm + qx) * C1 + c1;
        st1t(o, ld1t(o) - s);
    }
}

// X and SX on the matrix cores: rows = 64 channels of Y1 (two 32-row tiles, row i of tile e = channel 2i + e, one
// 8-byte load per lane), columns = offsets (NT 32-column tiles), contraction over pixels, two per v_mfma_f32_32x32x2.
// A block walks 8x32-pixel tiles (persistent, grid-strided), each wave two rows of the tile; the dO patch of the tile is
// in LDS and the column operand is gathered from it.  Partial sums per block go to a slab (deterministic reduction in
// uptail_mfinish_kernel).
constexpr int XTY = 8, XTX = 32;

template <int R>
__global__ void __launch_bounds__(256) uptail_xcorr_kernel(const float *__restrict__ y1, const float *__restrict__ dO,
                                                           float *__restrict__ Xs, int B, int Hm, int Wm, int C1, int TX, int TY) {
    constexpr int NOFF = 3 * R + 2, NO2 = NOFF * NOFF, NT = (NO2 + 31) / 32, PW = XTX * R + 2 * R + 2, PH = XTY * R + 2 * R + 2;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5, wave = threadIdx.x >> 6;
    const int cg = blockIdx.y, Hh = Hm * R, Wh = Wm * R;
    f32x16 acc[2][NT];
    float sb[NT], bm[NT];
    int boff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int off = 32 * j + l31;
        const bool ok = off < NO2;
        boff[j] = ok ? (off / NOFF) * PW + off % NOFF : 0;
        bm[j] = ok ? 1.f : 0.f;
        sb[j] = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) { acc[0][j][v] = 0.f; acc[1][j][v] = 0.f; }
    }
    const int ntiles = B * TX * TY;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int b = t / (TX * TY), trem = t - b * TX * TY, tyb = trem / TX, txb = trem - tyb * TX;
        const int y0 = tyb * XTY, x0 = txb * XTX, hy0 = R * y0 - (R + 1), hx0 = R * x0 - (R + 1);
        __syncthreads();
        for (int e = threadIdx.x; e < PH * PW; e += 256) {
            const int py = hy0 + e / PW, px = hx0 + e % PW;
            sm[e] = ((unsigned)py < (unsigned)Hh && (unsigned)px < (unsigned)Wh) ? dO[((long)b * Hh + py) * Wh + px] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int ly = 2 * wave + rr, qy = y0 + ly;
            if (qy >= Hm) break;
            const float *yrow = y1 + (((long)b * Hm + qy) * Wm) * C1 + cg * 64 + 2 * l31;
            const float *prow = sm + (R * ly) * PW;
#pragma unroll 4
            for (int s = 0; s < XTX / 2; ++s) {
                const int lx = 2 * s + kh, qx = x0 + lx;
                const bool pv = qx < Wm;
                float2 a = make_float2(0.f, 0.f);
                if (pv) a = *reinterpret_cast<const float2 *>(yrow + (long)qx * C1);
                float bv[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    bv[j] = pv ? prow[R * lx + boff[j]] * bm[j] : 0.f;
                    sb[j] += bv[j];
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bv[j], acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bv[j], acc[1][j], 0, 0, 0);
                }
            }
        }
    }
    // cross-wave sum in LDS ([offset][64 channels], then SX[offset]); the block's partial goes to its slab
    float *red = sm, *sred = sm + NT * 32 * 64;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int i = (v & 3) + 8 * (v >> 2) + 4 * kh;
                        float *p = red + (32 * j + l31) * 64 + 2 * i + e;
                        *p = (w == 0 ? 0.f : *p) + acc[e][j][v];
                    }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float v = sb[j] + __shfl_xor(sb[j], 32, 64);
                if (kh == 0) sred[32 * j + l31] = (w == 0 ? 0.f : sred[32 * j + l31]) + v;
            }
        }
    }
    __syncthreads();
    constexpr int SLAB = NT * 32 * 64 + NT * 32;
    float *slab = Xs + ((long)blockIdx.x * gridDim.y + cg) * SLAB;
    for (int e = threadIdx.x; e < SLAB; e += 256) slab[e] = sm[e];
}

// Cs[chunk][t2][dl][c1] = sum over the chunk's border pixels q' with q' - t2 outside of Y1[q'][c1] * dO[r*(q' - t2) + delta]
template <typename T>
__global__ void __launch_bounds__(256) uptail_mborder_kernel(const T *y1, const float *dO, float *Cs, int B, int Hm, int Wm, int C1,
                                                             int r, int ipc) {
    const int ND = r + 2, Hh = Hm * r, Wh = Wm * r, nper = 2 * (Hm + Wm), C4 = C1 >> 2;
    const int chunk = blockIdx.x, t2 = blockIdx.y, o = blockIdx.z * 256 + threadIdx.x;
    if (o >= ND * ND * C4) return;
    const int c4 = o % C4, dl = o / C4, ty = t2 / 3 - 1, tx = t2 % 3 - 1, ddy = dl / ND - 1, ddx = dl % ND - 1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int b1 = min(B, (chunk + 1) * ipc);
    (void)nper;
    // q' - t2 leaves the image exactly on one border row (ty != 0) and / or one border column (tx != 0): walk those, not all
    // 2 (Hm + Wm) border candidates
    const int rowy = ty < 0 ? Hm - 1 : 0, colx = tx < 0 ? Wm - 1 : 0;
    auto add = [&](int b, int qy, int qx) {
        const int py = (qy - ty) * r + ddy, px = (qx - tx) * r + ddx;
        if ((unsigned)py >= (unsigned)Hh || (unsigned)px >= (unsigned)Wh) return;
        const float d = dO[((long)b * Hh + py) * Wh + px];
        const float4 a = ld4t(y1 + (((long)b * Hm + qy) * Wm + qx) * C1 + c4 * 4);
        acc.x += d * a.x; acc.y += d * a.y; acc.z += d * a.z; acc.w += d * a.w;
    };
    for (int b = chunk * ipc; b < b1; ++b) {
        if (ty != 0)
            for (int qx = 0; qx < Wm; ++qx) add(b, rowy, qx);
        if (tx != 0)
            for (int qy = 0; qy < Hm; ++qy)
                if (ty == 0 || qy != rowy) add(b, qy, colx);
    }
    rnh_st4(Cs + (((long)chunk * 9 + t2) * ND * ND + dl) * C1 + c4 * 4, acc);
}

// M[dl][c1][t2] = sum_blocks X[dl - r*t2][c1] - sum_chunks Cs[t2][dl][c1];  S[dl] = sum_blocks SX[dl]   (fixed order)
__global__ void uptail_mfinish_kernel(const float *Xs, const float *Cs, float *M, float *S, int nblk, int ncg, int nchunk, int C1, int r,
                                      int NT) {
    const int ND = r + 2, NOFF = 3 * r + 2, nm = ND * ND * 9 * C1;
    const long slabsz = (long)NT * 32 * 64 + NT * 32;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nm + ND * ND; e += gridDim.x * blockDim.x) {
        if (e < nm) {
            const int c1 = e % C1, t2 = (e / C1) % 9, dl = e / (9 * C1);
            const int offy = dl / ND - 1 - r * (t2 / 3 - 1) + r + 1, offx = dl % ND - 1 - r * (t2 % 3 - 1) + r + 1;
            const float *xp = Xs + (long)(c1 >> 6) * slabsz + (long)(offy * NOFF + offx) * 64 + (c1 & 63);
            float s = 0.f;
            for (int k = 0; k < nblk; ++k) s += xp[(long)k * ncg * slabsz];
            if (t2 != 4) {
                const float *cp = Cs + ((long)t2 * ND * ND + dl) * C1 + c1;
                float c = 0.f;
                for (int k = 0; k < nchunk; ++k) c += cp[(long)k * 9 * ND * ND * C1];
                s -= c;
            }
            M[((long)dl * C1 + c1) * 9 + t2] = s;
        } else {
            const int dl = e - nm;
            const float *xp = Xs + (long)NT * 32 * 64 + (dl / ND + r) * NOFF + dl % ND + r;
            float s = 0.f;
            for (int k = 0; k < nblk; ++k) s += xp[(long)k * ncg * slabsz];
            S[dl] = s;
        }
    }
}

inline int grid_for(long n, int cap = 8192) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int64_t rnh_uptail_g_floats(int C1, int r, int Co) {
    return (int64_t)Co * 9 * (r + 2) * (r + 2) * C1 + (Co == 1 ? (int64_t)(3 * r + 2) * (3 * r + 2) * C1 : 0);
}

extern "C" int rnh_uptail_compose(const float *w2, const float *w3, float *G, int C1, int Cq, int r, int Co, void *stream) {
    if (!w2 || !w3 || !G || C1 < 1 || Cq < 1 || r < 2 || r > 4 || Co < 1) RNH_FAIL(RNH_E_ARG, "rnh_uptail_compose: bad arguments");
    const long ng = (long)Co * 9 * (r + 2) * (r + 2) * C1;
    hipLaunchKernelGGL(uptail_compose_kernel, dim3(grid_for(ng)), dim3(256), 0, (hipStream_t)stream, w2, w3, G, C1, Cq, r, Co);
    RNH_CHECK_LAUNCH("rnh_uptail_compose");
    if (Co == 1) {                                       // merged-offset kernel Kd behind G
        hipLaunchKernelGGL(uptail_compose_kd_kernel, dim3(grid_for((long)(3 * r + 2) * (3 * r + 2) * C1)), dim3(256), 0, (hipStream_t)stream,
                           G, G + ng, C1, r);
        RNH_CHECK_LAUNCH("rnh_uptail_compose(Kd)");
    }
    return 0;
}

extern "C" int64_t rnh_uptail_fwd_ws_floats(int C1, int Cq, int r, int Co) {
    const int64_t r2 = r * r, C1p = (C1 + 15) & ~15, NOP = (r2 + 1) & ~1;
    return Co * 9 * r2 * 9 * C1 + Co * r2 * 9 + 25 * C1p * NOP + r2 + 64;
}

extern "C" int rnh_uptail_fwd(const float *y1, const float *w2, const float *b2, const float *w3, const float *b3, float *out, float *ws,
                              int B, int Hm, int Wm, int C1, int Cq, int r, int Co, void *stream) {
    if (!y1 || !w2 || !b2 || !w3 || !b3 || !out || !ws || B < 1 || Hm < 1 || Wm < 1 || C1 < 1 || Cq < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_uptail_fwd: bad arguments");
    if (C1 & 3) RNH_FAIL(RNH_E_ALIGN, "rnh_uptail_fwd: C1 must be a multiple of 4");
    if (!((r == 2 || r == 3) && Co == 1)) RNH_FAIL(RNH_E_RANGE, "rnh_uptail_fwd: built for r in {2, 3} and out_channels == 1");
    const int r2 = r * r, C1p = (C1 + 15) & ~15, NOP = (r2 + 1) & ~1;
    float *Gf = ws, *beta = Gf + (long)9 * r2 * 9 * C1, *Kf = beta + r2 * 9, *bsum = Kf + (long)25 * C1p * NOP;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = rnh_uptail_fwd_compose_(w2, b2, w3, ws, C1, Cq, r, Co, st)) return rc;
    const int TX = (Wm + UT - 1) / UT, TY = (Hm + UT - 1) / UT;
    const size_t shm = (size_t)UH * UH * UROW * sizeof(float);
    const dim3 grid((unsigned)(B * TX * TY)), block(256);
    if (r == 2) hipLaunchKernelGGL((uptail_fwd_kernel<2>), grid, block, shm, st, y1, Kf, bsum, b3, out, B, Hm, Wm, C1, C1p, TX, TY);
    else hipLaunchKernelGGL((uptail_fwd_kernel<3>), grid, block, shm, st, y1, Kf, bsum, b3, out, B, Hm, Wm, C1, C1p, TX, TY);
    RNH_CHECK_LAUNCH("rnh_uptail_fwd");
    const long nborder = (long)B * (2 * Wm * r + 2 * (Hm * r - 2));
    hipLaunchKernelGGL(uptail_border_kernel<float>, dim3((unsigned)((nborder + 3) / 4)), dim3(256), 0, st, y1, Gf, beta, out, B, Hm, Wm, C1, r, Co);
    RNH_CHECK_LAUNCH("rnh_uptail_fwd(border)");
    return 0;
}

extern "C" int rnh_uptail_dgrad(const float *d_o, const float *G, float *dy1, int B, int Hm, int Wm, int C1, int Co, int r, void *stream) {
    if (!d_o || !G || !dy1 || B < 1 || Hm < 1 || Wm < 1 || Co < 1 || r < 2 || r > 4) RNH_FAIL(RNH_E_ARG, "rnh_uptail_dgrad: bad arguments");
    if (C1 & 3) RNH_FAIL(RNH_E_ALIGN, "rnh_uptail_dgrad: C1 must be a multiple of 4");
    if (Co == 1 && (C1 & 15) == 0) {                     // merged-offset tile kernel + border correction
        const int TX = (Wm + DT - 1) / DT, TY = (Hm + DT - 1) / DT, PW = DT * r + 2 * r + 2;
        const float *Kd = G + (long)9 * (r + 2) * (r + 2) * C1;
        const size_t shp = (size_t)PW * PW * sizeof(float);
        hipStream_t st = (hipStream_t)stream;
        if ((C1 & 63) == 0)
            hipLaunchKernelGGL((uptail_dgrad_tile_kernel<64>), dim3((unsigned)(B * TX * TY), C1 / 64), dim3(256), shp, st, d_o, Kd, dy1, Hm, Wm,
                               C1, r, TX, TY);
        else
            hipLaunchKernelGGL((uptail_dgrad_tile_kernel<16>), dim3((unsigned)(B * TX * TY), C1 / 16), dim3(256), shp, st, d_o, Kd, dy1, Hm, Wm,
                               C1, r, TX, TY);
        RNH_CHECK_LAUNCH("rnh_uptail_dgrad(tile)");
        const long nb = (long)B * 2 * (Hm + Wm);
        hipLaunchKernelGGL(uptail_dgrad_border_kernel<float>, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, st, d_o, G, dy1, B, Hm, Wm, C1, r);
        RNH_CHECK_LAUNCH("rnh_uptail_dgrad(border)");
        return 0;
    }
    const size_t shm = (size_t)Co * 9 * (r + 2) * (r + 2) * C1 * sizeof(float);
    if (shm > 64 * 1024) RNH_FAIL(RNH_E_RANGE, "rnh_uptail_dgrad: composed weights (%zu bytes) do not fit in LDS", shm);
    hipLaunchKernelGGL(uptail_dgrad_kernel, dim3(grid_for((long)B * Hm * Wm * (C1 / 4), 4096)), dim3(256), shm, (hipStream_t)stream, d_o, G,
                       dy1, B, Hm, Wm, C1, Co, r);
    RNH_CHECK_LAUNCH("rnh_uptail_dgrad");
    return 0;
}

extern "C" int rnh_uptail_expand(const float *d_o, float *D, int B, int Hm, int Wm, int Co, int r, int Dc, void *stream) {
    if (!d_o || !D || B < 1 || Hm < 1 || Wm < 1 || Co < 1 || r < 2 || r > 4 || Dc < Co * (r + 2) * (r + 2) || (Dc & 3))
        RNH_FAIL(RNH_E_ARG, "rnh_uptail_expand: bad arguments");
    hipLaunchKernelGGL(uptail_expand_kernel, dim3(grid_for((long)B * Hm * Wm * Dc)), dim3(256), 0, (hipStream_t)stream, d_o, D, B, Hm, Wm, Co,
                       r, Dc);
    RNH_CHECK_LAUNCH("rnh_uptail_expand");
    return 0;
}

extern "C" int rnh_uptail_wcontract(const float *M, const float *S, const float *w2, const float *b2, const float *w3, float *dw2,
                                    float *db2, float *dw3, float *db3, int C1, int Cq, int r, int Co, int accumulate2, int accumulate3,
                                    void *stream) {
    if (!M || !S || !w2 || !b2 || !w3 || !dw2 || !db2 || !dw3 || !db3 || C1 < 1 || Cq < 1 || r < 2 || r > 4 || Co < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_uptail_wcontract: bad arguments");
    const long total = (long)Cq * r * r * C1 * 9 + Cq * r * r + (long)Co * Cq * 9 + Co;
    hipLaunchKernelGGL(uptail_wcontract_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, M, S, w2, b2, w3, dw2, db2, dw3, db3,
                       C1, Cq, r, Co, accumulate2, accumulate3);
    RNH_CHECK_LAUNCH("rnh_uptail_wcontract");
    hipLaunchKernelGGL(uptail_wcontract3_kernel, dim3((Co * Cq * 9 + 3) / 4), dim3(256), 0, (hipStream_t)stream, M, S, w2, b2, dw3, C1, Cq, r, Co,
                       accumulate3);
    RNH_CHECK_LAUNCH("rnh_uptail_wcontract(dW3)");
    return 0;
}

extern "C" int rnh_uptail_xcorr_supported(int C1, int r, int Co) { return Co == 1 && (r == 2 || r == 3) && C1 > 0 && (C1 & 63) == 0; }

static void xcorr_shape(int B, int Hm, int Wm, int r, int *TX, int *TY, int *nblk, int *nchunk, int *NT) {
    *TX = (Wm + XTX - 1) / XTX;
    *TY = (Hm + XTY - 1) / XTY;
    const long nt = (long)B * *TX * *TY;
    *nblk = (int)(nt < 512 ? nt : 512);
    *nchunk = B < 256 ? B : 256;
    *NT = ((3 * r + 2) * (3 * r + 2) + 31) / 32;
}

extern "C" int64_t rnh_uptail_xcorr_ws_floats(int B, int Hm, int Wm, int C1, int r) {
    int TX, TY, nblk, nchunk, NT;
    xcorr_shape(B, Hm, Wm, r, &TX, &TY, &nblk, &nchunk, &NT);
    return (int64_t)nblk * (C1 / 64) * (NT * 32 * 64 + NT * 32) + (int64_t)nchunk * 9 * (r + 2) * (r + 2) * C1 + 64;
}

extern "C" int rnh_uptail_xcorr(const float *y1, const float *d_o, float *M, float *S, float *ws, int B, int Hm, int Wm, int C1, int r,
                                void *stream) {
    if (!y1 || !d_o || !M || !S || !ws || B < 1 || Hm < 1 || Wm < 1) RNH_FAIL(RNH_E_ARG, "rnh_uptail_xcorr: bad arguments");
    if (!rnh_uptail_xcorr_supported(C1, r, 1)) RNH_FAIL(RNH_E_RANGE, "rnh_uptail_xcorr: built for r in {2, 3} and C1 a multiple of 64");
    int TX, TY, nblk, nchunk, NT;
    xcorr_shape(B, Hm, Wm, r, &TX, &TY, &nblk, &nchunk, &NT);
    const int ncg = C1 / 64, ND = r + 2, ipc = (B + nchunk - 1) / nchunk;
    nchunk = (B + ipc - 1) / ipc;
    float *Xs = ws, *Cs = ws + (long)nblk * ncg * (NT * 32 * 64 + NT * 32);
    hipStream_t st = (hipStream_t)stream;
    const int PW = XTX * r + 2 * r + 2, PH = XTY * r + 2 * r + 2;
    const size_t need = (size_t)NT * 32 * 65, patch = (size_t)PH * PW;
    const size_t shm = (need > patch ? need : patch) * sizeof(float);
    if (r == 2) hipLaunchKernelGGL((uptail_xcorr_kernel<2>), dim3(nblk, ncg), dim3(256), shm, st, y1, d_o, Xs, B, Hm, Wm, C1, TX, TY);
    else hipLaunchKernelGGL((uptail_xcorr_kernel<3>), dim3(nblk, ncg), dim3(256), shm, st, y1, d_o, Xs, B, Hm, Wm, C1, TX, TY);
    RNH_CHECK_LAUNCH("rnh_uptail_xcorr");
    hipLaunchKernelGGL(uptail_mborder_kernel<float>, dim3(nchunk, 9, (ND * ND * (C1 / 4) + 255) / 256), dim3(256), 0, st, y1, d_o, Cs, B, Hm, Wm, C1,
                       r, ipc);
    RNH_CHECK_LAUNCH("rnh_uptail_xcorr(border)");
    hipLaunchKernelGGL(uptail_mfinish_kernel, dim3(grid_for((long)ND * ND * 9 * C1 + ND * ND)), dim3(256), 0, st, Xs, Cs, M, S, nblk, ncg,
                       nchunk, C1, r, NT);
    RNH_CHECK_LAUNCH("rnh_uptail_xcorr(finish)");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Internals shared with csrc/uptail_bf16.hip (declared in rnh_common.h; C++ linkage, not part of the C ABI): the bf16-storage
// tail reuses the fp32 composition of the weights, the three border corrections (typed on the tail input) and the fixed-order
// reduction of the cross-correlation slabs.
// ---------------------------------------------------------------------------------------------------------
int rnh_uptail_fwd_compose_(const float *w2, const float *b2, const float *w3, float *ws, int C1, int Cq, int r, int Co, hipStream_t st) {
    const int r2 = r * r, C1p = (C1 + 15) & ~15, NOP = (r2 + 1) & ~1;
    float *Gf = ws, *beta = Gf + (long)9 * r2 * 9 * C1, *Kf = beta + r2 * 9, *bsum = Kf + (long)25 * C1p * NOP;
    hipLaunchKernelGGL(uptail_compose_fwd1_kernel, dim3(grid_for((long)9 * r2 * 9 * C1 + r2 * 9)), dim3(256), 0, st, w2, b2, w3, Gf, beta, C1,
                       Cq, r, Co);
    RNH_CHECK_LAUNCH("rnh_uptail_fwd(compose 1)");
    hipLaunchKernelGGL(uptail_compose_fwd2_kernel, dim3(grid_for((long)25 * C1p * NOP + r2)), dim3(256), 0, st, Gf, beta, Kf, bsum, C1, C1p, r,
                       NOP);
    RNH_CHECK_LAUNCH("rnh_uptail_fwd(compose 2)");
    return 0;
}

int rnh_uptail_fwd_border_bf16_(const void *y1, const float *ws, float *out, int B, int Hm, int Wm, int C1, int r, int Co, hipStream_t st) {
    const int r2 = r * r;
    const float *Gf = ws, *beta = Gf + (long)9 * r2 * 9 * C1;
    const long nborder = (long)B * (2 * Wm * r + 2 * (Hm * r - 2));
    hipLaunchKernelGGL(uptail_border_kernel<unsigned short>, dim3((unsigned)((nborder + 3) / 4)), dim3(256), 0, st,
                       (const unsigned short *)y1, Gf, beta, out, B, Hm, Wm, C1, r, Co);
    RNH_CHECK_LAUNCH("rnh_uptail_fwd_bf16(border)");
    return 0;
}

int rnh_uptail_dgrad_border_bf16_(const float *d_o, const float *G, void *dy1, int B, int Hm, int Wm, int C1, int r, hipStream_t st) {
    const long nb = (long)B * 2 * (Hm + Wm);
    hipLaunchKernelGGL(uptail_dgrad_border_kernel<unsigned short>, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, st, d_o, G,
                       (unsigned short *)dy1, B, Hm, Wm, C1, r);
    RNH_CHECK_LAUNCH("rnh_uptail_dgrad_bf16(border)");
    return 0;
}

void rnh_uptail_xcorr_shape_(int B, int Hm, int Wm, int r, int *TX, int *TY, int *nblk, int *nchunk, int *NT) {
    xcorr_shape(B, Hm, Wm, r, TX, TY, nblk, nchunk, NT);
}

// border terms + fixed-order reduction of the slabs Xs (written by uptail_xcorr_bf16_kernel in the layout of uptail_xcorr_kernel)
int rnh_uptail_xcorr_finish_bf16_(const void *y1, const float *d_o, float *M, float *S, float *ws, int B, int Hm, int Wm, int C1, int r,
                                  hipStream_t st) {
    int TX, TY, nblk, nchunk, NT;
    xcorr_shape(B, Hm, Wm, r, &TX, &TY, &nblk, &nchunk, &NT);
    const int ncg = C1 / 64, ND = r + 2, ipc = (B + nchunk - 1) / nchunk;
    nchunk = (B + ipc - 1) / ipc;
    float *Xs = ws, *Cs = ws + (long)nblk * ncg * (NT * 32 * 64 + NT * 32);
    hipLaunchKernelGGL(uptail_mborder_kernel<unsigned short>, dim3(nchunk, 9, (ND * ND * (C1 / 4) + 255) / 256), dim3(256), 0, st,
                       (const unsigned short *)y1, d_o, Cs, B, Hm, Wm, C1, r, ipc);
    RNH_CHECK_LAUNCH("rnh_uptail_xcorr_bf16(border)");
    hipLaunchKernelGGL(uptail_mfinish_kernel, dim3(grid_for((long)ND * ND * 9 * C1 + ND * ND)), dim3(256), 0, st, Xs, Cs, M, S, nblk, ncg,
                       nchunk, C1, r, NT);
    RNH_CHECK_LAUNCH("rnh_uptail_xcorr_bf16(finish)");
    return 0;
}
