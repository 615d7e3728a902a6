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
        } else if (e < n2 + nb2 + n3) {                       // dW3[co][c2][t3]
            const int idx = e - n2 - nb2;
            const int t3 = idx % 9, c2 = (idx / 9) % Cq, co = idx / (9 * Cq);
            float s = 0.f;
            for (int ij = 0; ij < r2; ++ij) {
                const int dl = (ij / r - (t3 / 3 - 1) + 1) * ND + (ij % r - (t3 % 3 - 1) + 1);
                const float *Mp = M + (long)(co * ND * ND + dl) * C1 * 9;
                const float *Wp = w2 + (long)(c2 * r2 + ij) * C1 * 9;
                float ss = 0.f;
                for (int x = 0; x < C1 * 9; ++x) ss += Wp[x] * Mp[x];
                s += ss + b2[c2 * r2 + ij] * S[co * ND * ND + dl];
            }
            float *o = dw3 + idx;
            *o = acc3 ? *o + s : s;
        } else {                                              // db3[co] = sum over the r*r sub-positions of S[(co, delta = ij)]
            const int co = e - n2 - nb2 - n3;
            float s = 0.f;
            for (int ij = 0; ij < r2; ++ij) s += S[co * ND * ND + (ij / r + 1) * ND + ij % r + 1];
            db3[co] = acc3 ? db3[co] + s : s;
        }
    }
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

__global__ void uptail_compose_fwd2_kernel(const float *Gf, const float *beta, float *Kf, float *bsum, int C1, int r, int Co) {
    const int r2 = r * r;
    const int nk = Co * r2 * 25 * C1, nb = Co * r2;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nk + nb; e += gridDim.x * blockDim.x) {
        if (e < nk) {                                        // Kf[co][ij0][u][c1]
            const int c1 = e % C1;
            int q = e / C1;
            const int u = q % 25;
            q /= 25;
            const int ij0 = q % r2, co = q / r2;
            const int i0 = ij0 / r, j0 = ij0 % r, uy = u / 5 - 2, ux = u % 5 - 2;
            float s = 0.f;
            for (int t3 = 0; t3 < 9; ++t3) {
                const int ay = i0 + t3 / 3 - 1, ax = j0 + t3 % 3 - 1;
                const int fy = floordiv(ay, r), fx = floordiv(ax, r);
                const int ij = (ay - fy * r) * r + (ax - fx * r);
                const int t2y = uy - fy, t2x = ux - fx;
                if (t2y < -1 || t2y > 1 || t2x < -1 || t2x > 1) continue;
                const int t2 = (t2y + 1) * 3 + t2x + 1;
                s += Gf[((((long)co * 9 + t2) * r2 + ij) * 9 + t3) * C1 + c1];
            }
            Kf[e] = s;
        } else {
            const int idx = e - nk;
            const int ij0 = idx % r2, co = idx / r2, i0 = ij0 / r, j0 = ij0 % r;
            float s = 0.f;
            for (int t3 = 0; t3 < 9; ++t3) {
                const int ay = i0 + t3 / 3 - 1, ax = j0 + t3 % 3 - 1;
                const int fy = floordiv(ay, r), fx = floordiv(ax, r);
                s += beta[(co * r2 + (ay - fy * r) * r + (ax - fx * r)) * 9 + t3];
            }
            bsum[idx] = s;
        }
    }
}

constexpr int UT = 16;            // 16x16 mid-resolution pixels per block
constexpr int UROW = 20;          // 16 channels + 4 pad floats per halo pixel
constexpr int UH = UT + 4;        // halo tile edge

template <int R, int CO>
__global__ void __launch_bounds__(256) uptail_fwd_kernel(const float *y1, const float *Kf, const float *bsum, const float *b3,
                                                         float *out, int B, int Hm, int Wm, int C1, int TX, int TY) {
    constexpr int R2 = R * R, NO = R2 * CO;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *tile = sm;                                  // [UH*UH][UROW]
    float *skf = sm + UH * UH * UROW;                  // [NO][25][16]
    const int tb = blockIdx.x;
    const int b = tb / (TX * TY), trem = tb - b * TX * TY, tyb = trem / TX, txb = trem - tyb * TX;
    const int y0 = tyb * UT, x0 = txb * UT;
    const int ly = threadIdx.x / UT, lx = threadIdx.x % UT;
    const int qy = y0 + ly, qx = x0 + lx;
    float acc[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] = 0.f;
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
        for (int e = threadIdx.x; e < NO * 25 * 16; e += 256) {
            const int c = e & 15, ou = e >> 4;
            skf[e] = (c0 + c < C1) ? Kf[(long)ou * C1 + c0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 5
        for (int u = 0; u < 25; ++u) {
            const float *tp = tile + ((ly + u / 5) * UH + lx + u % 5) * UROW;
            const float4 x0v = rnh_ld4(tp), x1v = rnh_ld4(tp + 4), x2v = rnh_ld4(tp + 8), x3v = rnh_ld4(tp + 12);
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const float *kp = skf + (o * 25 + u) * 16;
                const float4 k0 = rnh_ld4(kp), k1 = rnh_ld4(kp + 4), k2 = rnh_ld4(kp + 8), k3 = rnh_ld4(kp + 12);
                acc[o] += x0v.x * k0.x + x0v.y * k0.y + x0v.z * k0.z + x0v.w * k0.w + x1v.x * k1.x + x1v.y * k1.y + x1v.z * k1.z +
                          x1v.w * k1.w + x2v.x * k2.x + x2v.y * k2.y + x2v.z * k2.z + x2v.w * k2.w + x3v.x * k3.x + x3v.y * k3.y +
                          x3v.z * k3.z + x3v.w * k3.w;
            }
        }
    }
    if (qy >= Hm || qx >= Wm) return;
    const int Hh = Hm * R, Wh = Wm * R;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        const int co = o / R2, ij0 = o % R2;
        out[(((long)b * Hh + qy * R + ij0 / R) * Wh + qx * R + ij0 % R) * CO + co] = acc[o] + bsum[o] + b3[co];
    }
}

// The composed 5x5 kernel sums every (t3, t2) path; for an output pixel ON the border of the r-times larger image the
// paths whose intermediate pixel p + t3 lies outside it do not exist (zero padding of the PixelShuffle output, not of
// y1).  One wave per border pixel subtracts exactly those paths: lanes split (t3, t2, 4-channel group), wave-reduce.
__global__ void __launch_bounds__(256) uptail_border_kernel(const float *y1, const float *Gf, const float *beta, float *out, int B, int Hm,
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
        for (int idx = lane; idx < 81 * C4; idx += 64) {
            const int c4 = idx % C4, tt = idx / C4, t2 = tt % 9, t3 = tt / 9;
            const int ppy = py + t3 / 3 - 1, ppx = px + t3 % 3 - 1;
            if ((unsigned)ppy < (unsigned)Hh && (unsigned)ppx < (unsigned)Wh) continue;      // this path exists
            const int sy = floordiv(ppy, r), sx = floordiv(ppx, r), ij = (ppy - sy * r) * r + (ppx - sx * r);
            const int yy = sy + t2 / 3 - 1, xx = sx + t2 % 3 - 1;
            if ((unsigned)yy >= (unsigned)Hm || (unsigned)xx >= (unsigned)Wm) continue;
            const float4 a = rnh_ld4(y1 + (((long)b * Hm + yy) * Wm + xx) * C1 + c4 * 4);
            const float4 g = rnh_ld4(Gf + ((((long)co * 9 + t2) * r2 + ij) * 9 + t3) * C1 + c4 * 4);
            s += a.x * g.x + a.y * g.y + a.z * g.z + a.w * g.w;
        }
        if (lane < 9) {
            const int t3 = lane, ppy = py + t3 / 3 - 1, ppx = px + t3 % 3 - 1;
            if (!((unsigned)ppy < (unsigned)Hh && (unsigned)ppx < (unsigned)Wh)) {
                const int sy = floordiv(ppy, r), sx = floordiv(ppx, r);
                s += beta[(co * r2 + (ppy - sy * r) * r + (ppx - sx * r)) * 9 + t3];
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        if (lane == 0) out[(((long)b * Hh + py) * Wh + px) * Co + co] -= s;
    }
}

inline int grid_for(long n, int cap = 8192) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int rnh_uptail_compose(const float *w2, const float *w3, float *G, int C1, int Cq, int r, int Co, void *stream) {
    if (!w2 || !w3 || !G || C1 < 1 || Cq < 1 || r < 2 || r > 4 || Co < 1) RNH_FAIL(RNH_E_ARG, "rnh_uptail_compose: bad arguments");
    hipLaunchKernelGGL(uptail_compose_kernel, dim3(grid_for((long)Co * 9 * (r + 2) * (r + 2) * C1)), dim3(256), 0, (hipStream_t)stream, w2,
                       w3, G, C1, Cq, r, Co);
    RNH_CHECK_LAUNCH("rnh_uptail_compose");
    return 0;
}

extern "C" int64_t rnh_uptail_fwd_ws_floats(int C1, int Cq, int r, int Co) {
    const int64_t r2 = r * r;
    return Co * 9 * r2 * 9 * C1 + Co * r2 * 9 + Co * r2 * 25 * C1 + Co * r2 + 64;
}

extern "C" int rnh_uptail_fwd(const float *y1, const float *w2, const float *b2, const float *w3, const float *b3, float *out, float *ws,
                              int B, int Hm, int Wm, int C1, int Cq, int r, int Co, void *stream) {
    if (!y1 || !w2 || !b2 || !w3 || !b3 || !out || !ws || B < 1 || Hm < 1 || Wm < 1 || C1 < 1 || Cq < 1)
        RNH_FAIL(RNH_E_ARG, "rnh_uptail_fwd: bad arguments");
    if (C1 & 3) RNH_FAIL(RNH_E_ALIGN, "rnh_uptail_fwd: C1 must be a multiple of 4");
    if (!((r == 2 || r == 3) && Co == 1)) RNH_FAIL(RNH_E_RANGE, "rnh_uptail_fwd: built for r in {2, 3} and out_channels == 1");
    const int r2 = r * r;
    float *Gf = ws, *beta = Gf + (long)Co * 9 * r2 * 9 * C1, *Kf = beta + Co * r2 * 9, *bsum = Kf + (long)Co * r2 * 25 * C1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(uptail_compose_fwd1_kernel, dim3(grid_for((long)Co * 9 * r2 * 9 * C1 + Co * r2 * 9)), dim3(256), 0, st, w2, b2, w3, Gf,
                       beta, C1, Cq, r, Co);
    RNH_CHECK_LAUNCH("rnh_uptail_fwd(compose 1)");
    hipLaunchKernelGGL(uptail_compose_fwd2_kernel, dim3(grid_for((long)Co * r2 * 25 * C1 + Co * r2)), dim3(256), 0, st, Gf, beta, Kf, bsum,
                       C1, r, Co);
    RNH_CHECK_LAUNCH("rnh_uptail_fwd(compose 2)");
    const int TX = (Wm + UT - 1) / UT, TY = (Hm + UT - 1) / UT;
    const size_t shm = ((size_t)UH * UH * UROW + (size_t)r2 * Co * 25 * 16) * sizeof(float);
    const dim3 grid((unsigned)(B * TX * TY)), block(256);
    if (r == 2) hipLaunchKernelGGL((uptail_fwd_kernel<2, 1>), grid, block, shm, st, y1, Kf, bsum, b3, out, B, Hm, Wm, C1, TX, TY);
    else hipLaunchKernelGGL((uptail_fwd_kernel<3, 1>), grid, block, shm, st, y1, Kf, bsum, b3, out, B, Hm, Wm, C1, TX, TY);
    RNH_CHECK_LAUNCH("rnh_uptail_fwd");
    const long nborder = (long)B * (2 * Wm * r + 2 * (Hm * r - 2));
    hipLaunchKernelGGL(uptail_border_kernel, dim3((unsigned)((nborder + 3) / 4)), dim3(256), 0, st, y1, Gf, beta, out, B, Hm, Wm, C1, r, Co);
    RNH_CHECK_LAUNCH("rnh_uptail_fwd(border)");
    return 0;
}

extern "C" int rnh_uptail_dgrad(const float *d_o, const float *G, float *dy1, int B, int Hm, int Wm, int C1, int Co, int r, void *stream) {
    if (!d_o || !G || !dy1 || B < 1 || Hm < 1 || Wm < 1 || Co < 1 || r < 2 || r > 4) RNH_FAIL(RNH_E_ARG, "rnh_uptail_dgrad: bad arguments");
    if (C1 & 3) RNH_FAIL(RNH_E_ALIGN, "rnh_uptail_dgrad: C1 must be a multiple of 4");
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
    return 0;
}
