// Probe (round 4): what does the ADDRESS PATTERN of a wave's 16-byte / 4-byte global stores cost on gfx950?  Every variant writes the same
// bytes (npix pixels x 256 B rows, like the fp32 cell state of a ConvLSTM cell: 64 channels) once; only the lane -> address map differs.
//   A  lane = pixel (stride 256 B), 16 B per lane, lane + 32 writes the adjacent 16 B  (transposed-product accumulators, as they stand)
//   B  4 adjacent lanes = 64 contiguous bytes of a pixel, 16 lanes a whole row          (what an LDS round trip buys)
//   C  fully contiguous (lane i writes bytes 16 i of a 1 KB run = 4 rows)
//   D  lane = pixel, 16 B per lane, ALL 64 lanes different pixels (no partner half)
//   E  4 B per lane, 32 lanes = 128 contiguous bytes of a pixel, lane + 32 another pixel   (non-transposed accumulators)
//   F  8 adjacent lanes = 128 contiguous bytes (a whole line), 16 B per lane
// build: hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip ; run: ./store_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int V>
__global__ void __launch_bounds__(256) k(float *out, long npix) {
    // one wave handles 32 pixels x 256 B = 8 KB per "tile"; tiles are dealt to waves grid-stride
    const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    const float4 v = make_float4(1.f + lane, 2.f, 3.f, 4.f);
    for (long t = wave; t < npix / 32; t += nw) {
        float *base = out + t * 32 * 64;                                  // 32 pixels x 64 floats
        if (V == 0) {                                                     // A: 8 instructions: piece q = 0..7 -> floats 8 q + 4 kh of pixel l31
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float4 *>(base + l31 * 64 + 8 * q + 4 * kh) = v;
        } else if (V == 1) {                                              // B: lanes 16 r + i: row r of 4 (of a group of 4 pixels), piece i of 16
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float4 *>(base + (4 * q + (lane >> 4)) * 64 + 4 * (lane & 15)) = v;
        } else if (V == 2) {                                              // C
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float4 *>(base + q * 256 + 4 * lane) = v;
        } else if (V == 3) {                                              // D: pixel = lane (64 pixels over two tiles' worth: use 2 passes of 32 rows x 2)
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float4 *>(base + (lane & 31) * 64 + 4 * ((2 * q + kh + (lane & 31)) & 15)) = v;
        } else if (V == 4) {                                              // E: 32 dword instructions
#pragma unroll
            for (int q = 0; q < 32; ++q) base[((q & 15) * 2 + kh) * 64 + (q >> 4) * 32 + l31] = v.x;
        } else {                                                          // F: 8 lanes = 128 B, 8 rows per instruction... lane>>3 = row of 8, two pieces rows of 32 floats
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float4 *>(base + ((q >> 1) * 8 + (lane >> 3)) * 64 + (q & 1) * 32 + 4 * (lane & 7)) = v;
        }
    }
}

int main() {
    const long npix = 8L * 128 * 128 * 8;                                 // 8 launches' worth: 268 MB
    float *out;
    hipMalloc(&out, npix * 64 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[6] = {"A lane=pixel, 16B, +32 partner", "B 4 lanes = 64B", "C contiguous", "D lane=pixel, scattered pieces", "E dword, 32 lanes = 128B", "F 8 lanes = 128B"};
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < 6; ++v) {
            hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) {
                switch (v) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, out, npix); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, out, npix); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, out, npix); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(2048), dim3(256), 0, 0, out, npix); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(2048), dim3(256), 0, 0, out, npix); break;
                    default: hipLaunchKernelGGL(k<5>, dim3(2048), dim3(256), 0, 0, out, npix); break;
                }
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-36s %8.1f us per 268 MB  = %6.2f TB/s\n", names[v], ms / 5 * 1e3, npix * 256.0 / (ms / 5 * 1e-3) / 1e12);
        }
    return 0;
}
