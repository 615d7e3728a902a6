// Probe (round 5): what does an LDS-DMA buffer load (buffer_load_dwordx4 ... offen lds) write for a lane whose offset is out of the buffer's range?
// The halo staging of conv_bf16d_kernel relies on out-of-range raw buffer loads returning 0 (zero padding without a branch); if the LDS-DMA form
// writes zeros for such lanes too, the halo can go straight to LDS.  Also: do EXEC-masked lanes leave their LDS slot untouched?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_oob.hip -o /tmp/lds_dma_oob && /tmp/lds_dma_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned *src, unsigned *out, int nbytes) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * 256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned long long u = (unsigned long long)src;
    i32x4 rs = {(int)(u & 0xffffffffu), (int)(u >> 32), nbytes, 0x00020000};
    rs[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);
    rs[2] = __builtin_amdgcn_readfirstlane(rs[2]); rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);
    // image 0: odd lanes out of range (offset -1); image 1: odd lanes EXEC-masked
    const int voff = (lane & 1) ? -1 : lane * 16;
    const unsigned l0 = (unsigned)(unsigned long long)(&lds[0]), l1 = (unsigned)(unsigned long long)(&lds[256]);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"s"(l0), "v"(voff), "s"(rs) : "memory");
    if (!(lane & 1)) {
        const int v2 = lane * 16;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" ::"s"(l1), "v"(v2), "s"(rs) : "memory");
    }
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}

int main() {
    unsigned *src, *out, h[512], hs[256];
    for (int i = 0; i < 256; ++i) hs[i] = 0x1000u + i;
    hipMalloc(&src, 1024); hipMalloc(&out, 2048);
    hipMemcpy(src, hs, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, 1024);
    hipMemcpy(h, out, 2048, hipMemcpyDeviceToHost);
    int ok_even = 1, oob_zero = 1, masked_untouched = 1, ok_even1 = 1;
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
            const unsigned a = h[l * 4 + k], b = h[256 + l * 4 + k];
            if (!(l & 1)) { ok_even &= a == 0x1000u + l * 4 + k; ok_even1 &= b == 0x1000u + l * 4 + k; }
            else { oob_zero &= a == 0; masked_untouched &= b == 0xdeadbeefu; }
        }
    printf("in-range lanes copied: %d / %d; out-of-range lanes wrote zeros: %d (lane 1 holds %08x); EXEC-masked lanes left their slot untouched: %d (lane 1 holds %08x)\n",
           ok_even, ok_even1, oob_zero, h[4], masked_untouched, h[256 + 4]);
    return 0;
}
