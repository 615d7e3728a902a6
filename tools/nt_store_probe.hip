// Do non-temporal (`nt`) stores of PIECES of one 128-byte line, written by workgroups on different XCDs, lose or corrupt data?
//
// Root-cause probe for the gate-store corruption of round 2 (csrc/conv_wino.hip, DESIGN.md section 4: with the `nt` bit the
// 16-byte pieces of one 128-byte line of the ConvLSTM gate tensor, written by the workgroups of the four column blocks, "came
// out corrupted now and then"; the kernel went back to plain stores).  This program isolates the pattern: PARTS workgroups -
// consecutive block ids, i.e. different XCDs under the round-robin placement (the XCC id of every writer is recorded and
// reported) - each store their own 128 / PARTS bytes of every 128-byte line of a buffer, with plain, `nt`, `sc1` or `sc0 sc1`
// stores, all at the same time (a spin barrier on a device counter lines the writers up so that the pieces of a line are in flight
// together); the kernel ends; a second kernel (or the host) reads everything back and counts wrong dwords.  Lines are also
// re-written LOOPS times with a changing value so that a stale or torn merge shows up as an old value.
//
//   hipcc --offload-arch=gfx950 -O2 tools/nt_store_probe.hip -o /tmp/nt_probe && /tmp/nt_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>      // 0 plain, 1 nt, 2 sc1, 3 sc0 sc1
__device__ __forceinline__ void store16(unsigned *p, u32x4 v) {
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// block b writes part (b % parts) of the lines of group (b / parts); every thread stores 16-byte pieces
template <int MODE>
__global__ void __launch_bounds__(256) writer(unsigned *buf, int lines_per_group, int parts, int loops, unsigned *arrive, int *xcc) {
    const int part = blockIdx.x % parts, group = blockIdx.x / parts;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = (int)(id & 0xf);
        // line the writers of a group up (bounded spin: never a hang)
        __hip_atomic_fetch_add(arrive + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int i = 0; i < 20000; ++i)
            if (__hip_atomic_load(arrive + group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)parts) break;
    }
    __syncthreads();
    const int pieces = (128 / parts) / 16;                       // 16-byte pieces of this block's part of a line
    for (int it = 0; it < loops; ++it)
        for (int e = threadIdx.x; e < lines_per_group * pieces; e += 256) {
            const int line = e / pieces, pc = e % pieces;
            const long dword = ((long)group * lines_per_group + line) * 32 + part * (32 / parts) + pc * 4;
            const unsigned v = (unsigned)(dword * 2654435761u) + (unsigned)it;
            store16<MODE>(buf + dword, u32x4{v, v + 1, v + 2, v + 3});
        }
}

__global__ void checker(const unsigned *buf, long ndwords, int loops, unsigned long long *bad) {
    unsigned long long n = 0;
    for (long d = (long)(blockIdx.x * blockDim.x + threadIdx.x) * 4; d < ndwords; d += (long)gridDim.x * blockDim.x * 4) {
        const unsigned v = (unsigned)(d * 2654435761u) + (unsigned)(loops - 1);
        for (int j = 0; j < 4; ++j) n += buf[d + j] != v + j;
    }
    if (n) atomicAdd(bad, n);
}

template <int MODE>
int run(const char *name, int parts, int loops) {
    const int groups = 512 / parts * 4, lines_per_group = 2048;            // 2048 blocks, 256 KB per group
    const long ndwords = (long)groups * lines_per_group * 32;
    unsigned *buf, *arrive;
    int *xcc;
    unsigned long long *bad;
    CK(hipMalloc(&buf, ndwords * 4));
    CK(hipMalloc(&arrive, groups * 4));
    CK(hipMalloc(&xcc, groups * parts * 4));
    CK(hipMalloc(&bad, 8));
    unsigned long long total_bad = 0, host_bad = 0;
    int distinct_xcc_groups = 0;
    unsigned *h = (unsigned *)malloc(ndwords * 4);
    int *hx = (int *)malloc(groups * parts * 4);
    for (int rep = 0; rep < 20; ++rep) {
        CK(hipMemset(buf, 0xee, ndwords * 4));
        CK(hipMemset(arrive, 0, groups * 4));
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(writer<MODE>, dim3(groups * parts), dim3(256), 0, 0, buf, lines_per_group, parts, loops, arrive, xcc);
        hipLaunchKernelGGL(checker, dim3(1024), dim3(256), 0, 0, buf, ndwords, loops, bad);
        CK(hipDeviceSynchronize());
        unsigned long long b;
        CK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
        total_bad += b;
        if (rep == 0) {
            CK(hipMemcpy(h, buf, ndwords * 4, hipMemcpyDeviceToHost));
            for (long d = 0; d < ndwords; ++d) host_bad += h[d] != (unsigned)((d & ~3L) * 2654435761u) + (unsigned)(loops - 1) + (unsigned)(d & 3);
            CK(hipMemcpy(hx, xcc, groups * parts * 4, hipMemcpyDeviceToHost));
            for (int g = 0; g < groups; ++g) {
                int mask = 0;
                for (int p = 0; p < parts; ++p) mask |= 1 << hx[g * parts + p];
                distinct_xcc_groups += __builtin_popcount(mask) == parts;
            }
        }
    }
    printf("%-8s pieces of %3d B from %d workgroups per line, %2d passes: wrong dwords (device check, 20 launches) %llu, (host check) %llu; "
           "groups whose writers sat on %d different XCDs: %d of %d\n", name, 128 / parts, parts, loops, total_bad, host_bad, parts,
           distinct_xcc_groups, groups);
    (void)hipFree(buf); (void)hipFree(arrive); (void)hipFree(xcc); (void)hipFree(bad); free(h); free(hx);
    return 0;
}

int main() {
    for (int loops = 1; loops <= 4; loops += 3)
        for (int parts = 2; parts <= 8; parts *= 2) {
            if (run<0>("plain", parts, loops) || run<1>("nt", parts, loops) || run<2>("sc1", parts, loops) || run<3>("sc0 sc1", parts, loops)) return 2;
        }
    return 0;
}
