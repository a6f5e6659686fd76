// ubench_rowstream.hip - the memory side of the emitted bit-plane kernel (cw_bits_jit) as a synthetic stream: 1 024 waves, one per
// SIMD, each stores ROWS rows of 256 bytes (a dword per lane) front to back with FILL valu instructions in between and re-reads
// some older rows, under different TABLE LAYOUTS:
//   A  T[chunk w][slot s][256 B]                         (the shipped layout: every wave walks its own 38 MB region)
//   B  T[tile][slot s][chunk in tile (C)][256 B]          (row s of C waves contiguous: C x 256 B runs written by waves in step)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_rowstream.hip -o gpurun_in/ubench_rowstream
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// LAYOUT 0: A; otherwise C = LAYOUT chunks per tile.  NT: streaming stores.  LOADS: re-reads per 5 stores (0, 1, 2).  FILL: valu ops per row
template <int LAYOUT, int NT, int LOADS, int FILL>
__global__ void __launch_bounds__(256) rows(uint32_t *T, uint32_t slots, uint32_t n_rows, uint32_t *sink) {
    extern __shared__ uint32_t hog[];                                   // one workgroup per CU: one wave per SIMD
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    size_t base, stride;                                                // in dwords
    if (LAYOUT == 0) { base = (size_t)w * slots * 64; stride = 64; }
    else { base = ((size_t)(w / LAYOUT) * slots * LAYOUT + w % LAYOUT) * 64; stride = (size_t)LAYOUT * 64; }
    uint32_t x = w * 2654435761u + lane, acc = 0;
    uint32_t *p = T + base + lane;
    for (uint32_t s = 0; s < n_rows; s++) {
#pragma unroll
        for (int f = 0; f < FILL; f++) x = (x ^ (x >> 7)) + 0x9E3779B9u;
        if (NT) __builtin_nontemporal_store(x, p + (size_t)s * stride); else p[(size_t)s * stride] = x;
        if (LOADS && s % 5 == 0 && s >= 24000) {
            acc += p[(size_t)(s - 3000) * stride];
            if (LOADS > 1) acc += p[(size_t)(s - 23456) * stride];
        }
    }
    if (acc == 0x12345u) sink[0] = acc + hog[0];
}

template <class F> static void run(const char *name, double bytes, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    float tot = 0, best = 1e30f;
    for (int r = 0; r < 4; r++) {
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; best = ms < best ? ms : best;
    }
    printf("%-58s avg %7.3f ms  best %7.3f ms  %7.1f GB/s stored\n", name, tot / 4, best, bytes / (tot / 4) * 1e-6);
    fflush(stdout);
}

int main() {
    const uint32_t slots = 149363, n_rows = 144040, waves = 1024;
    uint32_t *T, *sink;
    CK(hipMalloc(&T, (size_t)waves * slots * 256)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(T, 0, (size_t)waves * slots * 256));
    const double bytes = (double)waves * n_rows * 256;
    printf("%u waves x %u rows x 256 B = %.1f GB stored per launch (table %.1f GB)\n", waves, n_rows, bytes * 1e-9, (double)waves * slots * 256 * 1e-9);
    const size_t lds = 100 * 1024;
#define R(L, NT, LD, F, name) { CK(hipFuncSetAttribute((const void *)rows<L, NT, LD, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        run(name, bytes, [&] { hipLaunchKernelGGL((rows<L, NT, LD, F>), dim3(waves / 4), dim3(256), lds, 0, T, slots, n_rows, sink); }); }
    R(0, 1, 0, 0, "A  stores only, streaming, no filler")
    R(0, 0, 0, 0, "A  stores only, plain, no filler")
    R(32, 1, 0, 0, "B C=32  stores only, streaming, no filler")
    R(128, 1, 0, 0, "B C=128 stores only, streaming, no filler")
    R(1024, 1, 0, 0, "B C=1024 stores only, streaming, no filler")
    R(128, 0, 0, 0, "B C=128 stores only, plain, no filler")
    R(0, 1, 0, 10, "A  stores + 10 valu per row, streaming")
    R(128, 1, 0, 10, "B C=128 stores + 10 valu per row, streaming")
    R(1024, 1, 0, 10, "B C=1024 stores + 10 valu per row, streaming")
    R(0, 1, 2, 10, "A  stores + 0.4 re-reads + 10 valu per row, streaming")
    R(0, 0, 2, 10, "A  stores + 0.4 re-reads + 10 valu per row, plain")
    R(32, 1, 2, 10, "B C=32  stores + 0.4 re-reads + 10 valu, streaming")
    R(128, 1, 2, 10, "B C=128 stores + 0.4 re-reads + 10 valu, streaming")
    R(128, 0, 2, 10, "B C=128 stores + 0.4 re-reads + 10 valu, plain")
    R(1024, 1, 2, 10, "B C=1024 stores + 0.4 re-reads + 10 valu, streaming")
    R(0, 1, 2, 40, "A  stores + 0.4 re-reads + 40 valu per row, streaming")
    R(128, 1, 2, 40, "B C=128 stores + 0.4 re-reads + 40 valu, streaming")
    R(0, 1, 0, 0, "A  stores only, streaming, no filler (again)")
    return 0;
}
