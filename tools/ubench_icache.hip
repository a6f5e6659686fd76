// ubench_icache.hip — does a gfx950 wave sustain its issue rate on STRAIGHT-LINE code far larger than the 64 KB
// instruction cache?  (Feasibility of per-circuit emitted gate code: one v_bitop3_b32 per gate, 0.7 M gates = 6 MB.)
// Not on the product path.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_icache.hip -o tools/ubench_icache && tools/ubench_icache
//
// Each kernel is ONE inline-asm block of N instructions (8 independent chains of v_bitop3_b32, no loop), N = 2 K .. 1 M
// (16 KB .. 8 MB of code).  Launched with 1 / 2 / 4 waves per SIMD on every CU; s_memtime per wave and HIP events.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define BODY8                                                                                                         \
    "v_bitop3_b32 %0, %8, %9, %0 bitop3:0x96\n v_bitop3_b32 %1, %8, %9, %1 bitop3:0x96\n"                               \
    "v_bitop3_b32 %2, %8, %9, %2 bitop3:0x96\n v_bitop3_b32 %3, %8, %9, %3 bitop3:0x96\n"                               \
    "v_bitop3_b32 %4, %8, %9, %4 bitop3:0x96\n v_bitop3_b32 %5, %8, %9, %5 bitop3:0x96\n"                               \
    "v_bitop3_b32 %6, %8, %9, %6 bitop3:0x96\n v_bitop3_b32 %7, %8, %9, %7 bitop3:0x96\n"

#define S2(x) #x
#define S1(x) S2(x)

#define LINE_KERNEL(NAME, REPT8)                                                                                      \
    __global__ void __launch_bounds__(256) NAME(uint64_t *out, uint32_t seed) {                                         \
        uint32_t t = blockIdx.x * 256 + threadIdx.x;                                                                    \
        uint32_t a = t * 2654435761u + seed, b = (t ^ seed) * 40503u + 7u;                                              \
        uint32_t c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;            \
        uint64_t t0 = __builtin_readcyclecounter();                                                                     \
        asm volatile(".rept " S1(REPT8) "\n" BODY8 ".endr\n"                                                            \
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)                   \
                     : "v"(a), "v"(b));                                                                                 \
        uint64_t t1 = __builtin_readcyclecounter();                                                                     \
        out[t] = (uint64_t)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7) | ((t1 - t0) << 32);                                 \
    }

LINE_KERNEL(k_2k, 256)
LINE_KERNEL(k_8k, 1024)
LINE_KERNEL(k_32k, 4096)
LINE_KERNEL(k_128k, 16384)
LINE_KERNEL(k_512k, 65536)
LINE_KERNEL(k_1m, 131072)

typedef void (*kern_t)(uint64_t *, uint32_t);

int main() {
    struct { const char *name; kern_t k; double n; } ks[] = {
        {"2K", k_2k, 2048}, {"8K", k_8k, 8192}, {"32K", k_32k, 32768}, {"128K", k_128k, 131072}, {"512K", k_512k, 524288}, {"1M", k_1m, 1048576}};
    FILE *js = fopen("gpurun_out/ubench_icache.json", "w");
    if (js) fprintf(js, "[");
    bool first = true;
    for (auto &e : ks) {
        for (int wps : {1, 2, 4}) {
            for (int cus : {256, 32}) {          // all CUs busy / one CU in eight (less sharing of L2 lines, less contention)
                const int blocks = cus * wps;
                uint64_t *d;
                CHECK(hipMalloc(&d, (size_t)blocks * 256 * 8));
                hipEvent_t e0, e1;
                CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 1u);
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 2u);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<uint64_t> h((size_t)blocks * 256);
                CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
                double clk = 0;
                for (size_t i = 0; i < h.size(); i += 64) clk += (double)(h[i] >> 32);
                clk /= (double)(h.size() / 64);
                // s_memtime counts at 100 MHz on gfx950? report both: raw counter per instruction and wall-clock ns per instruction
                printf("straight-line %-5s inst  waves/SIMD %d  blocks %4d  %8.3f ms  %7.3f ns/inst/wave (events)  %7.3f memtime-ticks/inst\n",
                       e.name, wps, blocks, ms, ms * 1e6 / e.n, clk / e.n);
                if (js) {
                    fprintf(js, "%s{\"inst\":%.0f,\"code_bytes\":%.0f,\"waves_per_simd\":%d,\"blocks\":%d,\"ms\":%.4f,\"ns_per_inst_per_wave\":%.4f}",
                            first ? "" : ",\n", e.n, e.n * 8, wps, blocks, ms, ms * 1e6 / e.n);
                    first = false;
                }
                CHECK(hipFree(d));
            }
        }
    }
    if (js) { fprintf(js, "]\n"); fclose(js); }
    return 0;
}
