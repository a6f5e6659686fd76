// ubench_loop.hip - what does ROLLING the emitted bit-plane code buy?  (VERDICT r5 "next" #2: the micro-benchmark that comes
// before the emitter change.)  The emitted kernel `cw_bits_jit` is 13 MB of straight-line code per wave that is FETCHED, not
// cached; tools/ubench_fetch.hip showed such code running 2x slower beside a 6 TB/s reader and 16x slower beside a saturating
// writer.  Here the same 1 M instructions per wave (5 x v_bitop3_b32 + 3 x v_xor_b32 per group of 8 = 6.5 bytes each, the real
// code's mix) run as ONE straight line or as a loop over a body of 2 K ... 256 K instructions (13 KB ... 1.7 MB; the
// instruction cache is 64 KB, one L2 4 MB), one wave per SIMD on every CU:
//   * alone, beside a reader, beside a writer (another kernel on a second stream),
//   * with the kernel's OWN row stores (one 256-byte buffer_store_dword per 16 instructions, own 16 MB chunk per wave: the
//     emitted kernel's store stream), alone and beside a reader,
//   * as workgroups of 4 waves that meet at an s_barrier every 2 K instructions (do waves that stay together share fetches?).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_loop.hip -o gpurun_in/ubench_loop && gpurun_in/ubench_loop
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define S2(x) #x
#define S1(x) S2(x)
// one group of 8 instructions on 8 independent chains; with STORE every second group ends in a row store (soffset s24 walks
// 4 KB pages, the immediate walks the 16 rows of a page), with BAR every 256th group in an s_barrier
#define GROUP8                                                                                                        \
    "v_bitop3_b32 %0, %8, %9, %0 bitop3:0x96\n v_xor_b32 %1, %9, %1\n v_bitop3_b32 %2, %8, %9, %2 bitop3:0x96\n"        \
    "v_xor_b32 %3, %9, %3\n v_bitop3_b32 %4, %8, %9, %4 bitop3:0x96\n v_bitop3_b32 %5, %8, %9, %5 bitop3:0x96\n"        \
    "v_xor_b32 %6, %8, %6\n v_bitop3_b32 %7, %8, %9, %7 bitop3:0x96\n"
#define STORE_STEP                                                                                                    \
    ".if (ctr %% 2) == 1\n buffer_store_dword %0, %10, s[12:15], s24 offen offset:((ctr / 2) %% 16) * 256 nt\n"      \
    ".if ((ctr / 2) %% 16) == 15\n s_add_u32 s24, s24, 0x1000\n .endif\n .endif\n"
#define STORE_STEP8                                                                                                   \
    "buffer_store_dword %0, %10, s[12:15], s24 offen offset:(ctr %% 16) * 256 nt\n"                                    \
    ".if (ctr %% 16) == 15\n s_add_u32 s24, s24, 0x1000\n .endif\n"
#define BAR_STEP ".if (ctr %% 256) == 255\n s_barrier\n .endif\n"
#define NO_STEP ""

#define LOOP_KERNEL(NAME, N8, ITER, STEP, WG)                                                                         \
    __global__ void __launch_bounds__(WG) NAME(uint32_t *out, uint8_t *rows, uint32_t seed) {                           \
        uint32_t t = blockIdx.x * WG + threadIdx.x;                                                                     \
        uint32_t a = t * 2654435761u + seed, b = (t ^ seed) * 40503u + 7u;                                              \
        uint32_t c0 = a + 8, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;            \
        uint32_t lane4 = (threadIdx.x & 63) * 4;                                                                        \
        uint8_t *chunk = rows + (size_t)(t >> 6) * (32u << 20);                                                         \
        asm volatile("s_mov_b32 s12, %11\n s_and_b32 s13, %12, 0xffff\n s_mov_b32 s14, 0x02000000\n s_mov_b32 s15, 0x00020000\n" \
                     "s_mov_b32 s24, 0\n s_mov_b32 s20, " S1(ITER) "\n s_getpc_b64 s[22:23]\n"                          \
                     ".set ctr, 0\n .rept " S1(N8) "\n" GROUP8 STEP ".set ctr, ctr + 1\n .endr\n"                        \
                     "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc0 1f\n s_setpc_b64 s[22:23]\n 1:\n"      \
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)                    \
                     : "v"(a), "v"(b), "v"(lane4), "s"(__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)chunk)), "s"(__builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)chunk >> 32))) \
                     : "s12", "s13", "s14", "s15", "s20", "s22", "s23", "s24", "scc", "memory");                          \
        out[t] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;                                                                 \
    }
// (s_getpc_b64 returns the address of the instruction behind it: the first instruction of the body)

// 1 M instructions per wave = 131 072 groups
LOOP_KERNEL(line_1m, 131072, 1, NO_STEP, 64)
LOOP_KERNEL(loop_256k, 32768, 4, NO_STEP, 64)
LOOP_KERNEL(loop_64k, 8192, 16, NO_STEP, 64)
LOOP_KERNEL(loop_16k, 2048, 64, NO_STEP, 64)
LOOP_KERNEL(loop_8k, 1024, 128, NO_STEP, 64)
LOOP_KERNEL(loop_4k, 512, 256, NO_STEP, 64)
LOOP_KERNEL(loop_2k, 256, 512, NO_STEP, 64)
LOOP_KERNEL(st_line_1m, 131072, 1, STORE_STEP, 64)
LOOP_KERNEL(st_loop_256k, 32768, 4, STORE_STEP, 64)
LOOP_KERNEL(st_loop_64k, 8192, 16, STORE_STEP, 64)
LOOP_KERNEL(st_loop_16k, 2048, 64, STORE_STEP, 64)
LOOP_KERNEL(st_loop_8k, 1024, 128, STORE_STEP, 64)
LOOP_KERNEL(st_loop_4k, 512, 256, STORE_STEP, 64)
LOOP_KERNEL(st8_line_1m, 131072, 1, STORE_STEP8, 64)
LOOP_KERNEL(st8_loop_64k, 8192, 16, STORE_STEP8, 64)
LOOP_KERNEL(st8_loop_4k, 512, 256, STORE_STEP8, 64)
LOOP_KERNEL(bar_line_1m, 131072, 1, BAR_STEP, 256)
LOOP_KERNEL(bar_loop_256k, 32768, 4, BAR_STEP, 256)

__global__ void __launch_bounds__(256) writer(u32x4 *out, size_t n16, int passes) {
    const u32x4 v = {1u, 2u, 3u, 4u};
    for (int p = 0; p < passes; p++)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = v;
}
__global__ void __launch_bounds__(256) reader(const u32x4 *in, size_t n16, int passes, uint32_t *sink) {
    uint32_t acc = 0;
    for (int p = 0; p < passes; p++)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const u32x4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x1234567u) *sink = acc;
}

typedef void (*kern_t)(uint32_t *, uint8_t *, uint32_t);
int main() {
    uint32_t *out, *sink; u32x4 *big; uint8_t *rows;
    const size_t big_bytes = 8ull << 30, rows_bytes = 1024ull * (32u << 20);
    CK(hipMalloc(&out, 1024 * 64 * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&big, big_bytes)); CK(hipMemset(big, 0, big_bytes));
    CK(hipMalloc(&rows, rows_bytes)); CK(hipMemset(rows, 0, rows_bytes));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, b0, b1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    struct K { const char *name; kern_t k; int wg; int stores; };
    K ks[] = {{"straight line, 1 M instructions (6.8 MB)", line_1m, 64, 0}, {"loop 4 x 256 K (1.7 MB body)", loop_256k, 64, 0},
              {"loop 16 x 64 K (426 KB body)", loop_64k, 64, 0}, {"loop 64 x 16 K (106 KB body)", loop_16k, 64, 0},
              {"loop 128 x 8 K (53 KB body)", loop_8k, 64, 0}, {"loop 256 x 4 K (26 KB body)", loop_4k, 64, 0},
              {"loop 512 x 2 K (13 KB body)", loop_2k, 64, 0},
              {"+ own row stores: straight line", st_line_1m, 64, 1}, {"+ own row stores: 4 x 256 K", st_loop_256k, 64, 1},
              {"+ own row stores: 16 x 64 K", st_loop_64k, 64, 1}, {"+ own row stores: 64 x 16 K", st_loop_16k, 64, 1},
              {"+ own row stores: 128 x 8 K", st_loop_8k, 64, 1}, {"+ own row stores: 256 x 4 K", st_loop_4k, 64, 1},
              {"+ own row stores, one per 8: straight line", st8_line_1m, 64, 2}, {"+ own row stores, one per 8: 16 x 64 K", st8_loop_64k, 64, 2},
              {"+ own row stores, one per 8: 256 x 4 K", st8_loop_4k, 64, 2},
              {"4 waves + s_barrier / 2 K: straight line", bar_line_1m, 256, 0}, {"4 waves + s_barrier / 2 K: 4 x 256 K", bar_loop_256k, 256, 0}};
    const double n_ins = 131072.0 * 8;
    printf("one wave per SIMD on every CU (1 024 waves), %.0f K VALU instructions per wave (5 x 8-byte + 3 x 4-byte per 8)\n", n_ins / 1024);
    printf("own row stores = one 256-byte row per 16 instructions, 16 MB per wave, 17.2 GB per launch\n");
    for (auto &e : ks) {
        for (int mode = 0; mode < 3; mode++) {                       // 0 alone, 1 beside a writer, 2 beside a reader
            if (e.stores && mode == 1) continue;
            float best = 1e30f, tot = 0, bg_ms = 0;
            for (int r = 0; r < 4; r++) {
                CK(hipDeviceSynchronize());
                if (mode) {
                    CK(hipEventRecord(b0, s2));
                    if (mode == 1) hipLaunchKernelGGL(writer, dim3(1024), dim3(256), 0, s2, big, big_bytes / 16, 6);
                    else hipLaunchKernelGGL(reader, dim3(1024), dim3(256), 0, s2, big, big_bytes / 16, 6, sink);
                    CK(hipEventRecord(b1, s2));
                }
                CK(hipEventRecord(e0, s1));
                hipLaunchKernelGGL(e.k, dim3(1024 * 64 / e.wg), dim3(e.wg), 0, s1, out, rows, 7u + r);
                CK(hipEventRecord(e1, s1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (mode) { CK(hipEventSynchronize(b1)); CK(hipEventElapsedTime(&bg_ms, b0, b1)); }
                if (r) { tot += ms; best = ms < best ? ms : best; }
            }
            printf("%-44s %-16s avg %7.3f ms  best %7.3f ms = %5.2f ns per instruction", e.name, mode == 0 ? "alone" : mode == 1 ? "beside a writer" : "beside a reader",
                   tot / 3, best, best * 1e6 / n_ins);
            if (e.stores) printf("  own stores %.0f GB/s", 17.18 * e.stores / best * 1e3);
            if (mode) printf("   (the other kernel: %.1f ms for %.0f GB = %.0f GB/s)", bg_ms, 6 * big_bytes * 1e-9, 6 * big_bytes / bg_ms * 1e-6);
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
