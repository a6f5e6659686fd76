// ubench_fetch.hip - is a wave's straight-line code (fetched, not cached) slowed down by OTHER kernels' memory traffic, and does
// it speed up with fewer bytes per instruction?  One wave per SIMD on every CU runs 512 K independent-chain VALU instructions with
// no memory operation of its own: as 8-byte VOP3 (v_bitop3_b32) or 4-byte VOP2 (v_xor_b32); alone, beside a store stream, beside
// a load stream (second HIP stream).  The experiment behind DESIGN 4.0b "bytes of code are time".
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_fetch.hip -o gpurun_in/ubench_fetch && gpurun_in/ubench_fetch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define BODY8_VOP3                                                                                                    \
    "v_bitop3_b32 %0, %8, %9, %0 bitop3:0x96\n v_bitop3_b32 %1, %8, %9, %1 bitop3:0x96\n"                               \
    "v_bitop3_b32 %2, %8, %9, %2 bitop3:0x96\n v_bitop3_b32 %3, %8, %9, %3 bitop3:0x96\n"                               \
    "v_bitop3_b32 %4, %8, %9, %4 bitop3:0x96\n v_bitop3_b32 %5, %8, %9, %5 bitop3:0x96\n"                               \
    "v_bitop3_b32 %6, %8, %9, %6 bitop3:0x96\n v_bitop3_b32 %7, %8, %9, %7 bitop3:0x96\n"
#define BODY8_VOP2                                                                                                    \
    "v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %9, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %9, %3\n"                       \
    "v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %9, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %9, %7\n"
#define BODY8_MIX                                                                                                     \
    "v_bitop3_b32 %0, %8, %9, %0 bitop3:0x96\n v_xor_b32 %1, %9, %1\n v_bitop3_b32 %2, %8, %9, %2 bitop3:0x96\n"        \
    "v_xor_b32 %3, %9, %3\n v_bitop3_b32 %4, %8, %9, %4 bitop3:0x96\n v_bitop3_b32 %5, %8, %9, %5 bitop3:0x96\n"        \
    "v_xor_b32 %6, %8, %6\n v_bitop3_b32 %7, %8, %9, %7 bitop3:0x96\n"
#define S2(x) #x
#define S1(x) S2(x)
#define LINE_KERNEL(NAME, REPT8, BODY)                                                                                \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed) {                                         \
        uint32_t t = blockIdx.x * 256 + threadIdx.x;                                                                    \
        uint32_t a = t * 2654435761u + seed, b = (t ^ seed) * 40503u + 7u;                                              \
        uint32_t c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;            \
        asm volatile(".rept " S1(REPT8) "\n" BODY ".endr\n"                                                             \
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b)); \
        out[t] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;                                                                 \
    }
LINE_KERNEL(code_vop3, 65536, BODY8_VOP3)
LINE_KERNEL(code_vop2, 65536, BODY8_VOP2)
LINE_KERNEL(code_mix, 65536, BODY8_MIX)
#define BODY8_PRIO "s_setprio 3\n" BODY8_VOP3
LINE_KERNEL(code_vop3_prio, 65536, BODY8_VOP3)


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

typedef void (*kern_t)(uint32_t *, uint32_t);
int main() {
    uint32_t *out, *sink; u32x4 *big;
    const size_t big_bytes = 8ull << 30;
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&big, big_bytes)); CK(hipMemset(big, 0, big_bytes));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, b0, b1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    struct { const char *name; kern_t k; double bytes; } ks[] = {{"8-byte VOP3 (v_bitop3_b32)", code_vop3, 8}, {"4-byte VOP2 (v_xor_b32)", code_vop2, 4},
                                                                 {"5 x VOP3 + 3 x VOP2 (6.5 bytes)", code_mix, 6.5}};
    const double n_ins = 65536.0 * 8;
    printf("one wave per SIMD on every CU (256 workgroups x 4 waves), %.0f K instructions of straight-line code per wave\n", n_ins / 1024);
    for (auto &e : ks) {
        for (int mode = 0; mode < 3; mode++) {                       // 0 alone, 1 beside a writer, 2 beside a reader
            float best = 1e30f, tot = 0, bg_ms = 0;
            for (int r = 0; r < 4; r++) {
                CK(hipDeviceSynchronize());
                if (mode) {
                    CK(hipEventRecord(b0, s2));
                    if (mode == 1) hipLaunchKernelGGL(writer, dim3(1024), dim3(256), 0, s2, big, big_bytes / 16, 12);
                    else hipLaunchKernelGGL(reader, dim3(1024), dim3(256), 0, s2, big, big_bytes / 16, 12, sink);
                    CK(hipEventRecord(b1, s2));
                }
                CK(hipEventRecord(e0, s1));
                hipLaunchKernelGGL(e.k, dim3(256), dim3(256), 0, s1, out, 7u + r);
                CK(hipEventRecord(e1, s1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (mode) { CK(hipEventSynchronize(b1)); CK(hipEventElapsedTime(&bg_ms, b0, b1)); }
                if (r) { tot += ms; best = ms < best ? ms : best; }
            }
            printf("%-34s %-18s avg %7.3f ms  best %7.3f ms = %5.2f ns per instruction%s", e.name, mode == 0 ? "alone" : mode == 1 ? "beside a writer" : "beside a reader",
                   tot / 3, best, best * 1e6 / n_ins, mode ? "" : "\n");
            if (mode) printf("   (the other kernel: %.1f ms for %.0f GB = %.0f GB/s)\n", bg_ms, 12 * big_bytes * 1e-9, 12 * big_bytes / bg_ms * 1e-6);
            fflush(stdout);
        }
    }
    // how much write traffic does it take?  writers of 64 / 256 / 512 workgroups beside the 8-byte and the 4-byte code
    for (int wb : {64, 256, 512}) for (int which = 0; which < 2; which++) {
        float best = 1e30f, bg_ms = 0;
        const int passes = wb == 64 ? 2 : wb == 256 ? 6 : 10;
        for (int r = 0; r < 3; r++) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(b0, s2));
            hipLaunchKernelGGL(writer, dim3(wb), dim3(256), 0, s2, big, big_bytes / 16, passes);
            CK(hipEventRecord(b1, s2));
            CK(hipEventRecord(e0, s1));
            hipLaunchKernelGGL(ks[which].k, dim3(256), dim3(256), 0, s1, out, 9u + r);
            CK(hipEventRecord(e1, s1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipEventSynchronize(b1)); CK(hipEventElapsedTime(&bg_ms, b0, b1));
            if (r) best = ms < best ? ms : best;
        }
        printf("%-34s beside a writer of %3d workgroups: best %7.3f ms = %5.2f ns per instruction   (the writer: %.1f ms, %.0f GB/s)\n", ks[which].name, wb,
               best, best * 1e6 / n_ins, bg_ms, passes * big_bytes / bg_ms * 1e-6);
        fflush(stdout);
    }
    return 0;
}
