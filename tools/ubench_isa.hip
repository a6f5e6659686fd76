// ubench_isa.hip — ISA-level issue-rate micro-benchmarks for gfx950 (calibrates the VALU peak of bench.py's roofline and
// prices the vector-memory instructions of the bit-plane evaluation kernel).  Not on the product path.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_isa.hip -o tools/ubench_isa && tools/ubench_isa
//
// VALU part: every kernel body is an inline-asm block of REPT instructions of ONE opcode (the instruction count is what
// the assembler emits: REPT per loop trip, checked with `llvm-objdump -d`), on `CH` independent register chains, timed
// with s_memtime (shader clocks) per wave and with HIP events; 1, 2, 4, 8 waves per SIMD (blocks of 256 threads =
// one wave per SIMD, k blocks per CU).  Reported: shader clocks per wave-instruction per SIMD = the reciprocal of the issue
// rate one SIMD sustains, and the chip-wide rate that follows (1024 SIMDs).
// VMEM part: a wave loop issuing one buffer instruction per trip in several lane patterns (all lanes in range, all out
// of range, a few in range, EXEC-masked), four waves per CU; clocks per instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define TRIPS 512
#define REPT 64            // instructions per trip (8 chains x 8)

// 8 independent chains v[acc0..acc7]; OPSTR uses %N placeholders: dst/src2 = the chain register
#define VALU_KERNEL(NAME, BODY8)                                                                                      \
    __global__ void __launch_bounds__(256) NAME(uint64_t *out, uint32_t seed) {                                         \
        uint32_t t = blockIdx.x * 256 + threadIdx.x;                                                                    \
        uint32_t a = t * 2654435761u + seed, b = (t ^ seed) * 40503u + 7u;                                              \
        uint32_t c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;            \
        uint64_t t0 = __builtin_readcyclecounter();                                                                     \
        for (int it = 0; it < TRIPS; it++) {                                                                            \
            asm volatile(".rept 8\n" BODY8 ".endr\n"                                                                   \
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)               \
                         : "v"(a), "v"(b));                                                                             \
        }                                                                                                               \
        uint64_t t1 = __builtin_readcyclecounter();                                                                     \
        out[t] = (uint64_t)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7) | ((t1 - t0) << 32);                                 \
    }

#define B8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define S_(x) #x
#define OP_BFI(i) "v_bfi_b32 %" S_(i) ", %8, %9, %" S_(i) "\n"
#define OP_AND(i) "v_and_b32 %" S_(i) ", %8, %" S_(i) "\n"
#define OP_BITOP3(i) "v_bitop3_b32 %" S_(i) ", %8, %9, %" S_(i) " bitop3:0x96\n"
#define OP_BFE(i) "v_bfe_i32 %" S_(i) ", %" S_(i) ", 3, 1\n"
#define OP_ADD(i) "v_add_u32 %" S_(i) ", %8, %" S_(i) "\n"
#define OP_XOR3(i) "v_xad_u32 %" S_(i) ", %8, %9, %" S_(i) "\n"
#define OP_LSHLOR(i) "v_lshl_or_b32 %" S_(i) ", %" S_(i) ", 1, %9\n"
#define OP_ANDOR(i) "v_and_or_b32 %" S_(i) ", %" S_(i) ", %8, %9\n"
#define OP_MAD64(i) "v_mul_lo_u32 %" S_(i) ", %8, %" S_(i) "\n"
#define OP_FMA(i) "v_fma_f32 %" S_(i) ", %8, %9, %" S_(i) "\n"
#define OP_PKFMA(i) "v_mov_b32 %" S_(i) ", %8\n"
// one chain only (fully dependent): the same register 8 times
#define OP_BFI_DEP(i) "v_bfi_b32 %0, %8, %9, %0\n"
#define OP_BITOP3_DEP(i) "v_bitop3_b32 %0, %8, %9, %0 bitop3:0x96\n"

VALU_KERNEL(k_bfi, B8(OP_BFI))
VALU_KERNEL(k_and, B8(OP_AND))
VALU_KERNEL(k_bitop3, B8(OP_BITOP3))
VALU_KERNEL(k_bfe, B8(OP_BFE))
VALU_KERNEL(k_add, B8(OP_ADD))
VALU_KERNEL(k_xor3, B8(OP_XOR3))
VALU_KERNEL(k_lshlor, B8(OP_LSHLOR))
VALU_KERNEL(k_andor, B8(OP_ANDOR))
VALU_KERNEL(k_mullo, B8(OP_MAD64))
VALU_KERNEL(k_fma, B8(OP_FMA))
VALU_KERNEL(k_mov, B8(OP_PKFMA))
VALU_KERNEL(k_bfi_dep, B8(OP_BFI_DEP))
VALU_KERNEL(k_bitop3_dep, B8(OP_BITOP3_DEP))

// v_mad_u64_u32 (32 x 32 + 64 -> 64: the workhorse of the 29-bit-limb Montgomery product): 8 independent 64-bit accumulators
__global__ void __launch_bounds__(256) k_mad64(uint64_t *out, uint32_t seed) {
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint32_t a = t * 2654435761u + seed, b = (t ^ seed) * 40503u + 7u;
    uint64_t c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < TRIPS; it++) {
        asm volatile(".rept 8\n"
                     "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n"
                     "v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
                     "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                     ".endr\n"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
                     : "v"(a), "v"(b)
                     : "vcc");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[t] = (uint64_t)(uint32_t)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7) | ((t1 - t0) << 32);
}

typedef void (*valu_kern_t)(uint64_t *, uint32_t);

static void run_valu(const char *name, valu_kern_t k, FILE *js, bool &first) {
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        uint64_t *d;
        CHECK(hipMalloc(&d, (size_t)blocks * 256 * 8));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1u);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 2u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> h((size_t)blocks * 256);
        CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
        double clk = 0;
        for (size_t i = 0; i < h.size(); i += 64) clk += (double)(h[i] >> 32);
        clk /= (double)(h.size() / 64);                       // average shader clocks per wave for TRIPS*REPT instructions
        const double n_inst = (double)TRIPS * REPT;
        // a SIMD hosts `wps` waves concurrently: clocks per wave-instruction per SIMD = wave clocks / (instructions * wps)
        const double clk_per_inst = clk / (n_inst * wps);
        const double wall_inst_per_s = (double)blocks * 4 * n_inst / (ms * 1e-3);
        printf("VALU %-14s waves/SIMD %d  %7.3f ms  %6.2f clk/wave-inst/SIMD (s_memtime)  %8.1f G wave-inst/s chip (events)  eff clock %.2f GHz\n",
               name, wps, ms, clk_per_inst, wall_inst_per_s / 1e9, clk / (ms * 1e-3) / 1e9);
        if (js) {
            fprintf(js, "%s{\"kind\":\"valu\",\"op\":\"%s\",\"waves_per_simd\":%d,\"ms\":%.4f,\"clk_per_wave_inst_per_simd\":%.3f,"
                        "\"wave_inst_per_s\":%.4e,\"eff_clock_ghz\":%.3f}", first ? "" : ",\n", name, wps, ms, clk_per_inst,
                    wall_inst_per_s, clk / (ms * 1e-3) / 1e9);
            first = false;
        }
        CHECK(hipFree(d));
    }
}

// ---- vector-memory instruction cost ------------------------------------------------------------------------------------
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define VTRIPS 2048

// MODE 0: store b64, all lanes in range (coalesced 512 B)      1: store b64, all lanes out of range (dropped)
//      2: store b64, 8 lanes in range scattered, rest OOR        3: store b64, EXEC-masked to 8 lanes
//      4: load b128 coalesced (1 KiB)   5: load b96 (768 B)   6: load b64 (512 B)   7: load b64 all OOR   8: load b64 8 lanes in range
template <int MODE>
__global__ void __launch_bounds__(256) k_vmem(char *buf, uint32_t bytes_per_wave, uint64_t *out) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    char *base = buf + (size_t)wave * bytes_per_wave;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes_per_wave, 0x00020000);
    uint32_t acc = 0;
    const uint32_t OOR = 0xFFFFFFF0u;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < VTRIPS; it++) {
        const uint32_t row = (uint32_t)(it & 15) * 1024u;               // 16 KiB window per wave: L1/L2 resident
        if (MODE == 0) { u32x2 v = {acc + it, lane}; __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)(row + lane * 8), 0, 0); }
        if (MODE == 1) { u32x2 v = {acc + it, lane}; __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)(OOR), 0, 0); }
        if (MODE == 2) { u32x2 v = {acc + it, lane}; __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)((lane & 7) == 0 ? row + lane * 8 * 13 % 1024 : OOR), 0, 0); }
        if (MODE == 3) { if ((lane & 7) == 0) { u32x2 v = {acc + it, lane}; __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)(row + lane * 8), 0, 0); } }
        if (MODE == 4) { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(row + lane * 16), 0, 0); acc += v.x; }
        if (MODE == 5) { u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(r, (int)((row + lane * 12)), 0, 0); acc += v.x; }
        if (MODE == 6) { u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(row + lane * 8), 0, 0); acc += v.x; }
        if (MODE == 7) { u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(OOR), 0, 0); acc += v.x; }
        if (MODE == 8) { u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)((lane & 7) == 0 ? row + lane * 8 * 13 % 1024 : OOR), 0, 0); acc += v.x; }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = (uint64_t)acc | ((t1 - t0) << 32);
}

template <int MODE>
static void run_vmem(const char *name, FILE *js, bool &first) {
    for (int wpc : {4, 8, 16}) {                              // waves per CU
        const int blocks = 256 * wpc / 4;
        const uint32_t bpw = 16 * 1024;
        char *buf;
        uint64_t *d;
        CHECK(hipMalloc(&buf, (size_t)blocks * 4 * bpw));
        CHECK(hipMemset(buf, 1, (size_t)blocks * 4 * bpw));
        CHECK(hipMalloc(&d, (size_t)blocks * 256 * 8));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_vmem<MODE>, dim3(blocks), dim3(256), 0, 0, buf, bpw, d);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_vmem<MODE>, dim3(blocks), dim3(256), 0, 0, buf, bpw, d);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> h((size_t)blocks * 256);
        CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
        double clk = 0;
        for (size_t i = 0; i < h.size(); i += 64) clk += (double)(h[i] >> 32);
        clk /= (double)(h.size() / 64);
        // per CU: wpc waves each issue VTRIPS instructions in `clk` clocks -> clocks of the CU's memory path per instruction
        const double clk_per_inst_cu = clk / ((double)VTRIPS * wpc);
        printf("VMEM %-26s waves/CU %2d  %7.3f ms  %7.2f clk per instruction per CU   (%7.1f clk per instruction per wave)\n", name, wpc, ms,
               clk_per_inst_cu, clk / VTRIPS);
        if (js) {
            fprintf(js, "%s{\"kind\":\"vmem\",\"op\":\"%s\",\"waves_per_cu\":%d,\"ms\":%.4f,\"clk_per_inst_per_cu\":%.3f,\"clk_per_inst_per_wave\":%.2f}",
                    first ? "" : ",\n", name, wpc, ms, clk_per_inst_cu, clk / VTRIPS);
            first = false;
        }
        CHECK(hipFree(buf));
        CHECK(hipFree(d));
    }
}

// ---- LDS instruction cost and the evaluation kernel's step shape --------------------------------------------------------
// MODE 0: 8 independent ds_read_b64 per trip, one wait     1: 8 ds_write_b64 per trip
//      2: the step of cw_bits_eval_kernel (3 operand reads one step ahead, 2 mask expansions, 4 v_bitop3, 1 write), 8 steps/trip
//      3: the same with the operand reads TWO steps ahead
#define LTRIPS 1024
template <int MODE>
__global__ void __launch_bounds__(256) k_lds(uint64_t *out, uint32_t seed) {
    extern __shared__ uint64_t lds_buf[];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    typedef __attribute__((address_space(3))) uint64_t lds_u64;
    const uint32_t base = wv * 8192;                                     // 16 rows of 512 B per wave
    for (uint32_t o = lane * 8; o < 8192; o += 512) *(lds_u64 *)(uintptr_t)(base + o) = seed + o;
    uint32_t addr[8];
#pragma unroll
    for (int k = 0; k < 8; k++) addr[k] = base + ((lane * 7 + k * 5) & 63) * 8 + k * 512;
    uint64_t acc = seed, a = 1, b = 2, c = 3, a2 = 4, b2 = 5, c2 = 6;
    const uint32_t km = (lane & 1) ? ~0u : 0u;
    uint64_t t0 = __builtin_readcyclecounter();
    constexpr int UNR = MODE == 4 ? 16 : MODE == 5 ? 32 : 1;          // 4, 5: mode 2 as 15 / 30 KB of straight-line code
    for (int it0 = 0; it0 < LTRIPS; it0 += UNR) {
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        const int it = it0 + u;
        if (MODE == 0) {
            uint64_t v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = *(lds_u64 *)(uintptr_t)addr[k];
#pragma unroll
            for (int k = 0; k < 8; k++) acc ^= v[k];
        }
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 8; k++) *(lds_u64 *)(uintptr_t)(addr[k] + 8192 - 8192) = acc + k;
            acc += it;
        }
        if (MODE >= 2) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t w0 = addr[k] + (uint32_t)it * 0, w1 = addr[(k + 3) & 7];
                const uint64_t na = *(lds_u64 *)(uintptr_t)(w0 & 0xFFFFF8u), nb = *(lds_u64 *)(uintptr_t)(w1), nc = *(lds_u64 *)(uintptr_t)(addr[(k + 5) & 7]);
                const uint32_t k1 = (uint32_t)((int32_t)(w0 << 28) >> 31) ^ km, k2 = (uint32_t)((int32_t)(w1 << 27) >> 31) ^ km;
                uint32_t lo, hi;
                asm volatile("v_bitop3_b32 %0, %2, %4, %8 bitop3:0x94\n\tv_bitop3_b32 %1, %3, %5, %8 bitop3:0x94\n\t"
                             "v_bitop3_b32 %0, %0, %6, %9 bitop3:0xbc\n\tv_bitop3_b32 %1, %1, %7, %9 bitop3:0xbc"
                             : "=&v"(lo), "=&v"(hi)
                             : "v"((uint32_t)a), "v"((uint32_t)(a >> 32)), "v"((uint32_t)b), "v"((uint32_t)(b >> 32)), "v"((uint32_t)c),
                               "v"((uint32_t)(c >> 32)), "v"(k1), "v"(k2));
                *(lds_u64 *)(uintptr_t)(addr[(k + 1) & 7]) = ((uint64_t)hi << 32) | lo;
                if (MODE != 3) { a = na; b = nb; c = nc; }
                else { a = a2; b = b2; c = c2; a2 = na; b2 = nb; c2 = nc; }
            }
        }
      }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = (uint64_t)(uint32_t)(acc ^ a ^ b ^ c) | ((t1 - t0) << 32);
}

template <int MODE>
static void run_lds(const char *name, int per_trip, FILE *js, bool &first) {
    for (int wps : {1, 2}) {
        const int blocks = 256 * wps;
        uint64_t *d;
        CHECK(hipMalloc(&d, (size_t)blocks * 256 * 8));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_lds<MODE>, dim3(blocks), dim3(256), 32768, 0, d, 1u);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_lds<MODE>, dim3(blocks), dim3(256), 32768, 0, d, 2u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> h((size_t)blocks * 256);
        CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
        double clk = 0;
        for (size_t i = 0; i < h.size(); i += 64) clk += (double)(h[i] >> 32);
        clk /= (double)(h.size() / 64);
        printf("LDS  %-22s waves/SIMD %d  %7.3f ms  %7.1f clk per %s per wave\n", name, wps, ms, clk / ((double)LTRIPS * 8), per_trip ? "step" : "instruction");
        if (js) {
            fprintf(js, "%s{\"kind\":\"lds\",\"op\":\"%s\",\"waves_per_simd\":%d,\"ms\":%.4f,\"clk_per_unit_per_wave\":%.2f}", first ? "" : ",\n", name, wps, ms,
                    clk / ((double)LTRIPS * 8));
            first = false;
        }
        CHECK(hipFree(d));
    }
}

// ---- HBM write roof: what a pure writer can reach (calibrates canonical_egress in bench.py) -----------------------------
// MODE 0: plain 16-byte stores   1: nontemporal 16-byte stores   2: copy (16-byte load + store)
template <int MODE>
__global__ void __launch_bounds__(256) k_fill(u32x4 *__restrict__ dst, const u32x4 *__restrict__ src, size_t n16, uint32_t seed) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n16; i += stride) {
        u32x4 v = {seed, 0u, 0u, 0u};
        if (MODE == 2) v = src[i];
        if (MODE == 1) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}
template <int MODE>
static void run_fill(const char *name, FILE *js, bool &first) {
    const size_t bytes = (size_t)8 << 30;
    u32x4 *d, *s2 = nullptr;
    CHECK(hipMalloc(&d, bytes));
    if (MODE == 2) { CHECK(hipMalloc(&s2, bytes)); CHECK(hipMemset(s2, 1, bytes)); }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_fill<MODE>, dim3(256 * 32), dim3(256), 0, 0, d, s2, bytes / 16, 1u);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_fill<MODE>, dim3(256 * 32), dim3(256), 0, 0, d, s2, bytes / 16, 2u + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double gbs = 3.0 * bytes / (ms * 1e-3) / 1e9;
    printf("HBM  %-22s 8 GiB x 3  %8.3f ms  %7.1f GB/s written%s\n", name, ms, gbs, MODE == 2 ? " (+ as much read)" : "");
    if (js) {
        fprintf(js, "%s{\"kind\":\"hbm\",\"op\":\"%s\",\"ms\":%.4f,\"written_GBps\":%.1f}", first ? "" : ",\n", name, ms, gbs);
        first = false;
    }
    CHECK(hipFree(d));
    if (s2) CHECK(hipFree(s2));
}

int main(int argc, char **argv) {
    FILE *js = argc > 1 ? fopen(argv[1], "w") : nullptr;
    bool first = true;
    if (js) fprintf(js, "[\n");
    if (argc <= 2) {
    run_valu("v_bfi_b32", k_bfi, js, first);
    run_valu("v_bitop3_b32", k_bitop3, js, first);
    run_valu("v_and_b32", k_and, js, first);
    run_valu("v_bfe_i32", k_bfe, js, first);
    run_valu("v_add_u32", k_add, js, first);
    run_valu("v_xad_u32", k_xor3, js, first);
    run_valu("v_lshl_or_b32", k_lshlor, js, first);
    run_valu("v_and_or_b32", k_andor, js, first);
    run_valu("v_mul_lo_u32", k_mullo, js, first);
    run_valu("v_mad_u64_u32", k_mad64, js, first);
    run_valu("v_fma_f32", k_fma, js, first);
    run_valu("v_mov_b32", k_mov, js, first);
    run_valu("v_bfi_b32.dep", k_bfi_dep, js, first);
    run_valu("v_bitop3.dep", k_bitop3_dep, js, first);
    }
    run_lds<0>("ds_read_b64 x8", 0, js, first);
    run_lds<1>("ds_write_b64 x8", 0, js, first);
    run_lds<2>("eval step, reads 1 ahead", 1, js, first);
    run_lds<3>("eval step, reads 2 ahead", 1, js, first);
    run_lds<4>("eval step, 15 KB of code", 1, js, first);
    run_lds<5>("eval step, 30 KB of code", 1, js, first);
    run_fill<0>("fill, plain stores", js, first);
    run_fill<1>("fill, nt stores", js, first);
    run_fill<2>("copy", js, first);
    if (argc > 2) { if (js) { fprintf(js, "\n]\n"); fclose(js); } return 0; }
    run_vmem<0>("store_b64 coalesced", js, first);
    run_vmem<1>("store_b64 all-OOR", js, first);
    run_vmem<2>("store_b64 8-lanes+OOR", js, first);
    run_vmem<3>("store_b64 exec-8-lanes", js, first);
    run_vmem<4>("load_b128 coalesced", js, first);
    run_vmem<5>("load_b96 coalesced", js, first);
    run_vmem<6>("load_b64 coalesced", js, first);
    run_vmem<7>("load_b64 all-OOR", js, first);
    run_vmem<8>("load_b64 8-lanes+OOR", js, first);
    if (js) { fprintf(js, "\n]\n"); fclose(js); }
    return 0;
}
