// ubench_valu.hip — VALU instruction-throughput micro-benchmark for gfx950 (decides the limb strategy of
// the Montgomery multiplier).  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 2048
#define NACC 8

template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t *out, uint32_t seed) {
    uint32_t t = blockIdx.x * 256 + threadIdx.x;
    uint64_t acc[NACC];
    uint32_t a = t * 2654435761u + seed, b = (t ^ seed) * 40503u + 7;
    double da = (double)a, db = 1.0000001, dacc[NACC];
    for (int i = 0; i < NACC; i++) { acc[i] = a + i; dacc[i] = da + i; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (OP == 0) acc[i] = (uint64_t)(uint32_t)acc[i] * b + acc[i];                 // v_mad_u64_u32
            if (OP == 1) acc[i] = (uint32_t)acc[i] * b + (uint32_t)i;                      // v_mul_lo_u32 (+add)
            if (OP == 2) acc[i] = __umulhi((uint32_t)acc[i], b) + a;                      // v_mul_hi_u32
            if (OP == 3) acc[i] = ((uint32_t)acc[i] & 0xFFFFFF) * (b & 0xFFFFFF) + a;      // v_mul_u32_u24 / mad_u32_u24
            if (OP == 4) dacc[i] = __builtin_fma(dacc[i], db, da);                        // v_fma_f64
            if (OP == 5) acc[i] = (uint32_t)acc[i] + b + (uint32_t)(acc[i] >> 7);         // v_add3 / adds
            if (OP == 6) acc[i] = acc[i] + ((uint64_t)b << 32 | a);                       // 64-bit add (add_co + addc)
            if (OP == 7) { float f = __uint_as_float((uint32_t)acc[i]); f = __builtin_fmaf(f, 1.0001f, 0.5f); acc[i] = __float_as_uint(f); } // v_fma_f32
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i] + (uint64_t)dacc[i];
    out[t] = s;
}

template <int OP>
static void run(const char *name, int ops_per) {
    const int blocks = 256 * 8;   // 8 blocks of 256 per CU = 8 waves/SIMD
    uint64_t *d;
    hipMalloc(&d, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double n = (double)blocks * 256 * ITERS * NACC * ops_per;
    // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
    double wave_insts = n / 64.0;
    double cyc = ms * 1e-3 * 2.4e9 * 1024 / wave_insts;
    printf("%-28s %8.3f ms  %8.2f Gop/s (lane-ops)  ~%5.2f cyc/wave-inst/SIMD @2.4GHz\n", name, ms, n / ms / 1e6, cyc);
    hipFree(d);
}

int main() {
    run<0>("v_mad_u64_u32", 1);
    run<1>("v_mul_lo_u32(+add)", 1);
    run<2>("v_mul_hi_u32(+add)", 1);
    run<3>("v_mul_u32_u24(+add)", 1);
    run<4>("v_fma_f64", 1);
    run<5>("v_add3_u32-ish", 1);
    run<6>("add_u64 (co+addc)", 1);
    run<7>("v_fma_f32", 1);
    return 0;
}
