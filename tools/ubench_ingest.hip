// ubench_ingest.hip - variants of the 32-byte ingest (cw_bits.hip::cw_bits_ingest_kernel) on the default bench line's shape:
// 2^21 instances x 2 048 inputs x 32 bytes = 137 GB in, one 64-bit mask per (group of 64 instances, input) out (emitted layout, sh = 5).
// Every variant must leave the table and the flag words of variant 0 (checksums compared); the last line is a read-only
// ceiling (a grid-stride uint4 sum over the same buffer).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_ingest.hip -o tools/ubench_ingest && tools/ubench_ingest [log2 batch] [n_in]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t *bits_group(uint64_t *T, uint64_t slots, uint32_t sh, uint32_t g) {
    return T + (((size_t)(g >> sh) * slots) << sh) + (g & ((1u << sh) - 1u));
}

__global__ void fill_kernel(uint4 *in, size_t n_elems, uint32_t n_in) {        // element e = (instance, input): a bit, rarely a non-bit
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_elems; e += (size_t)gridDim.x * blockDim.x) {
        uint64_t h = e * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        uint4 lo = make_uint4((uint32_t)h & 1u, 0, 0, 0), hi = make_uint4(0, 0, 0, 0);
        if ((h >> 40) == 12345u) { if (h & 2) lo.x = 2; else hi.z = 1; }          // ~1 in 16 M elements is not a bit
        in[2 * e] = lo;
        in[2 * e + 1] = hi;
    }
}

// ---- variant 0: the shipped kernel -------------------------------------------------------------------------------------------
template <int SWAP>
__global__ void __launch_bounds__(64) ingest_v0(const uint4 *__restrict__ in, uint64_t *__restrict__ T, uint64_t slots, uint32_t sh,
                                                uint32_t input_slot0, uint32_t n_in, uint32_t batch, uint64_t *fbmask, uint32_t nchunk) {
    uint32_t g, c;
    if (SWAP) { c = blockIdx.x % nchunk; g = blockIdx.x / nchunk; } else { g = blockIdx.x; c = blockIdx.y; }
    const uint32_t lane = threadIdx.x, k = c * 64 + lane;
    const bool have = k < n_in;
    const uint32_t i0 = g * 64, ni = min(64u, batch - i0);
    uint64_t mine = 0, badmask = 0;
    for (uint32_t ii = 0; ii < ni; ii++) {
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (have) {
            const size_t src = ((size_t)(i0 + ii) * n_in + k) * 2;
            lo = in[src];
            hi = in[src + 1];
        }
        const bool isbit = (lo.x <= 1u) & ((lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) == 0u);
        mine |= (uint64_t)(lo.x & 1u) << ii;
        if (__any(!isbit)) badmask |= 1ull << ii;
    }
    if (have) bits_group(T, slots, sh, g)[(size_t)(input_slot0 + k) << sh] = mine;
    if (lane == 0 && badmask) atomicOr((unsigned long long *)&fbmask[g], (unsigned long long)badmask);
}

// ---- variants 1..: the flag stays per lane until the end, full groups run a constant-trip loop unrolled U times ---------------
template <int U, int SWAP, int NT>
__global__ void __launch_bounds__(64) ingest_v1(const uint4 *__restrict__ in, uint64_t *__restrict__ T, uint64_t slots, uint32_t sh,
                                                uint32_t input_slot0, uint32_t n_in, uint32_t batch, uint64_t *fbmask, uint32_t nchunk) {
    uint32_t g, c;
    if (SWAP) { c = blockIdx.x % nchunk; g = blockIdx.x / nchunk; } else { g = blockIdx.x; c = blockIdx.y; }
    const uint32_t lane = threadIdx.x, k = c * 64 + lane;
    const bool have = k < n_in;
    const uint32_t i0 = g * 64, ni = min(64u, batch - i0);
    uint64_t mine = 0, bad = 0;
    const uint4 *p = in + ((size_t)i0 * n_in + (have ? k : 0)) * 2;
    const size_t step = (size_t)n_in * 2;
    if (ni == 64) {
#pragma unroll U
        for (uint32_t ii = 0; ii < 64; ii++) {
            uint4 lo, hi;
            if (NT) {
                const u32x4 a = __builtin_nontemporal_load((const u32x4 *)p), b = __builtin_nontemporal_load((const u32x4 *)p + 1);
                lo = make_uint4(a.x, a.y, a.z, a.w); hi = make_uint4(b.x, b.y, b.z, b.w);
            } else { lo = p[0]; hi = p[1]; }
            p += step;
            const uint32_t rest = lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w;
            const bool notbit = (lo.x > 1u) | (rest != 0u);
            mine |= (uint64_t)(lo.x & 1u) << ii;
            bad |= (uint64_t)notbit << ii;
        }
    } else {
        for (uint32_t ii = 0; ii < ni; ii++) {
            const uint4 lo = p[0], hi = p[1];
            p += step;
            const uint32_t rest = lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w;
            mine |= (uint64_t)(lo.x & 1u) << ii;
            bad |= (uint64_t)((lo.x > 1u) | (rest != 0u)) << ii;
        }
    }
    if (have) {
        bits_group(T, slots, sh, g)[(size_t)(input_slot0 + k) << sh] = mine;
        if (bad) atomicOr((unsigned long long *)&fbmask[g], (unsigned long long)bad);
    }
}

// ---- variant: a workgroup of four waves, wave w takes chunk 4 c + w of the same group ------------------------------------------
template <int U>
__global__ void __launch_bounds__(256) ingest_v4w(const uint4 *__restrict__ in, uint64_t *__restrict__ T, uint64_t slots, uint32_t sh,
                                                  uint32_t input_slot0, uint32_t n_in, uint32_t batch, uint64_t *fbmask, uint32_t nchunk4) {
    const uint32_t c = blockIdx.x % nchunk4, g = blockIdx.x / nchunk4;
    const uint32_t k = c * 256 + threadIdx.x;
    const bool have = k < n_in;
    const uint32_t i0 = g * 64, ni = min(64u, batch - i0);
    uint64_t mine = 0, bad = 0;
    const uint4 *p = in + ((size_t)i0 * n_in + (have ? k : 0)) * 2;
    const size_t step = (size_t)n_in * 2;
    if (ni == 64) {
#pragma unroll U
        for (uint32_t ii = 0; ii < 64; ii++) {
            const uint4 lo = p[0], hi = p[1];
            p += step;
            const uint32_t rest = lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w;
            mine |= (uint64_t)(lo.x & 1u) << ii;
            bad |= (uint64_t)((lo.x > 1u) | (rest != 0u)) << ii;
        }
    } else {
        for (uint32_t ii = 0; ii < ni; ii++) {
            const uint4 lo = p[0], hi = p[1];
            p += step;
            const uint32_t rest = lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w;
            mine |= (uint64_t)(lo.x & 1u) << ii;
            bad |= (uint64_t)((lo.x > 1u) | (rest != 0u)) << ii;
        }
    }
    if (have) {
        bits_group(T, slots, sh, g)[(size_t)(input_slot0 + k) << sh] = mine;
        if (bad) atomicOr((unsigned long long *)&fbmask[g], (unsigned long long)bad);
    }
}


// ---- variant: every wave starts at its own instance of the group (rotation r): waves that run together then sit in different
// 64 KB records instead of marching through them in step -----------------------------------------------------------------------
template <int U, int SWAP, int ROT>
__global__ void __launch_bounds__(64) ingest_rot(const uint4 *__restrict__ in, uint64_t *__restrict__ T, uint64_t slots, uint32_t sh,
                                                 uint32_t input_slot0, uint32_t n_in, uint32_t batch, uint64_t *fbmask, uint32_t nchunk) {
    uint32_t g, c;
    if (SWAP) { c = blockIdx.x % nchunk; g = blockIdx.x / nchunk; } else { g = blockIdx.x; c = blockIdx.y; }
    const uint32_t lane = threadIdx.x, k = c * 64 + lane;
    const bool have = k < n_in;
    const uint32_t i0 = g * 64, ni = min(64u, batch - i0);
    uint64_t mine = 0, bad = 0;
    const uint4 *base = in + ((size_t)i0 * n_in + (have ? k : 0)) * 2;
    const size_t step = (size_t)n_in * 2;
    if (ni == 64) {
        const uint32_t r = ROT == 1 ? (g * 0x9E3779B1u) >> 26 : ROT == 2 ? ((g * 32u + c) * 0x9E3779B1u) >> 26 : ROT == 3 ? (g & 63u) : (c * 2u) & 63u;
#pragma unroll U
        for (uint32_t ii = 0; ii < 64; ii++) {
            const uint32_t inst = (ii + r) & 63u;
            const uint4 *p = base + inst * step;
            const uint4 lo = p[0], hi = p[1];
            const uint32_t rest = lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w;
            mine |= (uint64_t)(lo.x & 1u) << inst;
            bad |= (uint64_t)((lo.x > 1u) | (rest != 0u)) << inst;
        }
    } else {
        const uint4 *p = base;
        for (uint32_t ii = 0; ii < ni; ii++) {
            const uint4 lo = p[0], hi = p[1];
            p += step;
            const uint32_t rest = lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w;
            mine |= (uint64_t)(lo.x & 1u) << ii;
            bad |= (uint64_t)((lo.x > 1u) | (rest != 0u)) << ii;
        }
    }
    if (have) {
        bits_group(T, slots, sh, g)[(size_t)(input_slot0 + k) << sh] = mine;
        if (bad) atomicOr((unsigned long long *)&fbmask[g], (unsigned long long)bad);
    }
}

// ---- variant: a wave takes 128 inputs (4 KiB runs, two masks per lane) ---------------------------------------------------------
template <int U>
__global__ void __launch_bounds__(64) ingest_wide(const uint4 *__restrict__ in, uint64_t *__restrict__ T, uint64_t slots, uint32_t sh,
                                                  uint32_t input_slot0, uint32_t n_in, uint32_t batch, uint64_t *fbmask, uint32_t nchunk2) {
    const uint32_t g = blockIdx.x, c = blockIdx.y, lane = threadIdx.x, k = c * 128 + lane;      // and k + 64
    const uint32_t i0 = g * 64, ni = min(64u, batch - i0);
    uint64_t m0 = 0, m1 = 0, bad = 0;
    const uint4 *p = in + ((size_t)i0 * n_in + k) * 2;
    const size_t step = (size_t)n_in * 2;
    if (k + 64 < n_in + 0u && ni == 64) {
#pragma unroll U
        for (uint32_t ii = 0; ii < 64; ii++) {
            const uint4 lo = p[0], hi = p[1], lo2 = p[128], hi2 = p[129];
            p += step;
            const uint32_t rest = lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w, rest2 = lo2.y | lo2.z | lo2.w | hi2.x | hi2.y | hi2.z | hi2.w;
            m0 |= (uint64_t)(lo.x & 1u) << ii;
            m1 |= (uint64_t)(lo2.x & 1u) << ii;
            bad |= (uint64_t)((lo.x > 1u) | (rest != 0u) | (lo2.x > 1u) | (rest2 != 0u)) << ii;
        }
        uint64_t *Tg = bits_group(T, slots, sh, g);
        Tg[(size_t)(input_slot0 + k) << sh] = m0;
        Tg[(size_t)(input_slot0 + k + 64) << sh] = m1;
        if (bad) atomicOr((unsigned long long *)&fbmask[g], (unsigned long long)bad);
    }
}


// ---- variant: workgroups of NW waves on consecutive chunks of one group, rotated start ---------------------------------------
template <int U, int NW, int ROT>
__global__ void __launch_bounds__(64 * NW) ingest_nw(const uint4 *__restrict__ in, uint64_t *__restrict__ T, uint64_t slots, uint32_t sh,
                                                     uint32_t input_slot0, uint32_t n_in, uint32_t batch, uint64_t *fbmask, uint32_t nchunkw) {
    const uint32_t c = blockIdx.x % nchunkw, g = blockIdx.x / nchunkw;
    const uint32_t k = c * (64 * NW) + threadIdx.x;
    const bool have = k < n_in;
    const uint32_t i0 = g * 64, ni = min(64u, batch - i0);
    uint64_t mine = 0, bad = 0;
    const uint4 *base = in + ((size_t)i0 * n_in + (have ? k : 0)) * 2;
    const size_t step = (size_t)n_in * 2;
    if (ni == 64) {
        const uint32_t r = ROT ? (g * 0x9E3779B1u) >> 26 : 0u;
#pragma unroll U
        for (uint32_t ii = 0; ii < 64; ii++) {
            const uint32_t inst = (ii + r) & 63u;
            const uint4 *p = base + inst * step;
            const uint4 lo = p[0], hi = p[1];
            const uint32_t rest = lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w;
            mine |= (uint64_t)(lo.x & 1u) << inst;
            bad |= (uint64_t)((lo.x > 1u) | (rest != 0u)) << inst;
        }
    } else {
        const uint4 *p = base;
        for (uint32_t ii = 0; ii < ni; ii++) {
            const uint4 lo = p[0], hi = p[1];
            p += step;
            const uint32_t rest = lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w;
            mine |= (uint64_t)(lo.x & 1u) << ii;
            bad |= (uint64_t)((lo.x > 1u) | (rest != 0u)) << ii;
        }
    }
    if (have) {
        bits_group(T, slots, sh, g)[(size_t)(input_slot0 + k) << sh] = mine;
        if (bad) atomicOr((unsigned long long *)&fbmask[g], (unsigned long long)bad);
    }
}

// ---- read-only ceiling ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) read_only(const uint4 *__restrict__ in, size_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const uint4 a = in[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void checksum_kernel(const uint64_t *a, size_t n, unsigned long long *out) {
    unsigned long long s = 0, x = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        s += a[i] * (2 * i + 1);
        x ^= a[i] + i;
    }
    atomicAdd(&out[0], s);
    atomicXor(&out[1], x);
}

struct Ctx {
    uint4 *in; uint64_t *T, *fb; uint64_t slots; uint32_t sh, n_in, batch, groups, nchunk; size_t t_words;
    unsigned long long *cs;
};

static void sums(Ctx &c, unsigned long long out[4]) {
    CK(hipMemset(c.cs, 0, 32));
    hipLaunchKernelGGL(checksum_kernel, dim3(4096), dim3(256), 0, 0, c.T, c.t_words, c.cs);
    hipLaunchKernelGGL(checksum_kernel, dim3(256), dim3(256), 0, 0, c.fb, (size_t)c.groups, c.cs + 2);
    CK(hipMemcpy(out, c.cs, 32, hipMemcpyDeviceToHost));
}

template <class F> static void run(const char *name, Ctx &c, F launch, const unsigned long long *want, unsigned long long *got_out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(c.T, 0, c.t_words * 8)); CK(hipMemset(c.fb, 0, (size_t)c.groups * 8));
    launch();
    CK(hipDeviceSynchronize());
    unsigned long long got[4];
    sums(c, got);
    if (got_out) for (int i = 0; i < 4; i++) got_out[i] = got[i];
    bool same = true;
    if (want) for (int i = 0; i < 4; i++) same &= got[i] == want[i];
    float best = 1e30f, tot = 0;
    const int reps = 6;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; tot += ms;
    }
    const double bytes = (double)c.batch * c.n_in * 32.0;
    printf("%-44s avg %8.3f ms  best %8.3f ms  %7.1f GB/s (best %7.1f)  %s\n", name, tot / reps, best, bytes / (tot / reps) * 1e-6, bytes / best * 1e-6,
           want ? (same ? "same table + flags" : "DIFFERENT") : "reference");
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int lb = argc > 1 ? atoi(argv[1]) : 21;
    Ctx c;
    c.n_in = argc > 2 ? (uint32_t)atoi(argv[2]) : 2048u;
    c.batch = 1u << lb; c.sh = 5; c.slots = c.n_in + 3; c.groups = (c.batch + 63) / 64; c.nchunk = (c.n_in + 63) / 64;
    const size_t n_elems = (size_t)c.batch * c.n_in;
    c.t_words = (size_t)((c.groups + 31) / 32) * c.slots * 32;
    CK(hipMalloc(&c.in, n_elems * 32)); CK(hipMalloc(&c.T, c.t_words * 8)); CK(hipMalloc(&c.fb, (size_t)c.groups * 8)); CK(hipMalloc(&c.cs, 32));
    hipLaunchKernelGGL(fill_kernel, dim3(65536), dim3(256), 0, 0, c.in, n_elems, c.n_in);
    CK(hipDeviceSynchronize());
    printf("batch %u x %u inputs: %.1f GB in, table %.1f MB\n", c.batch, c.n_in, n_elems * 32.0 * 1e-9, c.t_words * 8.0 * 1e-6);
    unsigned long long ref[4];
    const dim3 g2(c.groups, c.nchunk), g1(c.groups * c.nchunk);
#define ARGS c.in, c.T, c.slots, c.sh, 3u, c.n_in, c.batch, c.fb
    run("v0 shipped (grid groups x chunks)", c, [&] { hipLaunchKernelGGL(ingest_v0<0>, g2, dim3(64), 0, 0, ARGS, c.nchunk); }, nullptr, ref);
    run("v0 chunks fastest", c, [&] { hipLaunchKernelGGL(ingest_v0<1>, g1, dim3(64), 0, 0, ARGS, c.nchunk); }, ref, nullptr);
    run("lane flags, unroll 4", c, [&] { hipLaunchKernelGGL((ingest_v1<4, 0, 0>), g2, dim3(64), 0, 0, ARGS, c.nchunk); }, ref, nullptr);
    run("lane flags, unroll 8", c, [&] { hipLaunchKernelGGL((ingest_v1<8, 0, 0>), g2, dim3(64), 0, 0, ARGS, c.nchunk); }, ref, nullptr);
    run("lane flags, unroll 8, chunks fastest", c, [&] { hipLaunchKernelGGL((ingest_v1<8, 1, 0>), g1, dim3(64), 0, 0, ARGS, c.nchunk); }, ref, nullptr);
    run("lane flags, unroll 8, nontemporal", c, [&] { hipLaunchKernelGGL((ingest_v1<8, 0, 1>), g2, dim3(64), 0, 0, ARGS, c.nchunk); }, ref, nullptr);
    run("lane flags, unroll 8, nontemporal, chunks fastest", c, [&] { hipLaunchKernelGGL((ingest_v1<8, 1, 1>), g1, dim3(64), 0, 0, ARGS, c.nchunk); }, ref, nullptr);
    const uint32_t nchunk4 = (c.n_in + 255) / 256;
    run("4 waves per workgroup, unroll 4", c, [&] { hipLaunchKernelGGL(ingest_v4w<4>, dim3(c.groups * nchunk4), dim3(256), 0, 0, ARGS, nchunk4); }, ref, nullptr);
    run("4 waves per workgroup, unroll 8", c, [&] { hipLaunchKernelGGL(ingest_v4w<8>, dim3(c.groups * nchunk4), dim3(256), 0, 0, ARGS, nchunk4); }, ref, nullptr);
    run("rot by hash(g), unroll 8", c, [&] { hipLaunchKernelGGL((ingest_rot<8, 0, 1>), g2, dim3(64), 0, 0, ARGS, c.nchunk); }, ref, nullptr);
    run("rot by hash(g, c), unroll 8", c, [&] { hipLaunchKernelGGL((ingest_rot<8, 0, 2>), g2, dim3(64), 0, 0, ARGS, c.nchunk); }, ref, nullptr);
    run("rot by hash(g), unroll 8, chunks fastest", c, [&] { hipLaunchKernelGGL((ingest_rot<8, 1, 1>), g1, dim3(64), 0, 0, ARGS, c.nchunk); }, ref, nullptr);
    run("rot by hash(g, c), unroll 8, chunks fastest", c, [&] { hipLaunchKernelGGL((ingest_rot<8, 1, 2>), g1, dim3(64), 0, 0, ARGS, c.nchunk); }, ref, nullptr);
    run("rot by g & 63, unroll 8, chunks fastest", c, [&] { hipLaunchKernelGGL((ingest_rot<8, 1, 3>), g1, dim3(64), 0, 0, ARGS, c.nchunk); }, ref, nullptr);
    run("rot by 2 c, unroll 8, chunks fastest", c, [&] { hipLaunchKernelGGL((ingest_rot<8, 1, 4>), g1, dim3(64), 0, 0, ARGS, c.nchunk); }, ref, nullptr);
    if (c.n_in % 128 == 0) {
        run("128 inputs per wave, unroll 8", c, [&] { hipLaunchKernelGGL(ingest_wide<8>, dim3(c.groups, c.n_in / 128), dim3(64), 0, 0, ARGS, c.n_in / 128); }, ref, nullptr);
    }
#define NWRUN(U, NW, ROT, name) { const uint32_t ncw = (c.n_in + 64 * NW - 1) / (64 * NW); \
        run(name, c, [&] { hipLaunchKernelGGL((ingest_nw<U, NW, ROT>), dim3(c.groups * ncw), dim3(64 * NW), 0, 0, ARGS, ncw); }, ref, nullptr); }
    NWRUN(8, 1, 1, "nw: 1 wave, rot hash(g), unroll 8")
    NWRUN(4, 1, 1, "nw: 1 wave, rot hash(g), unroll 4")
    NWRUN(16, 1, 1, "nw: 1 wave, rot hash(g), unroll 16")
    NWRUN(8, 2, 1, "nw: 2 waves, rot hash(g), unroll 8")
    NWRUN(8, 4, 1, "nw: 4 waves, rot hash(g), unroll 8")
    NWRUN(4, 4, 1, "nw: 4 waves, rot hash(g), unroll 4")
    NWRUN(2, 4, 1, "nw: 4 waves, rot hash(g), unroll 2")
    NWRUN(8, 8, 1, "nw: 8 waves, rot hash(g), unroll 8")
    NWRUN(4, 8, 1, "nw: 8 waves, rot hash(g), unroll 4")
    NWRUN(8, 16, 1, "nw: 16 waves, rot hash(g), unroll 8")
    NWRUN(4, 16, 1, "nw: 16 waves, rot hash(g), unroll 4")
    NWRUN(8, 8, 0, "nw: 8 waves, no rotation, unroll 8")
    NWRUN(8, 16, 0, "nw: 16 waves, no rotation, unroll 8")
    run("v0 shipped, again", c, [&] { hipLaunchKernelGGL(ingest_v0<0>, g2, dim3(64), 0, 0, ARGS, c.nchunk); }, ref, nullptr);
    {   // read-only ceiling
        uint32_t *sink; CK(hipMalloc(&sink, 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int blocks : {2048, 8192, 32768}) {
            float tot = 0;
            for (int r = 0; r < 5; r++) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(read_only, dim3(blocks), dim3(256), 0, 0, c.in, n_elems * 2, sink);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r) tot += ms;
            }
            printf("read-only uint4 sum, %5d workgroups x 256          avg %8.3f ms  %7.1f GB/s\n", blocks, tot / 4, n_elems * 32.0 / (tot / 4) * 1e-6);
        }
    }
    return 0;
}
