// ubench_egress.hip - variants of the 32-byte egress (cw_bits.hip::cw_bits_gather_kernel) on the shape of the default line's
// --O1 witness: 156 809 wires x 32 bytes per instance out of the emitted code's bit table (sh = 5), written chunk by chunk into
// two rotating buffers as cw_stream_witnesses_device does.  Every variant must write the bytes of variant 0 (checksums of the
// last two chunks); the last lines are pure writers of the same buffers.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_egress.hip -o gpurun_in/ubench_egress && gpurun_in/ubench_egress [n_wit] [chunk]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ const uint64_t *bits_group(const uint64_t *T, uint64_t slots, uint32_t sh, uint32_t g) {
    return T + (((size_t)(g >> sh) * slots) << sh) + (g & ((1u << sh) - 1u));
}

__global__ void fill_table(uint64_t *T, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t h = (i + 1) * 0x9E3779B97F4A7C15ull; h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29;
        T[i] = h;
    }
}
__global__ void fill_wslot(uint32_t *w, uint32_t n, uint32_t slots) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = (uint32_t)(((uint64_t)(i * 2654435761u) * slots) >> 32);
}

// RUN: consecutive 1 KiB pieces a wave writes per instance; NT: streaming stores; ROT: 0 none, 1 start instance by hash(blockIdx.x),
// 2 by hash(blockIdx.x, wave)
template <int RUN, int NT, int ROT>
__global__ void __launch_bounds__(256)
gather(const uint64_t *__restrict__ T, uint64_t slots, uint32_t lsh, const uint32_t *__restrict__ wslot, uint32_t n_wit,
       uint32_t first, uint32_t count, uint4 *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t k0 = (blockIdx.x * 4 + wv) * 32 * RUN;
    const uint32_t j0 = blockIdx.y * 64;
    if (k0 >= n_wit) return;
    const uint32_t i0 = first + j0, g = i0 >> 6, sh = i0 & 63u;
    const bool two = sh && (uint64_t)(i0 + 64 - sh) < (uint64_t)first + count;
    uint64_t win[RUN];
    bool have[RUN];
#pragma unroll
    for (int r = 0; r < RUN; r++) {
        const uint32_t k = k0 + r * 32 + (lane >> 1);
        have[r] = k < n_wit;
        win[r] = 0;
        if (have[r]) {
            const uint32_t sl = wslot[k];
            win[r] = bits_group(T, slots, lsh, g)[(size_t)sl << lsh] >> sh;
            if (two) win[r] |= bits_group(T, slots, lsh, g + 1)[(size_t)sl << lsh] << (64 - sh);
        }
    }
    const uint32_t nj = min(64u, count - j0);
    const bool low_half = !(lane & 1);
    uint4 *o = out + ((size_t)j0 * n_wit + k0) * 2 + lane;
    const size_t step = (size_t)n_wit * 2;
    const uint32_t r0 = ROT == 1 ? (blockIdx.x * 0x9E3779B1u) >> 26 : ROT == 2 ? ((blockIdx.x * 4 + wv) * 0x9E3779B1u) >> 26 : 0u;
    for (uint32_t t = 0; t < nj; t++) {
        uint32_t jj = t + r0;
        if (ROT) jj = jj >= nj ? jj - nj : jj, jj = jj >= nj ? jj % nj : jj;
#pragma unroll
        for (int r = 0; r < RUN; r++) {
            const uint32_t bit = low_half ? (uint32_t)(win[r] >> jj) & 1u : 0u;
            if (have[r]) {
                const u32x4 v = {bit, 0u, 0u, 0u};
                u32x4 *dst = (u32x4 *)(o + jj * step + r * 64);
                if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
            }
        }
    }
}

// instance groups fastest: blockIdx.x = instance group, blockIdx.y = element block; TPB threads
template <int RUN, int NT, int ROT = 0, int TPB = 256>
__global__ void __launch_bounds__(TPB)
gather_t(const uint64_t *__restrict__ T, uint64_t slots, uint32_t lsh, const uint32_t *__restrict__ wslot, uint32_t n_wit,
         uint32_t first, uint32_t count, uint4 *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t k0 = (blockIdx.y * (TPB / 64) + wv) * 32 * RUN;
    const uint32_t j0 = blockIdx.x * 64;
    if (k0 >= n_wit) return;
    const uint32_t i0 = first + j0, g = i0 >> 6, sh = i0 & 63u;
    const bool two = sh && (uint64_t)(i0 + 64 - sh) < (uint64_t)first + count;
    uint64_t win[RUN];
    bool have[RUN];
#pragma unroll
    for (int r = 0; r < RUN; r++) {
        const uint32_t k = k0 + r * 32 + (lane >> 1);
        have[r] = k < n_wit;
        win[r] = 0;
        if (have[r]) {
            const uint32_t sl = wslot[k];
            win[r] = bits_group(T, slots, lsh, g)[(size_t)sl << lsh] >> sh;
            if (two) win[r] |= bits_group(T, slots, lsh, g + 1)[(size_t)sl << lsh] << (64 - sh);
        }
    }
    const uint32_t nj = min(64u, count - j0);
    const bool low_half = !(lane & 1);
    uint4 *o = out + ((size_t)j0 * n_wit + k0) * 2 + lane;
    const size_t step = (size_t)n_wit * 2;
    const uint32_t r0 = ROT == 1 ? (blockIdx.x * 0x9E3779B1u) >> 26 : ROT == 2 ? (blockIdx.y * 0x9E3779B1u) >> 26 : 0u;
    for (uint32_t t = 0; t < nj; t++) {
        uint32_t jj = t;
        if (ROT) jj = (t + r0) % nj;
#pragma unroll
        for (int r = 0; r < RUN; r++) {
            const uint32_t bit = low_half ? (uint32_t)(win[r] >> jj) & 1u : 0u;
            if (have[r]) {
                const u32x4 v = {bit, 0u, 0u, 0u};
                u32x4 *dst = (u32x4 *)(o + jj * step + r * 64);
                if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
            }
        }
    }
}

template <int NT> __global__ void __launch_bounds__(256) pure_fill(u32x4 *out, size_t n16) {
    const u32x4 v = {1u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}


// pure writers, parameterised: TPB threads, U stores of 16 bytes in flight per thread, PER consecutive 16-byte pieces per thread
template <int TPB, int U, int NT> __global__ void __launch_bounds__(TPB) fill_u(u32x4 *out, size_t n16) {
    const u32x4 v = {1u, 0u, 0u, 0u};
    const size_t stride = (size_t)gridDim.x * TPB;
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(v, out + i + u * stride); else out[i + u * stride] = v; }
    }
    for (; i < n16; i += stride) out[i] = v;
}
// every workgroup owns ONE contiguous span of the buffer (n16 / gridDim pieces) and walks it front to back
template <int TPB, int U> __global__ void __launch_bounds__(TPB) fill_span(u32x4 *out, size_t n16) {
    const u32x4 v = {1u, 0u, 0u, 0u};
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
    size_t i = lo + threadIdx.x;
    for (; i + (U - 1) * TPB < hi; i += U * TPB) {
#pragma unroll
        for (int u = 0; u < U; u++) out[i + u * TPB] = v;
    }
    for (; i < hi; i += TPB) out[i] = v;
}


// one pass, PER consecutive 16-byte pieces per thread
template <int TPB, int PER, int NT> __global__ void __launch_bounds__(TPB) fill_c(u32x4 *out, size_t n16) {
    const u32x4 v = {1u, 0u, 0u, 0u};
    const size_t i = ((size_t)blockIdx.x * TPB + threadIdx.x) * PER;
#pragma unroll
    for (int p = 0; p < PER; p++) if (i + p < n16) { if (NT) __builtin_nontemporal_store(v, out + i + p); else out[i + p] = v; }
}
// one pass, the wave's PER instructions write PER consecutive KiB
template <int TPB, int PER, int NT> __global__ void __launch_bounds__(TPB) fill_w(u32x4 *out, size_t n16) {
    const u32x4 v = {1u, 0u, 0u, 0u};
    const size_t w = ((size_t)blockIdx.x * TPB + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const size_t i = w * 64 * PER + lane;
#pragma unroll
    for (int p = 0; p < PER; p++) if (i + p * 64 < n16) { if (NT) __builtin_nontemporal_store(v, out + i + p * 64); else out[i + p * 64] = v; }
}

// ---- the egress as a sweep in address order: a workgroup writes INST consecutive instances' pieces of 128 elements (4 KiB each) --
template <int INST, int NT>
__global__ void __launch_bounds__(256)
gather_lin(const uint64_t *__restrict__ T, uint64_t slots, uint32_t lsh, const uint32_t *__restrict__ wslot, uint32_t n_wit,
           uint32_t first, uint32_t count, uint4 *__restrict__ out) {
    const uint32_t k = blockIdx.x * 128 + (threadIdx.x >> 1);
    if (k >= n_wit) return;
    const bool low_half = !(threadIdx.x & 1);
    const uint32_t sl = low_half ? wslot[k] : 0u;
#pragma unroll
    for (int q = 0; q < INST; q++) {
        const uint32_t j = blockIdx.y * INST + q;
        if (j >= count) break;
        const uint32_t i = first + j;
        uint32_t bit = 0;
        if (low_half) bit = (uint32_t)(bits_group(T, slots, lsh, i >> 6)[(size_t)sl << lsh] >> (i & 63u)) & 1u;
        const u32x4 v = {bit, 0u, 0u, 0u};
        u32x4 *dst = (u32x4 *)(out + ((size_t)j * n_wit + k) * 2 + (threadIdx.x & 1));
        if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
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

struct Ctx { uint64_t *T; uint32_t *wslot; uint4 *buf[2]; uint64_t slots; uint32_t n_wit, chunk, instances; unsigned long long *cs; size_t buf_bytes; };

template <class F> static void run(const char *name, Ctx &c, uint32_t chunk, F launch, const unsigned long long *want, unsigned long long *got_out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(c.buf[0], 0xEE, c.buf_bytes)); CK(hipMemset(c.buf[1], 0xEE, c.buf_bytes));
    auto sweep = [&] {
        uint32_t n_chunk = 0;
        for (uint32_t done = 0; done < c.instances; done += chunk, n_chunk++) {
            const uint32_t n = chunk < c.instances - done ? chunk : c.instances - done;
            launch(done, n, c.buf[n_chunk & 1]);
        }
    };
    sweep();
    CK(hipDeviceSynchronize());
    unsigned long long got[4];
    CK(hipMemset(c.cs, 0, 32));
    if (chunk == c.chunk) {
        hipLaunchKernelGGL(checksum_kernel, dim3(4096), dim3(256), 0, 0, (const uint64_t *)c.buf[0], c.buf_bytes / 8, c.cs);
        hipLaunchKernelGGL(checksum_kernel, dim3(4096), dim3(256), 0, 0, (const uint64_t *)c.buf[1], c.buf_bytes / 8, c.cs + 2);
    }
    CK(hipMemcpy(got, c.cs, 32, hipMemcpyDeviceToHost));
    if (got_out) for (int i = 0; i < 4; i++) got_out[i] = got[i];
    bool same = true;
    if (want) for (int i = 0; i < 4; i++) same &= got[i] == want[i];
    float tot = 0, best = 1e30f;
    const int reps = 3;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0, 0));
        sweep();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        tot += ms; best = ms < best ? ms : best;
    }
    const double bytes = (double)c.instances * c.n_wit * 32.0;
    printf("%-56s avg %8.3f ms  %7.1f GB/s (best %7.1f)  %s\n", name, tot / reps, bytes / (tot / reps) * 1e-6, bytes / best * 1e-6,
           chunk != c.chunk ? "(other chunking: not compared)" : want ? (same ? "same bytes" : "DIFFERENT") : "reference");
    fflush(stdout);
}

int main(int argc, char **argv) {
    Ctx c;
    c.n_wit = argc > 1 ? (uint32_t)atoi(argv[1]) : 156809u;
    const size_t row = (size_t)c.n_wit * 32;
    c.chunk = argc > 2 ? (uint32_t)atoi(argv[2]) : (uint32_t)((1ull << 30) / row);
    c.instances = 16384; c.slots = 149363;
    const uint32_t big = 2048;                                   // the largest chunk any variant uses
    c.buf_bytes = (size_t)big * row;
    const size_t t_words = (size_t)(c.instances / 2048) * c.slots * 32;
    CK(hipMalloc(&c.T, t_words * 8)); CK(hipMalloc(&c.wslot, (size_t)c.n_wit * 4)); CK(hipMalloc(&c.cs, 32));
    CK(hipMalloc(&c.buf[0], c.buf_bytes)); CK(hipMalloc(&c.buf[1], c.buf_bytes));
    hipLaunchKernelGGL(fill_table, dim3(8192), dim3(256), 0, 0, c.T, t_words);
    hipLaunchKernelGGL(fill_wslot, dim3((c.n_wit + 255) / 256), dim3(256), 0, 0, c.wslot, c.n_wit, (uint32_t)c.slots);
    CK(hipDeviceSynchronize());
    printf("%u instances x %u wires x 32 B = %.1f GB per sweep, chunks of %u instances (%.2f GB)\n", c.instances, c.n_wit,
           c.instances * (double)row * 1e-9, c.chunk, c.chunk * (double)row * 1e-9);
    unsigned long long ref[4];
#define G(KERN, RUN) [&](uint32_t first, uint32_t n, uint4 *out) { \
        hipLaunchKernelGGL(KERN, dim3((c.n_wit + 128 * RUN - 1) / (128 * RUN), (n + 63) / 64), dim3(256), 0, 0, c.T, c.slots, 5u, c.wslot, c.n_wit, first, n, out); }
#define GT(KERN, RUN) GTB(KERN, RUN, 256)
#define GTB(KERN, RUN, TPB) [&](uint32_t first, uint32_t n, uint4 *out) { \
        hipLaunchKernelGGL(KERN, dim3((n + 63) / 64, (c.n_wit + TPB / 2 * RUN - 1) / (TPB / 2 * RUN)), dim3(TPB), 0, 0, c.T, c.slots, 5u, c.wslot, c.n_wit, first, n, out); }
    run("v0 shipped (run 4, streaming stores)", c, c.chunk, G((gather<4, 1, 0>), 4), nullptr, ref);
    run("plain stores", c, c.chunk, G((gather<4, 0, 0>), 4), ref, nullptr);
    run("instance groups fastest", c, c.chunk, GT((gather_t<4, 1>), 4), ref, nullptr);
    run("v0, chunks of 1024", c, 1024, G((gather<4, 1, 0>), 4), ref, nullptr);
    run("instance groups fastest, chunks of 1024", c, 1024, GT((gather_t<4, 1>), 4), ref, nullptr);
    run("instance groups fastest, plain stores", c, c.chunk, GT((gather_t<4, 0>), 4), ref, nullptr);
    run("instance groups fastest, plain stores, chunks of 1024", c, 1024, GT((gather_t<4, 0>), 4), ref, nullptr);
    run("plain stores, chunks of 1024", c, 1024, G((gather<4, 0, 0>), 4), ref, nullptr);
    run("plain stores, run 8", c, c.chunk, G((gather<8, 0, 0>), 8), ref, nullptr);
#define GL(KERN, INST, PAD) [&](uint32_t first, uint32_t n, uint4 *out) { uint32_t gx = (c.n_wit + 127) / 128; if (PAD) gx = (gx + 7) & ~7u; \
        hipLaunchKernelGGL(KERN, dim3(gx, (n + INST - 1) / INST), dim3(256), 0, 0, c.T, c.slots, 5u, c.wslot, c.n_wit, first, n, out); }
    run("address-order sweep, 1 instance, grid.x multiple of 8", c, c.chunk, GL((gather_lin<1, 0>), 1, 1), ref, nullptr);
    run("address-order sweep, 4 instances, x8", c, c.chunk, GL((gather_lin<4, 0>), 4, 1), ref, nullptr);
    run("address-order sweep, 8 instances, x8", c, c.chunk, GL((gather_lin<8, 0>), 8, 1), ref, nullptr);
    run("groups fastest, plain, chunks of 512", c, 512, GT((gather_t<4, 0>), 4), ref, nullptr);
    run("groups fastest, plain, chunks of 2048", c, 2048, GT((gather_t<4, 0>), 4), ref, nullptr);
    run("groups fastest, plain, run 8, chunks of 1024", c, 1024, GT((gather_t<8, 0>), 8), ref, nullptr);
    run("groups fastest, plain, run 2, chunks of 1024", c, 1024, GT((gather_t<2, 0>), 2), ref, nullptr);
    run("groups fastest, plain, run 1, chunks of 1024", c, 1024, GT((gather_t<1, 0>), 1), ref, nullptr);
    run("groups fastest, plain, run 8, chunks of 2048", c, 2048, GT((gather_t<8, 0>), 8), ref, nullptr);
    run("groups fastest, plain, rot hash(group), chunks of 1024", c, 1024, GT((gather_t<4, 0, 1>), 4), ref, nullptr);
    run("groups fastest, plain, rot hash(element block), 1024", c, 1024, GT((gather_t<4, 0, 2>), 4), ref, nullptr);
    run("groups fastest, plain, 512 threads, chunks of 1024", c, 1024, GTB((gather_t<4, 0, 0, 512>), 4, 512), ref, nullptr);
    run("groups fastest, plain, 1024 threads, chunks of 1024", c, 1024, GTB((gather_t<4, 0, 0, 1024>), 4, 1024), ref, nullptr);
    run("groups fastest, plain, 128 threads, chunks of 1024", c, 1024, GTB((gather_t<4, 0, 0, 128>), 4, 128), ref, nullptr);
    run("groups fastest, plain, 64 threads, chunks of 1024", c, 1024, GTB((gather_t<4, 0, 0, 64>), 4, 64), ref, nullptr);
    run("groups fastest, plain, 64 threads, run 8, chunks of 1024", c, 1024, GTB((gather_t<8, 0, 0, 64>), 8, 64), ref, nullptr);
    run("groups fastest, plain, 64 threads, run 16, chunks of 1024", c, 1024, GTB((gather_t<16, 0, 0, 64>), 16, 64), ref, nullptr);
    run("v0 shipped, again", c, c.chunk, G((gather<4, 1, 0>), 4), ref, nullptr);
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const size_t bytes = (size_t)c.chunk * row;
        for (int nt = 0; nt < 2; nt++) for (int blocks : {2048, 16384}) {
            float tot = 0;
            for (int r = 0; r < 4; r++) {
                CK(hipEventRecord(e0, 0));
                for (int k = 0; k < 16; k++) {
                    if (nt) hipLaunchKernelGGL(pure_fill<1>, dim3(blocks), dim3(256), 0, 0, (u32x4 *)c.buf[k & 1], bytes / 16);
                    else hipLaunchKernelGGL(pure_fill<0>, dim3(blocks), dim3(256), 0, 0, (u32x4 *)c.buf[k & 1], bytes / 16);
                }
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r) tot += ms;
            }
            printf("pure 16-byte fill, %s, %5d workgroups, 16 x %.2f GB            %7.1f GB/s\n", nt ? "streaming" : "plain    ", blocks, bytes * 1e-9, 16.0 * bytes / (tot / 3) * 1e-6);
        }

#define FILL(KERN, blocks, tpb, name) { float tot = 0; \
        for (int r = 0; r < 4; r++) { CK(hipEventRecord(e0, 0)); \
            for (int k = 0; k < 16; k++) hipLaunchKernelGGL(KERN, dim3(blocks), dim3(tpb), 0, 0, (u32x4 *)c.buf[k & 1], bytes / 16); \
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) tot += ms; } \
        printf("%-64s %7.1f GB/s\n", name, 16.0 * bytes / (tot / 3) * 1e-6); fflush(stdout); }
        FILL((fill_u<256, 1, 0>), 65536, 256, "fill_u 256 thr, 1 in flight, 65536 wg (one store per thread)")
        FILL((fill_c<256, 1, 0>), (bytes / 16 + 255) / 256, 256, "fill_c 256 thr, 16 B per thread, one pass")
        FILL((fill_c<256, 1, 1>), (bytes / 16 + 255) / 256, 256, "fill_c 256 thr, 16 B per thread, one pass, streaming")
        FILL((fill_c<64, 1, 0>), (bytes / 16 + 63) / 64, 64, "fill_c 64 thr, 16 B per thread, one pass")
        FILL((fill_c<1024, 1, 0>), (bytes / 16 + 1023) / 1024, 1024, "fill_c 1024 thr, 16 B per thread, one pass")
        FILL((fill_c<256, 2, 0>), (bytes / 32 + 255) / 256, 256, "fill_c 256 thr, 32 B contiguous per thread")
        FILL((fill_c<256, 4, 0>), (bytes / 64 + 255) / 256, 256, "fill_c 256 thr, 64 B contiguous per thread")
        FILL((fill_w<256, 2, 0>), (bytes / 32 + 255) / 256, 256, "fill_w 256 thr, 2 KiB per wave")
        FILL((fill_w<256, 4, 0>), (bytes / 64 + 255) / 256, 256, "fill_w 256 thr, 4 KiB per wave")
        FILL((fill_w<256, 8, 0>), (bytes / 128 + 255) / 256, 256, "fill_w 256 thr, 8 KiB per wave")
        FILL((fill_w<256, 8, 1>), (bytes / 128 + 255) / 256, 256, "fill_w 256 thr, 8 KiB per wave, streaming")
        FILL((fill_w<1024, 4, 0>), (bytes / 64 + 1023) / 1024, 1024, "fill_w 1024 thr, 4 KiB per wave")
        FILL((fill_u<256, 4, 0>), 16384, 256, "fill_u 256 thr, 4 in flight, 16384 wg")
        FILL((fill_u<256, 4, 0>), 4096, 256, "fill_u 256 thr, 4 in flight, 4096 wg")
        FILL((fill_u<256, 8, 0>), 2048, 256, "fill_u 256 thr, 8 in flight, 2048 wg")
        FILL((fill_u<256, 8, 0>), 8192, 256, "fill_u 256 thr, 8 in flight, 8192 wg")
        FILL((fill_u<1024, 4, 0>), 512, 1024, "fill_u 1024 thr, 4 in flight, 512 wg")
        FILL((fill_u<1024, 4, 0>), 2048, 1024, "fill_u 1024 thr, 4 in flight, 2048 wg")
        FILL((fill_u<1024, 4, 1>), 2048, 1024, "fill_u 1024 thr, 4 in flight, 2048 wg, streaming")
        FILL((fill_u<512, 4, 0>), 1024, 512, "fill_u 512 thr, 4 in flight, 1024 wg")
        FILL((fill_span<256, 4>), 1024, 256, "fill_span 256 thr, 4 in flight, 1024 wg")
        FILL((fill_span<256, 4>), 4096, 256, "fill_span 256 thr, 4 in flight, 4096 wg")
        FILL((fill_span<256, 4>), 16384, 256, "fill_span 256 thr, 4 in flight, 16384 wg")
        FILL((fill_span<1024, 4>), 512, 1024, "fill_span 1024 thr, 4 in flight, 512 wg")
        FILL((fill_span<1024, 4>), 2048, 1024, "fill_span 1024 thr, 4 in flight, 2048 wg")
        FILL((fill_span<64, 8>), 16384, 64, "fill_span 64 thr, 8 in flight, 16384 wg")
        FILL((fill_span<64, 8>), 65536, 64, "fill_span 64 thr, 8 in flight, 65536 wg")
        float tot = 0;
        for (int r = 0; r < 4; r++) {
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < 16; k++) CK(hipMemsetAsync(c.buf[k & 1], 1, bytes, 0));
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) tot += ms;
        }
        printf("hipMemsetAsync, 16 x %.2f GB                                        %7.1f GB/s\n", bytes * 1e-9, 16.0 * bytes / (tot / 3) * 1e-6);
    }
    return 0;
}
