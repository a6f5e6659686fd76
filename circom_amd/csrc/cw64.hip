// cw64.hip — the 64-bit runtime on the device: circuits compiled for `--prime goldilocks` (q = 2^64 - 2^32 + 1).
//
// Reference counterpart: code_producers/src/c_elements/goldilocks/fr.hpp (a field element is a plain uint64, no Montgomery
// form, no tagged representations) with common64/{main,calcwit}.cpp and the value-style emitted code (compute_bucket.rs:353,
// store_bucket.rs:575-657; constants are literals: value_bucket.rs:82-86).  The 256-bit engine (cw_kernels.hip) is built on
// "q is large" (short products without reduction, lazy integer sums, single-limb masks): none of that holds for a 64-bit
// prime, and none of its machinery is needed - a value is ONE register.  So this file is its own small engine:
//
//   value table   V[slot][instance] : uint64, canonical residues, slot 0 = the constant 1, signals, then temporaries
//                 (one coalesced 512-byte access per wave and operand)
//   program       the flat witness code, one 32-byte row per operation (hip_elements/lower64.py), wave-uniform
//   one lane = one instance; operators follow the reference's 64-bit library exactly as oracle/field.py restates it for
//   any q (tests/golden/reference_wtns_goldilocks.json: vectors from the reference's own 64-bit runtime, incl. the operator zoo)
#include <hip/hip_runtime.h>
#include "cw_kernels.h"

#define GL_P 0xFFFFFFFF00000001ull
#define GL_HALF (GL_P >> 1)

// ---- field arithmetic ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t gl_add(uint64_t a, uint64_t b) {
    const uint64_t s = a + b;
    const bool wrap = s < a;                       // a + b >= 2^64: subtract p = add 2^32 - 1
    uint64_t r = wrap ? s + 0xFFFFFFFFull : s;     // (a, b < p, so the corrected value is < p)
    return r >= GL_P ? r - GL_P : r;
}
__device__ __forceinline__ uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GL_P - b); }
__device__ __forceinline__ uint64_t gl_neg(uint64_t a) { return a ? GL_P - a : 0; }
// x = hi 2^64 + lo with 2^64 = 2^32 - 1, 2^96 = -1 (mod p): lo - hi_hi + hi_lo (2^32 - 1)
__device__ __forceinline__ uint64_t gl_reduce128(uint64_t hi, uint64_t lo) {
    const uint64_t hh = hi >> 32, hl = hi & 0xFFFFFFFFull;
    uint64_t t = lo - hh;
    if (lo < hh) t -= 0xFFFFFFFFull;               // borrowed 2^64 = p + 2^32 - 1: take the 2^32 - 1 back (t stays a residue mod p)
    const uint64_t m = hl * 0xFFFFFFFFull;         // < 2^64
    uint64_t r = t + m;
    if (r < t) r += 0xFFFFFFFFull;                 // carried 2^64
    return r >= GL_P ? r - GL_P : r;
}
__device__ __forceinline__ uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_reduce128(__umul64hi(a, b), a * b); }
__device__ __forceinline__ uint64_t gl_pow(uint64_t x, uint64_t e) {
    uint64_t r = 1;
    for (int i = 63; i >= 0; i--) {
        r = gl_mul(r, r);
        if ((e >> i) & 1ull) r = gl_mul(r, x);
    }
    return r;
}
__device__ __forceinline__ uint64_t gl_inv(uint64_t x) { return gl_pow(x, GL_P - 2); }            // Fr_inv: 0 -> 0
__device__ __forceinline__ uint64_t gl_wrap(uint64_t v) { return v >= GL_P ? v - GL_P : v; }     // lboMask is all ones: one subtraction
__device__ __forceinline__ bool gl_lt(uint64_t x, uint64_t y) {                                    // val(x) < val(y), val = x - p iff x > half
    const bool nx = x > GL_HALF, ny = y > GL_HALF;
    return nx == ny ? x < y : nx;
}
__device__ __forceinline__ uint64_t gl_shl(uint64_t x, uint64_t y) {
    if (y < 64) return gl_wrap(x << y);
    const uint64_t k = GL_P - y;                   // a "negative" amount shifts the other way
    return k >= 64 ? 0 : x >> k;
}
__device__ __forceinline__ uint64_t gl_shr(uint64_t x, uint64_t y) {
    if (y < 64) return x >> y;
    const uint64_t k = GL_P - y;
    return k >= 64 ? 0 : gl_wrap(x << k);
}

// ---- kernels ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cw64_init_kernel(uint64_t *V, uint32_t Bp, uint32_t *status, uint32_t *first_bad) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < Bp) {
        V[i] = 1;                                  // slot 0
        status[i] = 0;
        first_bad[i] = 0xFFFFFFFFu;
    }
}
// inputs arrive as the boundary's 32-byte little-endian values [batch][n_in][32]; a value that is not a canonical residue
// (upper words set, or >= p) is reduced, as Fr_str2element does for what loadJson reads
__global__ void __launch_bounds__(256) cw64_ingest_kernel(const uint64_t *__restrict__ in, uint64_t *V, uint32_t input_start, uint32_t n_in,
                                                          uint32_t batch, uint32_t Bp) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    if (i >= batch) return;
    const uint64_t *p = in + ((size_t)i * n_in + k) * 4;
    uint64_t r = p[0] >= GL_P ? p[0] - GL_P : p[0];
    // 2^64 = 2^32 - 1, 2^128 = (2^32 - 1)^2, 2^192 = (2^32 - 1)^3 (mod p)
    const uint64_t e = 0xFFFFFFFFull, e2 = gl_mul(e, e), e3 = gl_mul(e2, e);
    if (p[1] | p[2] | p[3]) {
        r = gl_add(r, gl_mul(p[1] >= GL_P ? p[1] - GL_P : p[1], e));
        r = gl_add(r, gl_mul(p[2] >= GL_P ? p[2] - GL_P : p[2], e2));
        r = gl_add(r, gl_mul(p[3] >= GL_P ? p[3] - GL_P : p[3], e3));
    }
    V[(size_t)(input_start + k) * Bp + i] = r;
}

// row = 8 x u32: w0 = op | dk << 8 | ak << 10 | bk << 12 | ck << 14 (kind 0 = table slot, 2 = constant, 3 = none); dst; a; b; c;
// index of the flat operation (failure reports); 0; 0
enum : uint32_t { O_COPY = 0, O_ADD, O_SUB, O_MUL, O_DIV, O_IDIV, O_MOD, O_POW, O_NEG, O_SHL, O_SHR, O_BAND, O_BOR, O_BXOR, O_BNOT,
                  O_LT, O_GT, O_LEQ, O_GEQ, O_EQ, O_NEQ, O_LAND, O_LOR, O_LNOT, O_SELECT, O_ASSERT_EQ, O_ASSERT_NZ };
// (Round 6 tried requesting the operands of row r + 1 before row r computes, with register forwarding of a value the next row
// reads: 1.15 ms instead of 0.96 ms for Poseidon(2) x 65 536 - the loads of one row already overlap across the waves of a SIMD,
// the extra bookkeeping does not pay; profiles/r06i_bench_poseidon2_goldilocks.json has that run.  The check below is what
// moved the line: 38 -> 59 M witnesses/s.)
__global__ void __launch_bounds__(64) cw64_eval_kernel(const uint4 *__restrict__ rows, uint32_t n_rows, const uint64_t *__restrict__ consts,
                                                       uint64_t *V, uint32_t Bp, uint32_t batch, uint32_t *status) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= batch) return;
    uint64_t *Vi = V + i;
    uint32_t st = 0;
    for (uint32_t r = 0; r < n_rows; r++) {
        const uint4 x = rows[2 * r], y = rows[2 * r + 1];               // wave-uniform: scalar loads
        const uint32_t op = x.x & 0xFF, ak = (x.x >> 10) & 3, bk = (x.x >> 12) & 3, ck = (x.x >> 14) & 3;
        const uint64_t a = ak == 0 ? Vi[(size_t)x.z * Bp] : ak == 2 ? consts[x.z] : 0;
        const uint64_t b = bk == 0 ? Vi[(size_t)x.w * Bp] : bk == 2 ? consts[x.w] : 0;
        uint64_t d = 0;
        bool fail = false;
        switch (op) {
        case O_COPY: d = a; break;
        case O_ADD: d = gl_add(a, b); break;
        case O_SUB: d = gl_sub(a, b); break;
        case O_MUL: d = gl_mul(a, b); break;
        case O_DIV: d = gl_mul(a, gl_inv(b)); break;
        case O_IDIV: if (b == 0) fail = true; else d = a / b; break;
        case O_MOD: if (b == 0) fail = true; else d = a % b; break;
        case O_POW: d = gl_pow(a, b); break;
        case O_NEG: d = gl_neg(a); break;
        case O_SHL: d = gl_shl(a, b); break;
        case O_SHR: d = gl_shr(a, b); break;
        case O_BAND: d = gl_wrap(a & b); break;
        case O_BOR: d = gl_wrap(a | b); break;
        case O_BXOR: d = gl_wrap(a ^ b); break;
        case O_BNOT: d = gl_wrap(~a); break;
        case O_LT: d = gl_lt(a, b); break;
        case O_GT: d = gl_lt(b, a); break;
        case O_LEQ: d = !gl_lt(b, a); break;
        case O_GEQ: d = !gl_lt(a, b); break;
        case O_EQ: d = a == b; break;
        case O_NEQ: d = a != b; break;
        case O_LAND: d = (a != 0) & (b != 0); break;
        case O_LOR: d = (a != 0) | (b != 0); break;
        case O_LNOT: d = a == 0; break;
        case O_SELECT: {
            const uint64_t cc = ck == 0 ? Vi[(size_t)y.x * Bp] : ck == 2 ? consts[y.x] : 0;
            d = a != 0 ? b : cc;
            break;
        }
        case O_ASSERT_EQ: fail = a != b; break;
        case O_ASSERT_NZ: fail = a == 0; break;
        default: break;
        }
        if (fail && !(st & 3u))                                      // the first failing check in program order
            st = (op == O_IDIV || op == O_MOD ? CW_ST_ARITH : CW_ST_ASSERT_FAILED) | (y.y << 8);
        if (((x.x >> 8) & 3) == 0 && op != O_ASSERT_EQ && op != O_ASSERT_NZ) Vi[(size_t)x.y * Bp] = d;
    }
    if (st) atomicOr(&status[i], st);
}

// R1CS: constraint k = three runs of (slot, coefficient) terms; A.w * B.w == C.w.  term = {slot, part | last of the constraint
// << 2, coefficient lo, hi}.  A workgroup checks ONE CHUNK of consecutive constraints (chunk = {first term, terms, first row, 0},
// cut at constraint boundaries by the host) for 64 instances: the launch is (groups x chunks) workgroups instead of one
// wave per 64 instances walking the whole system (Poseidon(2): 3 037 dependent load + multiply steps per wave, one wave per
// SIMD: 1.69 ms; chunked: the chip is full and the loads of different chunks overlap).
__global__ void __launch_bounds__(64) cw64_r1cs_kernel(const uint4 *__restrict__ chunks, uint32_t n_chunks, const uint4 *__restrict__ terms,
                                                       const uint64_t *__restrict__ V, uint32_t Bp, uint32_t batch, uint32_t *status,
                                                       uint32_t *first_bad) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= batch) return;
    const uint64_t *Vi = V + i;
    uint32_t bad = 0xFFFFFFFFu;
    for (uint32_t cix = blockIdx.y; cix < n_chunks; cix += gridDim.y) {
        const uint4 ch = chunks[cix];
        uint64_t acc[3] = {0, 0, 0};
        uint32_t row = ch.z;
        uint4 x = terms[ch.x];
        uint64_t w = Vi[(size_t)x.x * Bp];
        for (uint32_t t = 0; t < ch.y; t++) {
            const uint4 nx = terms[ch.x + (t + 1 < ch.y ? t + 1 : t)];      // the next term's wire is in flight during the product
            const uint64_t nw = Vi[(size_t)nx.x * Bp];
            const uint64_t pr = gl_mul(w, ((uint64_t)x.w << 32) | x.z);
            const uint32_t part = x.y & 3u;
            acc[0] = part == 0 ? gl_add(acc[0], pr) : acc[0];
            acc[1] = part == 1 ? gl_add(acc[1], pr) : acc[1];
            acc[2] = part == 2 ? gl_add(acc[2], pr) : acc[2];
            if (x.y & 4u) {
                if (gl_mul(acc[0], acc[1]) != acc[2] && row < bad) bad = row;
                row++;
                acc[0] = acc[1] = acc[2] = 0;
            }
            x = nx;
            w = nw;
        }
    }
    if (bad != 0xFFFFFFFFu) {
        atomicMin(&first_bad[i], bad);
        atomicOr(&status[i], CW_ST_R1CS_FAILED);
    }
}

// one instance's witness as 32-byte little-endian values (the boundary's element format; the files are written with n8 = 8)
__global__ void __launch_bounds__(256) cw64_gather_kernel(const uint64_t *__restrict__ V, const uint32_t *__restrict__ w2s, uint32_t n_wit,
                                                          uint32_t Bp, uint32_t first, uint32_t count, uint64_t *out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (k >= n_wit || j >= count) return;
    uint64_t *o = out + ((size_t)j * n_wit + k) * 4;
    o[0] = V[(size_t)w2s[k] * Bp + first + j];
    o[1] = o[2] = o[3] = 0;
}

// ---- launch wrappers -----------------------------------------------------------------------------------------------------
hipError_t cwk64_init(hipStream_t s, void *V, uint32_t Bp, uint32_t *status, uint32_t *first_bad) {
    hipLaunchKernelGGL(cw64_init_kernel, dim3((Bp + 255) / 256), dim3(256), 0, s, (uint64_t *)V, Bp, status, first_bad);
    return hipGetLastError();
}
hipError_t cwk64_ingest(hipStream_t s, const void *in, void *V, uint32_t input_start, uint32_t n_in, uint32_t batch, uint32_t Bp) {
    if (!n_in) return hipSuccess;
    if (n_in > 65535u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cw64_ingest_kernel, dim3((batch + 255) / 256, n_in), dim3(256), 0, s, (const uint64_t *)in, (uint64_t *)V, input_start,
                       n_in, batch, Bp);
    return hipGetLastError();
}
hipError_t cwk64_eval(hipStream_t s, const void *rows, uint32_t n_rows, const void *consts, void *V, uint32_t Bp, uint32_t batch,
                      uint32_t *status) {
    hipLaunchKernelGGL(cw64_eval_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, (const uint4 *)rows, n_rows, (const uint64_t *)consts,
                       (uint64_t *)V, Bp, batch, status);
    return hipGetLastError();
}
hipError_t cwk64_r1cs(hipStream_t s, const void *chunks, uint32_t n_chunks, const void *terms, const void *V, uint32_t Bp, uint32_t batch,
                      uint32_t *status, uint32_t *first_bad) {
    if (!n_chunks) return hipSuccess;
    hipLaunchKernelGGL(cw64_r1cs_kernel, dim3((batch + 63) / 64, n_chunks < 65535u ? n_chunks : 65535u), dim3(64), 0, s, (const uint4 *)chunks,
                       n_chunks, (const uint4 *)terms, (const uint64_t *)V, Bp, batch, status, first_bad);
    return hipGetLastError();
}
hipError_t cwk64_gather(hipStream_t s, const void *V, const uint32_t *w2s, uint32_t n_wit, uint32_t Bp, uint32_t first, uint32_t count,
                        void *out) {
    if (!n_wit || !count) return hipSuccess;
    for (uint32_t done = 0; done < count; done += 65535u) {
        const uint32_t n = count - done < 65535u ? count - done : 65535u;
        hipLaunchKernelGGL(cw64_gather_kernel, dim3((n_wit + 255) / 256, n), dim3(256), 0, s, (const uint64_t *)V, w2s, n_wit, Bp, first + done,
                           n, (uint64_t *)out + (size_t)done * n_wit * 4);
    }
    return hipGetLastError();
}
