// cw_rowops.hip.h - row-level device helpers shared by the interpreting kernels (cw_kernels.hip) and the row bodies of the
// emitted 256-bit code (hip_elements/fpjit.py compiles them once per ABI variant and calls them from straight-line code).
#pragma once
#include "cw_tape.h"
#include "fp256.hip.h"

// status word of an instance: failure bits | index of the flat operation << 8; the smallest index wins (the check the
// reference's sequential program would have stopped at: assert_bucket.rs:75-77, calcwit.cpp:104-114)
__device__ __forceinline__ void cw_fail(uint32_t &st, uint32_t bits, uint32_t idx) {
    if (st == 0 || idx < (st >> 8)) st = bits | (idx << 8);
}

__device__ __forceinline__ void cw_publish_status(uint32_t *status, uint32_t i, uint32_t st) {
    uint32_t old = status[i];
    while (old == 0 || (st >> 8) < (old >> 8)) {
        const uint32_t prev = atomicCAS(&status[i], old, st);
        if (prev == old) break;
        old = prev;
    }
}

// ---- D_LINSUM accumulators: two unsigned 192-bit sums (positive / negative terms), no modular reduction ----------------
struct Acc192 { uint64_t w0, w1, w2; };
__device__ __forceinline__ void acc192_add(Acc192 &a, uint64_t lo, uint64_t hi) {
    const uint64_t s0 = a.w0 + lo;
    const uint64_t c0 = s0 < lo;
    const uint64_t s1 = a.w1 + hi;
    const uint64_t c1 = s1 < hi;
    const uint64_t s1b = s1 + c0;
    const uint64_t c1b = s1b < c0;
    a.w0 = s0;
    a.w1 = s1b;
    a.w2 += c1 + c1b;
}
__device__ __forceinline__ fe acc192_to_fe(const Acc192 &a) {
    fe r = fe_zero();
    r.v[0] = (uint32_t)a.w0; r.v[1] = (uint32_t)(a.w0 >> 32);
    r.v[2] = (uint32_t)a.w1; r.v[3] = (uint32_t)(a.w1 >> 32);
    r.v[4] = (uint32_t)a.w2; r.v[5] = (uint32_t)(a.w2 >> 32);
    return r;
}
__device__ __forceinline__ void linsum_term(const fe &xc, uint64_t cf, fe &g, Acc192 &pos, Acc192 &neg, const FpParams &P) {
    const uint64_t mag = cf & 0x7FFFFFFFFFFFFFFFull;
    const bool cneg = cf >> 63;
    if (__all(fe_hi_or(xc) == 0)) {
        // 64x64 -> 128-bit product, accumulated without reduction
        const uint32_t a0 = xc.v[0], a1 = xc.v[1], b0 = (uint32_t)mag, b1 = (uint32_t)(mag >> 32);
        uint64_t t = (uint64_t)a0 * b0;
        const uint32_t r0 = (uint32_t)t;
        t = (uint64_t)a0 * b1 + (t >> 32);
        const uint64_t t2 = (uint64_t)a1 * b0 + (uint32_t)t;
        const uint64_t hi = (uint64_t)a1 * b1 + (t >> 32) + (t2 >> 32);
        const uint64_t lo = ((uint64_t)(uint32_t)t2 << 32) | r0;
        if (cneg) acc192_add(neg, lo, hi);
        else acc192_add(pos, lo, hi);
    } else if (mag) {
        // generic: coef * x in the field (coef as a canonical element)
        fe cm = fe_zero();
        cm.v[0] = (uint32_t)mag;
        cm.v[1] = (uint32_t)(mag >> 32);
        const fe p = fe_mul2_auto(xc, cm, P);
        g = cneg ? fe_sub(g, p, P) : fe_add(g, p, P);
    }
}
