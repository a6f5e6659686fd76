// cw_call.hip.h - tier 2 on the device: the per-lane interpreter of circom functions with run-time control flow (D_CALL) and
// the native routines of the big-integer witness hints.  Shared by the interpreting kernels (cw_kernels.hip) and the `call`
// body of the emitted 256-bit code (hip_elements/fpjit_bodies.py).
#pragma once
#include "cw_rowops.hip.h"

#ifndef CW_CALL_NATIVE
#define CW_CALL_NATIVE __noinline__      // (the emitted code's body is a leaf: it inlines the native routines)
#endif

__device__ __forceinline__ fe c_load(const uint32_t *__restrict__ consts, uint32_t idx) {
    return fe_from(consts + (size_t)idx * 8);
}

struct EvalCtx {                         // per-wave constants of the interpreter
    const char *Vb;                      // value table, bytes
    const char *Cb;                      // constant table, bytes
    const uint32_t *Lb;                  // limb-form constant table (D_DOTC), 12 words per entry
    const uint64_t *terms;               // D_LINSUM / D_DOTC term table of this strand ...
    uint32_t tp;                         // ... and the running position in it
    uint32_t vlo, vhi;                   // this lane's byte offsets of the lo/hi half inside a value slot
    uint32_t lane16;                     // lane * 16 (LDS)
    uint32_t lds_hi;                     // bytes from the lo half of an LDS slot to its hi half (16 x the lanes a slot holds)
    const uint4 *fcode;                  // bytecode of circom functions (D_CALL), all functions concatenated
    const uint4 *ftab;                   // per function {first instruction, n instructions, n registers, -}
    uint64_t slot_stride;                // bytes between consecutive value slots (2 * Bp * 16)
#ifdef CW_PROFILE
    uint32_t prof_level;                 // barriers this strand has passed (profiling build)
#endif
};

// ---- D_CALL: a circom function with run-time control flow, interpreted per lane ---------------------------------------
// Reference: the emitted C++ of a function is real control flow on Fr_isTrue / Fr_toInt (loop_bucket.rs:76-91,
// branch_bucket.rs:100-122, compute_bucket.rs:361-363, call_bucket.rs:466-533); the trip counts differ per input, so the
// trace cannot unroll it.  Every lane has its own program counter; per turn the wave executes the instruction of its
// unfinished lane with the LOWEST program counter for all lanes that sit on it (SIMT divergence, lanes elsewhere wait).  Registers are
// 256-bit values in the lane's column of consecutive temp slots (the call's window): operand loads and result stores
// go to the value table like any spilled temporary — this is the slow path by design (tier 2).
__device__ __forceinline__ fe fn_operand(uint32_t x, const char *regs, const EvalCtx &c) {
    if (x & FN_CONST) return c_load((const uint32_t *)c.Cb, x & 0x7FFFFFFFu);
    const char *p = regs + (uint64_t)x * c.slot_stride;
    const uint4 lo = *(const uint4 *)(p + c.vlo), hi = *(const uint4 *)(p + c.vhi);
    fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void fn_store(uint32_t r, char *regs, const EvalCtx &c, const fe &x) {
    char *p = regs + (uint64_t)r * c.slot_stride;
    *(uint4 *)(p + c.vlo) = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    *(uint4 *)(p + c.vhi) = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}
// Fr_toInt (generic/fr.cpp:1146-1170) restricted to what an array address can be: [0, n) or "bad"
__device__ __forceinline__ bool fn_index(const fe &v, uint32_t n, uint32_t *out) {
    uint32_t hi = 0;
    FE_UNROLL for (int k = 1; k < 8; k++) hi |= v.v[k];
    *out = v.v[0];
    return hi == 0 && v.v[0] < n;
}
// ---- native big-integer functions ---------------------------------------------------------------------------------------
// circom-ecdsa's witness hints (`mod_inv` = mod_exp(a, p - 2), `secp256k1_addunequal_func`, `secp256k1_double_func`) are pure
// functions of their arguments: k limbs of n bits per number, arithmetic modulo a FOREIGN prime (secp256k1's p or group
// order inside a BLS12-381 circuit).  Interpreting their bytecode costs ~10^6 instructions per modular inverse; the same
// values come from this file's own field code instantiated for the foreign prime - binary-GCD inverse, canonical products -
// whose parameters the loader appended to the function table.  tests/test_ecdsa.py: device native == device bytecode ==
// oracle == reference runtime.  ftab[fn].w = kind | k << 4 | n << 8 | (uint4 offset of the FpParams) << 16.
__device__ __forceinline__ fe big_pack(const char *regs, uint32_t first, uint32_t k, uint32_t n, const EvalCtx &c) {
    fe r = fe_zero();
    for (uint32_t i = 0; i < k; i++) {
        const fe l = fn_operand(first + i, regs, c);
        const uint64_t v = ((uint64_t)l.v[1] << 32) | l.v[0];
        const uint32_t sh = n * i, w = sh >> 5, bs = sh & 31u;                  // wave-uniform
        const uint64_t lo = v << bs;
        const uint32_t hi = bs ? (uint32_t)(v >> (64 - bs)) : 0u;
        FE_UNROLL for (int j = 0; j < 8; j++)
            r.v[j] |= ((uint32_t)j == w ? (uint32_t)lo : 0u) | ((uint32_t)j == w + 1 ? (uint32_t)(lo >> 32) : 0u) | ((uint32_t)j == w + 2 ? hi : 0u);
    }
    return r;
}
__device__ __forceinline__ void big_unpack(char *regs, uint32_t first, uint32_t k, uint32_t n, const EvalCtx &c, const fe &x) {
    for (uint32_t i = 0; i < k; i++) {
        fe l = fe_shr_raw(x, n * i);
        const uint64_t m = n >= 64 ? ~0ull : ((1ull << n) - 1);
        l.v[0] &= (uint32_t)m;
        l.v[1] &= (uint32_t)(m >> 32);
        FE_UNROLL for (int j = 2; j < 8; j++) l.v[j] = 0;
        fn_store(first + i, regs, c, l);
    }
}
// long_div (kind 4): a[k + m], b[k] -> div[m + 1] ++ mod[k] with a = div * b + mod, 0 <= mod < b (bigint_func.circom long_div:
// Knuth D in base 2^n on registers, ~10^5 interpreted instructions per call).  Contract of the tag (circuits/bigint_func.py):
// proper limbs (< 2^n), b[k-1] != 0 - then the quotient has m + 1 limbs and the pair is unique, whatever algorithm finds it.
// Here: Knuth D in base 2^32 on the packed numbers - the divisor normalised per lane to 256 bits (as fe_divmod), the dividend
// (up to CW_LD_WORDS words) shifted along; ftab[fn].w = 4 | k << 4 | n << 8 | m << 16.  A lane whose divisor breaks the
// contract is flagged (CW_ST_ARITH): templates only tag calls whose divisor is a compile-time constant.
#define CW_LD_WORDS 20
__device__ CW_CALL_NATIVE void eval_call_long_div(uint32_t w, char *regs, uint32_t row_id, uint32_t &st, const EvalCtx &c) {
    const uint32_t k = (w >> 4) & 15u, n = (w >> 8) & 255u, m = w >> 16;
    constexpr int NW = CW_LD_WORDS, NX = CW_LD_WORDS + 8;
    uint32_t X[NX];
    FE_UNROLL for (int j = 0; j < NX; j++) X[j] = 0;
    const fe B0 = big_pack(regs, k + m, k, n, c);
    {   // the top limb of the divisor (contract) - and with it B0 != 0
        const fe top = fn_operand(2 * k + m - 1, regs, c);
        uint32_t any = 0;
        FE_UNROLL for (int j = 0; j < 8; j++) any |= top.v[j];
        if (any == 0) cw_fail(st, CW_ST_ARITH, row_id);
    }
    fe B = B0;
    if (fe_is_zero(B)) B.v[0] = 1;
    // dividend: limb i at bit n * i (wave-uniform positions)
    if (n == 64) {
        FE_UNROLL for (int i = 0; i < NW / 2; i++)
            if ((uint32_t)i < k + m) {
                const fe l = fn_operand(i, regs, c);
                X[2 * i] = l.v[0];
                X[2 * i + 1] = l.v[1];
            }
    } else if (n == 32) {
        FE_UNROLL for (int i = 0; i < NW; i++)
            if ((uint32_t)i < k + m) X[i] = fn_operand(i, regs, c).v[0];
    } else {
        for (uint32_t i = 0; i < k + m; i++) {
            const fe l = fn_operand(i, regs, c);
            const uint64_t v = ((uint64_t)l.v[1] << 32) | l.v[0];
            const uint32_t sh = n * i, w0 = sh >> 5, bs = sh & 31u;                 // wave-uniform
            const uint64_t lo = v << bs;
            const uint32_t hi = bs ? (uint32_t)(v >> (64 - bs)) : 0u;
            FE_UNROLL for (int j = 0; j < NW; j++)
                X[j] |= ((uint32_t)j == w0 ? (uint32_t)lo : 0u) | ((uint32_t)j == w0 + 1 ? (uint32_t)(lo >> 32) : 0u) | ((uint32_t)j == w0 + 2 ? hi : 0u);
        }
    }
    // normalise: V = B << s has its top bit set; the dividend moves by the same (per-lane) amount
    const uint32_t s = fe_clz256(B) & 255u;
    const fe V = fe_shl_raw(B, s);
    {
        const bool s4 = s & 128, s2 = s & 64, s1 = s & 32;
        FE_UNROLL for (int j = NX - 1; j >= 0; j--) X[j] = s4 ? (j >= 4 ? X[j >= 4 ? j - 4 : 0] : 0u) : X[j];
        FE_UNROLL for (int j = NX - 1; j >= 0; j--) X[j] = s2 ? (j >= 2 ? X[j >= 2 ? j - 2 : 0] : 0u) : X[j];
        FE_UNROLL for (int j = NX - 1; j >= 0; j--) X[j] = s1 ? (j >= 1 ? X[j >= 1 ? j - 1 : 0] : 0u) : X[j];
        const uint32_t bs = s & 31u, rs = (32u - bs) & 31u;
        FE_UNROLL for (int j = NX - 1; j >= 1; j--) X[j] = bs ? __builtin_amdgcn_alignbit(X[j], X[j - 1], rs) : X[j];
        X[0] <<= bs;
    }
    // the quotient is below 2^(32 NW): the eight top words of the shifted dividend are the first partial remainder (< V)
    uint32_t r[9];
    FE_UNROLL for (int j = 0; j < 8; j++) r[j] = X[NW + j];
    r[8] = 0;
    const uint32_t v7 = V.v[7], v6 = V.v[6];
    // One digit per trip, as a LOOP (unrolled it is NW copies of ~300 instructions): the dividend leaves X at the top while the
    // quotient enters at the bottom, so every trip works on fixed registers and X[0 .. NW) ends as the quotient.
#pragma clang loop unroll(disable)
    for (int it = 0; it < NW; it++) {
        FE_UNROLL for (int t = 8; t >= 1; t--) r[t] = r[t - 1];
        r[0] = X[NW - 1];
        FE_UNROLL for (int t = NW - 1; t >= 1; t--) X[t] = X[t - 1];
        X[0] = 0;                                                     // this trip's quotient digit
        bool lt = false, decided = r[8] != 0;
        FE_UNROLL for (int t = 7; t >= 0; t--) {
            if (!decided && r[t] != V.v[t]) { lt = r[t] < V.v[t]; decided = true; }
        }
        if (!__any(!lt)) continue;                                    // every lane's digit is zero
        const uint64_t nn = ((uint64_t)r[8] << 32) | r[7];
        uint64_t qh = nn / v7;
        if (qh > 0xFFFFFFFFull) qh = 0xFFFFFFFFull;
        uint64_t rh = nn - qh * v7;
        FE_UNROLL for (int t = 0; t < 2; t++) {
            const bool dec = rh <= 0xFFFFFFFFull && qh * (uint64_t)v6 > ((rh << 32) | r[6]);
            qh -= dec ? 1u : 0u;
            rh += dec ? v7 : 0u;
        }
        const uint32_t qd = (uint32_t)qh;
        uint64_t carry = 0;
        int64_t br = 0;
        FE_UNROLL for (int t = 0; t < 8; t++) {
            const uint64_t p = (uint64_t)qd * V.v[t] + carry;
            carry = p >> 32;
            const int64_t d = (int64_t)r[t] - (int64_t)(uint32_t)p + br;
            r[t] = (uint32_t)d;
            br = d >> 32;
        }
        const int64_t t8 = (int64_t)r[8] - (int64_t)carry + br;
        r[8] = (uint32_t)t8;
        bool neg = t8 < 0;
        uint32_t qfix = 0;
        FE_UNROLL for (int t = 0; t < 2; t++) {
            uint64_t cc = 0;
            uint32_t a9[9];
            FE_UNROLL for (int u = 0; u < 8; u++) {
                cc += (uint64_t)r[u] + V.v[u];
                a9[u] = (uint32_t)cc;
                cc >>= 32;
            }
            cc += (uint64_t)r[8];
            a9[8] = (uint32_t)cc;
            const bool wrapped = (cc >> 32) != 0;
            FE_UNROLL for (int u = 0; u < 9; u++) r[u] = neg ? a9[u] : r[u];
            qfix += neg ? 1u : 0u;
            neg = neg && !wrapped;
        }
        X[0] = qd - qfix;
    }
    // results behind the arguments: div[m + 1], then mod[k]
    const uint32_t rb = 2 * k + m;
    fe rn;
    FE_UNROLL for (int j = 0; j < 8; j++) rn.v[j] = r[j];
    big_unpack(regs, rb + m + 1, k, n, c, fe_shr_raw(rn, s));
    if (n == 64) {
        FE_UNROLL for (int i = 0; i < NW / 2; i++)
            if ((uint32_t)i <= m) {
                fe l = fe_zero();
                l.v[0] = X[2 * i];
                l.v[1] = X[2 * i + 1];
                fn_store(rb + i, regs, c, l);
            }
    } else if (n == 32) {
        FE_UNROLL for (int i = 0; i < NW; i++)
            if ((uint32_t)i <= m) {
                fe l = fe_zero();
                l.v[0] = X[i];
                fn_store(rb + i, regs, c, l);
            }
    } else {
        for (uint32_t i = 0; i <= m; i++) {
            const uint32_t sh = n * i, w0 = sh >> 5, bs = sh & 31u;                 // wave-uniform
            uint32_t x0 = 0, x1 = 0, x2 = 0;
            FE_UNROLL for (int j = 0; j < NW; j++) {
                x0 |= (uint32_t)j == w0 ? X[j] : 0u;
                x1 |= (uint32_t)j == w0 + 1 ? X[j] : 0u;
                x2 |= (uint32_t)j == w0 + 2 ? X[j] : 0u;
            }
            const uint64_t lo = (((uint64_t)x1 << 32) | x0) >> bs;
            const uint64_t v = lo | (bs ? ((uint64_t)x2 << (64 - bs)) : 0ull);
            const uint64_t mk = n >= 64 ? ~0ull : ((1ull << n) - 1);
            fe l = fe_zero();
            l.v[0] = (uint32_t)(v & mk);
            l.v[1] = (uint32_t)((v & mk) >> 32);
            fn_store(rb + i, regs, c, l);
        }
    }
}

__device__ CW_CALL_NATIVE void eval_call_native(uint32_t w, char *regs, const EvalCtx &c) {
    const uint32_t kind = w & 15u, k = (w >> 4) & 15u, n = (w >> 8) & 255u;
    const FpParams P2 = *(const FpParams *)(c.ftab + (w >> 16));               // wave-uniform: scalar loads
    if (kind == 1) {                                                          // mod_inv(a) -> a^-1 (0 for 0)
        const fe a = fe_csub_q(big_pack(regs, 0, k, n, c), P2);
        big_unpack(regs, k, k, n, c, fe_inv(a, P2));
        return;
    }
    const fe x1 = big_pack(regs, 0, k, n, c), y1 = big_pack(regs, k, k, n, c);
    fe num, den, xo;
    if (kind == 2) {                                                          // chord through (x1, y1), (x2, y2)
        xo = big_pack(regs, 2 * k, k, n, c);
        den = fe_sub(xo, x1, P2);
        num = fe_sub(big_pack(regs, 3 * k, k, n, c), y1, P2);
    } else {                                                                  // tangent: 3 x1^2 / (2 y1)
        xo = x1;
        const fe xx = fe_mul2(x1, x1, P2);
        num = fe_add(fe_add(xx, xx, P2), xx, P2);
        den = fe_add(y1, y1, P2);
    }
    const fe lam = fe_mul2(num, fe_inv(den, P2), P2);
    const fe x3 = fe_sub(fe_sub(fe_mul2(lam, lam, P2), x1, P2), xo, P2);
    const fe y3 = fe_sub(fe_mul2(lam, fe_sub(x1, x3, P2), P2), y1, P2);
    const uint32_t rb = kind == 2 ? 4 * k : 2 * k;
    big_unpack(regs, rb, k, n, c, lam);
    big_unpack(regs, rb + k, k, n, c, x3);
    big_unpack(regs, rb + 2 * k, k, n, c, y3);
}

__device__ __forceinline__ void eval_call_body(uint32_t fn, uint64_t reg_off, uint32_t row_id, uint32_t &st, const EvalCtx &c, const FpParams &P) {
    const uint4 ft = c.ftab[fn];
    const uint4 *code = c.fcode + ft.x;
    char *regs = (char *)c.Vb + reg_off;
#ifndef CW_NO_NATIVE_LONG_DIV            // (the emitted code's `call` body is built without it: fpjit.py keeps such programs on the interpreting kernel)
    if ((ft.w & 15u) == 4u) {
        eval_call_long_div(ft.w, regs, row_id, st, c);
        return;
    }
#endif
    if (ft.w & 15u) {
        eval_call_native(ft.w, regs, c);
        return;
    }
    uint32_t pc = 0, steps = 0;
    bool done = false;
    const uint32_t lane = __lane_id();
    const int pc_bits = 32 - __builtin_clz(ft.y | 1u);                   // instruction indices are < ft.y (wave-uniform)
    while (__any(!done)) {
        // The instruction to issue = the LOWEST program counter among the unfinished lanes (found bit by bit with ballots:
        // no cross-lane data movement, lanes outside EXEC never matter).  The bytecode of rtcode.py is structured - an
        // `if` jumps forward over its body, a loop jumps back to its head - so lanes that took different sides of a branch
        // meet again at its join instead of running one after the other to the end of the function (the first version
        // followed the first unfinished lane: 32 lanes with 32 different paths through long_div cost 32 passes).
        uint64_t cand = __ballot(!done);
        uint32_t cur = 0;
        for (int bit = pc_bits - 1; bit >= 0; bit--) {
            const uint64_t z = __ballot(!done && ((cand >> lane) & 1ull) && !((pc >> bit) & 1u));
            if (z) cand = z;
            else cur |= 1u << bit;
        }
        if (!done) {
            if (pc == cur) {
                const uint4 ins = code[cur];                              // wave-uniform
                const uint32_t op = ins.x, d = ins.y;
                pc = cur + 1;
                if (++steps > CW_CALL_STEP_LIMIT) {
                    cw_fail(st, CW_ST_ARITH, row_id);
                    done = true;
                } else if (op == F_RET) {
                    done = true;
                } else if (op == F_JMP) {
                    pc = d;
                } else if (op == F_JZ) {
                    if (fe_is_zero(fn_operand(ins.z, regs, c))) pc = d;
                } else if (op == F_LDX || op == F_STX) {
                    uint32_t idx;
                    if (!fn_index(fn_operand(ins.w & 0xFFFFu, regs, c), ins.w >> 16, &idx)) {
                        cw_fail(st, CW_ST_ARITH, row_id);
                        idx = 0;
                    }
                    if (op == F_LDX) fn_store(d, regs, c, fn_operand(ins.z + idx, regs, c));
                    else fn_store(d + idx, regs, c, fn_operand(ins.z, regs, c));
                } else {
                    const fe a = fn_operand(ins.z, regs, c);
                    fe b = fe_zero();
                    if (op != D_COPY && op != D_NEG && op != D_BNOT && op != D_LNOT) b = fn_operand(ins.w, regs, c);
                    fe r = fe_zero();
                    switch (op) {
                    case D_COPY: r = a; break;
                    case D_ADD: r = fe_add(a, b, P); break;
                    case D_SUB: r = fe_sub(a, b, P); break;
                    case D_NEG: r = fe_neg(a, P); break;
                    case D_MUL2: r = fe_mul2_auto(a, b, P); break;
                    case F_DIV: r = fe_mul2_auto(a, fe_inv(b, P), P); break;
                    case D_IDIV:
                    case D_MOD: {
                        fe qq, rr;
                        if (fe_is_zero(b)) {
                            cw_fail(st, CW_ST_ARITH, row_id);
                        } else {
                            fe_divmod(a, b, &qq, &rr);
                            r = (op == D_IDIV) ? qq : rr;
                        }
                        break;
                    }
                    case D_POW: r = fe_pow(a, b, P); break;
                    case D_SHL: r = fe_shl(a, b, P); break;
                    case D_SHR: r = fe_shr(a, b, P); break;
                    case D_BAND: r = fe_band(a, b, P); break;
                    case D_BOR: r = fe_bor(a, b, P); break;
                    case D_BXOR: r = fe_bxor(a, b, P); break;
                    case D_BNOT: r = fe_bnot(a, P); break;
                    case D_LT: r = fe_small(fe_lt(a, b, P)); break;
                    case D_GT: r = fe_small(fe_lt(b, a, P)); break;
                    case D_LEQ: r = fe_small(!fe_lt(b, a, P)); break;
                    case D_GEQ: r = fe_small(!fe_lt(a, b, P)); break;
                    case D_EQ: r = fe_small(fe_eq(a, b)); break;
                    case D_NEQ: r = fe_small(!fe_eq(a, b)); break;
                    case D_LAND: r = fe_small(!fe_is_zero(a) & !fe_is_zero(b)); break;
                    case D_LOR: r = fe_small(!fe_is_zero(a) | !fe_is_zero(b)); break;
                    case D_LNOT: r = fe_small(fe_is_zero(a)); break;
                    default: break;
                    }
                    fn_store(d, regs, c, r);
                }
            }
        }
    }
}

