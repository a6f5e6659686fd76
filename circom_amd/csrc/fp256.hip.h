// fp256.hip.h — device arithmetic on 256-bit prime-field elements for gfx950 (CDNA4).
//
// The reference's native field library is x86-64 (`mulx/adcx/adox`, <prime>/fr.asm) or GMP mpn_*
// (generic/fr.cpp:19-376).  CDNA4 has no 64x64 multiplier and no carry flag chained across
// instructions for free, so the design is re-done for the VALU: stored elements are 8 x u32 limbs
// (canonical, what .wtns wants); add/sub/bitwise/compare work on those directly; products switch to
// 9 x 29-bit limbs with lazy carries so that every partial product is one v_mad_u64_u32 (see fe29_mmul).
// One witness instance per lane; the modulus and its constants are wave-uniform (SGPRs, struct FpParams).
//
// Semantics follow generic/fr.cpp (cited per function); everything operates on raw residues in
// [0,q): whether a residue is "canonical" or "Montgomery" is the schedule's business (lower.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cw_tape.h"

struct fe { uint32_t v[8]; };

#define FE_UNROLL _Pragma("unroll")

__device__ __forceinline__ fe fe_zero() {
    fe r;
    FE_UNROLL for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
__device__ __forceinline__ fe fe_small(uint32_t x) {
    fe r = fe_zero();
    r.v[0] = x;
    return r;
}
__device__ __forceinline__ fe fe_from(const uint32_t *p) {
    fe r;
    FE_UNROLL for (int i = 0; i < 8; i++) r.v[i] = p[i];
    return r;
}
__device__ __forceinline__ bool fe_is_zero(const fe &a) {
    uint32_t o = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
__device__ __forceinline__ bool fe_eq(const fe &a, const fe &b) {
    uint32_t o = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
// a < b as unsigned 256-bit integers (Fr_rawCmp, generic/fr.cpp:263)
__device__ __forceinline__ bool fe_ltu(const fe &a, const uint32_t *b) {
    int64_t br = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) {
        int64_t d = (int64_t)a.v[i] - (int64_t)b[i] + br;
        br = d >> 32;
    }
    return br != 0;
}
__device__ __forceinline__ bool fe_ltu(const fe &a, const fe &b) { return fe_ltu(a, b.v); }
// a > b
__device__ __forceinline__ bool fe_gtu(const fe &a, const uint32_t *b) {
    int64_t br = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) {
        int64_t d = (int64_t)b[i] - (int64_t)a.v[i] + br;
        br = d >> 32;
    }
    return br != 0;
}

// r = a - q if a >= q else a   (the single conditional subtraction of fr.cpp:24-27,297-301)
__device__ __forceinline__ fe fe_csub_q(const fe &a, const FpParams &P) {
    fe t;
    int64_t br = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) {
        int64_t d = (int64_t)a.v[i] - (int64_t)P.q[i] + br;
        t.v[i] = (uint32_t)d;
        br = d >> 32;
    }
    fe r;
    FE_UNROLL for (int i = 0; i < 8; i++) r.v[i] = br ? a.v[i] : t.v[i];
    return r;
}

// Fr_rawAdd, generic/fr.cpp:19-27
__device__ __forceinline__ fe fe_add(const fe &a, const fe &b, const FpParams &P) {
    fe s, t;
    uint64_t c = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        s.v[i] = (uint32_t)c;
        c >>= 32;
    }
    int64_t br = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) {
        int64_t d = (int64_t)s.v[i] - (int64_t)P.q[i] + br;
        t.v[i] = (uint32_t)d;
        br = d >> 32;
    }
    bool use_t = (c != 0) | (br == 0);
    fe r;
    FE_UNROLL for (int i = 0; i < 8; i++) r.v[i] = use_t ? t.v[i] : s.v[i];
    return r;
}

// Fr_rawSub, generic/fr.cpp:39-47
__device__ __forceinline__ fe fe_sub(const fe &a, const fe &b, const FpParams &P) {
    fe d;
    int64_t br = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) {
        int64_t x = (int64_t)a.v[i] - (int64_t)b.v[i] + br;
        d.v[i] = (uint32_t)x;
        br = x >> 32;
    }
    uint32_t m = br ? 0xFFFFFFFFu : 0u;
    uint64_t c = 0;
    fe r;
    FE_UNROLL for (int i = 0; i < 8; i++) {
        c += (uint64_t)d.v[i] + (P.q[i] & m);
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

// Fr_rawNeg, generic/fr.cpp:76-86
__device__ __forceinline__ fe fe_neg(const fe &a, const FpParams &P) {
    bool z = fe_is_zero(a);
    fe r;
    int64_t br = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) {
        int64_t x = (int64_t)P.q[i] - (int64_t)a.v[i] + br;
        r.v[i] = z ? 0u : (uint32_t)x;
        br = x >> 32;
    }
    return r;
}

// ---- Montgomery product  a*b*R'^-1 mod q,  R' = 2^261 ------------------------------------------------
// Role of Fr_rawMMul (generic/fr.cpp:110-164, <prime>/fr.asm:365), re-designed for the CDNA4 VALU.
// Measured on gfx950 (tools/ubench_valu): v_mad_u64_u32 issues at ~the same rate as a plain 32-bit add,
// so the cost of a multi-limb product is its INSTRUCTION COUNT, and carry handling (add_co/addc/mov
// pairs) is what dominated a 8x32-bit CIOS (589 VALU instructions, only 136 of them multiplies).
// Hence: 9 limbs of 29 bits with LAZY carries.  Every partial product is < 2^58 and each 64-bit column
// accumulator takes at most 9 (a*b) + 9 (m*q) of them plus one carry (< 2^63): a column update is
// exactly ONE v_mad_u64_u32 and carries are resolved once per row (shift+add) instead of per product.
// The radix change (2^261 instead of the reference's 2^256) is invisible outside the schedule: slots
// hold canonical residues, the lowering pre-scales constants by R' (hip_elements/lower.py).
#define FE29_MASK 0x1FFFFFFFu
struct fe29 { uint32_t l[9]; };

__device__ __forceinline__ fe29 fe_to29(const fe &a) {
    fe29 r;
    FE_UNROLL for (int k = 0; k < 9; k++) {
        const int bit = 29 * k, w = bit >> 5, sh = bit & 31;
        uint32_t v;
        if (sh == 0) v = a.v[w];
        else if (w + 1 < 8) v = __builtin_amdgcn_alignbit(a.v[w + 1], a.v[w], sh);
        else v = a.v[w] >> sh;
        r.l[k] = v & FE29_MASK;
    }
    return r;
}
__device__ __forceinline__ fe fe_from29(const fe29 &a) {
    fe r;
    FE_UNROLL for (int w = 0; w < 8; w++) {
        const int bit = 32 * w, k = bit / 29, o = bit - 29 * k;      // word w starts at bit o of limb k
        uint32_t v = a.l[k] >> o;
        v |= a.l[k + 1] << (29 - o);                                  // o <= 21: two limbs always cover the word
        r.v[w] = v;
    }
    return r;
}

// operands and result as 29-bit limbs, all < q
__device__ __forceinline__ fe29 fe29_mmul(const fe29 &a, const fe29 &b, const FpParams &P) {
    uint64_t acc[18];
    FE_UNROLL for (int i = 0; i < 18; i++) acc[i] = 0;
    FE_UNROLL for (int i = 0; i < 9; i++) {
        const uint32_t bi = b.l[i];
        FE_UNROLL for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)a.l[j] * bi;
        const uint32_t m = ((uint32_t)acc[i] * P.np29) & FE29_MASK;
        FE_UNROLL for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)m * P.q29[j];
        acc[i + 1] += acc[i] >> 29;                                   // low 29 bits of acc[i] are now zero
    }
    // normalise columns 9..17 into limbs, then one conditional subtraction of q (result < 2q)
    fe29 r;
    uint64_t c = 0;
    FE_UNROLL for (int k = 0; k < 9; k++) {
        c += acc[9 + k];
        r.l[k] = (uint32_t)c & FE29_MASK;
        c >>= 29;
    }
    fe29 d;
    int32_t br = 0;
    FE_UNROLL for (int k = 0; k < 9; k++) {
        int32_t t = (int32_t)r.l[k] - (int32_t)P.q29[k] + br;
        d.l[k] = (uint32_t)t & FE29_MASK;
        br = t >> 31;
    }
    FE_UNROLL for (int k = 0; k < 9; k++) r.l[k] = br ? r.l[k] : d.l[k];
    return r;
}

// (a*b + c) * R'^-1 mod q with c < 2^261 given as limbs: the addend enters the low columns before the reduction.
// Used by the R1CS check: A*B == C  <=>  (A*B + (q - C)) * R'^-1 == 0, one product instead of two.
__device__ __forceinline__ fe29 fe29_mmul_add(const fe29 &a, const fe29 &b, const fe29 &cadd, const FpParams &P) {
    uint64_t acc[18];
    FE_UNROLL for (int i = 0; i < 9; i++) acc[i] = cadd.l[i];
    FE_UNROLL for (int i = 9; i < 18; i++) acc[i] = 0;
    FE_UNROLL for (int i = 0; i < 9; i++) {
        const uint32_t bi = b.l[i];
        FE_UNROLL for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)a.l[j] * bi;
        const uint32_t m = ((uint32_t)acc[i] * P.np29) & FE29_MASK;
        FE_UNROLL for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)m * P.q29[j];
        acc[i + 1] += acc[i] >> 29;
    }
    fe29 r;
    uint64_t c = 0;
    FE_UNROLL for (int k = 0; k < 9; k++) {
        c += acc[9 + k];
        r.l[k] = (uint32_t)c & FE29_MASK;
        c >>= 29;
    }
    fe29 d;
    int32_t br = 0;
    FE_UNROLL for (int k = 0; k < 9; k++) {
        int32_t t = (int32_t)r.l[k] - (int32_t)P.q29[k] + br;
        d.l[k] = (uint32_t)t & FE29_MASK;
        br = t >> 31;
    }
    FE_UNROLL for (int k = 0; k < 9; k++) r.l[k] = br ? r.l[k] : d.l[k];
    return r;
}

// ---- dot products with one reduction ---------------------------------------------------------------------------
// acc (17 columns + carry) += a * c, the 81 unreduced partial products; c = wave-uniform limbs (SGPRs)
__device__ __forceinline__ void fe29_mac(uint64_t acc[18], const fe29 &a, const uint32_t *c29) {
    FE_UNROLL for (int i = 0; i < 9; i++) {
        const uint32_t ci = c29[i];
        FE_UNROLL for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)a.l[j] * ci;
    }
}
// Montgomery reduction of such an accumulator: (acc * R'^-1) mod q, valid while every column stays < 2^64, i.e. for
// up to 4 accumulated products (45 terms of < 2^58 each per column)
__device__ __forceinline__ fe29 fe29_reduce(uint64_t acc[18], const FpParams &P) {
    FE_UNROLL for (int i = 0; i < 9; i++) {
        const uint32_t m = ((uint32_t)acc[i] * P.np29) & FE29_MASK;
        FE_UNROLL for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)m * P.q29[j];
        acc[i + 1] += acc[i] >> 29;
    }
    fe29 r;
    uint64_t c = 0;
    FE_UNROLL for (int k = 0; k < 9; k++) {
        c += acc[9 + k];
        r.l[k] = (uint32_t)c & FE29_MASK;
        c >>= 29;
    }
    fe29 d;
    int32_t br = 0;
    FE_UNROLL for (int k = 0; k < 9; k++) {
        int32_t t = (int32_t)r.l[k] - (int32_t)P.q29[k] + br;
        d.l[k] = (uint32_t)t & FE29_MASK;
        br = t >> 31;
    }
    FE_UNROLL for (int k = 0; k < 9; k++) r.l[k] = br ? r.l[k] : d.l[k];
    return r;
}

__device__ __forceinline__ fe fe_mmul(const fe &a, const fe &b, const FpParams &P) {
    return fe_from29(fe29_mmul(fe_to29(a), fe_to29(b), P));
}
// canonical product a*b mod q = MMUL(MMUL(a,b), R'^2); the intermediate never leaves the 29-bit limb form and is
// not brought below q (inputs < 2q keep every bound of fe29_mmul: R' = 2^261 > 4q).  When every lane multiplies a value
// by itself (x*x and x^2*x^2 of the x^5 S-box: two of the three products of every Poseidon round) the first product
// is a squaring: 45 instead of 81 partial products and one limb conversion less.
__device__ __forceinline__ fe29 fe29_tail_nc(uint64_t acc[18]) {       // columns 9..17 -> limbs, value < 2q
    fe29 r;
    uint64_t c = 0;
    FE_UNROLL for (int k = 0; k < 9; k++) {
        c += acc[9 + k];
        r.l[k] = (uint32_t)c & FE29_MASK;
        c >>= 29;
    }
    return r;
}
__device__ __forceinline__ fe29 fe29_mmul_nc(const fe29 &a, const fe29 &b, const FpParams &P) {
    uint64_t acc[18];
    FE_UNROLL for (int i = 0; i < 18; i++) acc[i] = 0;
    FE_UNROLL for (int i = 0; i < 9; i++) {
        const uint32_t bi = b.l[i];
        FE_UNROLL for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)a.l[j] * bi;
        const uint32_t m = ((uint32_t)acc[i] * P.np29) & FE29_MASK;
        FE_UNROLL for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)m * P.q29[j];
        acc[i + 1] += acc[i] >> 29;
    }
    return fe29_tail_nc(acc);
}
__device__ __forceinline__ fe29 fe29_msqr_nc(const fe29 &a, const FpParams &P) {
    uint64_t acc[18];
    FE_UNROLL for (int i = 0; i < 18; i++) acc[i] = 0;
    FE_UNROLL for (int i = 0; i < 9; i++) {
        acc[2 * i] += (uint64_t)a.l[i] * a.l[i];
        const uint32_t a2 = a.l[i] << 1;                              // 30 bits: 2 a_i a_j < 2^59, <= 4 per column
        FE_UNROLL for (int j = i + 1; j < 9; j++) acc[i + j] += (uint64_t)a2 * a.l[j];
    }
    FE_UNROLL for (int i = 0; i < 9; i++) {
        const uint32_t m = ((uint32_t)acc[i] * P.np29) & FE29_MASK;
        FE_UNROLL for (int j = 0; j < 9; j++) acc[i + j] += (uint64_t)m * P.q29[j];
        acc[i + 1] += acc[i] >> 29;
    }
    return fe29_tail_nc(acc);
}
__device__ __forceinline__ fe fe_mul2(const fe &a, const fe &b, const FpParams &P) {
    fe29 r2;                                                        // wave-uniform limbs: stay in SGPRs
    FE_UNROLL for (int k = 0; k < 9; k++) r2.l[k] = P.r2_29[k];
    const fe29 a29 = fe_to29(a);
    fe29 t;
    if (__all(fe_eq(a, b))) t = fe29_msqr_nc(a29, P);
    else t = fe29_mmul_nc(a29, fe_to29(b), P);
    return fe_from29(fe29_mmul(t, r2, P));
}

// ---- run-time short path for products of small signed values -------------------------------------------
// The reference keeps small values as int32 "short" elements and multiplies them with one imul
// (mul_s1s2, generic/fr.cpp:416-439); bit-heavy circuits (SHA-256, Num2Bits, comparators) run almost
// entirely on that path.  The device representation is uniform (canonical 256-bit), so the short path is
// selected PER WAVE at run time: if every lane's operands are "signed small" (x < 2^64 or q - x < 2^64) the
// product is a 64x64 -> 128-bit multiply (4 v_mad_u64_u32) plus one conditional q - p; otherwise the wave
// takes the generic Montgomery path.  Both paths produce the same canonical residue, so the choice never
// changes a result.
__device__ __forceinline__ uint32_t fe_hi_or(const fe &x) { return x.v[2] | x.v[3] | x.v[4] | x.v[5] | x.v[6] | x.v[7]; }

// decode x as sign * mag with mag < 2^64; returns false for lanes where that is impossible
__device__ __forceinline__ bool fe_signed_small(const fe &x, const FpParams &P, uint64_t *mag, bool *neg) {
    const bool pos = fe_hi_or(x) == 0;
    fe d;
    int64_t br = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) {
        int64_t t = (int64_t)P.q[i] - (int64_t)x.v[i] + br;
        d.v[i] = (uint32_t)t;
        br = t >> 32;
    }
    const bool ng = fe_hi_or(d) == 0;
    *mag = pos ? (((uint64_t)x.v[1] << 32) | x.v[0]) : (((uint64_t)d.v[1] << 32) | d.v[0]);
    *neg = !pos;
    return pos | ng;
}
// sign * (ma * mb) as a canonical residue
__device__ __forceinline__ fe fe_small_product(uint64_t ma, uint64_t mb, bool neg, const FpParams &P) {
    const uint32_t a0 = (uint32_t)ma, a1 = (uint32_t)(ma >> 32), b0 = (uint32_t)mb, b1 = (uint32_t)(mb >> 32);
    uint64_t t = (uint64_t)a0 * b0;
    fe p = fe_zero();
    p.v[0] = (uint32_t)t;
    t = (uint64_t)a0 * b1 + (t >> 32);
    const uint64_t t2 = (uint64_t)a1 * b0 + (uint32_t)t;
    p.v[1] = (uint32_t)t2;
    t = (uint64_t)a1 * b1 + (t >> 32) + (t2 >> 32);
    p.v[2] = (uint32_t)t;
    p.v[3] = (uint32_t)(t >> 32);
    const fe n = fe_neg(p, P);                         // q - p, and 0 for p = 0
    fe r;
    FE_UNROLL for (int i = 0; i < 8; i++) r.v[i] = neg ? n.v[i] : p.v[i];
    return r;
}
// canonical a*b with the short path
__device__ __forceinline__ fe fe_mul2_auto(const fe &a, const fe &b, const FpParams &P) {
    if (__all((fe_hi_or(a) | fe_hi_or(b)) == 0)) {    // every lane: both operands < 2^64
        return fe_small_product(((uint64_t)a.v[1] << 32) | a.v[0], ((uint64_t)b.v[1] << 32) | b.v[0], false, P);
    }
    uint64_t ma, mb;
    bool na, nb;
    const bool ok = fe_signed_small(a, P, &ma, &na) & fe_signed_small(b, P, &mb, &nb);
    if (__all(ok)) return fe_small_product(ma, mb, na != nb, P);
    return fe_mul2(a, b, P);
}
// canonical a*c for a compile-time constant: bm = c*R' (generic path), (cmag, cneg) = |val(c)| when small
__device__ __forceinline__ fe fe_mulc_auto(const fe &a, const fe &bm, bool c_small, uint64_t cmag, bool cneg,
                                           const FpParams &P) {
    if (c_small) {
        if (__all(fe_hi_or(a) == 0)) return fe_small_product(((uint64_t)a.v[1] << 32) | a.v[0], cmag, cneg, P);
        uint64_t ma;
        bool na;
        const bool ok = fe_signed_small(a, P, &ma, &na);
        if (__all(ok)) return fe_small_product(ma, cmag, na != cneg, P);
    }
    return fe_mmul(a, bm, P);
}

// ---- bitwise operators on canonical values (Fr_rawAnd/Or/Xor/Not, generic/fr.cpp:293-327,366-376) ----
__device__ __forceinline__ fe fe_mask_wrap(fe r, const FpParams &P) {
    r.v[7] &= P.topmask;
    return fe_csub_q(r, P);
}
__device__ __forceinline__ fe fe_band(const fe &a, const fe &b, const FpParams &P) {
    fe r;
    FE_UNROLL for (int i = 0; i < 8; i++) r.v[i] = a.v[i] & b.v[i];
    return fe_mask_wrap(r, P);
}
__device__ __forceinline__ fe fe_bor(const fe &a, const fe &b, const FpParams &P) {
    fe r;
    FE_UNROLL for (int i = 0; i < 8; i++) r.v[i] = a.v[i] | b.v[i];
    return fe_mask_wrap(r, P);
}
__device__ __forceinline__ fe fe_bxor(const fe &a, const fe &b, const FpParams &P) {
    fe r;
    FE_UNROLL for (int i = 0; i < 8; i++) r.v[i] = a.v[i] ^ b.v[i];
    return fe_mask_wrap(r, P);
}
__device__ __forceinline__ fe fe_bnot(const fe &a, const FpParams &P) {
    fe r;
    FE_UNROLL for (int i = 0; i < 8; i++) r.v[i] = ~a.v[i];
    return fe_mask_wrap(r, P);
}

// 256-bit logical shifts by s in [0,255] (Fr_rawShl/Shr, generic/fr.cpp:329-364): barrel shifter,
// branch-free so a lane-varying amount costs the same as a uniform one.
__device__ __forceinline__ fe fe_shl_raw(const fe &a, uint32_t s) {
    const bool s4 = s & 128, s2 = s & 64, s1 = s & 32;
    fe t, u, w;
    FE_UNROLL for (int i = 0; i < 8; i++) t.v[i] = s4 ? ((i >= 4) ? a.v[(i >= 4) ? i - 4 : 0] : 0u) : a.v[i];
    FE_UNROLL for (int i = 0; i < 8; i++) u.v[i] = s2 ? ((i >= 2) ? t.v[(i >= 2) ? i - 2 : 0] : 0u) : t.v[i];
    FE_UNROLL for (int i = 0; i < 8; i++) w.v[i] = s1 ? ((i >= 1) ? u.v[(i >= 1) ? i - 1 : 0] : 0u) : u.v[i];
    const uint32_t bs = s & 31;
    const uint32_t rs = (32u - bs) & 31u;                 // v_alignbit(hi, lo, rs) = low 32 bits of (hi:lo) >> rs
    fe r;
    FE_UNROLL for (int i = 7; i >= 1; i--)
        r.v[i] = bs ? __builtin_amdgcn_alignbit(w.v[i], w.v[i - 1], rs) : w.v[i];
    r.v[0] = w.v[0] << bs;
    return r;
}
__device__ __forceinline__ fe fe_shr_raw(const fe &a, uint32_t s) {
    const bool s4 = s & 128, s2 = s & 64, s1 = s & 32;
    fe t, u, w;
    FE_UNROLL for (int i = 0; i < 8; i++) t.v[i] = s4 ? ((i + 4 < 8) ? a.v[(i + 4 < 8) ? i + 4 : 0] : 0u) : a.v[i];
    FE_UNROLL for (int i = 0; i < 8; i++) u.v[i] = s2 ? ((i + 2 < 8) ? t.v[(i + 2 < 8) ? i + 2 : 0] : 0u) : t.v[i];
    FE_UNROLL for (int i = 0; i < 8; i++) w.v[i] = s1 ? ((i + 1 < 8) ? u.v[(i + 1 < 8) ? i + 1 : 0] : 0u) : u.v[i];
    const uint32_t bs = s & 31;
    fe r;
    FE_UNROLL for (int i = 0; i < 7; i++) r.v[i] = __builtin_amdgcn_alignbit(w.v[i + 1], w.v[i], bs);
    r.v[7] = w.v[7] >> bs;
    return r;
}

// classify a shift amount y (canonical): returns 0 = forward by *amt, 1 = reversed by *amt, 2 = result 0
// (Fr_shl/Fr_shr + *_big_shift, generic/fr.cpp:2157-2307: y < qbits forward; else k = q - y reversed
//  if k < qbits; else 0)
__device__ __forceinline__ int fe_shift_kind(const fe &y, const FpParams &P, uint32_t *amt) {
    uint32_t hi = 0;
    FE_UNROLL for (int i = 1; i < 8; i++) hi |= y.v[i];
    if (hi == 0 && y.v[0] < P.qbits) { *amt = y.v[0]; return 0; }
    // k = q - y
    uint32_t k0 = 0, khi = 0;
    int64_t br = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) {
        int64_t d = (int64_t)P.q[i] - (int64_t)y.v[i] + br;
        if (i == 0) k0 = (uint32_t)d; else khi |= (uint32_t)d;
        br = d >> 32;
    }
    if (khi == 0 && k0 < P.qbits) { *amt = k0; return 1; }
    *amt = 0;
    return 2;
}
__device__ __forceinline__ fe fe_shl(const fe &x, const fe &y, const FpParams &P) {
    uint32_t amt;
    int k = fe_shift_kind(y, P, &amt);
    fe l = fe_mask_wrap(fe_shl_raw(x, amt), P);
    fe r = fe_shr_raw(x, amt);
    fe o;
    FE_UNROLL for (int i = 0; i < 8; i++) o.v[i] = (k == 0) ? l.v[i] : ((k == 1) ? r.v[i] : 0u);
    return o;
}
__device__ __forceinline__ fe fe_shr(const fe &x, const fe &y, const FpParams &P) {
    uint32_t amt;
    int k = fe_shift_kind(y, P, &amt);
    fe l = fe_mask_wrap(fe_shl_raw(x, amt), P);
    fe r = fe_shr_raw(x, amt);
    fe o;
    FE_UNROLL for (int i = 0; i < 8; i++) o.v[i] = (k == 0) ? r.v[i] : ((k == 1) ? l.v[i] : 0u);
    return o;
}

// relational operators compare val(x) = x - q if x > half else x   (rltL1L2, generic/fr.cpp:1208-1218)
__device__ __forceinline__ bool fe_lt(const fe &x, const fe &y, const FpParams &P) {
    bool nx = fe_gtu(x, P.half), ny = fe_gtu(y, P.half);
    bool ltu = fe_ltu(x, y);
    return (nx != ny) ? nx : ltu;
}

// ---- slow-path operators (only in the "full" kernel) -----------------------------------------
// Modular inverse, canonical -> canonical, inv(0) = 0 (Fr_inv -> mpz_invert, generic/fr.cpp:2895-2906).
// Lanes cannot branch independently without serialising the wave, so this is a constant-time binary extended
// GCD in the style of Pornin's "Optimized Binary GCD for Modular Inversion": INV_K = 30 halving steps at a
// time run on 64-bit approximations of (a, b) (low 30 bits exact + top 34 bits of the longer one) and yield
// update factors |f| + |g| <= 2^30, which are then applied to the full-width values
//     a, b <- |f0 a + g0 b| / 2^30, |f1 a + g1 b| / 2^30            (exact divisions)
//     u, v <- +-(f0 u + g0 v) / 2^30,  +-(f1 u + g1 v) / 2^30  mod q (Montgomery-style division)
// keeping a = u*y, b = v*y (mod q).  After ceil((2*qbits - 1) / 30) rounds (17 for a 254-bit prime) b = 1 and
// v = 1/y: ~22 K VALU instructions instead of ~100 K for the Fermat power y^(q-2).
// oracle/bingcd_model.py restates this routine limb for limb; tests pin both against pow(y, -1, q).
// slow-path operators: out of line in the interpreting kernels (one copy, registers of the lean variant unaffected), inlined
// into the row bodies of the emitted code (fpjit.py: a body is a leaf, it cannot call)
#ifndef CW_FE_SLOW
#define CW_FE_SLOW __noinline__
#endif
#define INV_K 30
struct w9 { uint32_t v[9]; };

// f*x + g*y as a 288-bit two's complement value (wrap-around).  Signed factors are offset to unsigned ones:
// f x + g y = (f + 2^30) x + (g + 2^30) y - 2^30 (x + y).
__device__ __forceinline__ w9 inv_lincomb(const fe &x, const fe &y, int32_t f, int32_t g) {
    const uint32_t fp = (uint32_t)(f + (1 << INV_K)), gp = (uint32_t)(g + (1 << INV_K));
    w9 U;
    uint64_t c = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) { c += (uint64_t)x.v[i] * fp; U.v[i] = (uint32_t)c; c >>= 32; }
    U.v[8] = (uint32_t)c;
    c = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) { c += (uint64_t)y.v[i] * gp + U.v[i]; U.v[i] = (uint32_t)c; c >>= 32; }
    U.v[8] += (uint32_t)c;
    uint32_t S[9];
    c = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) { c += (uint64_t)x.v[i] + y.v[i]; S[i] = (uint32_t)c; c >>= 32; }
    S[8] = (uint32_t)c;
    w9 D;
    uint32_t prev = 0, br = 0;
    FE_UNROLL for (int i = 0; i < 9; i++) {
        const uint32_t w = __builtin_amdgcn_alignbit(S[i], prev, 32 - INV_K);     // (S << 30) limb i
        prev = S[i];
        const uint64_t t = (uint64_t)U.v[i] - w - br;
        D.v[i] = (uint32_t)t;
        br = (uint32_t)(t >> 32) & 1u;
    }
    return D;
}
__device__ __forceinline__ void inv_cond_neg(w9 &d, bool neg) {
    const uint32_t m = neg ? 0xFFFFFFFFu : 0u;
    uint64_t c = neg ? 1u : 0u;
    FE_UNROLL for (int i = 0; i < 9; i++) { c += (uint64_t)(d.v[i] ^ m); d.v[i] = (uint32_t)c; c >>= 32; }
}
// 64-bit approximations: both exact if max(len a, len b) <= 64, else low 30 bits | top 34 bits at that length
__device__ __forceinline__ void inv_approx(const fe &a, const fe &b, uint64_t &xa, uint64_t &xb) {
    int t = 0;                                                      // index of the top non-zero limb of a | b
    uint32_t top = a.v[0] | b.v[0];
    FE_UNROLL for (int i = 1; i < 8; i++) {
        const uint32_t o = a.v[i] | b.v[i];
        t = o ? i : t;
        top = o ? o : top;
    }
    const int n = 32 * t + 32 - __clz(top | 1u);                    // bit length (>= 1)
    const int s = n - 34;                                           // top window starts at bit s
    const int w = s >> 5;                                           // (only used when n > 64, i.e. s >= 31, w >= 0)
    const uint32_t sh = (uint32_t)s & 31u;
    uint32_t a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
    FE_UNROLL for (int i = 0; i < 8; i++) {
        a0 = (w == i) ? a.v[i] : a0;     b0 = (w == i) ? b.v[i] : b0;
        a1 = (w + 1 == i) ? a.v[i] : a1; b1 = (w + 1 == i) ? b.v[i] : b1;
        a2 = (w + 2 == i) ? a.v[i] : a2; b2 = (w + 2 == i) ? b.v[i] : b2;
    }
    const uint32_t alo = __builtin_amdgcn_alignbit(a1, a0, sh), ahi = __builtin_amdgcn_alignbit(a2, a1, sh) & 3u;
    const uint32_t blo = __builtin_amdgcn_alignbit(b1, b0, sh), bhi = __builtin_amdgcn_alignbit(b2, b1, sh) & 3u;
    const uint64_t ta = (((uint64_t)ahi << 32 | alo) << INV_K) | (a.v[0] & 0x3FFFFFFFu);
    const uint64_t tb = (((uint64_t)bhi << 32 | blo) << INV_K) | (b.v[0] & 0x3FFFFFFFu);
    const bool exact = n <= 64;
    xa = exact ? ((uint64_t)a.v[1] << 32 | a.v[0]) : ta;
    xb = exact ? ((uint64_t)b.v[1] << 32 | b.v[0]) : tb;
}
__device__ CW_FE_SLOW fe fe_inv(const fe &y, const FpParams &P) {
    fe a = y, b = fe_from(P.q), u = fe_small(1), v = fe_zero();
    uint32_t ninv = 1;                                              // -q^-1 mod 2^30 (wave-uniform, scalar unit)
    for (int i = 0; i < 5; i++) ninv *= 2u - P.q[0] * ninv;
    ninv = (0u - ninv) & 0x3FFFFFFFu;
    w9 qs;                                                          // q << 30
    {
        uint32_t prev = 0;
        FE_UNROLL for (int i = 0; i < 8; i++) { qs.v[i] = __builtin_amdgcn_alignbit(P.q[i], prev, 32 - INV_K); prev = P.q[i]; }
        qs.v[8] = prev >> (32 - INV_K);
    }
    const int rounds = (2 * (int)P.qbits - 1 + INV_K - 1) / INV_K;
    for (int r = 0; r < rounds; r++) {
        // 2*qbits - 1 halvings is the worst case; ~1.4*qbits is typical.  Once a == 0 further rounds leave (b, v)
        // unchanged, so the wave stops as soon as every active lane is done.
        if (__all(fe_is_zero(a))) break;
        uint64_t xa, xb;
        inv_approx(a, b, xa, xb);
        int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
        for (int j = 0; j < INV_K; j++) {
            const bool odd = xa & 1u;
            const bool sw = odd & (xa < xb);
            const uint64_t ta = sw ? xb : xa, tb = sw ? xa : xb;
            const int32_t tf0 = sw ? f1 : f0, tf1 = sw ? f0 : f1, tg0 = sw ? g1 : g0, tg1 = sw ? g0 : g1;
            xa = (ta - (odd ? tb : 0ull)) >> 1;
            xb = tb;
            f0 = tf0 - (odd ? tf1 : 0);
            g0 = tg0 - (odd ? tg1 : 0);
            f1 = (int32_t)((uint32_t)tf1 << 1);
            g1 = (int32_t)((uint32_t)tg1 << 1);
        }
        fe na, nb, nu, nv;
        FE_UNROLL for (int h = 0; h < 2; h++) {
            const int32_t f = h ? f1 : f0, g = h ? g1 : g0;
            w9 d = inv_lincomb(a, b, f, g);
            const bool neg = d.v[8] >> 31;
            inv_cond_neg(d, neg);
            fe &oa = h ? nb : na;
            FE_UNROLL for (int i = 0; i < 8; i++) oa.v[i] = __builtin_amdgcn_alignbit(d.v[i + 1], d.v[i], INV_K);
            w9 e = inv_lincomb(u, v, f, g);
            inv_cond_neg(e, neg);
            uint64_t c = 0;                                         // + q 2^30: non-negative, < 2^31 q
            FE_UNROLL for (int i = 0; i < 9; i++) { c += (uint64_t)e.v[i] + qs.v[i]; e.v[i] = (uint32_t)c; c >>= 32; }
            const uint32_t k = (e.v[0] * ninv) & 0x3FFFFFFFu;       // + k q: the low 30 bits vanish
            c = 0;
            FE_UNROLL for (int i = 0; i < 8; i++) { c += (uint64_t)P.q[i] * k + e.v[i]; e.v[i] = (uint32_t)c; c >>= 32; }
            e.v[8] += (uint32_t)c;
            uint32_t t[9];                                          // / 2^30: < 3 q
            FE_UNROLL for (int i = 0; i < 8; i++) t[i] = __builtin_amdgcn_alignbit(e.v[i + 1], e.v[i], INV_K);
            t[8] = e.v[8] >> INV_K;
            FE_UNROLL for (int pass = 0; pass < 2; pass++) {          // two conditional subtractions of q
                uint32_t d2[9], br = 0;
                FE_UNROLL for (int i = 0; i < 9; i++) {
                    const uint64_t x = (uint64_t)t[i] - (i < 8 ? P.q[i] : 0u) - br;
                    d2[i] = (uint32_t)x;
                    br = (uint32_t)(x >> 32) & 1u;
                }
                FE_UNROLL for (int i = 0; i < 9; i++) t[i] = br ? t[i] : d2[i];
            }
            fe &ou = h ? nv : nu;
            FE_UNROLL for (int i = 0; i < 8; i++) ou.v[i] = t[i];
        }
        a = na; b = nb; u = nu; v = nv;
    }
    return v;
}
// x^y with a per-lane exponent (Fr_pow / mpz_powm, generic/fr.cpp:2877-2893; 0^0 = 1)
__device__ CW_FE_SLOW fe fe_pow(const fe &x, const fe &y, const FpParams &P) {
    fe29 r2;
    FE_UNROLL for (int k = 0; k < 9; k++) r2.l[k] = P.r2_29[k];
    const fe29 xm = fe29_mmul(fe_to29(x), r2, P);
    fe29 r = fe_to29(fe_from(P.one_m));
    // one loop over the 256 exponent bits; the word is picked by a select ladder on the (wave-uniform) index - indexing
    // y.v[] with a loop variable would make the compiler park the exponent in scratch / LDS
    for (int i = 255; i >= 0; i--) {
        const int w = i >> 5;
        const uint32_t ew = w == 0 ? y.v[0] : w == 1 ? y.v[1] : w == 2 ? y.v[2] : w == 3 ? y.v[3] : w == 4 ? y.v[4]
                            : w == 5 ? y.v[5] : w == 6 ? y.v[6] : y.v[7];
        r = fe29_mmul(r, r, P);
        const fe29 rx = fe29_mmul(r, xm, P);
        const bool bit = (ew >> (i & 31)) & 1;
        FE_UNROLL for (int k = 0; k < 9; k++) r.l[k] = bit ? rx.l[k] : r.l[k];
    }
    return fe_from29(fe29_mmul(r, fe_to29(fe_small(1)), P));
}
// floor(x / y), x mod y on canonical integers (Fr_idiv/Fr_mod via mpz_fdiv_q/r, generic/fr.cpp:2835-2875); y == 0 is
// reported by the caller.  Schoolbook long division in base 2^32 (Knuth's algorithm D) on operands normalised PER LANE:
// V = y << s with s = the leading zeros of y (so V's top bit is set), X = x << s (up to 511 bits: its high half is the first
// remainder, its low half supplies one digit per step); a quotient digit = one 64-by-32-bit estimate from the remainder's two
// top words, refined with V's second word (then at most one too large), a 9-word multiply-subtract and a conditional
// add-back.  ~150 instructions per digit instead of the 32 x ~80 of the restoring division it replaces (rounds 1-3), and a
// digit that is zero in every lane of the wave (r < V everywhere: numerators of bigint long division are 2-3 words) costs
// one comparison.
__device__ __forceinline__ uint32_t fe_clz256(const fe &a) {         // leading zeros of a 256-bit value (256 for zero)
    uint32_t n = 256;
    FE_UNROLL for (int k = 0; k < 8; k++)
        if (a.v[k]) n = (uint32_t)(32 * (7 - k)) + (uint32_t)__builtin_clz(a.v[k]);
    return n;
}
__device__ CW_FE_SLOW void fe_divmod(const fe &x, const fe &y, fe *quo, fe *rem) {
    const uint32_t s = fe_clz256(y) & 255u;                           // y != 0: 0..255
    const fe V = fe_shl_raw(y, s);
    const fe Xlo = fe_shl_raw(x, s);
    const fe Xhi_ = fe_shr_raw(x, (256u - s) & 255u);
    uint32_t r[9];
    FE_UNROLL for (int k = 0; k < 8; k++) r[k] = s ? Xhi_.v[k] : 0u;  // x >> (256 - s); nothing for s = 0
    r[8] = 0;
    fe q = fe_zero();
    const uint32_t v7 = V.v[7], v6 = V.v[6];
    FE_UNROLL for (int j = 7; j >= 0; j--) {
        FE_UNROLL for (int k = 8; k >= 1; k--) r[k] = r[k - 1];       // r = r * 2^32 + next digit of X  (r < V before)
        r[0] = Xlo.v[j];
        // r >= V in some lane?  (9-word compare; r[8] != 0 implies r >= V)
        bool lt = false, decided = r[8] != 0;
        FE_UNROLL for (int k = 7; k >= 0; k--) {
            if (!decided && r[k] != V.v[k]) { lt = r[k] < V.v[k]; decided = true; }
        }
        if (!__any(!lt)) continue;                                    // every lane's digit is zero
        const uint64_t n = ((uint64_t)r[8] << 32) | r[7];
        uint64_t qh = n / v7;
        if (qh > 0xFFFFFFFFull) qh = 0xFFFFFFFFull;
        uint64_t rh = n - qh * v7;
        FE_UNROLL for (int t = 0; t < 2; t++) {
            const bool dec = rh <= 0xFFFFFFFFull && qh * (uint64_t)v6 > ((rh << 32) | r[6]);
            qh -= dec ? 1u : 0u;
            rh += dec ? v7 : 0u;
        }
        const uint32_t qd = (uint32_t)qh;
        // r -= qd * V
        uint64_t carry = 0;
        int64_t br = 0;
        FE_UNROLL for (int k = 0; k < 8; k++) {
            const uint64_t p = (uint64_t)qd * V.v[k] + carry;
            carry = p >> 32;
            const int64_t t = (int64_t)r[k] - (int64_t)(uint32_t)p + br;
            r[k] = (uint32_t)t;
            br = t >> 32;
        }
        const int64_t t8 = (int64_t)r[8] - (int64_t)carry + br;
        r[8] = (uint32_t)t8;
        bool neg = t8 < 0;
        uint32_t qfix = 0;
        FE_UNROLL for (int t = 0; t < 2; t++) {                       // add V back while the remainder is negative (at most once
            uint64_t c = 0;                                           // after the refinement; twice costs nothing to allow)
            uint32_t a9[9];
            FE_UNROLL for (int k = 0; k < 8; k++) {
                c += (uint64_t)r[k] + V.v[k];
                a9[k] = (uint32_t)c;
                c >>= 32;
            }
            c += (uint64_t)r[8];
            a9[8] = (uint32_t)c;
            const bool wrapped = (c >> 32) != 0;                      // the sum crossed zero: non-negative again
            FE_UNROLL for (int k = 0; k < 9; k++) r[k] = neg ? a9[k] : r[k];
            qfix += neg ? 1u : 0u;
            neg = neg && !wrapped;
        }
        q.v[j] = qd - qfix;
    }
    fe rn;
    FE_UNROLL for (int k = 0; k < 8; k++) rn.v[k] = r[k];
    *quo = q;
    *rem = fe_shr_raw(rn, s);
}
