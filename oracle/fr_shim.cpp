// TEST INFRASTRUCTURE (oracle): extern "C" shim over the reference's Fr_* API
// (code_producers/src/c_elements/generic/fr.hpp:32-71, C++-mangled there) so that
// tests can drive the *compiled reference field library* through ctypes and pin
// oracle/field.py + the HIP device functions against it.
//
// Values cross this boundary as canonical 32-byte little-endian integers in [0,q).
// `rep` selects which tagged representation (fr.hpp:17-21) the operand is handed to
// the reference in, so every tag-dispatch path of e.g. Fr_mul (generic/fr.cpp:559-637)
// can be exercised:  0 = auto (short if it fits int32 around 0, else long normal),
// 1 = long normal, 2 = long Montgomery, 3 = short (fails if it does not fit).
#include "fr.hpp"
#include <cstring>
#include <cstdint>
#include <gmp.h>

void Fr_square(PFrElement r, PFrElement a);   // defined in fr.cpp, missing from generic fr.hpp
void Fr_rawMMul(FrRawElement pRawResult, const FrRawElement pRawA, const FrRawElement pRawB);

static int fits_short(const uint8_t le[32], int32_t *out) {
    // value < 2^31  or  q - value <= 2^31
    uint64_t v[4];
    memcpy(v, le, 32);
    if (v[1] == 0 && v[2] == 0 && v[3] == 0 && v[0] < 0x80000000ULL) { *out = (int32_t)v[0]; return 1; }
    // d = q - v
    uint64_t d[4]; unsigned __int128 bor = 0;
    for (int i = 0; i < 4; i++) {
        unsigned __int128 t = (unsigned __int128)Fr_q.longVal[i] - v[i] - (uint64_t)bor;
        d[i] = (uint64_t)t; bor = (t >> 64) & 1;
    }
    if (bor) return 0;
    if (d[1] == 0 && d[2] == 0 && d[3] == 0 && d[0] <= 0x80000000ULL && d[0] > 0) { *out = (int32_t)(-(int64_t)d[0]); return 1; }
    return 0;
}

static int load(FrElement *e, const uint8_t le[32], int rep) {
    int32_t s;
    memset(e, 0, sizeof(*e));
    if ((rep == 0 || rep == 3) && fits_short(le, &s)) { e->type = Fr_SHORT; e->shortVal = s; return 0; }
    if (rep == 3) return -1;
    e->type = Fr_LONG;
    memcpy(e->longVal, le, 32);
    if (rep == 2) Fr_toMontgomery(e, e);
    return 0;
}

static void store(uint8_t le[32], PFrElement e) {
    FrElement t;
    Fr_toLongNormal(&t, e);
    memcpy(le, t.longVal, 32);
}

typedef void (*binop_t)(PFrElement, PFrElement, PFrElement);
typedef void (*unop_t)(PFrElement, PFrElement);

static binop_t BIN[] = {Fr_add, Fr_sub, Fr_mul, Fr_div, Fr_idiv, Fr_mod, Fr_pow, Fr_shl, Fr_shr, Fr_band,
                        Fr_bor, Fr_bxor, Fr_eq,  Fr_neq, Fr_lt,   Fr_gt,  Fr_leq, Fr_geq, Fr_land, Fr_lor};
static unop_t UN[] = {Fr_neg, Fr_bnot, Fr_lnot, Fr_inv, Fr_square};

extern "C" {

int ofr_n64(void) { return Fr_N64; }
void ofr_q(uint8_t le[32]) { memcpy(le, Fr_q.longVal, 32); }

// op index order = BIN[] above
int ofr_binop(int op, uint8_t out[32], const uint8_t a[32], int repa, const uint8_t b[32], int repb) {
    FrElement ea, eb, er;
    if (op < 0 || op >= (int)(sizeof(BIN) / sizeof(BIN[0]))) return -2;
    if (load(&ea, a, repa) || load(&eb, b, repb)) return -1;
    BIN[op](&er, &ea, &eb);
    store(out, &er);
    return 0;
}

int ofr_unop(int op, uint8_t out[32], const uint8_t a[32], int repa) {
    FrElement ea, er;
    if (op < 0 || op >= (int)(sizeof(UN) / sizeof(UN[0]))) return -2;
    if (load(&ea, a, repa)) return -1;
    UN[op](&er, &ea);
    store(out, &er);
    return 0;
}

int ofr_is_true(const uint8_t a[32], int repa) {
    FrElement ea;
    if (load(&ea, a, repa)) return -1;
    return Fr_isTrue(&ea);
}

// Fr_toInt asserts when the value does not fit; callers only pass values that fit.
int ofr_to_int(const uint8_t a[32], int repa) {
    FrElement ea;
    load(&ea, a, repa);
    return Fr_toInt(&ea);
}

int ofr_str2element(uint8_t out[32], const char *s, unsigned base) {
    FrElement e;
    Fr_str2element(&e, s, base);
    store(out, &e);
    return 0;
}

// Raw Montgomery product on 4xu64 limbs (generic/fr.cpp:110-164), for the device MMul parity test.
void ofr_raw_mmul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) { Fr_rawMMul(r, a, b); }

// Dependent-chain multiply loop: the CPU "Fp mul/s" figure of BASELINE.md section 2.
void ofr_mul_chain(uint8_t out[32], const uint8_t a[32], const uint8_t b[32], uint64_t n) {
    FrElement ea, eb;
    load(&ea, a, 2);
    load(&eb, b, 2);
    for (uint64_t i = 0; i < n; i++) Fr_mul(&ea, &ea, &eb);
    store(out, &ea);
}
}
