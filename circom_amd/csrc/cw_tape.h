// cw_tape.h — schedule ("tape") row format and opcodes shared by host loader and HIP kernels.
// Must match circom_amd/hip_elements/lower.py (D_* numbering) and writers.py (.cwt layout).
#pragma once
#include <stdint.h>

enum : uint32_t {
    D_COPY = 0, D_ADD, D_SUB, D_NEG, D_MMUL, D_INV, D_IDIV, D_MOD, D_POW, D_SHL, D_SHR, D_BAND, D_BOR, D_BXOR,
    D_BNOT, D_LT, D_GT, D_LEQ, D_GEQ, D_EQ, D_NEQ, D_LAND, D_LOR, D_LNOT, D_SELECT, D_EXT, D_ASSERT_EQ,
    D_ASSERT_NZ, D_ALSO, D_BARRIER, D_NOPS
};

// operand kinds (2 bits each in row.w0: dst<<8, a<<10, b<<12); ALSO rows: bits 16-17 = number of dsts
enum : uint32_t { K_SIG = 0, K_TMP = 1, K_CONST = 2, K_PREV = 3 };
enum : uint32_t { KD_NONE = 2 };   // destination kind "no store" (value only forwarded through PREV)

struct CwRow {       // 16 bytes, read with one scalar dwordx4 load
    uint32_t w0;     // op | dk<<8 | ak<<10 | bk<<12
    uint32_t dst;
    uint32_t a;
    uint32_t b;
};

// Field parameters, passed by value as a kernel argument (lands in SGPRs).
// The device Montgomery radix is R' = 2^261 (9 limbs x 29 bits, see fp256.hip.h), NOT the reference's
// R = 2^256: which radix a residue is scaled by is private to the schedule (lower.py pre-scales constants).
#define CW_RBITS 261
struct FpParams {
    uint32_t q[8];      // modulus, little-endian 32-bit limbs
    uint32_t half[8];   // (q-1)/2 : val(x) = x - q iff x > half   (generic/fr.cpp:9)
    uint32_t r2[8];     // R'^2 mod q                                 (role of Fr_rawR2, generic/fr.cpp:14)
    uint32_t one_m[8];  // R' mod q  (1 in Montgomery form)
    uint32_t qm2[8];    // q - 2 (Fermat exponent for INV)
    uint32_t q29[9];    // modulus as 9 x 29-bit limbs
    uint32_t np29;      // -q^-1 mod 2^29
    uint32_t qbits;     // bit length of q
    uint32_t topmask;   // mask of the top limb = lboMask >> 32     (generic/fr.cpp:16)
};

#define CW_ST_ASSERT_FAILED 1u
#define CW_ST_ARITH 2u
#define CW_ST_R1CS_FAILED 4u
