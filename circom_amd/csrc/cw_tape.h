// cw_tape.h — schedule ("tape") row format and opcodes shared by host loader and HIP kernels.
// Must match circom_amd/hip_elements/lower.py (D_* numbering) and writers.py (.cwt layout).
#pragma once
#include <stdint.h>

enum : uint32_t {
    D_COPY = 0, D_ADD, D_SUB, D_NEG, D_MMUL, D_INV, D_IDIV, D_MOD, D_POW, D_SHL, D_SHR, D_BAND, D_BOR, D_BXOR,
    D_BNOT, D_LT, D_GT, D_LEQ, D_GEQ, D_EQ, D_NEQ, D_LAND, D_LOR, D_LNOT, D_SELECT, D_EXT, D_ASSERT_EQ,
    D_ASSERT_NZ, D_ALSO, D_BARRIER, D_MUL2, D_MADD, D_MULC, D_MADDC, D_LINSUM, D_BIT, D_DOTC, D_CALL, D_BITS, D_NOPS
};

// row.w0 = op[0:8) | dk[8:11) | ak[11:14) | bk[14:17) | n_extra[17:29) | const flags[29:31)
#define SH_DK 8
#define SH_AK 11
#define SH_BK 14
#define SH_NX 17
#define SH_FLAG 29   // D_MULC/D_MADDC: 1 = constant is a small positive integer, 2 = small negative
// operand kinds: value-table signal slot, value-table temp slot, constant index, PREV (result of the previous
// value-producing row of the strand, forwarded in registers), LDS slot of the workgroup
enum : uint32_t { K_SIG = 0, K_TMP = 1, K_CONST = 2, K_PREV = 3, K_LDS = 4 };
enum : uint32_t { KD_NONE = 2 };   // destination kind "no store" (value only forwarded); 0/1/4 as above
// extra-destination table entries: slot number | flags
#define X_TMP 0x80000000u
#define X_LDS 0x40000000u
// D_BITS: consecutive bits of operand a, first index in field b, one row instead of one D_BIT row per bit (Num2Bits).  It has no
// destination of its own: its extra-destination entries list, bit after bit, where each bit goes - an entry with X_NEXT set
// belongs to the next bit (so a bit can have several destinations: the signal and its elided copies).  Value-table
// destinations only; slot numbers of such a schedule stay below 2^29.  D_ALSO rows do nothing (a spacer the lowering puts
// between a D_BITS row and a row that reads one of its bits as a prefetched operand).
#define X_NEXT 0x20000000u
#define X_NEXT_DEV (1ull << 62)
// ... and on the device: this destination is a SIGNAL slot, whose upper 16 bytes stay zero for the life of the batch (the table
// is cleared when the batch is created, a signal's only writer is this row, a bit is 0 or 1): only the lower half is stored -
// the row's cost is its vector-memory instructions (the CU issues one per ~25 clocks: 2.2 M destinations per instance group
// of the ECDSA verifier)
#define X_LO_DEV (1ull << 61)

struct CwRow {       // 16 bytes, read with one scalar dwordx4 load
    uint32_t w0;     // see SH_* above; BARRIER rows: dst = 1 -> also drain global stores
    uint32_t dst;
    uint32_t a;
    uint32_t b;
};

// Device row: a CwRow resolved for one batch (cw_batch_create): operands as byte offsets, so the kernel forms an
// address with one 64-bit scalar add.  SIG/TMP: offset of the lo half of the slot inside the value table
// (hi half = +Bp*16); CONST: offset into the constant table; LDS: byte offset of the slot inside the
// workgroup's LDS block (slot * 2048).  Extra-destination entries are 64-bit: value-table byte offset, or
// bit 63 + LDS byte offset.  Streams are padded with 3 D_NOP rows.
struct CwDRow {      // 32 bytes, one scalar dwordx8 load
    uint32_t w0;     // as CwRow.w0
    uint32_t aux;    // BARRIER: 1 = FULL
    uint64_t dst_off, a_off, b_off;
};
#define D_NOP 255u

// D_CALL: a circom function with run-time control flow (frontend/rtcode.py; reference: call_bucket.rs:466-533,
// loop_bucket.rs:76-91, branch_bucket.rs:100-122).  row.a = function id, operand b = first of its registers
// (consecutive pinned temp slots).  Bytecode: 4 x u32 per instruction {opcode, dst, a, b}; opcode = D_* for arithmetic on
// canonical values (D_MUL2 = product) or F_*; operands = register number, or FN_CONST | constant index;
// F_LDX / F_STX: b = index register | array length << 16 (address through Fr_toInt, generic/fr.cpp:1146-1170).
enum : uint32_t { F_JZ = 100, F_JMP = 101, F_LDX = 102, F_STX = 103, F_RET = 104, F_DIV = 105 };
#define FN_CONST 0x80000000u
#define CW_CALL_STEP_LIMIT (1u << 24)   // instructions per call and lane before the instance is flagged (runaway loop)

// Field parameters, passed by value as a kernel argument (lands in SGPRs).
// The device Montgomery radix is R' = 2^261 (9 limbs x 29 bits, see fp256.hip.h), NOT the reference's
// R = 2^256: which radix a residue is scaled by is private to the schedule (lower.py pre-scales constants).
#define CW_RBITS 261
struct FpParams {
    uint32_t q[8];      // modulus, little-endian 32-bit limbs
    uint32_t half[8];   // (q-1)/2 : val(x) = x - q iff x > half   (generic/fr.cpp:9)
    uint32_t r2[8];     // R'^2 mod q                                 (role of Fr_rawR2, generic/fr.cpp:14)
    uint32_t one_m[8];  // R' mod q  (1 in Montgomery form)
    uint32_t q29[9];    // modulus as 9 x 29-bit limbs
    uint32_t r2_29[9];  // R'^2 mod q as 9 x 29-bit limbs (scalar operands of the second product of a canonical multiply)
    uint32_t np29;      // -q^-1 mod 2^29
    uint32_t qbits;     // bit length of q
    uint32_t topmask;   // mask of the top limb = lboMask >> 32     (generic/fr.cpp:16)
};

#define CW_ST_ASSERT_FAILED 1u
#define CW_ST_ARITH 2u
#define CW_ST_R1CS_FAILED 4u
