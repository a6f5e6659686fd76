// cw_host.cpp — host side of the C ABI (include/circom_amd.h): circuit loading, input ingest with the
// reference's semantics, kernel sequencing, result egress.  Mirrors the reference's C++ runtime
// (code_producers/src/c_elements/common/{main.cpp,calcwit.cpp}); each function cites what it replaces.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/circom_amd.h"
#include "cw_kernels.h"
#include "cw_r1cs_plan.h"
#include "cw_bits_host.h"

typedef unsigned __int128 u128;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return fail(CW_EDEVICE, std::string(#x ": ") + hipGetErrorString(e_)); \
    } while (0)

#define NEED_DEVICE(b) \
    if ((b)->device < 0) return fail(CW_EDEVICE, "host-only batch: no GPU attached (the hot path has no CPU fallback)")
// entry points the 64-bit runtime (--prime goldilocks, cw64.hip) does not serve yet
#define NOT_FOR_64(b, what) \
    if ((b)->c->is64) return fail(CW_ESTATE, what " is not available for circuits of the 64-bit runtime (--prime goldilocks)")

extern "C" const char *cw_last_error(void) { return g_err.c_str(); }
extern "C" const char *cw_version(void) { return "circom_amd 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------------------------------------
// 256-bit host integers (4 x u64, little endian).  Only what ingest/loader need: compare, add, sub,
// doubling mod q, small-multiplier accumulate.  (The reference uses GMP here: generic/fr.cpp:2766-2811.)
// ---------------------------------------------------------------------------------------------------------
struct U256 {
    uint64_t w[4];
};
static U256 u256_zero() { return U256{{0, 0, 0, 0}}; }
static int u256_cmp(const U256 &a, const U256 &b) {
    for (int i = 3; i >= 0; i--)
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
    return 0;
}
static bool u256_is_zero(const U256 &a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) == 0; }
static uint64_t u256_add(U256 &r, const U256 &a, const U256 &b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.w[i] + b.w[i];
        r.w[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
static uint64_t u256_sub(U256 &r, const U256 &a, const U256 &b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a.w[i] - b.w[i] - br;
        r.w[i] = (uint64_t)t;
        br = (uint64_t)(t >> 64) & 1;
    }
    return br;
}
static U256 addmod(const U256 &a, const U256 &b, const U256 &q) {
    U256 r;
    uint64_t c = u256_add(r, a, b);
    if (c || u256_cmp(r, q) >= 0) u256_sub(r, r, q);
    return r;
}
static U256 submod(const U256 &a, const U256 &b, const U256 &q) {
    U256 r;
    if (u256_sub(r, a, b)) u256_add(r, r, q);
    return r;
}
// (a * m + d) mod q for small m (<= 16) and d < m, a < q
static U256 mulsmall_add_mod(const U256 &a, uint32_t m, uint32_t d, const U256 &q) {
    uint64_t t[5];
    u128 c = d;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.w[i] * m;
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    t[4] = (uint64_t)c;
    // reduce: subtract q while t >= q (at most m+1 times)
    for (;;) {
        bool ge = t[4] != 0;
        if (!ge) {
            U256 x{{t[0], t[1], t[2], t[3]}};
            ge = u256_cmp(x, q) >= 0;
        }
        if (!ge) break;
        uint64_t br = 0;
        for (int i = 0; i < 4; i++) {
            u128 s = (u128)t[i] - q.w[i] - br;
            t[i] = (uint64_t)s;
            br = (uint64_t)(s >> 64) & 1;
        }
        t[4] -= br;
    }
    return U256{{t[0], t[1], t[2], t[3]}};
}
// a * 2^k mod q by repeated doubling
static U256 shrmod(U256 a, unsigned k, const U256 &q) {      // a / 2^k mod q (q odd)
    for (unsigned i = 0; i < k; i++) {
        uint64_t carry = 0;
        if (a.w[0] & 1) {                                        // a + q is even; the sum may need a 257th bit
            unsigned __int128 t = 0;
            for (int j = 0; j < 4; j++) {
                t += (unsigned __int128)a.w[j] + q.w[j];
                a.w[j] = (uint64_t)t;
                t >>= 64;
            }
            carry = (uint64_t)t;
        }
        for (int j = 0; j < 3; j++) a.w[j] = (a.w[j] >> 1) | (a.w[j + 1] << 63);
        a.w[3] = (a.w[3] >> 1) | (carry << 63);
    }
    return a;
}
static U256 shlmod(U256 a, unsigned k, const U256 &q) {
    for (unsigned i = 0; i < k; i++) a = addmod(a, a, q);
    return a;
}
static unsigned u256_bits(const U256 &a) {
    for (int i = 3; i >= 0; i--)
        if (a.w[i]) return 64 * i + 64 - __builtin_clzll(a.w[i]);
    return 0;
}

// Fr_str2element (generic/fr.cpp:2805-2811): int(s, base) floor-mod q; leading '-' allowed.
static bool str_to_fe(const char *s, size_t n, unsigned base, const U256 &q, U256 *out) {
    bool neg = false;
    size_t i = 0;
    if (n > 0 && (s[0] == '-' || s[0] == '+')) {
        neg = s[0] == '-';
        i = 1;
    }
    if (i >= n) return false;
    U256 v = u256_zero();
    for (; i < n; i++) {
        char c = s[i];
        unsigned d;
        if (c >= '0' && c <= '9') d = c - '0';
        else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
        else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
        else return false;
        if (d >= base) return false;
        v = mulsmall_add_mod(v, base, d, q);
    }
    if (neg && !u256_is_zero(v)) u256_sub(v, q, v);
    *out = v;
    return true;
}

// The device field code is written for the reference's 4-limb primes (bn128, bls12381, bls12377, grumpkin, pallas,
// vesta, secq256r1: 253..256 bits, program_structure/src/utils/constants.rs:3-13).  The 64-bit Goldilocks runtime
// (c_elements/common64, goldilocks/fr.hpp) is a different code path of the reference and is out of scope here.
static bool prime_supported(const U256 &q) {
    unsigned bits = u256_bits(q);
    return bits >= 225 && bits <= 256 && (q.w[0] & 1);
}

static FpParams make_params(const U256 &q) {
    FpParams P;
    memset(&P, 0, sizeof(P));
    memcpy(P.q, q.w, 32);
    U256 half = q;   // (q-1)/2 == q>>1 for odd q
    for (int i = 0; i < 4; i++) half.w[i] = (q.w[i] >> 1) | (i < 3 ? (q.w[i + 1] << 63) : 0);
    memcpy(P.half, half.w, 32);
    U256 one{{1, 0, 0, 0}};
    U256 r1 = shlmod(one, CW_RBITS, q);      // R' mod q
    U256 r2 = shlmod(r1, CW_RBITS, q);       // R'^2 mod q
    memcpy(P.one_m, r1.w, 32);
    memcpy(P.r2, r2.w, 32);
    // np29 = -q^-1 mod 2^29 by Newton iteration
    uint32_t q0 = (uint32_t)q.w[0], inv = 1;
    for (int i = 0; i < 5; i++) inv *= 2 - q0 * inv;
    P.np29 = (uint32_t)(0u - inv) & 0x1FFFFFFFu;
    for (int k = 0; k < 9; k++) {
        unsigned bit = 29 * k, w = bit / 64, sh = bit % 64;
        uint64_t v = q.w[w] >> sh;
        if (sh > 35 && w + 1 < 4) v |= q.w[w + 1] << (64 - sh);
        P.q29[k] = (uint32_t)(v & 0x1FFFFFFFu);
        uint64_t v2 = r2.w[w] >> sh;
        if (sh > 35 && w + 1 < 4) v2 |= r2.w[w + 1] << (64 - sh);
        P.r2_29[k] = (uint32_t)(v2 & 0x1FFFFFFFu);
    }
    P.qbits = u256_bits(q);
    unsigned topbits = P.qbits - 224;             // bits used in the top 32-bit limb
    P.topmask = topbits >= 32 ? 0xFFFFFFFFu : ((1u << topbits) - 1);
    return P;
}

// ---------------------------------------------------------------------------------------------------------
// circuit
// ---------------------------------------------------------------------------------------------------------
struct HashEntry {
    uint64_t hash, signalid, signalsize;   // HashSignalInfo, circom.hpp:17-21
};

struct Variant {               // one lowering of the schedule for a given strand count
    uint32_t n_strands = 1, n_tslots = 0, n_lds = 0;
    uint32_t prio_mask = 0;        // strands whose share of the work is within 20 % of the heaviest one (s_setprio)
    double work = 0, crit = 0;     // circuits with functions: total cost of the rows / sum over barrier epochs of the busiest strand's
    uint32_t n_active = 1;         // strands that carry work (a 3-lane circuit leaves 13 of 16 strands with barriers only)
    bool wide_linsum = false;      // schedule dominated by long small-coefficient sums -> 4 operand loads in flight
    // kind 1 = pipelined single-wave schedule (hip_elements/pipe.py): prows = 8 words per row, extras = the load lists
    uint32_t kind = 0, nb = 0, nld = 0;
    std::vector<uint32_t> prows;
    std::vector<CwRow> rows;
    std::vector<uint32_t> stream_off, extras, extra_off, term_off, terms;   // terms: 4 x u32 each
    std::vector<uint32_t> seq_off, seqs;    // per strand: flat-operation index of every row that can fail a check, in row order
};

// the row stream of one strand variant as EMITTED gfx950 code (hip_elements/fpjit.py): straight-line code that loads the
// operands of every row, calls the operator's body and stores the result - no descriptors, no dispatch, exact wait counts
struct FpJit {
    uint32_t n_strands = 1, lds_bytes = 0, scratch_bytes = 0, n_vgpr = 0;
    std::vector<uint32_t> covered;                                   // bitmap over the .r1cs rows: checked by the code itself
    uint32_t n_covered = 0;
    std::vector<uint8_t> code;                                       // ELF code object (hipModuleLoadData)
    std::map<int, std::pair<hipModule_t, hipFunction_t>> mod;        // device -> loaded module
    uint32_t r1cs_crc = 0, r1cs_len = 0;                             // the .r1cs constraint section `covered` refers to (0, 0: unknown)
};
constexpr const char *FPJIT_KERNEL = "cw_fp_jit";

// CRC-32 (IEEE, zlib's) of the constraint section of a .r1cs: the identity of the constraint system emitted checks were built
// from (hip_elements/writers.py write_r1cs returns it, write_tape stores it)
static uint32_t crc32_ieee(const uint8_t *p, size_t n) {
    static uint32_t tab[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            tab[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = tab[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

struct cw_circuit {
    U256 q;
    FpParams P;
    uint32_t n_signals = 0, n_witness = 0, n_consts = 0, input_start = 0, n_inputs = 0, n_pub_in = 0;
    uint64_t n_rows = 0, n_mmul = 0;
    bool need_full = false;
    bool mont = false;                     // the value table holds Montgomery forms x R' (lower.py pass A6)
    std::vector<Variant> variants;
    std::vector<FpJit> fpjit;              // emitted code of strand variants (at most one per strand count)
    uint32_t n_dat_consts = 0xFFFFFFFFu, n_io_templates = 0;   // sections of the .dat (0xFFFFFFFF: constants count unknown)
    struct IoDef { uint32_t offset = 0, size = 0, bus_id = 0; std::vector<uint32_t> lengths; };
    std::vector<std::vector<IoDef>> bus_map;           // the .dat's bus-field map: per bus instance its fields (load_dat)
    struct IoTemplate { uint32_t id = 0; std::vector<IoDef> defs; };
    std::vector<IoTemplate> io_map;        // TemplateInstanceIOMap read from the .dat (Mixed component clusters)
    std::vector<uint32_t> consts;          // n_consts * 8
    std::vector<uint32_t> lconsts;         // n_lconsts * 12 (29-bit limbs)
    std::vector<uint32_t> w2s;
    std::vector<HashEntry> hashmap;
    std::map<std::string, std::pair<uint32_t, uint32_t>> input_names;   // name -> (start, size)
    // r1cs (CSR)
    uint32_t n_constraints = 0;
    std::vector<uint32_t> r_ptr, r_slot, r_coef, r_ctab, r_orig;
    std::vector<uint8_t> r_bool;           // per row in processing order: the row is b * (b -+ 1) = 0 (cw_r1cs_plan.h T_BOOL)
    // circom functions with run-time control flow (D_CALL): concatenated bytecode + per function {first ins, n ins, n regs}
    std::vector<uint32_t> fn_code, fn_tab;
    // bit-plane program (cw_bits.hip) when every signal of the circuit is provably boolean for 0/1 inputs
    bool has_bits = false;
    cwbits::Program bits;
    // the same gate network as EMITTED gfx950 code (hip_elements/bitjit.py) for large batches: a code object, its own slot map
    // the 64-bit runtime (--prime goldilocks, cw64.hip): the flat witness program, one 32-byte row per operation
    bool is64 = false;
    std::vector<uint32_t> rows64;          // n_rows * 8
    std::vector<uint64_t> consts64;
    uint32_t n_slots64 = 0;
    std::vector<uint32_t> r1_terms64;      // R1CS terms {slot, part | end << 2, coefficient lo, hi}
    std::vector<uint32_t> r1_chunks64;     // chunks of consecutive constraints {first term, terms, first row, 0}: one workgroup each
    bool has_jit = false;
    cwbits::JitProgram jit;
    std::map<int, std::pair<hipModule_t, hipFunction_t>> jit_mod;   // device -> loaded module
    std::map<int, std::pair<hipModule_t, hipFunction_t>> jit_audit_mod;   // ... of the audit program (loaded on first use)
    std::vector<uint32_t> r_cc, r_cctab;   // per term: id of its canonical coefficient in r_cctab (8 words each)
    // log(...) statements (LogBucket): the LAST n_logv of n_signals are hidden signals holding their arguments
    uint32_t n_logv = 0;
    struct LogItem { bool is_value = false; uint32_t value = 0; std::string text; };
    struct LogStmt { uint32_t at = 0; std::vector<LogItem> items; };      // at = flat operation that ends the statement
    std::vector<LogStmt> logs;
    // batches alive on this circuit (side batches included): their device images of the witness list (d_w2s, d_wslot,
    // d_gather) are sized once at creation, so the list may only change while this is zero (cw_set_witness_list)
    std::atomic<int> live_batches{0};
    bool r1cs_matches_code = true;         // the loaded .r1cs is the constraint system the emitted checks were built from (load_r1cs)
};

static uint64_t fnv1a(const char *s, size_t n) {   // calcwit.cpp:17-24
    uint64_t h = 0xCBF29CE484222325ULL;
    for (size_t i = 0; i < n; i++) {
        h ^= (uint64_t)(unsigned char)s[i];
        h *= 0x100000001B3ULL;
    }
    return h;
}

static bool read_file(const char *path, std::vector<uint8_t> &buf) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize((size_t)n);
    size_t got = n ? fread(buf.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

// Every index a schedule variant carries is checked once at load time, so that a damaged or hostile file is rejected
// instead of indexing out of bounds later (here, in the per-batch row resolution, or on the device).
static const char *validate_variant(const Variant &v, uint32_t n_signals, uint32_t n_consts, uint32_t n_lconsts,
                                    const std::vector<uint32_t> &fn_regs) {
    const size_t nrows = v.rows.size(), nextras = v.extras.size(), nterms = v.terms.size() / 4;
    auto mono = [&](const std::vector<uint32_t> &o, size_t limit) {
        if (o.size() != v.n_strands + 1 || o[0] != 0) return false;
        for (size_t i = 0; i + 1 < o.size(); i++)
            if (o[i] > o[i + 1]) return false;
        return (size_t)o.back() <= limit;
    };
    if (!mono(v.stream_off, nrows) || !mono(v.extra_off, nextras) || !mono(v.term_off, nterms)) return "offset tables are not monotone";
    auto operand_ok = [&](uint32_t kind, uint32_t idx) {
        switch (kind) {
        case K_SIG: return idx < n_signals;
        case K_TMP: return idx < v.n_tslots;
        case K_CONST: return idx < n_consts;
        case K_PREV: return true;
        case K_LDS: return idx < v.n_lds;
        default: return false;
        }
    };
    for (uint32_t st = 0; st < v.n_strands; st++) {
        size_t xp = v.extra_off[st], tp = v.term_off[st];
        // the bits a D_BITS row stores are not there yet when the row right behind it requests its operands (one row ahead):
        // the lowering puts a spacer row in between (lower.py, "spacer"); a tape without it would read stale slots silently
        size_t bits_lo = 0, bits_hi = 0;                      // extras of the D_BITS row directly in front of this one
        for (size_t r = v.stream_off[st]; r < v.stream_off[st + 1]; r++) {
            const CwRow &row = v.rows[r];
            const uint32_t op = row.w0 & 0xFF, dk = (row.w0 >> SH_DK) & 7, ak = (row.w0 >> SH_AK) & 7, bk = (row.w0 >> SH_BK) & 7;
            const uint32_t nx = (row.w0 >> SH_NX) & 0xFFF;
            if (op >= D_NOPS) return "unknown opcode";
            if (op == D_BARRIER) { bits_lo = bits_hi = 0; continue; }
            if (bits_hi > bits_lo && op != D_LINSUM && op != D_DOTC && op != D_CALL) {
                auto stored_by_prev = [&](uint32_t kind, uint32_t idx) {
                    if (kind != K_SIG && kind != K_TMP) return false;
                    for (size_t e = bits_lo; e < bits_hi && e < nextras; e++) {
                        const uint32_t x = v.extras[e] & ~X_NEXT;
                        if ((x & X_TMP) ? (kind == K_TMP && (x & 0x1FFFFFFFu) == idx) : (kind == K_SIG && x == idx)) return true;
                    }
                    return false;
                };
                if (stored_by_prev(ak, row.a) || (op != D_BIT && op != D_BITS && stored_by_prev(bk, row.b)))
                    return "a row reads a bit of the bit-field row directly in front of it (spacer row missing)";
            }
            bits_lo = bits_hi = 0;
            if (op == D_BITS) { bits_lo = xp; bits_hi = xp + nx; }
            if (dk == K_SIG ? row.dst >= n_signals : dk == K_TMP ? row.dst >= v.n_tslots : dk == K_LDS ? row.dst >= v.n_lds : dk != KD_NONE)
                return "destination out of range";
            if (op == D_LINSUM || op == D_DOTC) {
                if (tp + row.a > v.term_off[st + 1]) return "term list overruns its strand";
                for (uint32_t t = 0; t < row.a; t++) {
                    const uint32_t *tm = &v.terms[(tp + t) * 4];
                    const uint32_t tk = tm[0] & 7;                    // the kinds term_load handles
                    if (tk != K_SIG && tk != K_TMP && tk != K_PREV && tk != K_LDS) return "term operand kind";
                    if (!operand_ok(tk, tm[1])) return "term operand out of range";
                    if (op == D_DOTC && tm[2] >= n_lconsts) return "term constant out of range";
                }
                tp += row.a;
                // c0 is a constant or absent (kind 0, index 0): the kernel's prefetch dereferences whatever is encoded
                if (bk == K_CONST ? row.b >= n_consts : (bk != 0 || row.b != 0)) return "constant out of range";
            } else if (op == D_BIT) {
                if (!operand_ok(ak, row.a)) return "operand out of range";
            } else if (op == D_BITS) {
                // b = first bit; the destinations are the extra entries (X_NEXT = next bit), value-table slots below 2^29
                if (!operand_ok(ak, row.a) || dk != KD_NONE || row.b >= 256 || nx == 0) return "bit-field row malformed";
                if (n_signals >= X_NEXT || v.n_tslots >= X_NEXT) return "bit-field rows need slot numbers below 2^29";
                uint32_t last = row.b;
                for (uint32_t e = 0; e < nx && xp + e < nextras; e++) last += (v.extras[xp + e] & X_NEXT) ? 1u : 0u;
                if (last >= 256) return "bit-field row runs past bit 255";
            } else if (op == D_CALL) {
                // a = function id; b = first register: the whole window must lie inside the temp slots
                if (row.a >= fn_regs.size() || bk != K_TMP || (uint64_t)row.b + fn_regs[row.a] > v.n_tslots) return "function call out of range";
            } else {
                if (!operand_ok(ak, row.a)) return "operand out of range";
                const bool pair = (op == D_MULC || op == D_MADDC);
                if (bk == K_CONST ? (uint64_t)row.b + (pair ? 1 : 0) >= n_consts : !operand_ok(bk, row.b)) return "operand out of range";
            }
            if (xp + nx > v.extra_off[st + 1]) return "extra destinations overrun their strand";
            for (uint32_t e = 0; e < nx; e++) {
                uint32_t x = v.extras[xp + e];
                if (op == D_BITS) {
                    if (x & X_LDS) return "bit-field destinations live in the value table";
                    x &= ~X_NEXT;
                }
                if (x & X_TMP ? (x & 0x3FFFFFFFu) >= v.n_tslots : x & X_LDS ? (x & 0x3FFFFFFFu) >= v.n_lds : x >= n_signals)
                    return "extra destination out of range";
            }
            xp += nx;
        }
        if (xp != v.extra_off[st + 1] || tp != v.term_off[st + 1]) return "strand tables do not add up";
        size_t can_fail = 0;
        for (size_t r = v.stream_off[st]; r < v.stream_off[st + 1]; r++) {
            const uint32_t op = v.rows[r].w0 & 0xFF;
            can_fail += (op == D_ASSERT_EQ || op == D_ASSERT_NZ || op == D_IDIV || op == D_MOD || op == D_CALL);
        }
        if (v.seq_off.size() != v.n_strands + 1 || v.seq_off[st] > v.seq_off[st + 1] ||
            v.seq_off[st + 1] - v.seq_off[st] != can_fail)
            return "check-index table does not match the rows that can fail";
    }
    return nullptr;
}

// pipelined variant: every index the kernel dereferences (LDS entries, value-table slots, constants, term tables) and the
// shape the kernel's fixed wait relies on (whole batches, NLD loads per batch, two store targets per row)
static const char *validate_pipe_variant(const Variant &v, uint32_t n_signals, uint32_t n_consts, uint32_t n_lconsts) {
    const uint32_t nb = v.nb, nld = v.nld, rr = 2 * nb, n_ent = rr + 2 * nld;
    if (!((nb == 8 && (nld == 8 || nld == 4)) || (nb == 4 && nld == 4))) return "unsupported batch shape";
    if (v.n_strands != 1 || v.n_lds != n_ent) return "pipelined variants are single-strand with 2*NB + 2*NLD LDS entries";
    const size_t nrows = v.prows.size() / 8, nterms = v.terms.size() / 4;
    if (nrows == 0 || nrows % nb || nrows > (1u << 30)) return "rows do not form whole batches";
    if (v.extras.size() != (nrows / nb + 2) * (size_t)nld) return "load lists do not match the batches";
    if (nterms < 4) return "term table lacks its padding";
    auto target_ok = [&](uint32_t t) {
        if (t == 0xFFFFFFFFu) return true;
        if (t & 0x40000000u) return false;
        return (t & X_TMP) ? (t & 0x3FFFFFFFu) < v.n_tslots : t < n_signals;
    };
    for (uint32_t lw : v.extras) {
        if (lw == 0xFFFFFFFFu) continue;
        if (lw & 0x40000000u) {
            if ((lw & X_TMP) || (lw & 0x3FFFFFFFu) >= n_consts) return "load list: constant out of range";
        } else if (!target_ok(lw)) return "load list: slot out of range";
    }
    size_t tp = 0;
    for (size_t r = 0; r < nrows; r++) {
        const uint32_t *w = &v.prows[r * 8];
        const uint32_t op = w[0] & 0xFF, ak = (w[0] >> 8) & 7, bk = (w[0] >> 11) & 7;
        const uint32_t ae = w[2] & 0xFF, be = (w[2] >> 8) & 0xFF, de = (w[2] >> 16) & 0xFF;
        const uint32_t half = (uint32_t)(r / nb) & 1u;
        auto entry_ok = [&](uint32_t e) { return e < rr || (e < n_ent && (e - rr) / nld == half); };
        if (w[0] & 0x1FFFC000u) return "row word 0 has unknown bits";
        if (!(op < D_NOPS || op == D_NOP) || op == D_BARRIER || op == D_CALL || op == D_ALSO) return "opcode not allowed in a pipelined schedule";
        if ((ak != 0 && ak != K_PREV && ak != K_LDS) || (bk != 0 && bk != K_PREV && bk != K_LDS)) return "operand kind";
        if ((ak == K_LDS && !entry_ok(ae)) || (bk == K_LDS && !entry_ok(be))) return "LDS entry out of range";
        if ((ak != K_LDS && ae >= n_ent) || (bk != K_LDS && be >= n_ent)) return "LDS entry out of range";   // still prefetched
        const bool value = !(op == D_NOP || op == D_ASSERT_EQ || op == D_ASSERT_NZ || op == D_SELECT);
        if (value ? !(de == 0xFF || de == r % rr) : (de != 0xFF || w[3] != 0xFFFFFFFFu || w[4] != 0xFFFFFFFFu))
            return "result entry / store targets";
        if (!target_ok(w[3]) || !target_ok(w[4])) return "store target out of range";
        if (op == D_LINSUM || op == D_DOTC) {
            if (w[1] > nterms - 4 - tp) return "term list overruns the table";
            for (uint32_t t = 0; t < w[1]; t++) {
                const uint32_t *tm = &v.terms[(tp + t) * 4];
                const uint32_t tk = tm[0] & 7;
                if (tk != K_PREV && tk != K_LDS) return "term operand kind";
                if (tm[0] & 0x7FFFFFF8u) return "term word has unknown bits";
                if (tk == K_LDS ? !entry_ok(tm[1]) : tm[1] != 0) return "term entry out of range";
                if (op == D_DOTC && tm[2] >= n_lconsts) return "term constant out of range";
            }
            tp += w[1];
        }
    }
    if (tp + 4 != nterms) return "term table not consumed exactly";
    return nullptr;
}

// ---- the 64-bit runtime's tape ("CW64", hip_elements/lower64.py) ------------------------------------------------------------
//   "CW64" | u32 version = 1 | u64 prime | 12 x u32: n_signals, n_witness, n_consts, input_start, n_inputs, n_input_names,
//   hashmap_size, n_public_inputs, n_slots, n_rows, io-map templates in the .dat, 0 | consts n_consts x u64 | witness2signal
//   n_witness x u32 | input names { u32 len | bytes | u32 start | u32 size } | rows n_rows x 8 x u32 (cw64.hip)
static const uint64_t GOLDILOCKS = 0xFFFFFFFF00000001ull;
static int load_tape64(cw_circuit *c, const std::vector<uint8_t> &b) {
    if (b.size() < 16 + 48) return fail(CW_EIO, "tape file truncated");
    uint32_t ver;
    memcpy(&ver, b.data() + 4, 4);
    uint64_t prime;
    memcpy(&prime, b.data() + 8, 8);
    if (ver != 1) return fail(CW_EIO, "unsupported 64-bit tape version");
    if (prime != GOLDILOCKS) return fail(CW_EIO, "the 64-bit runtime serves the Goldilocks prime only");
    c->is64 = true;
    c->q = U256{{prime, 0, 0, 0}};
    uint32_t m[12];
    memcpy(m, b.data() + 16, 48);
    size_t off = 64;
    c->n_signals = m[0];
    c->n_witness = m[1];
    c->n_consts = m[2];
    c->input_start = m[3];
    c->n_inputs = m[4];
    const uint32_t n_names = m[5], hsize = m[6];
    c->n_pub_in = m[7];
    c->n_slots64 = m[8];
    const uint32_t n_rows = m[9];
    c->n_dat_consts = 0;                                       // the 64-bit .dat carries no constants (c_code_generator.rs:838-841)
    c->n_io_templates = m[10];
    if (c->n_signals == 0 || c->n_signals >= (1u << 26) || c->n_slots64 < c->n_signals || c->n_slots64 >= (1u << 28) || m[11] ||
        c->input_start == 0 || (uint64_t)c->input_start + c->n_inputs > c->n_signals || c->n_pub_in > c->n_inputs || c->n_witness == 0 ||
        c->n_witness > c->n_signals || hsize < 256 || (hsize & (hsize - 1)) || n_names > hsize || n_names > c->n_inputs + 1u ||
        n_rows > (1u << 26) || c->n_consts > (1u << 26) || c->n_io_templates > (1u << 20))
        return fail(CW_EIO, "64-bit tape header: inconsistent circuit shape");
    if (b.size() - off < (size_t)c->n_consts * 8 + (size_t)c->n_witness * 4) return fail(CW_EIO, "tape file truncated");
    c->consts64.resize(c->n_consts);
    memcpy(c->consts64.data(), b.data() + off, (size_t)c->n_consts * 8);
    off += (size_t)c->n_consts * 8;
    for (uint64_t v : c->consts64)
        if (v >= prime) return fail(CW_EIO, "64-bit tape: constant is not a canonical residue");
    c->w2s.resize(c->n_witness);
    memcpy(c->w2s.data(), b.data() + off, (size_t)c->n_witness * 4);
    off += (size_t)c->n_witness * 4;
    for (uint32_t v : c->w2s)
        if (v >= c->n_signals) return fail(CW_EIO, "tape witness list refers to a signal out of range");
    for (uint32_t i = 0; i < n_names; i++) {
        if (off + 4 > b.size()) return fail(CW_EIO, "tape input names truncated");
        uint32_t len;
        memcpy(&len, b.data() + off, 4);
        off += 4;
        if (len > 4096 || off + len + 8 > b.size()) return fail(CW_EIO, "tape input names truncated");
        std::string nm((const char *)b.data() + off, len);
        off += len;
        uint32_t ss[2];
        memcpy(ss, b.data() + off, 8);
        off += 8;
        if (ss[0] < c->input_start || (uint64_t)ss[0] + ss[1] > (uint64_t)c->input_start + c->n_inputs) return fail(CW_EIO, "tape input name outside the main inputs");
        if (!c->input_names.emplace(nm, std::make_pair(ss[0], ss[1])).second) return fail(CW_EIO, "tape input name appears twice");
    }
    if (b.size() - off != (size_t)n_rows * 32) return fail(CW_EIO, "64-bit tape: row section size");
    c->rows64.resize((size_t)n_rows * 8);
    memcpy(c->rows64.data(), b.data() + off, (size_t)n_rows * 32);
    for (uint32_t r = 0; r < n_rows; r++) {                   // every operand is checked once, here
        const uint32_t *w = &c->rows64[(size_t)r * 8];
        const uint32_t op = w[0] & 0xFF, dk = (w[0] >> 8) & 3, ks[3] = {(w[0] >> 10) & 3, (w[0] >> 12) & 3, (w[0] >> 14) & 3};
        const uint32_t vs[3] = {w[2], w[3], w[4]};
        bool ok = op <= 26 && !(w[0] >> 16) && (dk == 3 || (dk == 0 && w[1] > 0 && w[1] < c->n_slots64));
        for (int j = 0; j < 3; j++) ok = ok && (ks[j] == 3 || (ks[j] == 0 && vs[j] < c->n_slots64) || (ks[j] == 2 && vs[j] < c->n_consts));
        if (!ok) return fail(CW_EIO, "64-bit tape: bad row");
    }
    c->n_rows = n_rows;
    // hash map as generate_hash_map builds it (replaced by the .dat's if one is given)
    c->hashmap.assign(hsize, HashEntry{0, 0, 0});
    std::vector<std::pair<uint32_t, std::string>> order;
    for (auto &kv : c->input_names) order.push_back({kv.second.first, kv.first});
    std::sort(order.begin(), order.end());
    for (auto &o : order) {
        const uint64_t h = fnv1a(o.second.data(), o.second.size());
        size_t pos = (size_t)(h % hsize);
        while (c->hashmap[pos].signalid != 0) pos = (pos + 1) % hsize;
        c->hashmap[pos] = HashEntry{h, o.first, c->input_names[o.second].second};
    }
    return CW_OK;
}
// .r1cs with 8-byte coefficients (field size 8: r1cs_writer.rs writes the prime's own byte length)
static int load_r1cs64(cw_circuit *c, const char *path) {
    std::vector<uint8_t> b;
    if (!read_file(path, b)) return fail(CW_EIO, std::string(".r1cs file not found: ") + path);
    if (b.size() < 12 || memcmp(b.data(), "r1cs", 4)) return fail(CW_EIO, "bad r1cs magic");
    uint32_t nsec;
    memcpy(&nsec, b.data() + 8, 4);
    size_t off = 12;
    const uint8_t *sec[4] = {0};
    uint64_t seclen[4] = {0};
    for (uint32_t s = 0; s < nsec; s++) {
        if (off + 12 > b.size()) return fail(CW_EIO, "r1cs truncated");
        uint32_t typ;
        uint64_t len;
        memcpy(&typ, b.data() + off, 4);
        memcpy(&len, b.data() + off + 4, 8);
        off += 12;
        if (len > b.size() - off) return fail(CW_EIO, "r1cs section truncated");
        if (typ < 4) {
            sec[typ] = b.data() + off;
            seclen[typ] = len;
        }
        off += len;
    }
    if (!sec[1] || !sec[2] || seclen[1] < 4 + 8 + 16 + 8 + 4) return fail(CW_EIO, "r1cs misses header or constraints section");
    uint32_t fs;
    memcpy(&fs, sec[1], 4);
    uint64_t prime;
    memcpy(&prime, sec[1] + 4, 8);
    if (fs != 8 || prime != GOLDILOCKS) return fail(CW_EIO, "r1cs field differs from the tape's 64-bit prime");
    uint32_t n_wires, n_cons;
    memcpy(&n_wires, sec[1] + 12, 4);
    memcpy(&n_cons, sec[1] + 12 + 16 + 8, 4);
    if (n_wires != c->n_witness) return fail(CW_EIO, "r1cs wire count differs from the witness size");
    const uint8_t *p = sec[2], *end = sec[2] + seclen[2];
    c->r1_terms64.clear();
    c->r1_chunks64.clear();
    const uint32_t CHUNK_TERMS = 48;                                  // a chunk closes at the first constraint boundary past this
    uint32_t chunk_first = 0, chunk_row = 0;
    for (uint32_t k = 0; k < n_cons; k++) {
        const size_t first = c->r1_terms64.size();
        for (uint32_t part = 0; part < 3; part++) {
            if (p + 4 > end) return fail(CW_EIO, "r1cs constraints truncated");
            uint32_t nnz;
            memcpy(&nnz, p, 4);
            p += 4;
            if ((size_t)(end - p) < (size_t)nnz * 12) return fail(CW_EIO, "r1cs constraints truncated");
            for (uint32_t t = 0; t < nnz; t++) {
                uint32_t wire;
                uint64_t co;
                memcpy(&wire, p, 4);
                memcpy(&co, p + 4, 8);
                p += 12;
                if (wire >= n_wires || co >= prime) return fail(CW_EIO, "r1cs term out of range");
                c->r1_terms64.insert(c->r1_terms64.end(), {c->w2s[wire], part, (uint32_t)co, (uint32_t)(co >> 32)});
            }
        }
        if (c->r1_terms64.size() == first)                       // an empty constraint still closes a row: 0 * 0 = 0 on the constant wire
            c->r1_terms64.insert(c->r1_terms64.end(), {0u, 2u, 0u, 0u});
        c->r1_terms64[c->r1_terms64.size() - 3] |= 4u;
        const uint32_t now = (uint32_t)(c->r1_terms64.size() / 4);
        if (now - chunk_first >= CHUNK_TERMS || k + 1 == n_cons) {
            c->r1_chunks64.insert(c->r1_chunks64.end(), {chunk_first, now - chunk_first, chunk_row, 0u});
            chunk_first = now;
            chunk_row = k + 1;
        }
    }
    c->n_constraints = n_cons;
    return CW_OK;
}

static int load_tape(cw_circuit *c, const char *path) {
    std::vector<uint8_t> b;
    if (!read_file(path, b)) return fail(CW_EIO, std::string("tape file not found: ") + path);
    if (b.size() >= 4 && !memcmp(b.data(), "CW64", 4)) return load_tape64(c, b);
    if (b.size() < 16 + 32 + 48 || memcmp(b.data(), "CWTP", 4)) return fail(CW_EIO, "bad tape magic");
    const uint32_t *h = (const uint32_t *)(b.data() + 4);
    if (h[0] != 11) return fail(CW_EIO, "unsupported tape version");
    if (h[1] != 4) return fail(CW_EIO, "only 4x64-bit primes are supported (bn128, bls12381, ...)");
    uint32_t n_variants = h[2];
    size_t off = 16;
    memcpy(c->q.w, b.data() + off, 32);
    off += 32;
    if (b.size() < off + 64) return fail(CW_EIO, "tape file truncated");
    const uint32_t *m = (const uint32_t *)(b.data() + off);
    off += 64;
    c->n_signals = m[0];
    if (c->n_signals == 0 || c->n_signals >= (1u << cwplan::SLOT_BITS))      // slot ids travel in 26-bit fields
        return fail(CW_EIO, "tape: signal count out of range (1 .. 2^26 - 1)");
    c->n_witness = m[1];
    c->n_consts = m[2];
    c->input_start = m[3];
    c->n_inputs = m[4];
    uint32_t n_names = m[5], hsize = m[6];
    if ((m[7] & 0xFFFFu) != CW_RBITS || (m[7] >> 17)) return fail(CW_EIO, "tape was lowered for a different Montgomery radix");
    c->mont = (m[7] >> 16) & 1;
    uint32_t n_lconsts = m[8];
    c->n_pub_in = m[9];
    const uint32_t n_bit_programs = m[10], n_functions = m[11];
    c->n_dat_consts = m[12];                 // constants / io-map templates of the matching .dat (load_dat checks its sections)
    c->n_io_templates = m[13];
    const uint32_t n_log_statements = m[14];
    c->n_logv = m[15];
    if (n_log_statements > (1u << 20) || c->n_logv >= c->n_signals || c->n_io_templates > (1u << 20) ||
        (c->n_dat_consts != 0xFFFFFFFFu && c->n_dat_consts > (1u << 26)))
        return fail(CW_EIO, "tape header: section counts");
    if (c->n_logv && n_bit_programs) return fail(CW_EIO, "tape header: log values cannot be combined with a bit-plane program");
    if (n_functions > (1u << 16)) return fail(CW_EIO, "tape header: too many functions");
    if (n_bit_programs > 2) return fail(CW_EIO, "tape header: more than two bit-plane programs");
    if (c->mont && (n_bit_programs || n_functions))
        return fail(CW_EIO, "tape header: Montgomery-form signals cannot be combined with a bit-plane program or run-time functions");
    // shape of the main component: slot 0 is the constant 1, outputs from slot 1, inputs right after them
    if (c->input_start == 0 || (uint64_t)c->input_start + c->n_inputs > c->n_signals || c->n_pub_in > c->n_inputs ||
        c->n_witness == 0 || c->n_witness > c->n_signals - c->n_logv || hsize < 256 || (hsize & (hsize - 1)) ||
        n_names > hsize || hsize > std::max<uint64_t>(256, 2 * (uint64_t)n_names) ||            // max(2^ceil(log2 n), 256), mod.rs:167
        (hsize > 256 && hsize / 2 >= n_names) || n_names > c->n_inputs + 1u)
        return fail(CW_EIO, "tape header: inconsistent circuit shape");
    if (b.size() < off + ((size_t)c->n_consts + n_lconsts) * 32 + (size_t)c->n_witness * 4)
        return fail(CW_EIO, "tape file truncated");
    c->consts.resize((size_t)c->n_consts * 8);
    memcpy(c->consts.data(), b.data() + off, (size_t)c->n_consts * 32);
    off += (size_t)c->n_consts * 32;
    // D_DOTC constants: kept as 9 x 29-bit limbs (+3 pad words = 48 B per entry, 16-byte aligned for scalar loads)
    c->lconsts.assign((size_t)std::max<uint32_t>(n_lconsts, 1) * 12, 0);
    for (uint32_t k = 0; k < n_lconsts; k++) {
        uint64_t w[5] = {0, 0, 0, 0, 0};
        memcpy(w, b.data() + off + (size_t)k * 32, 32);
        for (int l = 0; l < 9; l++) {
            unsigned bit = 29 * l, wi = bit / 64, sh = bit % 64;
            uint64_t v = w[wi] >> sh;
            if (sh > 35) v |= w[wi + 1] << (64 - sh);
            c->lconsts[(size_t)k * 12 + l] = (uint32_t)(v & 0x1FFFFFFFu);
        }
    }
    off += (size_t)n_lconsts * 32;
    c->w2s.resize(c->n_witness);
    memcpy(c->w2s.data(), b.data() + off, (size_t)c->n_witness * 4);
    off += (size_t)c->n_witness * 4;
    for (uint32_t s : c->w2s)
        if (s >= c->n_signals - c->n_logv) return fail(CW_EIO, "tape witness list refers to a signal out of range");
    for (uint32_t i = 0; i < n_names; i++) {
        if (off + 4 > b.size()) return fail(CW_EIO, "tape names truncated");
        uint32_t len;
        memcpy(&len, b.data() + off, 4);
        off += 4;
        if (off + len + 8 > b.size()) return fail(CW_EIO, "tape names truncated");
        std::string name((const char *)b.data() + off, len);
        off += len;
        uint32_t ss[2];
        memcpy(ss, b.data() + off, 8);
        off += 8;
        if (ss[0] < c->input_start || (uint64_t)ss[0] + ss[1] > (uint64_t)c->input_start + c->n_inputs)
            return fail(CW_EIO, "tape input name refers to slots outside the main inputs");
        if (!c->input_names.emplace(name, std::make_pair(ss[0], ss[1])).second)
            return fail(CW_EIO, "tape input name appears twice");
    }
    // circom functions (device bytecode): every register, constant, jump target and array window is checked here once
    std::vector<uint32_t> fn_regs;
    std::vector<FpParams> fn_aux;                 // field parameters of the foreign primes of native big-integer functions
    std::vector<std::pair<uint32_t, uint32_t>> fn_native;     // (function, aux index)
    for (uint32_t fi = 0; fi < n_functions; fi++) {
        if (off + 52 > b.size()) return fail(CW_EIO, "tape functions truncated");
        uint32_t fh[5];
        memcpy(fh, b.data() + off, 20);
        U256 fmod;
        memcpy(fmod.w, b.data() + off + 20, 32);
        off += 52;
        const uint32_t n_regs = fh[0], n_ins = fh[1], nat_kind = fh[2], nat_n = fh[3], nat_k = fh[4];
        if (n_regs == 0 || n_regs >= (1u << 16) || n_ins == 0 || n_ins > (1u << 24) || (size_t)n_ins * 16 > b.size() - off)
            return fail(CW_EIO, "tape function: bad size");
        const uint32_t first = (uint32_t)(c->fn_code.size() / 4);
        c->fn_code.resize(c->fn_code.size() + (size_t)n_ins * 4);
        memcpy(&c->fn_code[(size_t)first * 4], b.data() + off, (size_t)n_ins * 16);
        off += (size_t)n_ins * 16;
        auto opnd_ok = [&](uint32_t x) { return (x & FN_CONST) ? (x & 0x7FFFFFFFu) < c->n_consts : x < n_regs; };
        for (uint32_t i = 0; i < n_ins; i++) {
            const uint32_t *ins = &c->fn_code[((size_t)first + i) * 4];
            const uint32_t op = ins[0], d = ins[1], a = ins[2], bb = ins[3];
            bool ok;
            switch (op) {
            case F_RET: ok = true; break;
            case F_JMP: ok = d < n_ins; break;
            case F_JZ: ok = d < n_ins && opnd_ok(a); break;
            case F_LDX: ok = d < n_regs && (bb & 0xFFFFu) < n_regs && (uint64_t)a + (bb >> 16) <= n_regs && (bb >> 16) > 0; break;
            case F_STX: ok = opnd_ok(a) && (bb & 0xFFFFu) < n_regs && (uint64_t)d + (bb >> 16) <= n_regs && (bb >> 16) > 0; break;
            case F_DIV: case D_MUL2: case D_ADD: case D_SUB: case D_IDIV: case D_MOD: case D_POW: case D_SHL: case D_SHR:
            case D_BAND: case D_BOR: case D_BXOR: case D_LT: case D_GT: case D_LEQ: case D_GEQ: case D_EQ: case D_NEQ:
            case D_LAND: case D_LOR:
                ok = d < n_regs && opnd_ok(a) && opnd_ok(bb); break;
            case D_COPY: case D_NEG: case D_BNOT: case D_LNOT:
                ok = d < n_regs && opnd_ok(a); break;
            default: ok = false;
            }
            if (!ok) return fail(CW_EIO, "tape function: bad instruction");
        }
        if (c->fn_code[((size_t)first + n_ins - 1) * 4] != F_RET) return fail(CW_EIO, "tape function does not end with a return");
        c->fn_tab.push_back(first);
        c->fn_tab.push_back(n_ins);
        c->fn_tab.push_back(n_regs);
        c->fn_tab.push_back(0);
        if (nat_kind == 4) {
            // long_div(a[k + m], b[k]) -> div[m + 1] ++ mod[k] (csrc/cw_call.hip.h eval_call_long_div): m travels in the modulus field
            const uint32_t nat_m = (uint32_t)fmod.w[0];
            const bool small = fmod.w[0] < 16 && fmod.w[1] == 0 && fmod.w[2] == 0 && fmod.w[3] == 0;
            if (!small || nat_n < 32 || nat_n > 64 || nat_k == 0 || nat_k > 15 || nat_n * nat_k > 256 || nat_m == 0 || nat_m > 15 ||
                (nat_k + nat_m) * nat_n > 640 || 3 * nat_k + 2 * nat_m + 1 > n_regs)
                return fail(CW_EIO, "tape function: bad native tag");
            c->fn_tab[c->fn_tab.size() - 1] = 4u | (nat_k << 4) | (nat_n << 8) | (nat_m << 16);
        } else if (nat_kind) {
            // a pure big-integer function with a closed form (circuits/bigint_func.py): mod_inv(a[k]) -> [k];
            // ec_add(x1, y1, x2, y2) / ec_double(x1, y1) -> lambda, x3, y3 - arguments in the first registers, results behind them
            const uint32_t n_args = nat_kind == 1 ? nat_k : nat_kind == 2 ? 4 * nat_k : 2 * nat_k;
            const uint32_t n_ret = nat_kind == 1 ? nat_k : 3 * nat_k;
            if (nat_kind > 3 || nat_n == 0 || nat_n > 64 || nat_k == 0 || nat_k > 15 || nat_n * nat_k > 256 || n_args + n_ret > n_regs ||
                !prime_supported(fmod) || u256_bits(fmod) > nat_n * nat_k)
                return fail(CW_EIO, "tape function: bad native tag");
            fn_native.push_back({fi, (uint32_t)fn_aux.size()});
            fn_aux.push_back(make_params(fmod));
            c->fn_tab[c->fn_tab.size() - 1] = nat_kind | (nat_k << 4) | (nat_n << 8);
        }
        fn_regs.push_back(n_regs);
        c->need_full = true;               // the interpreter lives in the full-operator kernel variant
    }
    // the field parameters of the native functions travel behind the function table (uint4 units from its start)
    {
        const uint32_t aux_words = (uint32_t)((sizeof(FpParams) + 15) / 16 * 4);
        for (auto &na : fn_native) {
            const uint32_t at = n_functions + na.second * (aux_words / 4);
            if (at >= (1u << 16)) return fail(CW_EIO, "tape: too many native functions");
            c->fn_tab[(size_t)na.first * 4 + 3] |= at << 16;
        }
        for (auto &P2 : fn_aux) {
            const size_t w0 = c->fn_tab.size();
            c->fn_tab.resize(w0 + aux_words, 0);
            memcpy(&c->fn_tab[w0], &P2, sizeof(FpParams));
        }
    }
    // log statements: strings and references to the hidden signals
    {
        uint32_t seen = 0;
        for (uint32_t li = 0; li < n_log_statements; li++) {
            if (off + 8 > b.size()) return fail(CW_EIO, "tape log program truncated");
            uint32_t lh[2];
            memcpy(lh, b.data() + off, 8);
            off += 8;
            cw_circuit::LogStmt st;
            st.at = lh[0];
            if (lh[1] > (1u << 16) || (!c->logs.empty() && st.at <= c->logs.back().at)) return fail(CW_EIO, "tape log program: bad statement");
            for (uint32_t k = 0; k < lh[1]; k++) {
                if (off + 8 > b.size()) return fail(CW_EIO, "tape log program truncated");
                uint32_t ih[2];
                memcpy(ih, b.data() + off, 8);
                off += 8;
                cw_circuit::LogItem it;
                if (ih[0] == 0) {
                    if (ih[1] > b.size() - off) return fail(CW_EIO, "tape log program truncated");
                    it.text.assign((const char *)b.data() + off, ih[1]);
                    off += ih[1];
                } else if (ih[0] == 1 && ih[1] == seen) {       // values are numbered in statement order
                    it.is_value = true;
                    it.value = ih[1];
                    seen++;
                } else return fail(CW_EIO, "tape log program: bad item");
                st.items.push_back(std::move(it));
            }
            c->logs.push_back(std::move(st));
        }
        if (seen != c->n_logv) return fail(CW_EIO, "tape log program: value count differs from the header");
    }
    if (n_variants == 0) return fail(CW_EIO, "tape holds no schedule");
    for (uint32_t v = 0; v < n_variants; v++) {
        if (off + 32 > b.size()) return fail(CW_EIO, "tape variant truncated");
        uint32_t vh[8];
        memcpy(vh, b.data() + off, 32);
        off += 32;
        Variant var;
        var.n_strands = vh[0];
        var.n_tslots = vh[1];
        uint32_t nrows = vh[2], nextras = vh[3], nterms = vh[5];
        var.n_lds = vh[4];
        if (var.n_strands == 0 || var.n_strands > 16) return fail(CW_EIO, "tape variant: bad strand count");
        if (var.n_lds > 72) return fail(CW_EIO, "tape variant: too many LDS slots");
        if (vh[6] > 1) return fail(CW_EIO, "tape variant: unknown kind");
        if (vh[6] == 1) {       // pipelined single-wave schedule: rows of 8 words, `extras` = load lists
            var.kind = 1;
            var.nb = vh[7] & 0xFF;
            var.nld = (vh[7] >> 8) & 0xFF;
            if (var.n_strands != 1) return fail(CW_EIO, "tape variant: bad strand count");
            size_t need = 24 + (size_t)nrows * 32 + (size_t)nextras * 4 + (size_t)nterms * 16;
            if (off + need > b.size()) return fail(CW_EIO, "tape variant truncated");
            uint32_t offs[6];
            memcpy(offs, b.data() + off, 24);
            off += 24;
            if (offs[0] != 0 || offs[1] != nrows || offs[2] != 0 || offs[3] != nextras || offs[4] != 0 || nterms < 4 || offs[5] != nterms - 4)
                return fail(CW_EIO, "tape variant: bad offsets");
            var.stream_off = {0, nrows};
            var.extra_off = {0, nextras};
            var.term_off = {0, nterms - 4};
            var.prows.resize((size_t)nrows * 8);
            memcpy(var.prows.data(), b.data() + off, (size_t)nrows * 32);
            off += (size_t)nrows * 32;
            var.extras.resize(nextras);
            memcpy(var.extras.data(), b.data() + off, (size_t)nextras * 4);
            off += (size_t)nextras * 4;
            var.terms.resize((size_t)nterms * 4);
            memcpy(var.terms.data(), b.data() + off, (size_t)nterms * 16);
            off += (size_t)nterms * 16;
            if (const char *why = validate_pipe_variant(var, c->n_signals, c->n_consts, n_lconsts))
                return fail(CW_EIO, std::string("tape variant (pipelined): ") + why);
            uint64_t mm = 0;
            for (size_t r = 0; r < nrows; r++) {
                const uint32_t op = var.prows[r * 8] & 0xFF, n = var.prows[r * 8 + 1];
                if (op == D_MMUL || op == D_MADD || op == D_MULC || op == D_MADDC) mm++;
                if (op == D_MUL2) mm += 2;
                if (op == D_DOTC || op == D_LINSUM) mm += n;
                if (op == D_INV || op == D_IDIV || op == D_MOD || op == D_POW) c->need_full = true;
            }
            if (v == 0) {
                c->n_rows = nrows;
                c->n_mmul = mm;
            }
            c->variants.push_back(std::move(var));
            continue;
        }
        size_t need = (size_t)(var.n_strands + 1) * 12 + (size_t)nrows * 16 + (size_t)nextras * 4 + (size_t)nterms * 16;
        if (off + need > b.size()) return fail(CW_EIO, "tape variant truncated");
        var.stream_off.resize(var.n_strands + 1);
        memcpy(var.stream_off.data(), b.data() + off, (size_t)(var.n_strands + 1) * 4);
        off += (size_t)(var.n_strands + 1) * 4;
        var.extra_off.resize(var.n_strands + 1);
        memcpy(var.extra_off.data(), b.data() + off, (size_t)(var.n_strands + 1) * 4);
        off += (size_t)(var.n_strands + 1) * 4;
        var.term_off.resize(var.n_strands + 1);
        memcpy(var.term_off.data(), b.data() + off, (size_t)(var.n_strands + 1) * 4);
        off += (size_t)(var.n_strands + 1) * 4;
        var.rows.resize(nrows);
        memcpy(var.rows.data(), b.data() + off, (size_t)nrows * 16);
        off += (size_t)nrows * 16;
        var.extras.resize(nextras);
        memcpy(var.extras.data(), b.data() + off, (size_t)nextras * 4);
        off += (size_t)nextras * 4;
        var.terms.resize((size_t)nterms * 4);
        memcpy(var.terms.data(), b.data() + off, (size_t)nterms * 16);
        off += (size_t)nterms * 16;
        {   // which flat operation every row that can fail comes from (reported in the status word)
            const uint32_t nseq = vh[7];
            if (nseq > nrows || off + ((size_t)var.n_strands + 1 + nseq) * 4 > b.size()) return fail(CW_EIO, "tape variant truncated");
            var.seq_off.resize(var.n_strands + 1);
            memcpy(var.seq_off.data(), b.data() + off, (size_t)(var.n_strands + 1) * 4);
            off += (size_t)(var.n_strands + 1) * 4;
            var.seqs.resize(nseq);
            memcpy(var.seqs.data(), b.data() + off, (size_t)nseq * 4);
            off += (size_t)nseq * 4;
            if (var.seq_off[0] != 0 || var.seq_off[var.n_strands] != nseq) return fail(CW_EIO, "tape variant: bad check-index offsets");
            for (uint32_t sq : var.seqs)
                if (sq > 0xFFFFFFu) return fail(CW_EIO, "tape variant: check index exceeds 24 bits");
        }
        if (nterms < 4 || nextras < 4) return fail(CW_EIO, "tape variant: tables lack their padding");
        if (const char *why = validate_variant(var, c->n_signals, c->n_consts, n_lconsts, fn_regs))
            return fail(CW_EIO, std::string("tape variant: ") + why);
        if (var.term_off[var.n_strands] + 4 != nterms) return fail(CW_EIO, "tape variant: bad term offsets");
        {   // DOTC terms must index the limb-form constant table
            size_t tpos = 0, lin_terms = 0;
            for (auto &r : var.rows) {
                uint32_t op = r.w0 & 0xFF;
                if (op == D_LINSUM || op == D_DOTC) {
                    if (tpos + r.a > nterms) return fail(CW_EIO, "tape variant: term list overruns the table");
                    if (op == D_DOTC)
                        for (uint32_t t = 0; t < r.a; t++)
                            if (var.terms[(tpos + t) * 4 + 2] >= n_lconsts) return fail(CW_EIO, "tape variant: bad constant index");
                    if (op == D_LINSUM) lin_terms += r.a;
                    tpos += r.a;
                }
            }
            var.wide_linsum = lin_terms * 4 > (size_t)nrows;        // more than a quarter of a term per row on average
            // schedules of circuits with run-time functions are interpreted on 16 strands, where every term of a sum is a dependent
            // table read: four in flight instead of two took the ECDSA verifier from 293 to 245 ms per launch
            if (!c->fn_tab.empty()) var.wide_linsum = true;
            if (const char *e = getenv("CW_WIDE_LINSUM")) var.wide_linsum = atoi(e) != 0;     // (experiments)
        }
        if (var.extra_off[var.n_strands] + 4 != nextras) return fail(CW_EIO, "tape variant: bad extra offsets");
        if (var.stream_off[var.n_strands] != nrows) return fail(CW_EIO, "tape variant: bad stream offsets");
        uint64_t mm = 0;
        for (auto &r : var.rows) {
            uint32_t op = r.w0 & 0xFF;
            if (op >= D_NOPS) return fail(CW_EIO, "tape contains an unknown opcode");
            if (op == D_MMUL || op == D_MADD || op == D_MULC || op == D_MADDC) mm++;
            if (op == D_MUL2) mm += 2;
            if (op == D_DOTC || op == D_LINSUM) mm += r.a;           // one product per term
            if (op == D_INV || op == D_IDIV || op == D_MOD || op == D_POW) c->need_full = true;
        }
        var.n_active = 0;
        std::vector<double> load(var.n_strands, 0.0);              // rough instruction counts (/32), as in lower.py
        for (uint32_t st = 0; st < var.n_strands; st++) {
            for (uint32_t r = var.stream_off[st]; r < var.stream_off[st + 1]; r++) {
                const uint32_t op = var.rows[r].w0 & 0xFF;
                if (op == D_BARRIER || op == D_NOP) continue;
                double cst = 2.0;
                if (op == D_MUL2) cst = 20.0;
                else if (op == D_MMUL || op == D_MULC || op == D_MADD || op == D_MADDC) cst = 10.0;
                else if (op == D_DOTC) cst = 5.0 + 4.5 * var.rows[r].a;
                else if (op == D_LINSUM) cst = 3.0 + 1.2 * var.rows[r].a;
                else if (op == D_INV) cst = 1000.0;
                else if (op == D_POW || op == D_IDIV || op == D_MOD) cst = 6000.0;
                load[st] += cst;
            }
            var.n_active += load[st] > 0;
        }
        if (!c->fn_tab.empty()) {
            // Circuits with run-time functions: how much shorter than the work is the variant's critical path?  Per barrier epoch the
            // busiest strand's cost (the MEASURED costs of interpreted rows, lower.py _COST_INTERP: a call is 130 - 20 000 units
            // on ONE strand while the others wait), summed; cw_batch_create skips a multi-strand variant that does not at least
            // halve the single strand's time (BigMultModP = one long_div call + a few rows: 16 strands only take wave slots).
            auto cost_of = [&](const CwRow &rw) -> double {
                const uint32_t op = rw.w0 & 0xFF, nx = (rw.w0 >> SH_NX) & 0xFFF;
                switch (op) {
                case D_CALL: {
                    const uint32_t kind = rw.a < n_functions ? (c->fn_tab[(size_t)rw.a * 4 + 3] & 15u) : 0u;
                    return kind == 0 ? 20000.0 : kind == 4 ? 130.0 : 170.0;
                }
                case D_BITS: return 6.0 + 0.3 * nx;
                case D_LINSUM: return 6.0 + 4.8 * rw.a;
                case D_DOTC: return 6.0 + 6.0 * rw.a;
                case D_MULC: case D_MADDC: case D_MADD: return 8.0;
                case D_MUL2: return 10.0;
                case D_IDIV: case D_MOD: return 23.0;
                case D_INV: return 180.0;
                case D_POW: return 6000.0;
                default: return 6.0;
                }
            };
            std::vector<std::vector<double>> ep(var.n_strands);
            size_t n_ep = 0;
            for (uint32_t st = 0; st < var.n_strands; st++) {
                double cur = 0;
                for (uint32_t r = var.stream_off[st]; r < var.stream_off[st + 1]; r++) {
                    const uint32_t op = var.rows[r].w0 & 0xFF;
                    if (op == D_BARRIER) { ep[st].push_back(cur); cur = 0; continue; }
                    cur += cost_of(var.rows[r]);
                }
                ep[st].push_back(cur);
                n_ep = std::max(n_ep, ep[st].size());
            }
            var.work = var.crit = 0;
            for (size_t e = 0; e < n_ep; e++) {
                double mx = 0;
                for (uint32_t st = 0; st < var.n_strands; st++) {
                    const double x = e < ep[st].size() ? ep[st][e] : 0.0;
                    var.work += x;
                    mx = std::max(mx, x);
                }
                var.crit += mx;
            }
        }
        const double heaviest = *std::max_element(load.begin(), load.end());
        var.prio_mask = 0;
        for (uint32_t st = 0; st < var.n_strands && st < 32; st++)
            if (load[st] > 0 && load[st] >= 0.8 * heaviest) var.prio_mask |= 1u << st;
        if (v == 0) {
            c->n_rows = nrows;
            c->n_mmul = mm;
        }
        c->variants.push_back(std::move(var));
    }
    if (n_bit_programs) {
        // bit-plane program: 8 x u32 {ring, n_vrows, n_slots lo, hi, cache, n_asserts, format version, 0}, records
        // (n_vrows * 64 x 2 u32), command blocks (n_vrows / 8 x 24 u32), signal -> slot map, assertion slots
        if (off + 32 > b.size()) return fail(CW_EIO, "tape bit program truncated");
        uint32_t bh[8];
        memcpy(bh, b.data() + off, 32);
        off += 32;
        if (bh[6] != 2 || bh[7]) return fail(CW_EIO, "tape bit program: unknown format version (lowered by another release)");
        cwbits::Program &bp = c->bits;
        bp.ring = bh[0];
        bp.n_vrows = bh[1];
        bp.n_slots = (uint64_t)bh[2] | ((uint64_t)bh[3] << 32);
        bp.cache = bh[4];
        const uint32_t n_asserts = bh[5];
        if (bp.n_vrows > (1u << 24) || bp.n_vrows % cwbits::BATCH || n_asserts > (1u << 24)) return fail(CW_EIO, "tape bit program truncated");
        const uint64_t rwords = (uint64_t)bp.n_vrows * 64 * 2, cwords = (uint64_t)(bp.n_vrows / cwbits::BATCH) * cwbits::CMD_WORDS;
        if ((rwords + cwords + c->n_signals + n_asserts) * 4 > b.size() - off) return fail(CW_EIO, "tape bit program truncated");
        bp.recs.resize((size_t)rwords);
        memcpy(bp.recs.data(), b.data() + off, (size_t)rwords * 4);
        off += (size_t)rwords * 4;
        bp.cmds.resize((size_t)cwords);
        memcpy(bp.cmds.data(), b.data() + off, (size_t)cwords * 4);
        off += (size_t)cwords * 4;
        bp.sig_slot.resize(c->n_signals);
        memcpy(bp.sig_slot.data(), b.data() + off, (size_t)c->n_signals * 4);
        off += (size_t)c->n_signals * 4;
        bp.assert_slots.resize(n_asserts);
        memcpy(bp.assert_slots.data(), b.data() + off, (size_t)n_asserts * 4);
        off += (size_t)n_asserts * 4;
        if (const char *why = cwbits::validate(bp, c->n_signals, c->n_inputs)) return fail(CW_EIO, std::string("tape: ") + why);
        c->has_bits = true;
    }
    if (n_bit_programs > 1) {
        // emitted code: 8 x u32 {format 1, n_slots lo, hi, code bytes, flags (bit 0: the fused R1CS check covers every
        // constraint), VGPRs, AccVGPRs, 0}, signal -> slot map, the code object (padded to 4 bytes)
        if (off + 32 > b.size()) return fail(CW_EIO, "tape emitted program truncated");
        uint32_t jh[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        memcpy(jh, b.data() + off, 32);
        off += 32;
        if (jh[0] == 2) {                          // format 2: + identity of the constraint system the checks were built from
            if (off + 8 > b.size()) return fail(CW_EIO, "tape emitted program truncated");
            memcpy(jh + 8, b.data() + off, 8);
            off += 8;
        } else if (jh[0] != 1 || jh[7]) {
            return fail(CW_EIO, "tape emitted program: unknown format version (lowered by another release)");
        }
        cwbits::JitProgram &jp = c->jit;
        jp.r1cs_crc = jh[8];
        jp.r1cs_len = jh[9];
        jp.n_slots = (uint64_t)jh[1] | ((uint64_t)jh[2] << 32);
        const uint64_t code_bytes = jh[3], padded = (code_bytes + 3) & ~3ull;
        jp.check_complete = jh[4] & 1u;
        jp.n_vgpr = jh[5];
        jp.n_agpr = jh[6];
        if ((uint64_t)c->n_signals * 4 + padded > b.size() - off) return fail(CW_EIO, "tape emitted program truncated");
        jp.sig_slot.resize(c->n_signals);
        memcpy(jp.sig_slot.data(), b.data() + off, (size_t)c->n_signals * 4);
        off += (size_t)c->n_signals * 4;
        jp.code.assign(b.data() + off, b.data() + off + code_bytes);
        off += (size_t)padded;
        const uint64_t audit_bytes = jh[7], audit_padded = (audit_bytes + 3) & ~3ull;     // the audit's code object (may be absent)
        if (audit_padded > b.size() - off) return fail(CW_EIO, "tape emitted program truncated");
        jp.audit_code.assign(b.data() + off, b.data() + off + audit_bytes);
        off += (size_t)audit_padded;
        if (const char *why = cwbits::validate_jit(jp, c->n_signals, c->n_inputs)) return fail(CW_EIO, std::string("tape: ") + why);
        c->has_jit = true;
    }
    // emitted 256-bit code (optional trailing section, hip_elements/fpjit.py): "FPJT" | u32 format 1 | u32 n, then per program
    // 8 x u32 {n_strands, code bytes, LDS bytes, scratch bytes, VGPRs, words of the fused-check bitmap, 0, 0}, the bitmap (one bit
    // per .r1cs row: the emitted code recomputes that row itself) and the code object padded to 4 bytes
    if (off + 12 <= b.size() && memcmp(b.data() + off, "FPJT", 4) == 0) {
        uint32_t fh[2];
        memcpy(fh, b.data() + off + 4, 8);
        off += 12;
        if (fh[0] != 1 || fh[1] > 8) return fail(CW_EIO, "tape emitted 256-bit code: unknown format version (lowered by another release)");
        for (uint32_t k = 0; k < fh[1]; k++) {
            if (off + 32 > b.size()) return fail(CW_EIO, "tape emitted 256-bit code truncated");
            uint32_t ph[8];
            memcpy(ph, b.data() + off, 32);
            off += 32;
            const uint64_t padded = ((uint64_t)ph[1] + 3) & ~3ull;
            if (ph[0] == 0 || ph[0] > 16 || (ph[0] & (ph[0] - 1)) || ph[2] > 160 * 1024 || ph[3] > 4096 || ph[4] > 512 ||
                ph[1] < 64 || (uint64_t)ph[5] * 4 > b.size() - off || padded > b.size() - off - (uint64_t)ph[5] * 4)
                return fail(CW_EIO, "tape emitted 256-bit code: bad header");
            std::vector<uint32_t> cov(ph[5]);
            if (ph[5]) memcpy(cov.data(), b.data() + off, (size_t)ph[5] * 4);
            off += (size_t)ph[5] * 4;
            if (memcmp(b.data() + off, "\x7f" "ELF", 4) != 0) return fail(CW_EIO, "tape emitted 256-bit code: not a code object");
            bool have_variant = false;
            for (auto &v : c->variants) have_variant |= (v.kind == 0 && v.n_strands == ph[0]);
            if (!have_variant) return fail(CW_EIO, "tape emitted 256-bit code: no schedule variant with that strand count");
            FpJit fj;
            fj.n_strands = ph[0];
            fj.lds_bytes = ph[2];
            fj.scratch_bytes = ph[3];
            fj.n_vgpr = ph[4];
            fj.covered = std::move(cov);
            fj.r1cs_crc = ph[6];                   // (0, 0 in tapes of earlier releases: the row count is then the only guard)
            fj.r1cs_len = ph[7];
            for (uint32_t w : fj.covered) fj.n_covered += (uint32_t)__builtin_popcount(w);
            fj.code.assign(b.data() + off, b.data() + off + ph[1]);
            off += (size_t)padded;
            c->fpjit.push_back(std::move(fj));
        }
    }
    // hash map as generate_hash_map builds it (c_code_generator.rs:575-587); replaced by the .dat's if given
    c->hashmap.assign(hsize, HashEntry{0, 0, 0});
    // insertion order must be the reference's (main input list order = slot order)
    std::vector<std::pair<uint32_t, std::string>> order;
    for (auto &kv : c->input_names) order.push_back({kv.second.first, kv.first});
    std::sort(order.begin(), order.end());
    for (auto &o : order) {
        uint64_t hsh = fnv1a(o.second.data(), o.second.size());
        size_t p = hsh % hsize, probes = 0;
        while (c->hashmap[p].signalid != 0) {
            if (++probes > hsize) return fail(CW_EIO, "tape input hash map is full");
            p = (p + 1) % hsize;
        }
        c->hashmap[p] = HashEntry{hsh, o.first, c->input_names[o.second].second};
    }
    if (!prime_supported(c->q))
        return fail(CW_EINVAL, "unsupported prime: the device field code handles the 253..256-bit primes of circom "
                               "(4 x 64-bit limbs); the 64-bit Goldilocks runtime is out of scope");
    c->P = make_params(c->q);
    return CW_OK;
}

// <name>.dat, reference layout (reader main.cpp:22-124): hash map | witness2signal u64[] | constants ...
static int load_dat(cw_circuit *c, const char *path) {
    std::vector<uint8_t> b;
    if (!read_file(path, b)) return fail(CW_EIO, std::string(".dat file not found: ") + path);
    size_t hs = c->hashmap.size();
    size_t need = hs * 24 + (size_t)c->n_witness * 8;
    if (b.size() < need) return fail(CW_EIO, ".dat file too small for this tape");
    for (size_t i = 0; i < hs; i++) {
        memcpy(&c->hashmap[i], b.data() + i * 24, 24);
        const HashEntry &h = c->hashmap[i];                  // setInputSignal writes signalValues[signalid + idx]
        // overflow-free: signalid in [input_start, input_start + n_inputs], size <= what is left behind it
        const uint64_t in_end = (uint64_t)c->input_start + c->n_inputs;
        if (h.signalid != 0 && (h.signalid < c->input_start || h.signalid > in_end || h.signalsize > in_end - h.signalid))
            return fail(CW_EIO, ".dat input hash map refers to slots outside the main inputs");
    }
    const uint8_t *w = b.data() + hs * 24;
    for (uint32_t i = 0; i < c->n_witness; i++) {
        uint64_t s;
        memcpy(&s, w + (size_t)i * 8, 8);
        if (s >= c->n_signals) return fail(CW_EIO, ".dat witness list refers to a signal out of range");
        c->w2s[i] = (uint32_t)s;
    }
    // constants (40 bytes each, c_code_generator.rs:616-679), then the io-map of the Mixed component clusters
    // (c_code_generator.rs:681-738, main.cpp:60-92): u32 template ids, then per template the number of io signals and per
    // signal {offset, number of lengths - 1, lengths[1..], element size, bus id}.  The reference binary knows both counts
    // from compiled-in constants (get_size_of_constants(), get_size_of_io_map()); here the tape header carries them.  The
    // evaluator never needs the table (every access was resolved when the circuit was traced); it is validated, kept
    // for cw_io_map_size / cw_io_map_offset, and a damaged one is rejected.
    c->io_map.clear();
    if (c->n_dat_consts == 0xFFFFFFFFu) return CW_OK;          // tape without section counts
    const size_t tail0 = need + (size_t)c->n_dat_consts * 40;
    if (b.size() < tail0) return fail(CW_EIO, ".dat file too small for the constants of this circuit");
    if ((b.size() - tail0) % 4) return fail(CW_EIO, ".dat io-map section is not a whole number of words");
    const size_t nw = (b.size() - tail0) / 4, n = c->n_io_templates;
    std::vector<uint32_t> w32(nw);
    if (nw) memcpy(w32.data(), b.data() + tail0, nw * 4);
    if (nw < n) return fail(CW_EIO, ".dat io-map section truncated");
    size_t at = n;
    for (size_t i = 0; i < n; i++) {
        if (i && w32[i] <= w32[i - 1]) return fail(CW_EIO, ".dat io-map: template ids are not increasing");
        if (at >= nw) return fail(CW_EIO, ".dat io-map section truncated");
        const uint32_t nd = w32[at++];
        cw_circuit::IoTemplate t;
        t.id = w32[i];
        for (uint32_t d = 0; d < nd; d++) {
            if (at + 2 > nw) return fail(CW_EIO, ".dat io-map section truncated");
            cw_circuit::IoDef def;
            def.offset = w32[at];
            const uint32_t nl = w32[at + 1];
            at += 2;
            if (nl > 16 || at + nl + 2 > nw) return fail(CW_EIO, ".dat io-map section truncated");
            def.lengths.assign(w32.begin() + at, w32.begin() + at + nl);
            at += nl;
            def.size = w32[at];
            def.bus_id = w32[at + 1];
            at += 2;
            uint64_t span = def.size;
            for (uint32_t l : def.lengths) span *= std::max<uint32_t>(l, 1);
            if (def.size == 0 || def.offset >= c->n_signals || span > c->n_signals)
                return fail(CW_EIO, ".dat io-map: signal definition out of range");
            t.defs.push_back(std::move(def));
        }
        c->io_map.push_back(std::move(t));
    }
    // the bus-field map (c_code_generator.rs:740-794, main.cpp:95-121): per bus instance the number of fields and per field
    // {offset inside the bus, number of dimensions - 1, dimensions[1..], size of one element, id of the field's own bus}.  The
    // reference binary reads get_size_of_bus_field_map() entries; here the section runs to the end of the file.  Like the
    // io-map it is validated and kept (cw_bus_map_size / cw_bus_field), never needed by the evaluator.
    c->bus_map.clear();
    while (at < nw) {
        const uint32_t nf = w32[at++];
        if (nf == 0 || nf > 65536) return fail(CW_EIO, ".dat bus-field map: a bus has at least one field (and not more than 65 536)");
        std::vector<cw_circuit::IoDef> fields;
        for (uint32_t d = 0; d < nf; d++) {
            if (at + 2 > nw) return fail(CW_EIO, ".dat bus-field map truncated");
            cw_circuit::IoDef def;
            def.offset = w32[at];
            const uint32_t nl = w32[at + 1];
            at += 2;
            if (nl > 16 || at + nl + 2 > nw) return fail(CW_EIO, ".dat bus-field map truncated");
            def.lengths.assign(w32.begin() + at, w32.begin() + at + nl);
            at += nl;
            def.size = w32[at];
            def.bus_id = w32[at + 1];
            at += 2;
            uint64_t span = def.size;
            for (uint32_t l : def.lengths) span *= std::max<uint32_t>(l, 1);
            if (def.size == 0 || def.offset >= c->n_signals || span > c->n_signals)
                return fail(CW_EIO, ".dat bus-field map: field definition out of range");
            fields.push_back(std::move(def));
        }
        c->bus_map.push_back(std::move(fields));
    }
    // a field's own bus is an EARLIER entry (a nested bus is completed before the bus that holds it; 0 also stands for "a signal")
    for (size_t bi = 0; bi < c->bus_map.size(); bi++)
        for (const cw_circuit::IoDef &f : c->bus_map[bi])
            if (f.bus_id >= c->bus_map.size() || (f.bus_id && f.bus_id >= bi))
                return fail(CW_EIO, ".dat bus-field map: a field refers to a bus that is not an earlier entry");
    return CW_OK;
}

// <name>.r1cs (constraint_writers/src/r1cs_writer.rs; sections may come in any order — the writer emits 2,1,3)
static int load_r1cs(cw_circuit *c, const char *path) {
    std::vector<uint8_t> b;
    if (!read_file(path, b)) return fail(CW_EIO, std::string(".r1cs file not found: ") + path);
    if (b.size() < 12 || memcmp(b.data(), "r1cs", 4)) return fail(CW_EIO, "bad r1cs magic");
    uint32_t nsec;
    memcpy(&nsec, b.data() + 8, 4);
    size_t off = 12;
    const uint8_t *sec[6] = {0};
    uint64_t seclen[6] = {0};
    for (uint32_t s = 0; s < nsec; s++) {
        if (off + 12 > b.size()) return fail(CW_EIO, "r1cs truncated");
        uint32_t typ;
        uint64_t len;
        memcpy(&typ, b.data() + off, 4);
        memcpy(&len, b.data() + off + 4, 8);
        off += 12;
        if (len > b.size() - off) return fail(CW_EIO, "r1cs section truncated");
        if (typ < 6) {
            if (sec[typ]) return fail(CW_EIO, "r1cs section appears twice");
            sec[typ] = b.data() + off;
            seclen[typ] = len;
        }
        off += len;
    }
    if (!sec[1] || !sec[2]) return fail(CW_EIO, "r1cs misses header or constraints section");
    if (seclen[1] < 64) return fail(CW_EIO, "r1cs header section too short");      // 4 + 32 + 4*4 + 8 + 4
    uint32_t fs;
    memcpy(&fs, sec[1], 4);
    if (fs != 32) return fail(CW_EIO, "r1cs field size must be 32 bytes");
    if (memcmp(sec[1] + 4, c->q.w, 32)) return fail(CW_EIO, "r1cs prime differs from the tape's");
    uint32_t hdr[4];
    memcpy(hdr, sec[1] + 36, 16);   // nWires, nPubOut, nPubIn, nPrvIn
    uint32_t n_wires = hdr[0], n_cons;
    memcpy(&n_cons, sec[1] + 36 + 16 + 8, 4);
    if (n_wires != c->n_witness) return fail(CW_EIO, "r1cs wire count differs from the witness size");
    if ((uint64_t)n_cons > seclen[2] / 12) return fail(CW_EIO, "r1cs constraint count exceeds its section");   // 3 x u32 nnz each
    c->n_constraints = n_cons;
    {
        // Checks baked into emitted code (the fused check and the audit of the bit-plane code, the `covered` rows of the 256-bit
        // code) were built from ONE constraint system: they are trusted only for a .r1cs whose constraint section is byte for byte
        // that one (same CRC-32, same length).  Any other file - even one with the same number of rows - is checked in full by the
        // stand-alone kernels, and the programs with baked-in checks are not used.
        const uint32_t crc = crc32_ieee(sec[2], (size_t)seclen[2]), len = (uint32_t)seclen[2];
        c->r1cs_matches_code = true;
        if (c->has_jit && (c->jit.r1cs_crc || c->jit.r1cs_len) && (c->jit.r1cs_crc != crc || c->jit.r1cs_len != len)) {
            c->jit.check_complete = false;
            c->jit.audit_code.clear();
            c->r1cs_matches_code = false;
        }
        for (size_t k = 0; k < c->fpjit.size();) {
            FpJit &fj = c->fpjit[k];
            if (fj.n_covered && (fj.r1cs_crc || fj.r1cs_len) && (fj.r1cs_crc != crc || fj.r1cs_len != len)) {
                c->fpjit.erase(c->fpjit.begin() + (long)k);
                c->r1cs_matches_code = false;
                continue;
            }
            k++;
        }
    }
    c->r_ptr.assign(1, 0);
    c->r_ptr.reserve((size_t)n_cons * 3 + 1);
    // coefficient table: id 0 = +1, id 1 = -1, others = c*R mod q
    c->r_ctab.assign(16, 0);
    U256 one{{1, 0, 0, 0}}, minus1;
    u256_sub(minus1, c->q, one);
    std::map<std::array<uint64_t, 4>, uint32_t> cid, ccid;
    const uint8_t *p = sec[2], *end = sec[2] + seclen[2];
    for (uint32_t k = 0; k < n_cons; k++) {
        for (int part = 0; part < 3; part++) {
            if (p + 4 > end) return fail(CW_EIO, "r1cs constraints truncated");
            uint32_t nnz;
            memcpy(&nnz, p, 4);
            p += 4;
            if (p + (size_t)nnz * 36 > end) return fail(CW_EIO, "r1cs constraints truncated");
            for (uint32_t t = 0; t < nnz; t++) {
                uint32_t wire;
                U256 co;
                memcpy(&wire, p, 4);
                memcpy(co.w, p + 4, 32);
                p += 36;
                if (wire >= n_wires) return fail(CW_EIO, "r1cs wire id out of range");
                uint32_t id;
                const bool on_one = (c->w2s[wire] == 0);           // the constant-1 wire: coefficient * 1 needs no multiply
                if (u256_cmp(co, one) == 0) id = 0;
                else if (u256_cmp(co, minus1) == 0) id = 1;
                else {
                    std::array<uint64_t, 4> key{co.w[0], co.w[1], co.w[2], co.w[3] ^ (on_one ? (1ull << 63) : 0)};
                    auto it = cid.find(key);
                    if (it == cid.end()) {
                        id = (uint32_t)(c->r_ctab.size() / 8);
                        // the constant-1 wire holds 1 (R' when the table holds Montgomery forms): its term is the coefficient itself
                        U256 cm = (on_one && !c->mont) ? co : shlmod(co, CW_RBITS, c->q);
                        uint32_t limbs[8];
                        memcpy(limbs, cm.w, 32);
                        c->r_ctab.insert(c->r_ctab.end(), limbs, limbs + 8);
                        if (!on_one) {
                            // entry id + 1: what the term is worth on a wire that holds "one" (cw_r1cs_plan.h COEF_BITSEL) - c on a
                            // canonical table, c R' on a table of Montgomery forms
                            U256 cr = co;                         // (a coefficient >= q is legal in the file: reduce it)
                            for (int it = 0; it < 64 && u256_cmp(cr, c->q) >= 0; it++) u256_sub(cr, cr, c->q);
                            if (u256_cmp(cr, c->q) >= 0) return fail(CW_EIO, "r1cs coefficient is far above the prime");
                            memcpy(limbs, c->mont ? cm.w : cr.w, 32);
                            c->r_ctab.insert(c->r_ctab.end(), limbs, limbs + 8);
                        }
                        cid[key] = id;
                    } else id = it->second;
                    if (on_one) id |= cwplan::COEF_CONST;
                }
                c->r_slot.push_back(c->w2s[wire]);     // wire id = witness position -> value slot
                c->r_coef.push_back(id);
                if (c->has_bits) {                     // the bit-plane check works on canonical coefficients
                    std::array<uint64_t, 4> ck{co.w[0], co.w[1], co.w[2], co.w[3]};
                    auto it = ccid.find(ck);
                    uint32_t cci;
                    if (it == ccid.end()) {
                        if (u256_cmp(co, c->q) >= 0) return fail(CW_EIO, "r1cs coefficient is not reduced modulo the prime");
                        cci = (uint32_t)(c->r_cctab.size() / 8);
                        uint32_t limbs[8];
                        memcpy(limbs, co.w, 32);
                        c->r_cctab.insert(c->r_cctab.end(), limbs, limbs + 8);
                        ccid[ck] = cci;
                    } else cci = it->second;
                    c->r_cc.push_back(cci);
                }
            }
            c->r_ptr.push_back((uint32_t)c->r_slot.size());
        }
    }
    // ---- processing order: by the schedule position at which a row's youngest wire is produced ----------------
    // (variant 0 = one strand = program order).  A wire is then re-read by the check shortly after another row
    // touched it, while it is still in L2, instead of three times from HBM.
    std::vector<uint32_t> defpos(c->n_signals, 0);
    {
        const Variant &v0 = c->variants[0];
        size_t xp = 0;
        for (size_t r = 0; r * 8 < v0.prows.size(); r++)            // a pipelined variant: the two store targets of each row
            for (int k = 3; k <= 4; k++) {
                const uint32_t t = v0.prows[r * 8 + k];
                if (!(t & X_TMP) && t < c->n_signals) defpos[t] = (uint32_t)r + 1;
            }
        for (size_t r = 0; r < v0.rows.size(); r++) {
            const CwRow &row = v0.rows[r];
            uint32_t op = row.w0 & 0xFF, dk = (row.w0 >> SH_DK) & 7, nx = (row.w0 >> SH_NX) & 0xFFF;
            if (op == D_BARRIER) continue;
            if (dk == K_SIG && row.dst < c->n_signals) defpos[row.dst] = (uint32_t)r + 1;
            for (uint32_t e = 0; e < nx; e++) {
                uint32_t x = v0.extras[xp + e];
                if (op == D_BITS) x &= ~X_NEXT;
                if (!(x & (X_TMP | X_LDS)) && x < c->n_signals) defpos[x] = (uint32_t)r + 1;
            }
            xp += nx;
        }
    }
    std::vector<uint64_t> key(n_cons);
    for (uint32_t k = 0; k < n_cons; k++) {
        uint32_t mx = 0;
        for (uint32_t t = c->r_ptr[3 * k]; t < c->r_ptr[3 * k + 3]; t++) mx = std::max(mx, defpos[c->r_slot[t]]);
        key[k] = ((uint64_t)mx << 32) | k;
    }
    std::sort(key.begin(), key.end());
    std::vector<uint32_t> n_ptr(1, 0), n_slot, n_coef, n_cc;
    n_slot.reserve(c->r_slot.size());
    n_coef.reserve(c->r_coef.size());
    n_cc.reserve(c->r_cc.size());
    c->r_orig.resize(n_cons);
    c->r_bool.clear();
    c->r_bool.reserve(n_cons);
    for (uint32_t j = 0; j < n_cons; j++) {
        uint32_t k = (uint32_t)key[j];
        for (int part = 0; part < 3; part++) {
            for (uint32_t t = c->r_ptr[3 * k + part]; t < c->r_ptr[3 * k + part + 1]; t++) {
                n_slot.push_back(c->r_slot[t]);
                n_coef.push_back(c->r_coef[t]);
                if (c->has_bits) n_cc.push_back(c->r_cc[t]);
            }
            n_ptr.push_back((uint32_t)n_slot.size());
        }
        uint32_t a0 = c->r_ptr[3 * k], a1 = c->r_ptr[3 * k + 1], b1 = c->r_ptr[3 * k + 2], c1 = c->r_ptr[3 * k + 3];
        bool eq2 = (a0 == a1) && (a1 == b1) && (c1 - b1 == 2) &&
                   ((c->r_coef[b1] == 0 && c->r_coef[b1 + 1] == 1) || (c->r_coef[b1] == 1 && c->r_coef[b1 + 1] == 0));
        c->r_orig[j] = k | (eq2 ? 0x80000000u : 0u);
        // b * (b - 1) = 0 / b * (1 - b) = 0 (either order of the factors, C empty): the stream plan checks "b is 0 or 1" directly
        {
            auto single = [&](uint32_t lo, uint32_t hi, uint32_t &w) {
                if (hi - lo != 1 || c->r_coef[lo] != 0 || c->r_slot[lo] == 0) return false;
                w = c->r_slot[lo];
                return true;
            };
            auto minus_one = [&](uint32_t lo, uint32_t hi, uint32_t w) {       // {1: -1, w: +1} or {1: +1, w: -1}, any order
                if (hi - lo != 2) return false;
                for (int sw = 0; sw < 2; sw++) {
                    const uint32_t tc = lo + sw, tw = lo + 1 - sw;
                    if (c->r_slot[tc] == 0 && c->r_slot[tw] == w &&
                        ((c->r_coef[tc] == 1 && c->r_coef[tw] == 0) || (c->r_coef[tc] == 0 && c->r_coef[tw] == 1)))
                        return true;
                }
                return false;
            };
            uint32_t w = 0;
            const bool isbool = c1 == b1 && ((single(a0, a1, w) && minus_one(a1, b1, w)) || (single(a1, b1, w) && minus_one(a0, a1, w)));
            c->r_bool.push_back(isbool ? 1 : 0);
        }
    }
    if (c->r_ctab.size() / 8 >= (1u << 30)) return fail(CW_EIO, "r1cs: too many distinct coefficients");   // ids share a word with COEF_CONST / COEF_BITSEL
    c->r_ptr.swap(n_ptr);
    c->r_slot.swap(n_slot);
    c->r_coef.swap(n_coef);
    c->r_cc.swap(n_cc);
    return CW_OK;
}

extern "C" int cw_load(const char *tape_path, const char *dat_path, const char *r1cs_path, cw_circuit **out) {
    if (!tape_path || !out) return fail(CW_EINVAL, "cw_load: null argument");
    cw_circuit *c = new cw_circuit();
    int rc;
    try {                                   // a hostile size field must not take the host process down (no C++ exception crosses the C ABI)
        rc = load_tape(c, tape_path);
        if (rc == CW_OK && dat_path) rc = load_dat(c, dat_path);
        if (rc == CW_OK && r1cs_path) rc = c->is64 ? load_r1cs64(c, r1cs_path) : load_r1cs(c, r1cs_path);
    } catch (const std::bad_alloc &) {
        rc = fail(CW_EIO, "cw_load: a table size in the files exceeds available memory");
    } catch (const std::exception &e) {
        rc = fail(CW_EIO, std::string("cw_load: ") + e.what());
    }
    if (rc != CW_OK) {
        delete c;
        return rc;
    }
    *out = c;
    return CW_OK;
}
extern "C" void cw_free(cw_circuit *c) {
    if (!c) return;
    for (auto &kv : c->jit_mod) {               // modules of the emitted bit-plane code, one per device that ran it
        if (hipSetDevice(kv.first) == hipSuccess) hipModuleUnload(kv.second.first);
    }
    for (auto &kv : c->jit_audit_mod)
        if (hipSetDevice(kv.first) == hipSuccess) hipModuleUnload(kv.second.first);
    for (auto &fj : c->fpjit)
        for (auto &kv : fj.mod)
            if (hipSetDevice(kv.first) == hipSuccess) hipModuleUnload(kv.second.first);
    delete c;
}
extern "C" uint32_t cw_io_map_size(const cw_circuit *c) { return c ? (uint32_t)c->io_map.size() : 0; }
extern "C" int64_t cw_io_map_offset(const cw_circuit *c, uint32_t template_id, uint32_t signal_code) {
    if (!c) return -1;
    for (const auto &t : c->io_map)
        if (t.id == template_id) return signal_code < t.defs.size() ? (int64_t)t.defs[signal_code].offset : -1;
    return -1;
}
// get_size_of_bus_field_map() and one field of the map (Circom_Circuit::busInsId2FieldInfo, circom.hpp:42; main.cpp:95-121)
extern "C" uint32_t cw_bus_map_size(const cw_circuit *c) { return c ? (uint32_t)c->bus_map.size() : 0; }
extern "C" int cw_bus_field(const cw_circuit *c, uint32_t bus_id, uint32_t field, uint32_t *offset, uint32_t *size, uint32_t *field_bus_id,
                            uint32_t *n_lengths) {
    if (!c || bus_id >= c->bus_map.size() || field >= c->bus_map[bus_id].size()) return fail(CW_EINVAL, "cw_bus_field: no such bus / field");
    const cw_circuit::IoDef &f = c->bus_map[bus_id][field];
    if (offset) *offset = f.offset;
    if (size) *size = f.size;
    if (field_bus_id) *field_bus_id = f.bus_id;
    if (n_lengths) *n_lengths = (uint32_t)f.lengths.size();
    return CW_OK;
}
extern "C" uint32_t cw_n_signals(const cw_circuit *c) { return c->n_signals - c->n_logv; }     // the circuit's signals (hidden log values excluded)
extern "C" uint32_t cw_n_log_statements(const cw_circuit *c) { return c ? (uint32_t)c->logs.size() : 0; }
extern "C" uint32_t cw_n_witness(const cw_circuit *c) { return c->n_witness; }
extern "C" uint32_t cw_n_inputs(const cw_circuit *c) { return c->n_inputs; }
// The witness of a SIMPLIFIED constraint system.  Without a flag the reference simplifies at --O1 (constraint_list/src/
// constraint_simplification.rs): the `.r1cs` a prover takes keeps a subset of the signals, and the emitted calculator writes
// exactly those (witness2signal of the `.dat`, calcwit.hpp:54-56).  The evaluation and the R1CS check of this library work
// on the full system the circuit was loaded with (every term of the `.r1cs` was resolved to a value slot by cw_load); this
// call only changes what the egress paths hand out - cw_get_witness(es)(_device), cw_write_wtns(_many), cw_write_wtnsb -
// for batches created AFTER it.  `signals`: strictly increasing signal ids starting with 0 whose first 1 + cw_n_public
// entries are the ones of the current list (outputs and public inputs are never simplified away).
extern "C" int cw_set_witness_list(cw_circuit *c, const uint32_t *signals, uint32_t n) {
    if (!c || !signals || n == 0) return fail(CW_EINVAL, "cw_set_witness_list: bad argument");
    if (c->live_batches.load() > 0)
        return fail(CW_EINVAL, "cw_set_witness_list: the circuit has live batches (their device images of the list are sized at creation): "
                               "set the list before cw_batch_create, or free the batches first");
    const uint32_t np = 1 + cw_n_public(c);
    if (n < np || n > c->n_signals - c->n_logv) return fail(CW_EINVAL, "cw_set_witness_list: list length out of range");
    for (uint32_t i = 0; i < n; i++) {
        if (signals[i] >= c->n_signals - c->n_logv || (i && signals[i] <= signals[i - 1]))
            return fail(CW_EINVAL, "cw_set_witness_list: signal ids must increase and stay below cw_n_signals");
        if (i < np && (i >= c->w2s.size() || signals[i] != c->w2s[i]))
            return fail(CW_EINVAL, "cw_set_witness_list: the constant, the outputs and the public inputs must stay where they are");
    }
    c->w2s.assign(signals, signals + n);
    c->n_witness = n;
    return CW_OK;
}
extern "C" int cw_emitted_checks_match(const cw_circuit *c) { return c && c->r1cs_matches_code ? 1 : 0; }
extern "C" uint32_t cw_input_start(const cw_circuit *c) { return c->input_start; }
extern "C" uint32_t cw_n_constraints(const cw_circuit *c) { return c->n_constraints; }
// public signals = main's outputs then its public inputs = witness positions 1 .. n_public (r1cs header nPubOut/nPubIn)
extern "C" uint32_t cw_n_public(const cw_circuit *c) { return c->input_start - 1 + c->n_pub_in; }
extern "C" uint64_t cw_n_rows(const cw_circuit *c) { return c->n_rows; }
extern "C" uint64_t cw_n_mmul(const cw_circuit *c) { return c->n_mmul; }
extern "C" void cw_prime(const cw_circuit *c, uint8_t le32[32]) { memcpy(le32, c->q.w, 32); }

// getInputSignalHashPosition (calcwit.cpp:51-69): open addressing, empty slot = signalid 0
static int64_t hash_pos(const cw_circuit *c, uint64_t h) {
    size_t n = c->hashmap.size();
    size_t pos = h % n;
    if (c->hashmap[pos].hash != h) {
        size_t ini = pos;
        pos = (pos + 1) % n;
        while (pos != ini) {
            if (c->hashmap[pos].hash == h) return (int64_t)pos;
            if (c->hashmap[pos].signalid == 0) return -1;
            pos = (pos + 1) % n;
        }
        return -1;
    }
    return (int64_t)pos;
}

extern "C" int64_t cw_input_size(const cw_circuit *c, const char *name, uint32_t *start_slot) {
    int64_t p = hash_pos(c, fnv1a(name, strlen(name)));
    if (p < 0 || c->hashmap[p].signalid == 0) return -1;
    if (start_slot) *start_slot = (uint32_t)c->hashmap[p].signalid;
    return (int64_t)c->hashmap[p].signalsize;
}

// Staging plan parameters of the R1CS check for a batch: enough (instance group x chunk) workgroups for ~3
// rounds over the chip at the occupancy the LDS entries leave (160 KB / (entries x 2 KiB) waves per CU).
static void r1cs_plan_defaults(uint32_t batch, uint32_t *chunks, uint32_t *entries) {
    uint32_t e = 10;
    if (const char *s = getenv("CW_R1CS_ENTRIES")) e = (uint32_t)std::max(1, atoi(s));
    e = std::min<uint32_t>(std::max<uint32_t>(e, cwplan::DEPTH + 2), 64);
    uint64_t groups = ((uint64_t)batch + 63) / 64;
    uint32_t ch = (uint32_t)std::max<uint64_t>(1, (6144 + groups - 1) / groups);
    if (const char *s = getenv("CW_R1CS_CHUNKS")) ch = (uint32_t)std::max(1, atoi(s));
    *chunks = ch;
    *entries = e;
}

extern "C" int cw_r1cs_plan_stats(const cw_circuit *c, uint32_t batch, uint32_t chunks, uint32_t entries, uint64_t out[8]) {
    if (!c || !out) return fail(CW_EINVAL, "null argument");
    if (c->n_constraints == 0) return fail(CW_ESTATE, "no .r1cs was loaded for this circuit");
    uint32_t dc, de;
    r1cs_plan_defaults(batch ? batch : 65536, &dc, &de);
    if (!chunks) chunks = dc;
    if (!entries) entries = de;
    cwplan::Plan p = cwplan::build(c->r_ptr, c->r_slot, c->r_coef, c->r_orig, c->n_signals, chunks, entries);
    std::string err = cwplan::verify(p, c->r_slot);
    if (!err.empty()) return fail(CW_ESTATE, ("r1cs plan: " + err).c_str());
    uint64_t v[8] = {p.n_chunks, p.n_loads, p.n_terms, p.n_filler, p.n_unique, p.entries, cwplan::DEPTH, 0};
    memcpy(out, v, sizeof v);
    return CW_OK;
}

// host-only: the term stream cw_batch_create hands to cw_r1cs_stream_kernel, for replaying it on the CPU (tests/test_r1cs_plan.py)
extern "C" int cw_r1cs_stream_plan(const cw_circuit *c, uint32_t terms_per_chunk, uint32_t flags, uint64_t sizes[8], uint32_t *chunk,
                                   uint32_t *terms, uint32_t *row_orig, uint32_t *ctab) {
    if (!c || !sizes) return fail(CW_EINVAL, "null argument");
    if (c->n_constraints == 0) return fail(CW_ESTATE, "no .r1cs was loaded for this circuit");
    cwplan::Plan p = cwplan::build_stream(c->r_ptr, c->r_slot, c->r_coef, c->r_orig, terms_per_chunk ? terms_per_chunk : 192, nullptr,
                                          (flags & 1u) ? nullptr : &c->r_bool, !(flags & 2u));
    uint64_t v[8] = {p.chunk.size(), p.terms.size(), p.row_orig.size(), c->r_ctab.size(), p.n_terms, p.n_folded, p.n_bitsel, p.n_chunks};
    memcpy(sizes, v, sizeof v);
    if (chunk) memcpy(chunk, p.chunk.data(), p.chunk.size() * 4);
    if (terms) memcpy(terms, p.terms.data(), p.terms.size() * 4);
    if (row_orig) memcpy(row_orig, p.row_orig.data(), p.row_orig.size() * 4);
    if (ctab) memcpy(ctab, c->r_ctab.data(), c->r_ctab.size() * 4);
    return CW_OK;
}

// ---------------------------------------------------------------------------------------------------------
// batch
// ---------------------------------------------------------------------------------------------------------
static const uint64_t PIPE_MAX_GROUPS = 4096;

struct cw_batch {
    cw_circuit *c = nullptr;
    int device = 0;
    uint32_t batch = 0, Bp = 0, lanes = 64, prio_mask = 0;
    hipStream_t stream = nullptr;
    void *d_V = nullptr;
    size_t v_bytes = 0;
    CwDRow *d_rows = nullptr;
    const Variant *var = nullptr;
    uint32_t *d_stream_off = nullptr, *d_extra_off = nullptr;
    uint64_t *d_extras = nullptr, *d_terms = nullptr;
    uint32_t *d_term_off = nullptr;
    uint32_t *d_prows = nullptr, *d_ploads = nullptr;   // pipelined variant: device rows (CwPRow) and load lists
    uint32_t *d_lconsts = nullptr;
    uint32_t *d_fncode = nullptr, *d_fntab = nullptr;   // circom functions (D_CALL)
    uint32_t *d_consts = nullptr, *d_w2s = nullptr, *d_status = nullptr, *d_first_bad = nullptr;
    // R1CS check plan (cw_r1cs_plan.h) on the device; r1_entries != 0 selects the LDS-staged kernel
    uint32_t *d_rctab = nullptr, *d_rctab29 = nullptr, *d_pchunk = nullptr, *d_prec = nullptr, *d_pterms = nullptr, *d_prow = nullptr;
    uint32_t r1_chunks = 0, r1_entries = 0;
    void *d_in = nullptr;          // AoS staging [batch][n_in][32]
    void *d_pmask = nullptr;       // cw_set_inputs_bits: the caller's packed masks on the device (8 bytes per input and group)
    void *d_gather = nullptr;      // [n_witness][32]
    void *d_bulk = nullptr;        // staging of cw_get_witnesses: [bulk_rows][n_witness][32]
    uint32_t bulk_rows = 0;
    const void *ext_in = nullptr;  // caller-owned device inputs (cw_set_inputs_device)
    // cw_run_check: cw_run + cw_check_r1cs captured once as a HIP graph and replayed (one launch per step instead of ~10)
    hipGraphExec_t rc_graph = nullptr;
    const void *rc_in = nullptr, *rc_packed = nullptr;   // the input pointers the captured launches carry
    uint32_t rc_calls = 0;                                // plain calls since the inputs last changed (the first loads modules)
    bool rc_failed = false;                               // a capture of these launches failed once: cw_run_check stays on the plain calls
    // argument blocks of the emitted kernels (hipModuleLaunchKernel's `extra` form): members, not locals - a captured launch
    // (cw_run_check) may keep the POINTERS it was given, and a replay must find the block where the capture saw it
    struct FpArgs { void *V; uint32_t *status; uint32_t Bp, batch, lanes, pad; FpParams P; uint32_t pad2; const void *consts, *fcode, *ftab; } fp_args;
    struct JitArgs { void *T, *fb, *r1; } jit_args, audit_args;
    size_t fp_args_size = sizeof(FpArgs), jit_args_size = sizeof(JitArgs);
    void *fp_cfg[5], *jit_cfg[5], *audit_cfg[5];
    hipStream_t rc_stream = nullptr;                      // the launches are recorded on a private stream (the batch's may be the null
                                                          // stream, which cannot be captured) and replayed on the batch's own
    std::vector<uint32_t> h_stream_begin;
    std::vector<uint8_t> h_in;     // host staging for per-signal assignment
    std::vector<uint8_t> assigned; // [batch][n_in] flags (inputSignalAssigned, calcwit.cpp:28-32)
    std::vector<uint32_t> remaining;
    bool all_set = false, host_dirty = false, ran = false;
    // ---- bit-plane mode (cw_bits.hip): one bit per signal per instance instead of a 32-byte slot ----
    bool bitmode = false;
    uint64_t *d_T = nullptr, *d_fbmask = nullptr;   // bit table [groups][slots]; per-group mask of instances to re-run wide
    uint64_t t_bytes = 0;
    uint32_t n_groups = 0;
    uint32_t *d_brecs = nullptr, *d_bcmds = nullptr, *d_aslots = nullptr;   // the gate program: records, command blocks, assertion slots
    uint32_t *d_wslot = nullptr;                      // witness position -> bit-table slot (sig_slot o w2s)
    uint32_t *d_fbinst = nullptr;                     // instances of the side batch (device copy of fb_inst)
    uint32_t fbinst_cap = 0;
    const void *packed_in = nullptr;                  // cw_set_inputs_bits_device: uint64 masks [groups][n_inputs] instead of ext_in
    uint32_t *d_erecs = nullptr, *d_wchunk = nullptr, *d_wterms = nullptr, *d_wctab = nullptr, *d_wrow = nullptr;
    uint32_t *d_ichunk = nullptr, *d_iterms = nullptr, *d_itab = nullptr, *d_irow = nullptr, *d_sigslot = nullptr;
    uint32_t n_ichunks = 0;
    uint32_t n_evrows = 0, n_wchunks = 0, bits_steps = 0, bits_width = 64;
    // instances whose inputs are not all 0/1 (or that tripped an assertion gate) are re-run by the 256-bit schedule
    cw_batch *fb = nullptr;                           // side batch (classic variant) holding them
    std::vector<uint32_t> fb_inst;                    // side-batch position -> instance
    std::vector<int32_t> fb_index;                    // instance -> side-batch position or -1
    bool resolved = false, checked = false;
    // layout of the bit table (cw_bits.hip): slots per group / per chunk, shift (0 = interpreter, 5 = emitted code)
    uint64_t bits_slots = 0;
    uint32_t bits_sh = 0, n_groups_padded = 0;
    const std::vector<uint32_t> *bits_sigslot = nullptr;
    hipFunction_t fp_fn = nullptr;                     // emitted code of the chosen strand variant (nullptr: the interpreter runs it)
    uint32_t fp_lds = 0;
    const std::vector<uint32_t> *fp_covered = nullptr; // .r1cs rows that code checks itself (findings: second half of d_status)
    bool jit = false, table_dirty = false;             // emitted code runs this batch; the caller holds a raw pointer to the table
    hipFunction_t jit_fn = nullptr;
    // 64-bit runtime: V64[slot][Bp], its program and R1CS terms
    uint64_t *d_V64 = nullptr, *d_consts64 = nullptr;
    uint32_t *d_rows64 = nullptr, *d_terms64 = nullptr, *d_chunks64 = nullptr;
    uint64_t *d_r1flag = nullptr;                      // per group: instances whose fused R1CS check fired (emitted code)
    // cw_batch_set_timing: events on the batch's stream around the parts of cw_run / cw_check_r1cs (their own intervals, measured
    // where they run - bench.py's roofline figures): 0 run begins | 1 inputs ingested | 2 evaluation done | 3 check begins | 4 check done
    // cw_batch_set_timing(b, 2) keeps the marks of the last CW_TIMING_RING runs (a new set per cw_run): cw_batch_kernel_ms_mean
    // averages a part over every set recorded since timing was switched on - the figure a timed region produces, not one launch
    bool timing = false;
    struct TSet { hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; bool set[5] = {false, false, false, false, false}; };
    std::vector<TSet> tring = std::vector<TSet>(1);   // [0] alone: the marks of the LAST run / check
    size_t tcur = 0;
    bool tfresh = true;                                // no run since timing was switched on: the first one takes set 0
};
#define CW_TIMING_RING 64
static inline void cw_tmark(cw_batch *b, int k) {
    if (!b->timing) return;
    if (k == 0 && b->tring.size() > 1) {               // a run begins: its marks (and its check's) go to the next set of the ring
        if (!b->tfresh) b->tcur = (b->tcur + 1) % b->tring.size();
        for (bool &x : b->tring[b->tcur].set) x = false;
    }
    b->tfresh = false;
    cw_batch::TSet &t = b->tring[b->tcur];
    hipEventRecord(t.ev[k], b->stream);
    t.set[k] = true;
}
#define TMARK(b, k) cw_tmark((b), (k))

template <typename T>
static hipError_t upload(T **dst, const std::vector<T> &src, hipStream_t s) {
    size_t n = std::max<size_t>(src.size(), 1) * sizeof(T);
    hipError_t e = hipMalloc((void **)dst, n);
    if (e != hipSuccess) return e;
    if (!src.empty()) e = hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s);
    return e;
}

extern "C" void cw_batch_free(cw_batch *b) {
    if (!b) return;
    if (b->rc_graph) hipGraphExecDestroy(b->rc_graph);
    b->rc_graph = nullptr;
    if (b->rc_stream) hipStreamDestroy(b->rc_stream);
    b->rc_stream = nullptr;
    if (b->c) b->c->live_batches--;
    if (b->device < 0) {
        delete b;
        return;
    }
    hipSetDevice(b->device);
    hipStreamSynchronize(b->stream);
    for (cw_batch::TSet &t : b->tring)
        for (hipEvent_t e : t.ev)
            if (e) hipEventDestroy(e);
    if (b->fb) cw_batch_free(b->fb);
    void *bptrs[] = {b->d_V64, b->d_consts64, b->d_rows64, b->d_terms64, b->d_chunks64, b->d_T, b->d_fbmask, b->d_r1flag, b->d_brecs, b->d_bcmds, b->d_aslots, b->d_wslot, b->d_fbinst, b->d_erecs, b->d_wchunk, b->d_wterms, b->d_wctab, b->d_wrow,
                     b->d_ichunk, b->d_iterms, b->d_itab, b->d_irow, b->d_sigslot, b->d_pmask};
    for (void *p : bptrs)
        if (p) hipFree(p);
    if (b->d_fncode) hipFree(b->d_fncode);
    if (b->d_fntab) hipFree(b->d_fntab);
    void *ptrs[] = {b->d_V, b->d_prows, b->d_ploads, b->d_rows, b->d_stream_off, b->d_extras, b->d_extra_off, b->d_terms, b->d_term_off, b->d_lconsts, b->d_consts, b->d_w2s, b->d_status, b->d_first_bad,
                    b->d_rctab, b->d_rctab29, b->d_pchunk, b->d_prec, b->d_pterms, b->d_prow,
                    b->d_in, b->d_gather, b->d_bulk};
    for (void *p : ptrs)
        if (p) hipFree(p);
    delete b;
}

static int bits_batch_setup(cw_batch *b);
static int batch_setup64(cw_batch *b);
// device staging of host-side inputs: [batch][n_inputs][32]; bit-plane batches allocate it on first use
static int ensure_d_in(cw_batch *b) {
    if (b->d_in) return CW_OK;
    const size_t n = std::max<size_t>((size_t)b->batch * b->c->n_inputs * 32, 32);
    hipError_t e = hipMalloc(&b->d_in, n);
    if (e != hipSuccess)
        return fail(CW_EDEVICE, "hipMalloc of the input staging image failed (" + std::to_string(n) + " bytes): " + hipGetErrorString(e));
    return CW_OK;
}

static int batch_create_impl(cw_circuit *c, int device, uint32_t batch, void *stream, bool allow_bits, cw_batch **out) {
    if (!c || !out || batch == 0) return fail(CW_EINVAL, "cw_batch_create: bad argument");
    if (batch > (1u << 26)) return fail(CW_EINVAL, "cw_batch_create: batch exceeds 2^26 instances (32-bit lane offsets)");
    if (device < 0) {
        // host-only batch: input staging and its error semantics can be exercised without a GPU;
        // anything that computes fails loudly.
        cw_batch *hb = new cw_batch();
        hb->c = c;
        c->live_batches++;
        hb->device = -1;
        hb->batch = batch;
        hb->Bp = (batch + 255) / 256 * 256;
        hb->remaining.assign(batch, c->n_inputs);
        *out = hb;
        return CW_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(CW_EDEVICE, "no HIP device available: the witness calculator requires a gfx950 GPU (no CPU fallback)");
    HIPCHK(hipSetDevice(device));
    cw_batch *b = new cw_batch();
    b->c = c;
    c->live_batches++;
    b->device = device;
    b->batch = batch;
    b->Bp = (batch + 255) / 256 * 256;
    b->stream = (hipStream_t)stream;
    if (c->is64) {
        int rc = batch_setup64(b);
        if (rc != CW_OK) {
            cw_batch_free(b);
            return rc;
        }
        b->remaining.assign(batch, c->n_inputs);
        *out = b;
        return CW_OK;
    }
    if (allow_bits && c->has_bits) {
        // every signal is a bit: the bit-plane program replaces the 256-bit schedule (which stays in the file for the
        // instances whose inputs turn out not to be 0/1)
        b->bitmode = true;
        int rc = bits_batch_setup(b);
        if (rc != CW_OK) {
            cw_batch_free(b);
            return rc;
        }
        b->remaining.assign(batch, c->n_inputs);
        *out = b;
        return CW_OK;
    }
    // pick the schedule variant.  Up to 8192 waves (two rounds of the chip's 4096 wave slots) a variant with more
    // strands that carry work shortens the critical path of every instance group; a variant whose extra strands
    // only wait at barriers (Poseidon(2) has 3 independent lanes: S = 16 is S = 4 plus 12 idle waves) just takes
    // wave slots.  Measured on Poseidon(2): 512 groups S = 4: 0.81 ms, S = 16: 1.39 ms; 2048 groups S = 4: 2.12 ms,
    // S = 1: 2.53 ms; 4096 groups: S = 1 wins.  CW_STRANDS overrides.
    {
        uint64_t groups = (batch + 63) / 64;
        const Variant *best = nullptr;
        double work1 = 0;                              // circuits with functions: the single strand's time (any variant's work is the same rows)
        for (auto &v : c->variants)
            if (!v.kind && v.work > 0 && (work1 == 0 || v.n_strands == 1)) work1 = v.work;
        for (auto &v : c->variants) {
            if (v.kind) continue;
            if (groups * v.n_strands > 8192 && v.n_strands > 1) continue;
            if (v.n_strands > 1 && v.crit > 0 && work1 < 2.0 * v.crit && !getenv("CW_STRANDS")) {
                bool have1 = false;
                for (auto &u : c->variants) have1 |= !u.kind && u.n_strands == 1;
                if (have1) continue;                   // its strands mostly wait for one chain (a call): no shorter than one strand
            }
            if (!best || v.n_active > best->n_active || (v.n_active == best->n_active && v.n_strands < best->n_strands))
                best = &v;
        }
        if (!best)
            for (auto &v : c->variants)
                if (!v.kind && (!best || v.n_strands < best->n_strands)) best = &v;
        // A batch that fills every SIMD with ONE strand (>= 1 024 groups) and whose R1CS check can ride in the rows runs the
        // single-strand emitted program with the check fused in: no barriers, no hand-offs, no waiting for the slowest strand -
        // a lone wave per SIMD issues at one instruction per ~4.5 clocks, and the next batch in flight (another wave per SIMD)
        // fills the gaps.  Poseidon(2) x 65 536, rows + check: 59-62 M witnesses/s against 52 M with four strands
        // (tools/fpjit_poseidon_strands.sh, fpjit_poseidon_inflight.sh).  CW_STRANDS / CW_FP_JIT = 0 / CW_FP_FUSED = 0 override.
        {
            const char *fj0 = getenv("CW_FP_JIT"), *ff0 = getenv("CW_FP_FUSED");
            if (best && best->n_strands > 1 && !getenv("CW_STRANDS") && !(fj0 && atoi(fj0) == 0) && !(ff0 && atoi(ff0) == 0) &&
                c->n_constraints && groups >= 1024) {
                bool have_code = false;
                for (auto &fj : c->fpjit)
                    have_code |= fj.n_strands == 1 && !fj.covered.empty() && fj.covered.size() == (c->n_constraints + 31) / 32 &&
                                 (uint64_t)fj.n_covered * 2 >= c->n_constraints;
                if (have_code)
                    for (auto &v : c->variants)
                        if (!v.kind && v.n_strands == 1) best = &v;
            }
        }
        if (const char *e = best ? getenv("CW_STRANDS") : nullptr) {
            uint32_t want = (uint32_t)std::max(1, atoi(e));
            for (auto &v : c->variants) {
                if (v.kind) continue;
                bool better = (v.n_strands <= want && v.n_strands > best->n_strands) ||
                              (best->n_strands > want && v.n_strands < best->n_strands);
                if (better) best = &v;
            }
        }
        // The pipelined single-wave variant hides the value-table latency inside ONE wave (LDS ring + load lists a batch
        // ahead), so it wins wherever the strand variants run at one or two waves per SIMD: up to PIPE_MAX_GROUPS groups
        // of 64 instances.  Beyond that the plain single-strand schedule has enough waves per SIMD to hide the latency
        // by itself and needs no LDS.  CW_PIPE = 0 / 1 overrides.
        {
            const Variant *pv = nullptr;
            for (auto &v : c->variants)
                if (v.kind == 1) pv = &v;
            bool use = pv && (!best || (groups <= PIPE_MAX_GROUPS && !getenv("CW_STRANDS")));   // CW_STRANDS asks for a strand variant
            if (const char *e = getenv("CW_PIPE")) use = pv && (atoi(e) != 0 || !best);
            if (use) best = pv;
        }
        if (!best) {
            delete b;
            return fail(CW_EINVAL, "the tape holds no usable schedule variant");
        }
        b->var = best;
        // instances per workgroup: when there are fewer workgroups than CUs (256), a batch is spread over more of them
        // by leaving the upper lanes of the waves idle - never beyond one workgroup per CU, because idle lanes
        // still cost VALU issue (measured on Poseidon(2): 32 lanes are 16 % slower as soon as every CU is busy)
        uint32_t lanes = 64;
        while (lanes > 16 && 2 * (((uint64_t)batch + lanes - 1) / lanes) <= 256) lanes >>= 1;
        if (const char *e = getenv("CW_LANES")) {
            int v = atoi(e);
            if (v == 16 || v == 32 || v == 64) lanes = (uint32_t)v;
        }
        b->lanes = lanes;
        b->prio_mask = best->prio_mask;
        if (const char *e = getenv("CW_PRIO_MASK")) b->prio_mask = (uint32_t)strtoul(e, nullptr, 0);   // diagnostics
        // the variant's emitted code, when the tape carries it (CW_FP_JIT = 0: interpret the rows instead)
        const char *fe_ = getenv("CW_FP_JIT");
        if (best->kind == 0 && !(fe_ && atoi(fe_) == 0)) {
            // A variant may come in two programs: the rows alone, and the rows with the R1CS check fused in (recomputed behind
            // the rows that produce the wires).  The fused one does about twice the arithmetic in one launch and saves the
            // check's pass over the table: it wins where the launch is throughput-bound (4 waves per SIMD and more), the plain
            // one + the stand-alone check kernel where a batch waits on its dependency chain (measured, evaluation + check:
            // Poseidon(2) x 65 536, 4 waves per SIMD: 1.64 vs 2.07 ms; Semaphore-style x 8 192, 2 per SIMD: 21.1 vs 18.8 ms; its
            // 1 024-instance shard: 18.0 vs 14.3 ms).  CW_FP_FUSED = 0 / 1 overrides.
            const uint64_t waves = ((uint64_t)batch + lanes - 1) / lanes * best->n_strands;
            bool want_fused = (waves >= 4096 || (best->n_strands == 1 && waves >= 1024)) && c->n_constraints != 0;
            if (const char *e2 = getenv("CW_FP_FUSED")) want_fused = atoi(e2) != 0;
            FpJit *pick = nullptr;
            for (auto &fj : c->fpjit) {
                if (fj.n_strands != best->n_strands) continue;
                const bool fused = !fj.covered.empty();
                if (fused && !(c->n_constraints && fj.covered.size() == (c->n_constraints + 31) / 32)) continue;   // another .r1cs
                if (fused && !getenv("CW_FP_FUSED") && (uint64_t)fj.n_covered * 2 < c->n_constraints)
                    continue;                       // the code covers a minority of the rows: not worth its steps (unless asked for)
                if (!pick || fused == want_fused) pick = &fj;
            }
            if (pick) {
                FpJit &fj = *pick;
                auto it = fj.mod.find(b->device);
                if (it == fj.mod.end()) {
                    hipModule_t mod = nullptr;
                    hipFunction_t fn = nullptr;
                    hipError_t e1 = hipModuleLoadData(&mod, fj.code.data());
                    if (e1 == hipSuccess) e1 = hipModuleGetFunction(&fn, mod, FPJIT_KERNEL);
                    if (e1 != hipSuccess) {
                        delete b;
                        return fail(CW_EDEVICE, std::string("loading the emitted 256-bit code failed: ") + hipGetErrorString(e1));
                    }
                    it = fj.mod.emplace(b->device, std::make_pair(mod, fn)).first;
                }
                b->fp_fn = it->second.second;
                b->fp_lds = fj.lds_bytes;
                if (!fj.covered.empty()) b->fp_covered = &fj.covered;   // findings: second half of d_status (CW_R1CS_AUDIT: re-checked)
            }
        }
    }
    size_t slots = (size_t)c->n_signals + b->var->n_tslots;
    b->v_bytes = slots * 2 * b->Bp * 16;
    hipError_t e = hipMalloc(&b->d_V, b->v_bytes);
    if (e != hipSuccess) {
        delete b;
        return fail(CW_EDEVICE, "hipMalloc of the value table failed (" + std::to_string(slots) + " slots): " +
                                    hipGetErrorString(e));
    }
#define TRY(x)                                                              \
    do {                                                                    \
        hipError_t e2 = (x);                                                \
        if (e2 != hipSuccess) {                                             \
            cw_batch_free(b);                                               \
            return fail(CW_EDEVICE, std::string(#x ": ") + hipGetErrorString(e2)); \
        }                                                                   \
    } while (0)
    if (b->var->kind == 1) {
        // pipelined variant: value-table targets become unified slot numbers (signals, then temps), LDS entries of the terms
        // byte offsets; the row table is padded with 3 NOPs (the kernel reads three rows ahead)
        const Variant &v = *b->var;
        std::vector<uint32_t> prow(v.prows.size() + 3 * 8);
        const size_t nrows = v.prows.size() / 8;
        auto unify = [&](uint32_t t) { return t == 0xFFFFFFFFu ? t : (t & X_TMP) ? c->n_signals + (t & 0x3FFFFFFFu) : t; };
        for (size_t r = 0; r < nrows; r++) {
            const uint32_t *w = &v.prows[r * 8];
            uint32_t *o = &prow[r * 8];
            o[0] = w[0]; o[1] = w[1]; o[2] = w[2];
            o[3] = unify(w[3]); o[4] = unify(w[4]);
            o[5] = w[5]; o[6] = w[6]; o[7] = 0;
        }
        for (int k = 0; k < 3; k++) {
            uint32_t *o = &prow[(nrows + k) * 8];
            o[0] = D_NOP; o[1] = 0; o[2] = 0xFFu << 16; o[3] = o[4] = 0xFFFFFFFFu; o[5] = o[6] = o[7] = 0;
        }
        std::vector<uint32_t> pl(v.extras.size());
        for (size_t k = 0; k < v.extras.size(); k++) {
            const uint32_t lw = v.extras[k];
            pl[k] = lw == 0xFFFFFFFFu ? lw : (lw & 0x40000000u) ? (0x80000000u | (lw & 0x3FFFFFFFu)) : unify(lw);
        }
        std::vector<uint64_t> dterms(v.terms.size() / 2);
        for (size_t k = 0; k + 3 < v.terms.size(); k += 4) {
            const uint32_t kw = v.terms[k], kind = kw & 7;
            dterms[k / 2] = ((uint64_t)kind << 61) | ((uint64_t)v.terms[k + 1] * 2048);
            dterms[k / 2 + 1] = ((uint64_t)(kw >> 31) << 63) | ((uint64_t)v.terms[k + 3] << 32) | v.terms[k + 2];
        }
        TRY(upload(&b->d_prows, prow, b->stream));
        TRY(upload(&b->d_ploads, pl, b->stream));
        TRY(upload(&b->d_terms, dterms, b->stream));
        TRY(hipStreamSynchronize(b->stream));
    } else {
        // resolve the schedule for this batch: slot numbers -> byte offsets (CwDRow), streams padded with NOPs
        const Variant &v = *b->var;
        const uint64_t stride = (uint64_t)2 * b->Bp * 16;            // bytes per value slot
        // an LDS hand-over slot holds the lanes IN USE (32 bytes each): a workgroup of 16 instances needs 36 KB for 72 slots,
        // not 144 KB, so that the workgroups of several batches in flight share a CU (bench.py --in-flight)
        const uint64_t lds_slot = (uint64_t)b->lanes * 32;
        auto resolve = [&](uint32_t kind, uint32_t idx) -> uint64_t {
            switch (kind) {
            case K_SIG: return (uint64_t)idx * stride;
            case K_TMP: return ((uint64_t)c->n_signals + idx) * stride;
            case K_CONST: return (uint64_t)idx * 32;
            case K_LDS: return (uint64_t)idx * lds_slot;
            default: return 0;
            }
        };
        std::vector<CwDRow> drows;
        std::vector<uint32_t> doff(1, 0);
        std::vector<uint8_t> bits_entry(v.extras.size(), 0);          // extra-destination entries of D_BITS rows
        drows.reserve(v.rows.size() + 3 * v.n_strands);
        for (uint32_t st = 0; st < v.n_strands; st++) {
            size_t sq = v.seq_off[st], xq = v.extra_off[st];
            for (uint32_t r = v.stream_off[st]; r < v.stream_off[st + 1]; r++) {
                const CwRow &row = v.rows[r];
                uint32_t op = row.w0 & 0xFF, dk = (row.w0 >> SH_DK) & 7, ak = (row.w0 >> SH_AK) & 7,
                         bk = (row.w0 >> SH_BK) & 7;
                CwDRow d;
                d.w0 = row.w0;
                const bool can_fail = op == D_ASSERT_EQ || op == D_ASSERT_NZ || op == D_IDIV || op == D_MOD || op == D_CALL;
                const uint32_t seq = can_fail ? v.seqs[sq++] : 0;
                if (op == D_BARRIER) {
                    d.aux = row.dst;
                    d.dst_off = d.a_off = d.b_off = 0;
                } else if (op == D_LINSUM || op == D_DOTC) {
                    d.aux = row.a;                                   // number of terms
                    d.dst_off = dk == KD_NONE ? 0 : resolve(dk, row.dst);
                    d.a_off = 0;
                    d.b_off = resolve(bk, row.b);                    // constant term c0 (kind CONST) or nothing
                } else if (op == D_CALL) {
                    d.aux = row.a;                                   // function id
                    d.dst_off = seq;                                 // reported if the function fails (no destination)
                    d.a_off = 0;
                    d.b_off = resolve(K_TMP, row.b);                 // first register of the call's window
                } else if (op == D_BIT || op == D_BITS) {
                    d.aux = row.b;                                   // bit index k (D_BITS: of the first bit)
                    d.dst_off = dk == KD_NONE ? 0 : resolve(dk, row.dst);
                    d.a_off = resolve(ak, row.a);
                    d.b_off = 0;
                    if (op == D_BITS) {                              // its entries carry X_NEXT (bit 29 is not part of their slot number)
                        const uint32_t nx = (row.w0 >> SH_NX) & 0xFFF;
                        for (uint32_t e = 0; e < nx; e++) bits_entry[xq + e] = 1;
                    }
                } else {
                    d.aux = seq;                                     // flat operation, reported in the status word
                    d.dst_off = dk == KD_NONE ? 0 : resolve(dk, row.dst);
                    d.a_off = resolve(ak, row.a);
                    d.b_off = resolve(bk, row.b);
                }
                if (op != D_BARRIER) xq += (row.w0 >> SH_NX) & 0xFFF;
                drows.push_back(d);
            }
            doff.push_back((uint32_t)drows.size());
            for (int k = 0; k < 3; k++) drows.push_back(CwDRow{D_NOP, 0, 0, 0, 0});
        }
        // LINSUM terms: {kind<<61 | byte offset, sign<<63 | |coef|}
        std::vector<uint64_t> dterms(v.terms.size() / 2);
        for (size_t k = 0; k + 3 < v.terms.size(); k += 4) {
            uint32_t kw = v.terms[k], kind = kw & 7;
            dterms[k / 2] = ((uint64_t)kind << 61) | resolve(kind, v.terms[k + 1]);
            // LINSUM: sign | |coef| ; DOTC: index into the limb-form constant table (the kernel multiplies by 48)
            dterms[k / 2 + 1] = ((uint64_t)(kw >> 31) << 63) | ((uint64_t)v.terms[k + 3] << 32) | v.terms[k + 2];
        }
        std::vector<uint64_t> dex(v.extras.size() + 16, 0);            // (padded: a D_BITS row reads sixteen entries per trip)
        for (size_t k = 0; k < v.extras.size(); k++) {
            uint32_t x = v.extras[k];
            if (bits_entry[k])
                dex[k] = ((x & X_NEXT) ? X_NEXT_DEV : 0ull) | ((x & X_TMP) ? 0ull : X_LO_DEV) | resolve((x & X_TMP) ? K_TMP : K_SIG, x & 0x1FFFFFFFu);
            else if (x & X_LDS) dex[k] = (1ull << 63) | ((uint64_t)(x & 0x3FFFFFFFu) * lds_slot);
            else dex[k] = resolve((x & X_TMP) ? K_TMP : K_SIG, x & 0x3FFFFFFFu);
        }
        // stream offsets now refer to the padded array: stream s starts at doff[s] + 3*s ... keep explicit table
        std::vector<uint32_t> soff(v.n_strands + 1);
        for (uint32_t st = 0; st <= v.n_strands; st++) soff[st] = 0;
        {
            uint32_t pos = 0;
            for (uint32_t st = 0; st < v.n_strands; st++) {
                soff[st] = pos;
                pos += (v.stream_off[st + 1] - v.stream_off[st]) + 3;
            }
            soff[v.n_strands] = pos;
        }
        b->h_stream_begin = soff;
        // the kernel needs [begin, end) of real rows per stream: pass begin in stream_off[s] and end in a second table
        std::vector<uint32_t> tab(2 * v.n_strands);
        for (uint32_t st = 0; st < v.n_strands; st++) {
            tab[2 * st] = soff[st];
            tab[2 * st + 1] = soff[st] + (v.stream_off[st + 1] - v.stream_off[st]);
        }
        if (std::find(bits_entry.begin(), bits_entry.end(), (uint8_t)1) != bits_entry.end()) {
            // D_BITS rows store the lower half of their signal destinations only (cw_tape.h X_LO_DEV): the table starts cleared.
            // INVARIANT for the life of the batch: nothing else ever writes the upper 16 bytes of such a slot - every writer of
            // the value table is a row of this schedule (a slot has one producer: lower.py's clobber replay), inputs land in
            // input slots (never bit destinations), and the bit-plane fallback re-runs whole instances through these same rows
            if (c->mont) return fail(CW_EIO, "tape: bit-field rows in a schedule of Montgomery-form signals");
            TRY(hipMemsetAsync(b->d_V, 0, b->v_bytes, b->stream));
        }
        TRY(upload(&b->d_rows, drows, b->stream));
        TRY(upload(&b->d_stream_off, tab, b->stream));
        TRY(upload(&b->d_extras, dex, b->stream));
        TRY(upload(&b->d_extra_off, v.extra_off, b->stream));
        TRY(upload(&b->d_terms, dterms, b->stream));
        TRY(upload(&b->d_term_off, v.term_off, b->stream));
        TRY(hipStreamSynchronize(b->stream));                        // host vectors go out of scope
    }
    TRY(upload(&b->d_consts, c->consts, b->stream));
    TRY(upload(&b->d_fncode, c->fn_code, b->stream));
    TRY(upload(&b->d_fntab, c->fn_tab, b->stream));
    TRY(upload(&b->d_lconsts, c->lconsts, b->stream));
    TRY(upload(&b->d_w2s, c->w2s, b->stream));
    TRY(hipMalloc((void **)&b->d_status, (size_t)b->Bp * 4 * 2));      // second half: findings of the emitted code's fused R1CS check
    TRY(hipMalloc((void **)&b->d_first_bad, (size_t)b->Bp * 4));
    if (c->n_constraints) {
        TRY(upload(&b->d_rctab, c->r_ctab, b->stream));
        {   // the same coefficients as 9 x 29-bit limbs, the form the multiplier consumes
            std::vector<uint32_t> t29(c->r_ctab.size() / 8 * 9);
            for (size_t e = 0; e < c->r_ctab.size() / 8; e++) {
                uint64_t w[4];
                memcpy(w, &c->r_ctab[e * 8], 32);
                for (int k = 0; k < 9; k++) {
                    unsigned bit = 29 * k, wi = bit / 64, sh = bit % 64;
                    uint64_t v = w[wi] >> sh;
                    if (sh > 35 && wi + 1 < 4) v |= w[wi + 1] << (64 - sh);
                    t29[e * 9 + k] = (uint32_t)(v & 0x1FFFFFFFu);
                }
            }
            TRY(upload(&b->d_rctab29, t29, b->stream));
        }
        const char *mode = getenv("CW_R1CS_MODE");
        cwplan::Plan p;
        if (mode && !strcmp(mode, "staged")) {
            uint32_t chunks, entries;
            r1cs_plan_defaults(batch, &chunks, &entries);
            p = cwplan::build(c->r_ptr, c->r_slot, c->r_coef, c->r_orig, c->n_signals, chunks, entries);
            TRY(upload(&b->d_prec, p.rec, b->stream));
            b->r1_entries = p.entries;
        } else {
            uint32_t tpc = 192;
            if (const char *e = getenv("CW_R1CS_TERMS")) tpc = (uint32_t)std::max(8, atoi(e));
            const bool audit = getenv("CW_R1CS_AUDIT") != nullptr;
            p = cwplan::build_stream(c->r_ptr, c->r_slot, c->r_coef, c->r_orig, tpc, b->fp_fn && !audit ? b->fp_covered : nullptr,
                                     getenv("CW_R1CS_NO_BOOL") ? nullptr : &c->r_bool, getenv("CW_R1CS_NO_FOLD") == nullptr);
        }
        if (p.chunk.empty()) p.chunk.assign(4, 0);                   // every row is checked by the emitted code: nothing to stream
        if (p.row_orig.empty()) p.row_orig.assign(1, 0);
        TRY(upload(&b->d_pchunk, p.chunk, b->stream));
        TRY(upload(&b->d_pterms, p.terms, b->stream));
        TRY(upload(&b->d_prow, p.row_orig, b->stream));
        TRY(hipStreamSynchronize(b->stream));                        // the plan goes out of scope
        b->r1_chunks = p.n_chunks;
    }
    TRY(hipMalloc(&b->d_in, std::max<size_t>((size_t)batch * c->n_inputs * 32, 32)));
    TRY(hipMalloc(&b->d_gather, std::max<size_t>((size_t)c->n_witness * 32, 32)));
    TRY(cwk_init(b->stream, b->d_V, b->Bp, b->d_status, b->d_first_bad, c->mont, c->P));
#undef TRY
    b->remaining.assign(batch, c->n_inputs);
    *out = b;
    return CW_OK;
}
extern "C" int cw_batch_create(cw_circuit *c, int device, uint32_t batch, void *stream, cw_batch **out) {
    const char *e = getenv("CW_BITS");                // CW_BITS=0 forces the 256-bit schedule (diagnostics, A/B timing)
    return batch_create_impl(c, device, batch, stream, !(e && e[0] == '0'), out);
}
extern "C" int cw_batch_bitmode(const cw_batch *b) { return b && b->bitmode; }
extern "C" int cw_bits_info(const cw_circuit *c, uint64_t out[8]) {
    if (!c || !out) return fail(CW_EINVAL, "null argument");
    memset(out, 0, 64);
    if (!c->has_bits) return CW_OK;
    const cwbits::Program &bp = c->bits;
    // gate lanes = records that are not the idle pattern (all three operands the constant 0, result into the ring)
    const uint32_t const_off = (bp.ring + bp.cache) * 512u;
    uint64_t gates = 0, loads = 0, flushes = 0;
    for (size_t i = 0; i < (size_t)bp.n_vrows * 64; i++)
        if (bp.recs[i * 2] != (const_off | (const_off << 16)) || (bp.recs[i * 2 + 1] & 0xFFFFu) != const_off) gates++;
    for (size_t b = 0; b < bp.n_vrows / cwbits::BATCH; b++) {
        loads += bp.cmds[b * cwbits::CMD_WORDS] & 0xFFu;
        flushes += (bp.cmds[b * cwbits::CMD_WORDS] >> 8) & 0xFFu;
    }
    out[0] = 1; out[1] = bp.n_vrows; out[2] = bp.n_slots; out[3] = bp.ring; out[4] = gates; out[5] = loads; out[6] = flushes;
    out[7] = bp.cache;
    return CW_OK;
}
// host-only: shape of the R1CS check plan over the bit table (cw_bits_host.h::build_r1cs): how the slot assignment of
// the bit program serves the check (whole-word terms need 32 consecutive slots)
extern "C" int cw_bits_r1cs_plan_stats(const cw_circuit *c, uint64_t out[8]) {
    if (!c || !out) return fail(CW_EINVAL, "null argument");
    memset(out, 0, 64);
    if (!c->has_bits || !c->n_constraints) return CW_OK;
    cwbits::R1Plan p = cwbits::build_r1cs(c->r_ptr, c->r_slot, c->r_cc, c->r_cctab, c->r_orig, c->bits.sig_slot, c->q.w, 1024);
    out[0] = p.n_trivial; out[1] = p.n_lut; out[2] = p.n_int; out[3] = p.n_word_terms; out[4] = p.n_int_blocks;
    out[5] = p.n_contig_blocks; out[6] = p.n_wide; out[7] = p.iwords.size();
    return CW_OK;
}
extern "C" uint32_t cw_batch_size(const cw_batch *b) { return b->batch; }
extern "C" int cw_circuit_montgomery(const cw_circuit *c) { return c && c->mont ? 1 : 0; }
extern "C" uint32_t cw_batch_strands(const cw_batch *b) { return b->var ? b->var->n_strands : 0; }
extern "C" uint32_t cw_batch_pipelined(const cw_batch *b) { return b && b->var && b->var->kind == 1 ? b->var->nb | (b->var->nld << 8) : 0; }
extern "C" uint32_t cw_batch_emitted(const cw_batch *b) { return b && b->fp_fn ? (b->fp_covered ? 2 : 1) : 0; }
extern "C" uint32_t cw_batch_lanes(const cw_batch *b) { return b->bitmode ? b->bits_width : b->lanes; }

static int ensure_host_staging(cw_batch *b) {
    size_t n = (size_t)b->batch * b->c->n_inputs;
    if (b->h_in.size() != n * 32) b->h_in.assign(n * 32, 0);
    if (b->assigned.size() != n) b->assigned.assign(n, 0);
    return CW_OK;
}

// setInputSignal (calcwit.cpp:77-97) with the same four error conditions
static int set_input_hashed(cw_batch *b, uint32_t inst, uint64_t h, uint32_t idx, const uint8_t val[32],
                            const char *name_for_msg) {
    cw_circuit *c = b->c;
    if (inst >= b->batch) return fail(CW_EINVAL, "instance out of range");
    ensure_host_staging(b);
    if (b->remaining[inst] == 0) return fail(CW_EINPUT, "No more signals to be assigned");
    int64_t pos = hash_pos(c, h);
    if (pos < 0 || c->hashmap[pos].signalid == 0)
        return fail(CW_EINPUT, std::string("Signal not found: ") + name_for_msg);
    if (idx >= c->hashmap[pos].signalsize) return fail(CW_EINPUT, "Input signal array access exceeds the size");
    uint32_t si = (uint32_t)c->hashmap[pos].signalid + idx;
    uint32_t k = si - c->input_start;
    size_t cell = (size_t)inst * c->n_inputs + k;
    if (b->assigned[cell]) return fail(CW_EINPUT, "Signal assigned twice: " + std::to_string(si));
    memcpy(&b->h_in[cell * 32], val, 32);
    b->assigned[cell] = 1;
    b->remaining[inst]--;
    b->host_dirty = true;
    b->ext_in = nullptr;
    b->packed_in = nullptr;
    return CW_OK;
}

extern "C" int cw_set_input_signal(cw_batch *b, uint32_t instance, const char *name, uint32_t idx,
                                   const uint8_t val[32]) {
    if (!b || !name || !val) return fail(CW_EINVAL, "null argument");
    U256 v;
    memcpy(v.w, val, 32);
    if (u256_cmp(v, b->c->q) >= 0) return fail(CW_EINVAL, "input value is not reduced modulo the prime");
    return set_input_hashed(b, instance, fnv1a(name, strlen(name)), idx, val, name);
}

extern "C" int cw_get_staged_input(cw_batch *b, uint32_t instance, uint32_t k, uint8_t out[32]) {
    if (!b || !out || instance >= b->batch || k >= b->c->n_inputs) return fail(CW_EINVAL, "bad argument");
    size_t cell = (size_t)instance * b->c->n_inputs + k;
    if (b->assigned.size() <= cell || !b->assigned[cell]) return fail(CW_ESTATE, "input not assigned");
    memcpy(out, &b->h_in[cell * 32], 32);
    return CW_OK;
}

extern "C" int64_t cw_remaining_inputs(const cw_batch *b, uint32_t instance) {
    if (!b || instance >= b->batch) return -1;
    if (b->all_set) return 0;
    return b->remaining[instance];
}

extern "C" int cw_set_inputs(cw_batch *b, const uint8_t *le32) {
    if (!b || !le32) return fail(CW_EINVAL, "null argument");
    size_t n = (size_t)b->batch * b->c->n_inputs * 32;
    if (b->device < 0) {                                     // host-only batch: stage (every cell counts as assigned)
        ensure_host_staging(b);
        memcpy(b->h_in.data(), le32, n);
        std::fill(b->assigned.begin(), b->assigned.end(), 1);
        std::fill(b->remaining.begin(), b->remaining.end(), 0);
        b->all_set = true;
        b->host_dirty = true;
        b->ext_in = nullptr;
        return CW_OK;
    }
    HIPCHK(hipSetDevice(b->device));
    if (int rc = ensure_d_in(b)) return rc;
    HIPCHK(hipMemcpyAsync(b->d_in, le32, n, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));   // caller may free le32 on return
    b->packed_in = nullptr;
    b->ext_in = nullptr;
    b->host_dirty = false;
    b->all_set = true;
    std::fill(b->remaining.begin(), b->remaining.end(), 0);
    return CW_OK;
}

extern "C" int cw_set_inputs_bits_device(cw_batch *b, const void *d_masks) {
    if (!b || !d_masks) return fail(CW_EINVAL, "null argument");
    NOT_FOR_64(b, "packed boolean inputs");
    NEED_DEVICE(b);
    if (!b->bitmode) return fail(CW_ESTATE, "packed boolean inputs need a bit-plane batch (cw_batch_bitmode)");
    b->packed_in = d_masks;
    b->ext_in = nullptr;
    b->host_dirty = false;
    b->all_set = true;
    std::fill(b->remaining.begin(), b->remaining.end(), 0);
    return CW_OK;
}
extern "C" int cw_set_inputs_bits(cw_batch *b, const uint64_t *masks) {
    if (!b || !masks) return fail(CW_EINVAL, "null argument");
    NOT_FOR_64(b, "packed boolean inputs");
    NEED_DEVICE(b);
    if (!b->bitmode) return fail(CW_ESTATE, "packed boolean inputs need a bit-plane batch (cw_batch_bitmode)");
    HIPCHK(hipSetDevice(b->device));
    // a buffer of their own: the masks are 1 bit per input and instance, the 32-byte staging image (d_in) 256 times that - 226 GB
    // for 2^18 instances of the 27 008-input SHA-256, which the masks must not need
    const size_t nbytes = std::max<size_t>((size_t)b->n_groups * b->c->n_inputs * 8, 8);
    if (!b->d_pmask) {
        hipError_t e = hipMalloc(&b->d_pmask, nbytes);
        if (e != hipSuccess) return fail(CW_EDEVICE, "hipMalloc of the packed input masks failed (" + std::to_string(nbytes) + " bytes): " + hipGetErrorString(e));
    }
    HIPCHK(hipMemcpyAsync(b->d_pmask, masks, nbytes, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return cw_set_inputs_bits_device(b, b->d_pmask);
}

extern "C" int cw_set_inputs_device(cw_batch *b, const void *d_le32) {
    if (!b || !d_le32) return fail(CW_EINVAL, "null argument");
    b->packed_in = nullptr;
    b->ext_in = d_le32;
    b->host_dirty = false;
    b->all_set = true;
    std::fill(b->remaining.begin(), b->remaining.end(), 0);
    return CW_OK;
}

// ---------------------------------------------------------------------------------------------------------
// JSON ingest — loadJson / qualify_input / json2FrElements (main.cpp:144-286)
// ---------------------------------------------------------------------------------------------------------
struct JVal {
    enum T { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
    std::string s;          // STR: text, NUM: source token
    bool integral = false;  // NUM: token has no fraction/exponent
    std::vector<JVal> a;
    std::vector<std::pair<std::string, JVal>> o;
};
struct JParser {
    const char *p, *e;
    std::string err;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool parse(JVal &v) {
        ws();
        if (p >= e) return bad("unexpected end");
        char c = *p;
        if (c == '{') {
            v.t = JVal::OBJ;
            p++;
            ws();
            if (p < e && *p == '}') { p++; return true; }
            for (;;) {
                ws();
                JVal k;
                if (p >= e || *p != '"' || !str(k.s)) return bad("expected key");
                ws();
                if (p >= e || *p != ':') return bad("expected ':'");
                p++;
                JVal x;
                if (!parse(x)) return false;
                v.o.push_back({k.s, std::move(x)});
                ws();
                if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == '}') { p++; return true; }
                return bad("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.t = JVal::ARR;
            p++;
            ws();
            if (p < e && *p == ']') { p++; return true; }
            for (;;) {
                JVal x;
                if (!parse(x)) return false;
                v.a.push_back(std::move(x));
                ws();
                if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == ']') { p++; return true; }
                return bad("expected ',' or ']'");
            }
        }
        if (c == '"') { v.t = JVal::STR; return str(v.s); }
        if (c == 't' && e - p >= 4 && !memcmp(p, "true", 4)) { v.t = JVal::BOOL; p += 4; return true; }
        if (c == 'f' && e - p >= 5 && !memcmp(p, "false", 5)) { v.t = JVal::BOOL; p += 5; return true; }
        if (c == 'n' && e - p >= 4 && !memcmp(p, "null", 4)) { v.t = JVal::NUL; p += 4; return true; }
        if (c == '-' || (c >= '0' && c <= '9')) {
            const char *s0 = p;
            v.integral = true;
            if (*p == '-') p++;
            while (p < e && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) {
                if (*p == '.' || *p == 'e' || *p == 'E') v.integral = false;
                p++;
            }
            v.t = JVal::NUM;
            v.s.assign(s0, p - s0);
            return true;
        }
        return bad("unexpected character");
    }
    bool str(std::string &out) {
        p++;   // opening quote
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) {
                p++;
                switch (*p) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (e - p < 5) return bad("bad \\u escape");
                    unsigned cp = (unsigned)strtoul(std::string(p + 1, 4).c_str(), nullptr, 16);
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                    else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                    p += 4;
                    break;
                }
                default: out += *p;
                }
                p++;
            } else out += *p++;
        }
        if (p >= e) return bad("unterminated string");
        p++;
        return true;
    }
    bool bad(const char *m) { err = m; return false; }
};

// check_type (main.cpp:190-208): leaf kind of an array (numbers and strings count as the same kind)
static int leaf_kind(const JVal &v) {
    if (v.t != JVal::ARR) return (v.t == JVal::NUM || v.t == JVal::STR) ? 100 : (int)v.t;
    if (v.a.empty()) return (int)JVal::NUL;
    return leaf_kind(v.a[0]);
}
static void qualify(const std::string &prefix, const JVal &in, std::vector<std::pair<std::string, const JVal *>> &out);
static void qualify_list(const std::string &prefix, const JVal &in, std::vector<std::pair<std::string, const JVal *>> &out) {
    if (in.t == JVal::ARR) {
        for (size_t i = 0; i < in.a.size(); i++) qualify_list(prefix + "[" + std::to_string(i) + "]", in.a[i], out);
    } else qualify(prefix, in, out);
}
// qualify_input (main.cpp:221-241): nested objects / arrays of objects -> dotted, indexed keys
static void qualify(const std::string &prefix, const JVal &in, std::vector<std::pair<std::string, const JVal *>> &out) {
    if (in.t == JVal::ARR) {
        if (!in.a.empty() && leaf_kind(in) == (int)JVal::OBJ) qualify_list(prefix, in, out);
        else out.push_back({prefix, &in});
    } else if (in.t == JVal::OBJ) {
        for (auto &kv : in.o) qualify(prefix.empty() ? kv.first : prefix + "." + kv.first, kv.second, out);
    } else out.push_back({prefix, &in});
}
// json2FrElements (main.cpp:144-188)
static int json_to_fes(const JVal &v, const U256 &q, std::vector<U256> &out) {
    if (v.t == JVal::ARR) {
        for (auto &x : v.a) {
            int rc = json_to_fes(x, q, out);
            if (rc) return rc;
        }
        return CW_OK;
    }
    std::string s;
    unsigned base = 10;
    if (v.t == JVal::STR) {
        const std::string &sa = v.s;
        std::string pre = sa.substr(0, 2);
        if (pre == "0b" || pre == "0B") { s = sa.substr(2); base = 2; }
        else if (pre == "0o" || pre == "0O") { s = sa.substr(2); base = 8; }
        else if (pre == "0x" || pre == "0X") { s = sa.substr(2); base = 16; }
        else s = sa;
        // check_valid_number (main.cpp:126-142): digits only, no sign
        bool ok = true;
        for (char ch : s) {
            if (base == 16) ok &= (ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'f') || (ch >= 'A' && ch <= 'F');
            else ok &= (ch >= '0' && ch < (char)('0' + base));
        }
        if (!ok) return fail(CW_EINPUT, "Invalid number in JSON input: " + sa);
        if (s.empty()) { out.push_back(u256_zero()); return CW_OK; }   // mpz_init_set_str("") leaves 0
    } else if (v.t == JVal::NUM) {
        // the reference goes through double and prints it with fixed precision 0 (main.cpp:170-175)
        double vd = strtod(v.s.c_str(), nullptr);
        char buf[400];
        snprintf(buf, sizeof buf, "%.0f", vd);
        s = buf;
    } else return fail(CW_EINPUT, "Invalid JSON type");
    U256 x;
    if (!str_to_fe(s.data(), s.size(), base, q, &x)) return fail(CW_EINPUT, "Invalid number in JSON input: " + s);
    out.push_back(x);
    return CW_OK;
}

extern "C" int cw_set_inputs_json(cw_batch *b, uint32_t instance, const char *json_text) {
    if (!b || !json_text) return fail(CW_EINVAL, "null argument");
    JParser jp{json_text, json_text + strlen(json_text), ""};
    JVal root;
    if (!jp.parse(root)) return fail(CW_EINPUT, "JSON parse error: " + jp.err);
    std::vector<std::pair<std::string, const JVal *>> items;
    qualify("", root, items);
    // nlohmann's object is an ordered map: keys are visited in sorted order (main.cpp:261)
    std::stable_sort(items.begin(), items.end(), [](auto &x, auto &y) { return x.first < y.first; });
    cw_circuit *c = b->c;
    for (auto &it : items) {
        std::vector<U256> vals;
        int rc = json_to_fes(*it.second, c->q, vals);
        if (rc) return rc;
        uint64_t h = fnv1a(it.first.data(), it.first.size());
        int64_t pos = hash_pos(c, h);
        if (pos < 0 || c->hashmap[pos].signalid == 0) return fail(CW_EINPUT, "Signal not found: " + it.first);
        uint64_t sz = c->hashmap[pos].signalsize;
        if (vals.size() < sz) return fail(CW_EINPUT, "Error loading signal " + it.first + ": Not enough values");
        if (vals.size() > sz) return fail(CW_EINPUT, "Error loading signal " + it.first + ": Too many values");
        for (size_t i = 0; i < vals.size(); i++) {
            rc = set_input_hashed(b, instance, h, (uint32_t)i, (const uint8_t *)vals[i].w, it.first.c_str());
            if (rc) return fail(rc, "Error setting signal: " + it.first + "\n" + g_err);
        }
    }
    return CW_OK;
}

// ---------------------------------------------------------------------------------------------------------
// run / check / egress
// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// bit-plane mode
// ---------------------------------------------------------------------------------------------------------
#define BTRY(x)                                                                \
    do {                                                                       \
        hipError_t e2 = (x);                                                   \
        if (e2 != hipSuccess) return fail(CW_EDEVICE, std::string(#x ": ") + hipGetErrorString(e2)); \
    } while (0)

// 64-bit runtime (cw64.hip): value table [slot][Bp] of uint64, the flat program, the R1CS terms
static int batch_setup64(cw_batch *b) {
    cw_circuit *c = b->c;
    b->v_bytes = (size_t)c->n_slots64 * b->Bp * 8;
    hipError_t e = hipMalloc((void **)&b->d_V64, b->v_bytes);
    if (e != hipSuccess)
        return fail(CW_EDEVICE, "hipMalloc of the value table failed (" + std::to_string(b->v_bytes) + " bytes): " + hipGetErrorString(e));
    BTRY(hipMemsetAsync(b->d_V64, 0, b->v_bytes, b->stream));
    BTRY(upload(&b->d_rows64, c->rows64, b->stream));
    BTRY(upload(&b->d_consts64, c->consts64, b->stream));
    BTRY(upload(&b->d_terms64, c->r1_terms64, b->stream));
    BTRY(upload(&b->d_chunks64, c->r1_chunks64, b->stream));
    BTRY(upload(&b->d_w2s, c->w2s, b->stream));
    BTRY(hipMalloc((void **)&b->d_status, (size_t)b->Bp * 4));
    BTRY(hipMalloc((void **)&b->d_first_bad, (size_t)b->Bp * 4));
    BTRY(hipMalloc(&b->d_in, std::max<size_t>((size_t)b->batch * c->n_inputs * 32, 32)));
    BTRY(hipMalloc(&b->d_gather, std::max<size_t>((size_t)c->n_witness * 32, 32)));
    BTRY(cwk64_init(b->stream, b->d_V64, b->Bp, b->d_status, b->d_first_bad));
    BTRY(hipStreamSynchronize(b->stream));
    return CW_OK;
}

static int bits_batch_setup(cw_batch *b) {
    cw_circuit *c = b->c;
    const cwbits::Program &bp = c->bits;
    b->n_groups = (b->batch + 63) / 64;
    // Which engine: the interpreter (cw_bits_eval_kernel) finishes a batch of 65 536 in about a millisecond and scales
    // linearly beyond; the emitted code needs ~4 ms for its 1.5 M instructions whatever the batch (a wave = 2 048 instances)
    // and is then bound by the table's bytes: it wins from a few hundred thousand instances.  CW_BITS_JIT = 0 / 1 overrides.
    b->jit = c->has_jit && b->batch >= cwbits::JIT_MIN_BATCH;
    if (const char *ev = getenv("CW_BITS_JIT")) b->jit = c->has_jit && atoi(ev) != 0;
    if (b->jit) {
        auto it = c->jit_mod.find(b->device);
        if (it == c->jit_mod.end()) {
            hipModule_t mod = nullptr;
            hipFunction_t fn = nullptr;
            hipError_t e1 = hipModuleLoadData(&mod, c->jit.code.data());
            if (e1 == hipSuccess) e1 = hipModuleGetFunction(&fn, mod, cwbits::JIT_KERNEL);
            if (e1 != hipSuccess)
                return fail(CW_EDEVICE, std::string("loading the emitted bit-plane code failed: ") + hipGetErrorString(e1));
            it = c->jit_mod.emplace(b->device, std::make_pair(mod, fn)).first;
        }
        b->jit_fn = it->second.second;
        b->bits_slots = c->jit.n_slots;
        b->bits_sh = 5;
        b->bits_sigslot = &c->jit.sig_slot;
        b->n_groups_padded = (b->n_groups + 31) / 32 * 32;
    } else {
        b->bits_slots = bp.n_slots;
        b->bits_sh = 0;
        b->bits_sigslot = &bp.sig_slot;
        b->n_groups_padded = b->n_groups;
    }
    const std::vector<uint32_t> &sig_slot = *b->bits_sigslot;
    b->t_bytes = (uint64_t)b->n_groups_padded * b->bits_slots * 8;
    hipError_t e = hipMalloc((void **)&b->d_T, b->t_bytes);
    if (e != hipSuccess)
        return fail(CW_EDEVICE, "hipMalloc of the bit table failed (" + std::to_string(b->t_bytes) + " bytes): " + hipGetErrorString(e));
    BTRY(hipMalloc((void **)&b->d_fbmask, (size_t)b->n_groups_padded * 8));
    BTRY(hipMalloc((void **)&b->d_r1flag, (size_t)b->n_groups_padded * 8));
    if (!b->jit) {
        std::vector<uint32_t> dev, cmds;
        b->bits_steps = cwbits::device_stream(bp, dev, cmds);
        BTRY(upload(&b->d_brecs, dev, b->stream));
        BTRY(upload(&b->d_bcmds, cmds, b->stream));
        BTRY(upload(&b->d_aslots, bp.assert_slots, b->stream));
        BTRY(hipStreamSynchronize(b->stream));                       // the vectors go out of scope
    }
    // instances per wave: small batches are spread over more CUs by giving every group of 64 instances to 2 or 4
    // independent waves (each evaluates the whole program on its 32 / 16 bits of every mask)
    b->bits_width = b->n_groups * 4 <= 512 ? 16 : b->n_groups * 2 <= 512 ? 32 : 64;
    if (const char *ev = getenv("CW_BITS_WIDTH")) {
        const int w = atoi(ev);
        if (w == 16 || w == 32 || w == 64) b->bits_width = (uint32_t)w;
    }
    BTRY(upload(&b->d_w2s, c->w2s, b->stream));
    BTRY(upload(&b->d_sigslot, sig_slot, b->stream));
    {
        std::vector<uint32_t> wslot(c->w2s.size());
        for (size_t k = 0; k < wslot.size(); k++) wslot[k] = sig_slot[c->w2s[k]];
        BTRY(upload(&b->d_wslot, wslot, b->stream));
        BTRY(hipStreamSynchronize(b->stream));
    }
    BTRY(hipMalloc((void **)&b->d_status, (size_t)b->Bp * 4));
    BTRY(hipMalloc((void **)&b->d_first_bad, (size_t)b->Bp * 4));
    if (c->n_constraints) {
        uint32_t tpc = 1024;         // terms per chunk = per wave (measured on Sha256(2048) x 65 536: 256 -> 1.35 ms, 1024 -> 1.26, 4096 -> 1.32)
        if (const char *ev = getenv("CW_R1CS_TERMS")) tpc = (uint32_t)std::max(8, atoi(ev));
        cwbits::R1Plan p = cwbits::build_r1cs(c->r_ptr, c->r_slot, c->r_cc, c->r_cctab, c->r_orig, sig_slot, c->q.w, tpc);
        BTRY(upload(&b->d_erecs, p.erecs, b->stream));
        BTRY(upload(&b->d_wchunk, p.chunk, b->stream));
        BTRY(upload(&b->d_wterms, p.terms, b->stream));
        BTRY(upload(&b->d_wctab, p.ctab, b->stream));
        BTRY(upload(&b->d_wrow, p.row_orig, b->stream));
        BTRY(upload(&b->d_ichunk, p.ichunk, b->stream));
        BTRY(upload(&b->d_iterms, p.iwords, b->stream));
        BTRY(upload(&b->d_itab, p.itab, b->stream));
        BTRY(upload(&b->d_irow, p.irow_orig, b->stream));
        b->n_ichunks = p.n_ichunks;
        BTRY(hipStreamSynchronize(b->stream));                       // the plan goes out of scope
        b->n_evrows = p.n_evrows;
        if (getenv("CW_VERBOSE"))
            fprintf(stderr, "[cw] bit-plane R1CS plan: %llu trivial, %llu lut, %llu int (%llu terms in whole 32-bit words, %llu blocks of 8, "
                            "%llu contiguous), %llu field rows\n",
                    (unsigned long long)p.n_trivial, (unsigned long long)p.n_lut, (unsigned long long)p.n_int,
                    (unsigned long long)p.n_word_terms, (unsigned long long)p.n_int_blocks, (unsigned long long)p.n_contig_blocks,
                    (unsigned long long)p.n_wide);
        b->n_wchunks = p.n_chunks;
    }
    // the 32-byte staging image (d_in) of a bit-plane batch is allocated when a HOST-side setter first needs it: a caller that
    // hands over device buffers (cw_set_inputs_device / cw_set_inputs_bits_device) never does - 137 GB for 2 M instances of
    // Sha256(2048)
    BTRY(hipMalloc(&b->d_gather, std::max<size_t>((size_t)c->n_witness * 32, 32)));
    BTRY(cwk_bits_init(b->stream, b->d_T, b->bits_slots, b->bits_sh, b->n_groups_padded, b->d_fbmask, b->d_r1flag, b->d_status, b->d_first_bad, b->Bp));
    BTRY(hipStreamSynchronize(b->stream));
    b->fb_index.assign(b->batch, -1);
    return CW_OK;
}

static int bits_run(cw_batch *b, const void *in) {
    cw_circuit *c = b->c;
    const cwbits::Program &bp = c->bits;
    TMARK(b, 0);
    BTRY(cwk_bits_init(b->stream, b->d_T, b->bits_slots, b->bits_sh, b->n_groups_padded, b->d_fbmask, b->d_r1flag, b->d_status, b->d_first_bad, b->Bp));
    if (b->packed_in)
        BTRY(cwk_bits_ingest_packed(b->stream, b->packed_in, b->d_T, b->bits_slots, b->bits_sh, cwbits::IN_BASE, c->n_inputs, b->batch));
    else
        BTRY(cwk_bits_ingest(b->stream, in, b->d_T, b->bits_slots, b->bits_sh, cwbits::IN_BASE, c->n_inputs, b->batch, b->d_fbmask));
    TMARK(b, 1);
    if (b->jit) {
        // one wave per chunk of 2 048 instances runs the circuit's emitted code: gates on registers, every signal value stored
        // once, assertion gates and the fused R1CS check OR-ed into the two flag arrays
        b->jit_args = {b->d_T, b->d_fbmask, b->d_r1flag};
        void **cfg = b->jit_cfg;
        cfg[0] = HIP_LAUNCH_PARAM_BUFFER_POINTER; cfg[1] = &b->jit_args; cfg[2] = HIP_LAUNCH_PARAM_BUFFER_SIZE; cfg[3] = &b->jit_args_size;
        cfg[4] = HIP_LAUNCH_PARAM_END;
        BTRY(hipModuleLaunchKernel(b->jit_fn, b->n_groups_padded / 32, 1, 1, 64, 1, 1, 0, b->stream, nullptr, cfg));
        b->table_dirty = false;
    } else {
        BTRY(cwk_bits_eval(b->stream, b->d_brecs, b->d_bcmds, b->bits_steps, bp.ring, bp.cache, b->d_T, bp.n_slots, b->n_groups, b->bits_width,
                           b->d_aslots, (uint32_t)bp.assert_slots.size(), b->d_fbmask));
    }
    TMARK(b, 2);
    b->resolved = false;
    b->checked = false;
    return CW_OK;
}

// After the stream has drained: find the instances the bit-plane path could not serve (inputs other than 0/1, or an
// assertion gate fired) and compute them with the 256-bit schedule in a side batch.  Their status words, witnesses
// and public signals are then served from there, so every answer is the reference's for every input.
static int bits_resolve(cw_batch *b) {
    if (!b->bitmode || b->resolved || !b->ran) return CW_OK;
    cw_circuit *c = b->c;
    BTRY(hipStreamSynchronize(b->stream));
    std::vector<uint64_t> m(b->n_groups);
    BTRY(hipMemcpy(m.data(), b->d_fbmask, (size_t)b->n_groups * 8, hipMemcpyDeviceToHost));
    b->fb_inst.clear();
    std::fill(b->fb_index.begin(), b->fb_index.end(), -1);
    for (uint32_t g = 0; g < b->n_groups; g++)
        for (uint64_t x = m[g]; x; x &= x - 1) {
            const uint32_t i = g * 64 + (uint32_t)__builtin_ctzll(x);
            if (i < b->batch) b->fb_inst.push_back(i);
        }
    // when a large share of the batch is not boolean (a circuit class the bit program does not suit, or a caller feeding
    // field-valued inputs) the whole batch goes through the 256-bit schedule: one dense side batch instead of a sparse one
    if (b->fb_inst.size() > b->batch / 4) {
        b->fb_inst.resize(b->batch);
        for (uint32_t i = 0; i < b->batch; i++) b->fb_inst[i] = i;
    }
    const size_t n_fb = b->fb_inst.size();
    for (size_t k = 0; k < n_fb; k++) b->fb_index[b->fb_inst[k]] = (int32_t)k;
    // the side batch is kept while it is large enough and not more than 4x too large (its tail then repeats the first
    // instance): a caller whose batches have a few odd instances each does not pay a table allocation per run
    if (b->fb && (n_fb == 0 || b->fb->batch < n_fb || b->fb->batch > 4 * n_fb)) {
        cw_batch_free(b->fb);
        b->fb = nullptr;
    }
    if (n_fb) {
        if (!b->fb) {
            int rc = batch_create_impl(c, b->device, (uint32_t)n_fb, b->stream, false, &b->fb);
            if (rc != CW_OK) return rc;
        }
        std::vector<uint32_t> inst(b->fb_inst);
        inst.resize(b->fb->batch, b->fb_inst[0]);
        if (b->fbinst_cap < inst.size()) {
            if (b->d_fbinst) hipFree(b->d_fbinst);
            b->d_fbinst = nullptr;
            b->fbinst_cap = 0;
            BTRY(hipMalloc((void **)&b->d_fbinst, inst.size() * 4));
            b->fbinst_cap = (uint32_t)inst.size();
        }
        BTRY(hipMemcpyAsync(b->d_fbinst, inst.data(), inst.size() * 4, hipMemcpyHostToDevice, b->stream));
        // one kernel gathers the inputs of the listed instances (from the packed masks or the 32-byte image the caller
        // handed over: that buffer must stay unmodified until the first cw_sync / getter after cw_run, see circom_amd.h)
        BTRY(cwk_bits_collect_inputs(b->stream, b->packed_in, b->packed_in ? nullptr : (b->ext_in ? b->ext_in : b->d_in), b->d_fbinst,
                                     (uint32_t)inst.size(), c->n_inputs, b->fb->d_in));
        BTRY(hipStreamSynchronize(b->stream));                       // `inst` goes out of scope
        b->fb->ext_in = nullptr;
        b->fb->host_dirty = false;
        b->fb->all_set = true;
        std::fill(b->fb->remaining.begin(), b->fb->remaining.end(), 0);
        int rc = cw_run(b->fb);
        if (rc == CW_OK && b->checked && c->n_constraints) rc = cw_check_r1cs(b->fb);
        if (rc != CW_OK) return rc;
        BTRY(hipStreamSynchronize(b->stream));
    }
    b->resolved = true;
    return CW_OK;
}

extern "C" int cw_run(cw_batch *b) {
    if (!b) return fail(CW_EINVAL, "null batch");
    NEED_DEVICE(b);
    cw_circuit *c = b->c;
    if (!b->all_set) {
        uint64_t missing = 0;
        for (uint32_t r : b->remaining) missing += r;
        if (missing) {
            // main.cpp:352-355
            return fail(CW_ESTATE, "Not all inputs have been set. " + std::to_string(missing) + " values missing over the batch");
        }
    }
    HIPCHK(hipSetDevice(b->device));
    if (b->host_dirty) {
        if (int rc = ensure_d_in(b)) return rc;
        HIPCHK(hipMemcpyAsync(b->d_in, b->h_in.data(), b->h_in.size(), hipMemcpyHostToDevice, b->stream));
        b->host_dirty = false;
    }
    const void *in = b->ext_in ? b->ext_in : b->d_in;
    if (c->is64) {
        TMARK(b, 0);
        HIPCHK(cwk64_init(b->stream, b->d_V64, b->Bp, b->d_status, b->d_first_bad));
        HIPCHK(cwk64_ingest(b->stream, in, b->d_V64, c->input_start, c->n_inputs, b->batch, b->Bp));
        TMARK(b, 1);
        HIPCHK(cwk64_eval(b->stream, b->d_rows64, (uint32_t)(c->rows64.size() / 8), b->d_consts64, b->d_V64, b->Bp, b->batch, b->d_status));
        TMARK(b, 2);
        b->ran = true;
        return CW_OK;
    }
    if (b->bitmode) {
        int rc = bits_run(b, in);
        if (rc == CW_OK) b->ran = true;
        return rc;
    }
    TMARK(b, 0);
    HIPCHK(cwk_init(b->stream, b->d_V, b->Bp, b->d_status, b->d_first_bad, c->mont, c->P));
    HIPCHK(cwk_ingest(b->stream, in, b->d_V, c->input_start, c->n_inputs, b->batch, b->Bp, c->mont, c->P));
    TMARK(b, 1);
    if (b->var->kind == 1) {
        HIPCHK(cwk_eval_pipe(b->stream, c->need_full, false, b->var->nb, b->var->nld, b->d_prows, (uint32_t)(b->var->prows.size() / 8),
                             b->d_ploads, b->d_terms, b->d_V, b->d_consts, b->d_lconsts, (uint64_t)2 * b->Bp * 16, b->Bp, b->batch,
                             b->lanes, b->d_status, c->P));
        TMARK(b, 2);
        b->ran = true;
        return CW_OK;
    }
    if (b->fp_fn) {
        // the variant's rows as straight-line code: one workgroup of n_strands waves per `lanes` instances, as cwk_eval
        cw_batch::FpArgs &args = b->fp_args;                        // (the tables of tier 2: the D_CALL body loads their addresses)
        static_assert(sizeof(args) == 272, "argument block of the emitted code (fpjit.KERNARG_BYTES)");
        static_assert(sizeof(FpParams) == 53 * 4, "the emitted code loads 53 parameter words");
        memset(&args, 0, sizeof(args));
        args.V = b->d_V;
        args.status = b->d_status;
        args.Bp = b->Bp;
        args.batch = b->batch;
        args.lanes = b->lanes;
        args.P = c->P;
        args.consts = b->d_consts;
        args.fcode = b->d_fncode;
        args.ftab = b->d_fntab;
        HIPCHK(cwk_fill32(b->stream, b->d_status + b->Bp, 0xFFFFFFFFu, b->Bp));               // "no constraint found violated"
        void **cfg = b->fp_cfg;
        cfg[0] = HIP_LAUNCH_PARAM_BUFFER_POINTER; cfg[1] = &b->fp_args; cfg[2] = HIP_LAUNCH_PARAM_BUFFER_SIZE; cfg[3] = &b->fp_args_size;
        cfg[4] = HIP_LAUNCH_PARAM_END;
        HIPCHK(hipModuleLaunchKernel(b->fp_fn, (b->batch + b->lanes - 1) / b->lanes, 1, 1, 64 * b->var->n_strands, 1, 1, 0, b->stream,
                                     nullptr, cfg));
        TMARK(b, 2);
        b->ran = true;
        return CW_OK;
    }
    HIPCHK(cwk_eval(b->stream, c->need_full, b->var->wide_linsum, b->d_rows, b->d_stream_off, b->d_extras, b->d_extra_off, b->d_terms,
                    b->d_term_off, b->var->n_strands, b->var->n_lds, b->d_V, b->d_consts, b->d_lconsts,
                    c->fn_tab.empty() ? nullptr : b->d_fncode /* non-null selects the single-wave interpreter build */, b->d_fntab,
                    (uint64_t)2 * b->Bp * 16, b->Bp, b->batch, b->lanes, b->prio_mask, b->d_status, c->P));
    TMARK(b, 2);
    b->ran = true;
    return CW_OK;
}

extern "C" int cw_batch_set_timing(cw_batch *b, int on) {
    if (!b) return fail(CW_EINVAL, "null batch");
    NEED_DEVICE(b);
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (on) {
        const size_t want = on >= 2 ? CW_TIMING_RING : 1;
        if (b->tring.size() < want) b->tring.resize(want);
        for (cw_batch::TSet &t : b->tring)
            for (hipEvent_t &e : t.ev)
                if (!e) HIPCHK(hipEventCreate(&e));
        if (on < 2 && b->tring.size() > 1) {               // back to "last run only": keep one set
            for (size_t k = 1; k < b->tring.size(); k++)
                for (hipEvent_t e : b->tring[k].ev)
                    if (e) hipEventDestroy(e);
            b->tring.resize(1);
        }
    }
    b->timing = on != 0;
    b->tcur = 0;
    b->tfresh = true;
    for (cw_batch::TSet &t : b->tring)
        for (bool &x : t.set) x = false;
    return CW_OK;
}

static const int cw_tpairs[3][2] = {{0, 1}, {1, 2}, {3, 4}};

// ms[0] = table init + input ingest, ms[1] = the evaluation kernel(s), ms[2] = cw_check_r1cs, of the LAST run / check of the batch
// (the stream is drained first); a part that has not run since timing was switched on reads -1
extern "C" int cw_batch_kernel_ms(cw_batch *b, float ms[3]) {
    if (!b || !ms) return fail(CW_EINVAL, "cw_batch_kernel_ms: bad argument");
    NEED_DEVICE(b);
    if (!b->timing) return fail(CW_ESTATE, "cw_batch_kernel_ms: timing is off (cw_batch_set_timing)");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    const cw_batch::TSet &ts = b->tring[b->tcur];
    for (int k = 0; k < 3; k++) {
        ms[k] = -1.0f;
        if (ts.set[cw_tpairs[k][0]] && ts.set[cw_tpairs[k][1]]) {
            float t = 0;
            if (hipEventElapsedTime(&t, ts.ev[cw_tpairs[k][0]], ts.ev[cw_tpairs[k][1]]) == hipSuccess) ms[k] = t;
        }
    }
    return CW_OK;
}

// the same three parts averaged over every run recorded since cw_batch_set_timing(b, 2) (at most the last CW_TIMING_RING);
// counts[k] = the runs part k was averaged over (0: ms[k] = -1)
extern "C" int cw_batch_kernel_ms_mean(cw_batch *b, float ms[3], int counts[3]) {
    if (!b || !ms || !counts) return fail(CW_EINVAL, "cw_batch_kernel_ms_mean: bad argument");
    NEED_DEVICE(b);
    if (!b->timing) return fail(CW_ESTATE, "cw_batch_kernel_ms_mean: timing is off (cw_batch_set_timing)");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    for (int k = 0; k < 3; k++) {
        double sum = 0;
        counts[k] = 0;
        for (const cw_batch::TSet &ts : b->tring)
            if (ts.set[cw_tpairs[k][0]] && ts.set[cw_tpairs[k][1]]) {
                float t = 0;
                if (hipEventElapsedTime(&t, ts.ev[cw_tpairs[k][0]], ts.ev[cw_tpairs[k][1]]) == hipSuccess) { sum += t; counts[k]++; }
            }
        ms[k] = counts[k] ? (float)(sum / counts[k]) : -1.0f;
    }
    return CW_OK;
}

extern "C" int cw_check_r1cs(cw_batch *b) {
    if (!b) return fail(CW_EINVAL, "null batch");
    cw_circuit *c = b->c;
    NEED_DEVICE(b);
    if (!b->ran) return fail(CW_ESTATE, "cw_check_r1cs before cw_run");
    if (c->n_constraints == 0) return fail(CW_ESTATE, "no .r1cs was loaded for this circuit");
    HIPCHK(hipSetDevice(b->device));
    if (c->is64) {
        TMARK(b, 3);
        HIPCHK(cwk64_r1cs(b->stream, b->d_chunks64, (uint32_t)(c->r1_chunks64.size() / 4), b->d_terms64, b->d_V64, b->Bp, b->batch, b->d_status,
                          b->d_first_bad));
        TMARK(b, 4);
        return CW_OK;
    }
    if (b->bitmode) {
        // emitted code checked every constraint on its registers while it generated the witness: only the groups it flagged
        // are audited (to name the first violated row of each instance).  A caller that took the raw table pointer
        // (cw_device_bits) may have changed it: then, and with CW_R1CS_AUDIT=1, every group is audited from the table.
        const void *only = b->jit && c->jit.check_complete && !b->table_dirty && !getenv("CW_R1CS_AUDIT") ? b->d_r1flag : nullptr;
        TMARK(b, 3);
        if (!only && b->jit && c->jit.check_complete && !c->jit.audit_code.empty() && !getenv("CW_R1CS_AUDIT_GENERAL")) {
            // the audit as emitted code: every constraint recomputed from the table's rows (one coalesced row per wire and wave),
            // flags into the (cleared) R1CS flag array; the general kernels below then only name the first bad row of the
            // instances it flagged
            auto it = c->jit_audit_mod.find(b->device);
            if (it == c->jit_audit_mod.end()) {
                hipModule_t mod = nullptr;
                hipFunction_t fn = nullptr;
                hipError_t e1 = hipModuleLoadData(&mod, c->jit.audit_code.data());
                if (e1 == hipSuccess) e1 = hipModuleGetFunction(&fn, mod, cwbits::JIT_KERNEL);
                if (e1 != hipSuccess) return fail(CW_EDEVICE, std::string("loading the emitted audit code failed: ") + hipGetErrorString(e1));
                it = c->jit_audit_mod.emplace(b->device, std::make_pair(mod, fn)).first;
            }
            HIPCHK(cwk_fill32(b->stream, (uint32_t *)b->d_r1flag, 0u, (size_t)b->n_groups_padded * 2));
            b->audit_args = {b->d_T, b->d_fbmask, b->d_r1flag};
            void **cfg = b->audit_cfg;
            cfg[0] = HIP_LAUNCH_PARAM_BUFFER_POINTER; cfg[1] = &b->audit_args; cfg[2] = HIP_LAUNCH_PARAM_BUFFER_SIZE; cfg[3] = &b->jit_args_size;
            cfg[4] = HIP_LAUNCH_PARAM_END;
            HIPCHK(hipModuleLaunchKernel(it->second.second, b->n_groups_padded / 32, 1, 1, 64, 1, 1, 0, b->stream, nullptr, cfg));
            only = b->d_r1flag;
        }
        HIPCHK(cwk_bits_r1cs(b->stream, b->d_erecs, b->n_evrows, b->d_wchunk, b->n_wchunks, b->d_wterms, b->d_wctab, b->d_wrow,
                             b->d_ichunk, b->n_ichunks, b->d_iterms, b->d_itab, b->d_irow, b->d_T, b->bits_slots, b->bits_sh, only,
                             b->n_groups, b->batch, b->d_status, b->d_first_bad, c->P));
        TMARK(b, 4);
        b->checked = true;
        if (b->resolved && b->fb) return cw_check_r1cs(b->fb);       // the side batch was already computed: check it too
        return CW_OK;
    }
    TMARK(b, 3);
    // rows the emitted evaluation code recomputed itself (hip_elements/fpjit.py plan_checks): its findings join the words the
    // stand-alone kernel reports through; that kernel then only streams the rows the code left to it
    if (b->fp_fn) HIPCHK(cwk_fused_merge(b->stream, b->d_status + b->Bp, b->batch, b->d_status, b->d_first_bad));
    if (b->r1_entries)
        HIPCHK(cwk_r1cs_staged(b->stream, b->d_pchunk, b->r1_chunks, b->d_prec, b->d_pterms, b->d_rctab, b->d_rctab29, b->d_prow,
                               b->r1_entries, b->d_V, b->Bp, b->batch, b->d_status, b->d_first_bad, c->mont, c->P));
    else
        HIPCHK(cwk_r1cs(b->stream, b->d_pchunk, b->r1_chunks, b->d_pterms, b->d_rctab, b->d_rctab29, b->d_prow, b->d_V, b->Bp, b->batch,
                        b->d_status, b->d_first_bad, c->mont, c->P));
    TMARK(b, 4);
    return CW_OK;
}

// cw_run + cw_check_r1cs as ONE launch.  A step of a small batch is a dozen launches of microseconds each (table init, ingest,
// evaluation, two or three check kernels, a merge); one graph launch instead is worth +6 .. +15 % on Sha256(512) x 4 096 (35.7 ->
// 41.1 M witnesses/s with 16 hardware queues - the rest of that step's 0.10 ms is the device's, profiles/r06s_*).  Both calls only
// enqueue work on the batch's stream - KERNELS only, see cwk_fill32 - so the second call with unchanged input pointers records them
// with stream capture (on a private stream: the batch's may be the null stream), and every later call replays the graph on the
// batch's own stream.  Falls back to the two plain calls while timing marks are on (events are not captured), while inputs set on
// the host wait for their copy, and for good after a capture error.
extern "C" int cw_run_check(cw_batch *b) {
    if (!b) return fail(CW_EINVAL, "null batch");
    NEED_DEVICE(b);
    cw_circuit *c = b->c;
    if (c->n_constraints == 0) return fail(CW_ESTATE, "no .r1cs was loaded for this circuit");
    auto plain = [&]() {
        int rc = cw_run(b);
        return rc == CW_OK ? cw_check_r1cs(b) : rc;
    };
    const void *in = b->ext_in ? b->ext_in : b->d_in;
    if (b->rc_graph && (b->rc_in != in || b->rc_packed != b->packed_in || b->timing || b->host_dirty)) {
        hipGraphExecDestroy(b->rc_graph);
        b->rc_graph = nullptr;
        b->rc_calls = 0;
    }
    if (b->timing || b->host_dirty || b->rc_failed || getenv("CW_NO_GRAPH")) return plain();
    if (!b->rc_graph) {
        if (b->rc_in != in || b->rc_packed != b->packed_in) {
            b->rc_in = in;
            b->rc_packed = b->packed_in;
            b->rc_calls = 0;
        }
        if (b->rc_calls++ == 0) return plain();           // loads the code objects, makes every lazy allocation
        HIPCHK(hipSetDevice(b->device));
        hipGraph_t g = nullptr;
        if (!b->rc_stream && hipStreamCreateWithFlags(&b->rc_stream, hipStreamNonBlocking) != hipSuccess) {
            b->rc_stream = nullptr;
            b->rc_failed = true;
            (void)hipGetLastError();
            return plain();
        }
        if (hipStreamBeginCapture(b->rc_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            b->rc_failed = true;
            (void)hipGetLastError();
            return plain();
        }
        hipStream_t own = b->stream;
        b->stream = b->rc_stream;                         // every launch of the two calls names b->stream
        const int rc = plain();
        b->stream = own;
        const hipError_t e = hipStreamEndCapture(b->rc_stream, &g);
        if (rc != CW_OK || e != hipSuccess || !g) {
            if (g) hipGraphDestroy(g);
            (void)hipGetLastError();
            b->rc_failed = rc == CW_OK;                   // (an error of the calls themselves is the caller's to see, every time)
            return rc != CW_OK ? rc : plain();
        }
        const hipError_t e2 = hipGraphInstantiate(&b->rc_graph, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (e2 != hipSuccess) {
            b->rc_failed = true;
            b->rc_graph = nullptr;
            (void)hipGetLastError();
            return plain();
        }
    }
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipGraphLaunch(b->rc_graph, b->stream));
    b->ran = true;
    if (b->bitmode) {
        if (b->jit) b->table_dirty = false;
        b->resolved = false;
        b->checked = true;
    }
    return CW_OK;
}
extern "C" int cw_batch_graph_captured(const cw_batch *b) { return b && b->rc_graph != nullptr; }

extern "C" int cw_sync(cw_batch *b) {
    if (!b) return fail(CW_EINVAL, "null batch");
    NEED_DEVICE(b);
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    return bits_resolve(b);
}

// status words / first bad rows of the instances that were re-run by the 256-bit schedule come from the side batch
static int bits_patch_words(cw_batch *b, uint32_t *dst, bool first_bad) {
    if (!b->bitmode || b->fb_inst.empty()) return CW_OK;
    std::vector<uint32_t> w(b->fb->batch);
    int rc = first_bad ? cw_get_r1cs_first_bad(b->fb, w.data()) : cw_get_status(b->fb, w.data());
    if (rc != CW_OK) return rc;
    for (size_t k = 0; k < b->fb_inst.size(); k++) dst[b->fb_inst[k]] = w[k];
    return CW_OK;
}

extern "C" int cw_get_status(cw_batch *b, uint32_t *status) {
    if (!b || !status) return fail(CW_EINVAL, "null argument");
    NEED_DEVICE(b);
    HIPCHK(hipSetDevice(b->device));
    if (int rc = bits_resolve(b)) return rc;
    HIPCHK(hipMemcpyAsync(status, b->d_status, (size_t)b->batch * 4, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return bits_patch_words(b, status, false);
}
extern "C" int cw_get_r1cs_first_bad(cw_batch *b, uint32_t *row) {
    if (!b || !row) return fail(CW_EINVAL, "null argument");
    NEED_DEVICE(b);
    HIPCHK(hipSetDevice(b->device));
    if (int rc = bits_resolve(b)) return rc;
    HIPCHK(hipMemcpyAsync(row, b->d_first_bad, (size_t)b->batch * 4, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return bits_patch_words(b, row, true);
}

extern "C" int cw_get_witness(cw_batch *b, uint32_t instance, uint8_t *out) {
    if (!b || !out) return fail(CW_EINVAL, "null argument");
    if (instance >= b->batch) return fail(CW_EINVAL, "instance out of range");
    NEED_DEVICE(b);
    if (!b->ran) return fail(CW_ESTATE, "cw_get_witness before cw_run");
    cw_circuit *c = b->c;
    HIPCHK(hipSetDevice(b->device));
    if (c->is64) {
        HIPCHK(cwk64_gather(b->stream, b->d_V64, b->d_w2s, c->n_witness, b->Bp, instance, 1, b->d_gather));
    } else if (b->bitmode) {
        if (int rc = bits_resolve(b)) return rc;
        if (b->fb_index[instance] >= 0) return cw_get_witness(b->fb, (uint32_t)b->fb_index[instance], out);
        HIPCHK(cwk_bits_gather(b->stream, b->d_T, b->bits_slots, b->bits_sh, b->d_wslot, c->n_witness, instance, 1, b->d_gather));
    } else
    HIPCHK(cwk_gather(b->stream, b->d_V, b->d_w2s, c->n_witness, b->Bp, instance, b->d_gather, c->mont, c->P));
    HIPCHK(hipMemcpyAsync(out, b->d_gather, (size_t)c->n_witness * 32, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return CW_OK;
}

// Bulk form: `count` instances starting at `first`, [count][n_witness][32 B], transposed on the device and
// copied in pieces of at most 256 MiB through a staging buffer that is allocated on first use.
extern "C" int cw_get_witnesses(cw_batch *b, uint32_t first, uint32_t count, uint8_t *out) {
    if (!b || !out) return fail(CW_EINVAL, "null argument");
    if ((uint64_t)first + count > b->batch) return fail(CW_EINVAL, "instance range out of the batch");
    NEED_DEVICE(b);
    if (!b->ran) return fail(CW_ESTATE, "cw_get_witnesses before cw_run");
    cw_circuit *c = b->c;
    HIPCHK(hipSetDevice(b->device));
    if (int rc = bits_resolve(b)) return rc;
    const size_t row = (size_t)c->n_witness * 32;
    uint32_t per = (uint32_t)std::max<size_t>(1, std::min<size_t>(count, ((size_t)256 << 20) / std::max<size_t>(row, 1)));
    per = std::max<uint32_t>(64, per / 64 * 64);
    if (b->bulk_rows < per) {
        if (b->d_bulk) hipFree(b->d_bulk);
        b->d_bulk = nullptr;
        b->bulk_rows = 0;
        HIPCHK(hipMalloc(&b->d_bulk, (size_t)per * row));
        b->bulk_rows = per;
    }
    for (uint32_t done = 0; done < count; done += per) {
        const uint32_t n = std::min(per, count - done);
        if (c->is64)
            HIPCHK(cwk64_gather(b->stream, b->d_V64, b->d_w2s, c->n_witness, b->Bp, first + done, n, b->d_bulk));
        else if (b->bitmode)
            HIPCHK(cwk_bits_gather(b->stream, b->d_T, b->bits_slots, b->bits_sh, b->d_wslot, c->n_witness, first + done, n, b->d_bulk));
        else
            HIPCHK(cwk_gather_many(b->stream, b->d_V, b->d_w2s, c->n_witness, b->Bp, first + done, n, b->d_bulk, c->mont, c->P));
        HIPCHK(hipMemcpyAsync(out + (size_t)done * row, b->d_bulk, (size_t)n * row, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    if (b->bitmode)
        for (size_t k = 0; k < b->fb_inst.size();) {              // runs of consecutive re-run instances: one bulk call each
            size_t e = k + 1;
            while (e < b->fb_inst.size() && b->fb_inst[e] == b->fb_inst[e - 1] + 1) e++;
            const uint32_t lo = std::max(b->fb_inst[k], first), hi = (uint32_t)std::min<uint64_t>((uint64_t)b->fb_inst[e - 1] + 1, (uint64_t)first + count);
            if (lo < hi)
                if (int rc = cw_get_witnesses(b->fb, (uint32_t)(k + (lo - b->fb_inst[k])), hi - lo, out + (size_t)(lo - first) * row)) return rc;
            k = e;
        }
    return CW_OK;
}

// Device-side form for GPU provers: canonical 32-byte values of `count` instances written to DEVICE memory
// ([count][n_witness][32]); no host copy.  In bit-plane batches this is where a bit becomes a field element again.
extern "C" int cw_get_witnesses_device(cw_batch *b, uint32_t first, uint32_t count, void *d_out) {
    if (!b || !d_out) return fail(CW_EINVAL, "null argument");
    NOT_FOR_64(b, "cw_get_witnesses_device");
    if ((uint64_t)first + count > b->batch) return fail(CW_EINVAL, "instance range out of the batch");
    NEED_DEVICE(b);
    if (!b->ran) return fail(CW_ESTATE, "cw_get_witnesses_device before cw_run");
    cw_circuit *c = b->c;
    HIPCHK(hipSetDevice(b->device));
    if (!b->bitmode) {
        HIPCHK(cwk_gather_many(b->stream, b->d_V, b->d_w2s, c->n_witness, b->Bp, first, count, d_out, c->mont, c->P));
        return CW_OK;
    }
    // instances the bit-plane program could not serve (inputs that are not 0/1, assertion gates) are known only after
    // the evaluation: the first egress of a run waits for it (one stream synchronisation + an 8-byte-per-group copy),
    // then everything below is asynchronous on the batch's stream
    if (int rc = bits_resolve(b)) return rc;
    // the whole batch went through the 256-bit schedule (more than a quarter of it was not boolean: fb_inst is the identity):
    // the side batch serves the range in ONE gather; the bit table is not consulted
    if (b->resolved && b->fb && b->fb_inst.size() == b->batch) return cw_get_witnesses_device(b->fb, first, count, d_out);
    HIPCHK(cwk_bits_gather(b->stream, b->d_T, b->bits_slots, b->bits_sh, b->d_wslot, c->n_witness, first, count, d_out));
    if (b->resolved) {
        // re-run instances: consecutive positions of the side batch that are consecutive instances leave in one gather
        const size_t row = (size_t)c->n_witness * 32;
        for (size_t k = 0; k < b->fb_inst.size();) {
            size_t e = k + 1;
            while (e < b->fb_inst.size() && b->fb_inst[e] == b->fb_inst[e - 1] + 1) e++;
            const uint32_t lo = std::max(b->fb_inst[k], first), hi = std::min<uint64_t>((uint64_t)b->fb_inst[e - 1] + 1, (uint64_t)first + count);
            if (lo < hi) {
                int rc = cw_get_witnesses_device(b->fb, (uint32_t)(k + (lo - b->fb_inst[k])), hi - lo, (char *)d_out + (size_t)(lo - first) * row);
                if (rc != CW_OK) return rc;
            }
            k = e;
        }
    }
    return CW_OK;
}

// Chunked device-side egress for provers: the canonical image of `count` instances does not fit anywhere for a
// million-signal circuit (32 MB per instance), so it is produced `chunk` instances at a time into two caller-owned
// device buffers in turn; after each chunk's transpose has been ENQUEUED on the batch's stream, `consume` is called with
// that stream: work the consumer enqueues on it (or on its own stream behind an event recorded on it) sees the chunk
// complete, and the library's next write to the same buffer - two chunks later - is ordered behind that work.
// Another batch object of the same circuit, on another stream, can be evaluating the next inputs meanwhile (bench.py).
extern "C" int cw_stream_witnesses_device(cw_batch *b, uint32_t first, uint32_t count, uint32_t chunk, void *d_buf0, void *d_buf1,
                                          cw_chunk_fn consume, void *user) {
    if (!b || !d_buf0 || !d_buf1 || !consume || chunk == 0) return fail(CW_EINVAL, "null argument / zero chunk");
    NOT_FOR_64(b, "cw_stream_witnesses_device");
    if ((uint64_t)first + count > b->batch) return fail(CW_EINVAL, "instance range out of the batch");
    uint32_t n_chunk = 0;
    for (uint32_t done = 0; done < count; done += chunk, n_chunk++) {
        const uint32_t n = std::min(chunk, count - done);
        void *buf = (n_chunk & 1) ? d_buf1 : d_buf0;
        int rc = cw_get_witnesses_device(b, first + done, n, buf);
        if (rc != CW_OK) return rc;
        rc = consume(user, first + done, n, buf, (void *)b->stream);
        if (rc != 0) return fail(CW_ESTATE, "the chunk consumer returned " + std::to_string(rc));
    }
    return CW_OK;
}

// Public signals of every instance, [batch][n_public][32], written to DEVICE memory: what a multi-GPU job gathers to
// the root next to the status words and what a verifier needs (snarkjs public.json).
extern "C" int cw_get_public_device(cw_batch *b, void *d_out) {
    if (!b || !d_out) return fail(CW_EINVAL, "null argument");
    NEED_DEVICE(b);
    if (!b->ran) return fail(CW_ESTATE, "cw_get_public_device before cw_run");
    cw_circuit *c = b->c;
    const uint32_t np = cw_n_public(c);
    if (np == 0) return CW_OK;
    if (np >= c->n_witness) return fail(CW_ESTATE, "public signal count exceeds the witness");
    HIPCHK(hipSetDevice(b->device));
    if (b->bitmode) {
        if (int rc = bits_resolve(b)) return rc;
        HIPCHK(cwk_bits_gather(b->stream, b->d_T, b->bits_slots, b->bits_sh, b->d_wslot + 1, np, 0, b->batch, d_out));
        if (!b->fb_inst.empty()) {
            void *tmp = nullptr;
            const size_t prow = (size_t)np * 32;
            HIPCHK(hipMalloc(&tmp, (size_t)b->fb->batch * prow));
            int rc = cw_get_public_device(b->fb, tmp);
            for (size_t k = 0; rc == CW_OK && k < b->fb_inst.size(); k++)
                if (hipMemcpyAsync((char *)d_out + (size_t)b->fb_inst[k] * prow, (char *)tmp + k * prow, prow, hipMemcpyDeviceToDevice,
                                   b->stream) != hipSuccess)
                    rc = fail(CW_EDEVICE, "copy of re-run public signals failed");
            hipStreamSynchronize(b->stream);
            hipFree(tmp);
            return rc;
        }
        return CW_OK;
    }
    if (c->is64) {
        HIPCHK(cwk64_gather(b->stream, b->d_V64, b->d_w2s + 1, np, b->Bp, 0, b->batch, d_out));
        return CW_OK;
    }
    HIPCHK(cwk_gather_many(b->stream, b->d_V, b->d_w2s + 1, np, b->Bp, 0, b->batch, d_out, c->mont, c->P));
    return CW_OK;
}
extern "C" int cw_get_public(cw_batch *b, uint8_t *out) {
    if (!b || !out) return fail(CW_EINVAL, "null argument");
    NEED_DEVICE(b);
    const size_t bytes = (size_t)b->batch * cw_n_public(b->c) * 32;
    if (bytes == 0) return CW_OK;
    void *d = nullptr;
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipMalloc(&d, bytes));
    int rc = cw_get_public_device(b, d);
    if (rc == CW_OK) {
        hipError_t e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        if (e != hipSuccess) rc = fail(CW_EDEVICE, hipGetErrorString(e));
    }
    hipFree(d);
    return rc;
}

extern "C" int cw_get_signal(cw_batch *b, uint32_t instance, uint32_t slot, uint8_t out[32]) {
    if (!b || !out) return fail(CW_EINVAL, "null argument");
    if (instance >= b->batch || slot >= b->c->n_signals) return fail(CW_EINVAL, "instance or slot out of range");
    NEED_DEVICE(b);
    HIPCHK(hipSetDevice(b->device));
    if (b->c->is64) {
        memset(out, 0, 32);
        HIPCHK(hipMemcpyAsync(out, b->d_V64 + (size_t)slot * b->Bp + instance, 8, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        return CW_OK;
    }
    if (b->bitmode) {
        if (int rc = bits_resolve(b)) return rc;
        if (b->fb_index[instance] >= 0) return cw_get_signal(b->fb, (uint32_t)b->fb_index[instance], slot, out);
        uint64_t m = 0;
        const uint32_t g = instance >> 6, sh = b->bits_sh;
        const size_t at = ((((size_t)(g >> sh) * b->bits_slots) + (*b->bits_sigslot)[slot]) << sh) + (g & ((1u << sh) - 1u));
        HIPCHK(hipMemcpyAsync(&m, b->d_T + at, 8, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        memset(out, 0, 32);
        out[0] = (uint8_t)((m >> (instance & 63)) & 1);
        return CW_OK;
    }
    const uint8_t *V = (const uint8_t *)b->d_V;
    size_t base = ((size_t)slot * 2 * b->Bp + instance) * 16;
    HIPCHK(hipMemcpyAsync(out, V + base, 16, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipMemcpyAsync(out + 16, V + base + (size_t)b->Bp * 16, 16, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->c->mont) {                       // x R' -> x: halve CW_RBITS times modulo q
        U256 x;
        memcpy(x.w, out, 32);
        x = shrmod(x, CW_RBITS, b->c->q);
        memcpy(out, x.w, 32);
    }
    return CW_OK;
}

// writeBinWitness (main.cpp:288-334)
extern "C" int cw_write_wtns(cw_batch *b, uint32_t instance, const char *path) {
    if (!b || !path) return fail(CW_EINVAL, "null argument");
    cw_circuit *c = b->c;
    std::vector<uint8_t> w((size_t)c->n_witness * 32);
    int rc = cw_get_witness(b, instance, w.data());
    if (rc) return rc;
    FILE *f = fopen(path, "wb");
    if (!f) return fail(CW_EIO, std::string("cannot open for writing: ") + path);
    // n8 = the prime's byte length: 32 for the 4-limb primes, 8 for the 64-bit runtime (common64/main.cpp writeBinWitness)
    uint32_t version = 2, nsec = 2, id1 = 1, n8 = c->is64 ? 8 : 32, id2 = 2, nw = c->n_witness;
    uint64_t len1 = 8 + n8, len2 = (uint64_t)n8 * nw;
    fwrite("wtns", 4, 1, f);
    fwrite(&version, 4, 1, f);
    fwrite(&nsec, 4, 1, f);
    fwrite(&id1, 4, 1, f);
    fwrite(&len1, 8, 1, f);
    fwrite(&n8, 4, 1, f);
    fwrite(c->q.w, n8, 1, f);
    fwrite(&nw, 4, 1, f);
    fwrite(&id2, 4, 1, f);
    fwrite(&len2, 8, 1, f);
    if (c->is64)
        for (uint32_t k = 0; k < nw; k++) fwrite(&w[(size_t)k * 32], 8, 1, f);
    else
        fwrite(w.data(), 1, w.size(), f);
    fclose(f);
    return CW_OK;
}

// The whole batch as ONE compact container (`<name>.wtnsb`, format below; reader + expander: circom_amd/wtnsb.py).  The
// reference's product is one `.wtns` per witness (main.cpp:288-334: 32 bytes per element); for a boolean circuit that image
// is 256x the information (Sha256(2048): 32 MB per instance against 18.7 KB of bit planes), and producing it bounds a
// bit-plane batch to ~150 K witnesses/s however fast it was generated.  A prover that ingests batches takes the container
// and widens the elements it needs where it needs them; every `.wtns` of the batch is recoverable from it bit for bit.
//   "wtnb" | u32 version = 1 | u32 kind (0 = field elements, 1 = bit planes) | u32 n8 | prime, n8 bytes | u32 n_witness | u32 batch
//   kind 0:  batch x n_witness x n8 bytes, canonical little-endian, instance-major (= the section-2 bodies of the .wtns files)
//   kind 1:  u64 slots | u32 shift | u32 groups | n_witness x u32 (slot of every witness element) |
//            groups x slots x u64: the bit table; element (group g, slot s) = word (((g >> shift) * slots + s) << shift) +
//            (g & ((1 << shift) - 1)), bit i = instance 64 g + i; slot 0 / 1 = the constants 0 / 1 |
//            u32 n_wide | n_wide x { u32 instance | n_witness x n8 bytes }: instances re-run by the 256-bit schedule (inputs
//            that are not 0/1, tripped assertions) carry their field elements
extern "C" int cw_write_wtnsb(cw_batch *b, const char *path) {
    if (!b || !path) return fail(CW_EINVAL, "null argument");
    NOT_FOR_64(b, "cw_write_wtnsb");
    NEED_DEVICE(b);
    if (!b->ran) return fail(CW_ESTATE, "cw_write_wtnsb before cw_run");
    cw_circuit *c = b->c;
    HIPCHK(hipSetDevice(b->device));
    if (int rc = bits_resolve(b)) return rc;
    FILE *f = fopen(path, "wb");
    if (!f) return fail(CW_EIO, std::string("cannot open for writing: ") + path);
    const uint32_t version = 1, kind = b->bitmode ? 1u : 0u, n8 = 32, nw = c->n_witness, batch = b->batch;
    bool ok = fwrite("wtnb", 4, 1, f) == 1;
    ok &= fwrite(&version, 4, 1, f) == 1 && fwrite(&kind, 4, 1, f) == 1 && fwrite(&n8, 4, 1, f) == 1;
    ok &= fwrite(c->q.w, 32, 1, f) == 1 && fwrite(&nw, 4, 1, f) == 1 && fwrite(&batch, 4, 1, f) == 1;
    const size_t row = (size_t)nw * 32;
    int rc = CW_OK;
    if (!b->bitmode) {
        const uint32_t per = (uint32_t)std::max<size_t>(1, std::min<size_t>(batch, ((size_t)256 << 20) / std::max<size_t>(row, 1)));
        std::vector<uint8_t> buf((size_t)per * row);
        for (uint32_t done = 0; done < batch && rc == CW_OK && ok; done += per) {
            const uint32_t n = std::min(per, batch - done);
            rc = cw_get_witnesses(b, done, n, buf.data());
            if (rc == CW_OK) ok &= fwrite(buf.data(), row, n, f) == n;
        }
    } else {
        const uint64_t slots = b->bits_slots;
        const uint32_t shift = b->bits_sh, groups = b->n_groups_padded;
        ok &= fwrite(&slots, 8, 1, f) == 1 && fwrite(&shift, 4, 1, f) == 1 && fwrite(&groups, 4, 1, f) == 1;
        std::vector<uint32_t> wslot(nw);
        for (uint32_t k = 0; k < nw; k++) wslot[k] = (*b->bits_sigslot)[c->w2s[k]];
        ok &= fwrite(wslot.data(), 4, nw, f) == nw;
        const size_t total = (size_t)b->t_bytes, piece = (size_t)256 << 20;
        std::vector<uint8_t> buf(std::min(total, piece));
        for (size_t at = 0; at < total && ok; at += piece) {
            const size_t n = std::min(piece, total - at);
            hipError_t e = hipMemcpy(buf.data(), (const uint8_t *)b->d_T + at, n, hipMemcpyDeviceToHost);
            if (e != hipSuccess) {
                fclose(f);
                return fail(CW_EDEVICE, std::string("copying the bit table: ") + hipGetErrorString(e));
            }
            ok &= fwrite(buf.data(), 1, n, f) == n;
        }
        const uint32_t n_wide = (uint32_t)b->fb_inst.size();
        ok &= fwrite(&n_wide, 4, 1, f) == 1;
        std::vector<uint8_t> w(row);
        for (uint32_t k = 0; k < n_wide && rc == CW_OK && ok; k++) {
            rc = cw_get_witness(b, b->fb_inst[k], w.data());
            if (rc == CW_OK) ok &= fwrite(&b->fb_inst[k], 4, 1, f) == 1 && fwrite(w.data(), 1, row, f) == row;
        }
    }
    ok &= fclose(f) == 0;
    if (rc != CW_OK) return rc;
    return ok ? CW_OK : fail(CW_EIO, std::string("short write: ") + path);
}

// Many instances -> many .wtns files (one bulk device transpose per 256 MiB instead of one gather per instance):
// `pattern` is a printf pattern with one %u / %d (the instance number).  What a prover farm consumes.
extern "C" int cw_write_wtns_many(cw_batch *b, uint32_t first, uint32_t count, const char *pattern) {
    if (!b || !pattern) return fail(CW_EINVAL, "null argument");
    NOT_FOR_64(b, "cw_write_wtns_many");
    if ((uint64_t)first + count > b->batch) return fail(CW_EINVAL, "instance range out of the batch");
    {   // exactly one integer conversion, nothing else
        int convs = 0;
        for (const char *p = pattern; *p; p++)
            if (*p == '%') {
                if (p[1] == '%') { p++; continue; }
                const char *q = p + 1;
                while (*q >= '0' && *q <= '9') q++;
                if (*q != 'u' && *q != 'd') return fail(CW_EINVAL, "pattern may only hold one %u / %d conversion");
                convs++;
            }
        if (convs != 1) return fail(CW_EINVAL, "pattern must hold exactly one %u / %d conversion");
    }
    cw_circuit *c = b->c;
    const size_t row = (size_t)c->n_witness * 32;
    const uint32_t per = (uint32_t)std::max<size_t>(1, std::min<size_t>(count, ((size_t)256 << 20) / std::max<size_t>(row, 1)));
    std::vector<uint8_t> buf((size_t)per * row);
    for (uint32_t done = 0; done < count; done += per) {
        const uint32_t n = std::min(per, count - done);
        int rc = cw_get_witnesses(b, first + done, n, buf.data());
        if (rc) return rc;
        for (uint32_t k = 0; k < n; k++) {
            char path[4096];
            snprintf(path, sizeof path, pattern, first + done + k);
            FILE *f = fopen(path, "wb");
            if (!f) return fail(CW_EIO, std::string("cannot open for writing: ") + path);
            uint32_t version = 2, nsec = 2, id1 = 1, n8 = 32, id2 = 2, nw = c->n_witness;
            uint64_t len1 = 8 + n8, len2 = (uint64_t)n8 * nw;
            fwrite("wtns", 4, 1, f); fwrite(&version, 4, 1, f); fwrite(&nsec, 4, 1, f);
            fwrite(&id1, 4, 1, f); fwrite(&len1, 8, 1, f); fwrite(&n8, 4, 1, f); fwrite(c->q.w, 32, 1, f); fwrite(&nw, 4, 1, f);
            fwrite(&id2, 4, 1, f); fwrite(&len2, 8, 1, f);
            fwrite(buf.data() + (size_t)k * row, 1, row, f);
            fclose(f);
        }
    }
    return CW_OK;
}

// Debug trace of one instance (the reference prints template, line and the component trace of a failed `===` and
// aborts: c_code_generator.rs:461-468, calcwit.cpp:104-114).  Here: status word decoded, and for a violated
// constraint its index, every wire with its .sym name (constraint_writers sym format: "s,w,c,name" per line) and value.
static std::string u256_dec(const uint8_t le[32]) {
    uint32_t limb[8];
    memcpy(limb, le, 32);
    std::string out;
    bool nz = true;
    while (nz) {
        uint64_t rem = 0;
        nz = false;
        for (int i = 7; i >= 0; i--) {
            uint64_t cur = (rem << 32) | limb[i];
            limb[i] = (uint32_t)(cur / 1000000000u);
            rem = cur % 1000000000u;
            if (limb[i]) nz = true;
        }
        char tmp[16];
        snprintf(tmp, sizeof tmp, nz ? "%09u" : "%u", (unsigned)rem);
        out = std::string(tmp) + out;
    }
    return out;
}
extern "C" int cw_explain(cw_batch *b, uint32_t instance, const char *sym_path, char *out, size_t out_len) {
    if (!b || !out || out_len == 0) return fail(CW_EINVAL, "null argument");
    NOT_FOR_64(b, "cw_explain");
    if (instance >= b->batch) return fail(CW_EINVAL, "instance out of range");
    cw_circuit *c = b->c;
    std::vector<uint32_t> st(b->batch), fb(b->batch);
    int rc = cw_get_status(b, st.data());
    if (rc == CW_OK) rc = cw_get_r1cs_first_bad(b, fb.data());
    if (rc) return rc;
    std::vector<std::string> names;
    if (sym_path) {
        std::vector<uint8_t> buf;
        if (!read_file(sym_path, buf)) return fail(CW_EIO, std::string(".sym file not found: ") + sym_path);
        names.assign(c->n_signals, std::string());
        size_t i = 0;
        while (i < buf.size()) {
            size_t e = i;
            while (e < buf.size() && buf[e] != '\n') e++;
            std::string line((const char *)buf.data() + i, e - i);
            i = e + 1;
            size_t c1 = line.find(','), c2 = line.find(',', c1 + 1), c3 = line.find(',', c2 + 1);
            if (c1 == std::string::npos || c2 == std::string::npos || c3 == std::string::npos) continue;
            unsigned long sid = strtoul(line.c_str(), nullptr, 10);
            if (sid < names.size()) names[sid] = line.substr(c3 + 1);
        }
    }
    auto name_of = [&](uint32_t sgn) {
        if (sgn == 0) return std::string("one");
        if (sgn < names.size() && !names[sgn].empty()) return names[sgn];
        return "signal " + std::to_string(sgn);
    };
    std::string t = "instance " + std::to_string(instance) + ": ";
    const uint32_t s = st[instance];
    if (s == 0) t += "ok\n";
    if (s & CW_ST_ASSERT_FAILED) t += "a run-time check (=== / assert) failed: operation " + std::to_string(s >> 8) + " of the witness program (the first failing one in program order)\n";
    if (s & CW_ST_ARITH) t += "integer division or modulo by zero (or a run-away function): operation " + std::to_string(s >> 8) + " of the witness program\n";
    if (s & CW_ST_R1CS_FAILED) {
        const uint32_t k = fb[instance];
        t += "constraint " + std::to_string(k) + " of the .r1cs is violated: A*B - C != 0 with\n";
        for (size_t j = 0; j < c->r_orig.size(); j++) {
            if ((c->r_orig[j] & 0x7FFFFFFFu) != k) continue;
            for (int part = 0; part < 3; part++) {
                t += std::string("  ") + "ABC"[part] + ":";
                for (uint32_t q = c->r_ptr[3 * j + part]; q < c->r_ptr[3 * j + part + 1]; q++) {
                    uint8_t v[32];
                    rc = cw_get_signal(b, instance, c->r_slot[q], v);
                    if (rc) return rc;
                    t += " " + name_of(c->r_slot[q]) + " = " + u256_dec(v) + ";";
                }
                t += "\n";
            }
            break;
        }
    }
    snprintf(out, out_len, "%s", t.c_str());
    return CW_OK;
}

// What the reference binary prints on stdout for this instance (LogBucket code, log_bucket.rs:105-162: arguments through
// printf, values as Fr_element2str = canonical residue in decimal, one blank between arguments, newline at the end), up
// to the first failed run-time check (the reference process exits there: statements that end behind that operation of the
// witness program are not printed).  Returns the length of the text (without the terminator) or a negative error code;
// at most out_len - 1 characters are stored.
// A log string reaches the reference binary as printf("<string>") (log_bucket.rs:126-133): the text is used as a FORMAT, so
// "%%" prints one '%'.  (The string table holds the TEXT - the front-end resolved the source's escapes, and the emitters write
// it back as a C literal - so only printf's own rule is applied here; any other conversion would read printf arguments that
// do not exist: undefined in the reference, passed through here.)
static std::string log_literal(const std::string &t) {
    std::string o;
    for (size_t i = 0; i < t.size(); i++) {
        if (t[i] == '%' && i + 1 < t.size() && t[i + 1] == '%') i++;
        o += t[i];
    }
    return o;
}

extern "C" int64_t cw_get_log(cw_batch *b, uint32_t instance, char *out, size_t out_len) {
    if (!b || (!out && out_len)) return fail(CW_EINVAL, "null argument");
    if (instance >= b->batch) return fail(CW_EINVAL, "instance out of range");
    cw_circuit *c = b->c;
    std::string t;
    if (!c->logs.empty()) {
        // circuits with log statements never run as bit-plane batches (compiler.lower_bitplane), so the status word of ONE
        // instance is one 4-byte copy (the CLI asks per instance: fetching the whole vector made a batch O(batch^2))
        NEED_DEVICE(b);
        if (!b->ran) return fail(CW_ESTATE, "cw_get_log before cw_run");
        if (b->bitmode) return fail(CW_ESTATE, "log output of a bit-plane batch");
        uint32_t s = 0;
        HIPCHK(hipSetDevice(b->device));
        HIPCHK(hipMemcpyAsync(&s, b->d_status + instance, 4, hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        const uint32_t stop = (s & (CW_ST_ASSERT_FAILED | CW_ST_ARITH)) ? (s >> 8) : 0xFFFFFFFFu;
        for (const auto &stmt : c->logs) {
            if (stmt.at >= stop) break;
            for (size_t k = 0; k < stmt.items.size(); k++) {
                const auto &it = stmt.items[k];
                if (it.is_value) {
                    uint8_t v[32];
                    if (int rc = cw_get_signal(b, instance, c->n_signals - c->n_logv + it.value, v)) return rc;
                    t += u256_dec(v);
                } else t += log_literal(it.text);
                if (k + 1 < stmt.items.size()) t += " ";
            }
            t += "\n";
        }
    }
    if (out_len) snprintf(out, out_len, "%s", t.c_str());
    return (int64_t)t.size();
}

extern "C" void *cw_device_values(cw_batch *b, uint64_t *n_bytes, uint32_t *padded_batch) {
    if (!b) return nullptr;
    if (b->bitmode) {                       // no 256-bit table exists: see cw_device_bits
        if (n_bytes) *n_bytes = 0;
        if (padded_batch) *padded_batch = b->Bp;
        return nullptr;
    }
    if (n_bytes) *n_bytes = b->v_bytes;
    if (padded_batch) *padded_batch = b->Bp;
    return b->d_V;
}

extern "C" const uint32_t *cw_signal_slots(const cw_circuit *c) { return c && c->has_bits ? c->bits.sig_slot.data() : nullptr; }
extern "C" void *cw_device_bits(cw_batch *b, uint64_t *n_bytes, uint64_t *slots_per_group) {
    if (!b || !b->bitmode) return nullptr;
    if (n_bytes) *n_bytes = b->t_bytes;
    if (slots_per_group) *slots_per_group = b->bits_slots;
    if (b->rc_graph) {                      // (cw_run_check's graph holds the check that trusts the table)
        hipGraphExecDestroy(b->rc_graph);
        b->rc_graph = nullptr;
        b->rc_calls = 0;
    }
    b->table_dirty = true;                  // the caller may write through the pointer: the next R1CS check audits the table itself
    return b->d_T;
}
extern "C" int cw_batch_bits_layout(const cw_batch *b, uint64_t out[4]) {
    if (!b || !out) return fail(CW_EINVAL, "null argument");
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!b->bitmode) return CW_OK;
    out[0] = b->bits_slots;
    out[1] = b->bits_sh;
    out[2] = b->n_groups_padded;
    out[3] = b->jit ? 1 : 0;
    return CW_OK;
}
extern "C" const uint32_t *cw_batch_signal_slots(const cw_batch *b) { return b && b->bitmode && b->bits_sigslot ? b->bits_sigslot->data() : nullptr; }

// ---------------------------------------------------------------------------------------------------------
// field micro-benchmark and unit-test hook
// ---------------------------------------------------------------------------------------------------------
// Measurement hook (tools/bits_shape_bench.py): time the bit-plane evaluation kernel on an arbitrary (validated) program
// - synthetic programs separate what a vrow costs from what THIS circuit's operand pattern adds (LDS bank conflicts,
// row traffic).  The table is zero-filled; results are not read back.
extern "C" int cw_bits_eval_bench(int device, uint32_t ring, uint32_t cache, uint32_t n_vrows, uint64_t n_slots, const uint32_t *recs,
                                  const uint32_t *cmds, uint32_t n_groups, uint32_t width, uint32_t iters, float *ms) {
    if (!recs || !cmds || !ms || n_groups == 0 || iters == 0 || (width != 16 && width != 32 && width != 64)) return fail(CW_EINVAL, "bad argument");
    cwbits::Program p;
    p.ring = ring;
    p.cache = cache;
    p.n_vrows = n_vrows;
    p.n_slots = n_slots;
    p.recs.assign(recs, recs + (size_t)n_vrows * 64 * 2);
    p.cmds.assign(cmds, cmds + (size_t)(n_vrows / cwbits::BATCH) * cwbits::CMD_WORDS);
    if (n_vrows % cwbits::BATCH) return fail(CW_EINVAL, "not a whole number of batches");
    if (const char *why = cwbits::validate(p, 0, 0)) return fail(CW_EINVAL, why);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(CW_EDEVICE, "no HIP device available");
    HIPCHK(hipSetDevice(device));
    std::vector<uint32_t> dev, dcmds;
    const uint32_t steps = cwbits::device_stream(p, dev, dcmds);
    uint32_t *d_recs = nullptr, *d_cmds = nullptr;
    void *d_T = nullptr;
    HIPCHK(upload(&d_recs, dev, nullptr));
    HIPCHK(upload(&d_cmds, dcmds, nullptr));
    HIPCHK(hipMalloc(&d_T, (size_t)n_groups * n_slots * 8));
    HIPCHK(hipMemset(d_T, 0, (size_t)n_groups * n_slots * 8));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(cwk_bits_eval(nullptr, d_recs, d_cmds, steps, ring, cache, d_T, n_slots, n_groups, width, nullptr, 0, nullptr));
    float t = 0;
    if (getenv("CW_BENCH_COLD")) {
        // every launch behind 6 GB of unrelated writes (what the ingest and check kernels of a real step leave in the
        // caches and TLBs): the launches are timed one by one
        void *d_junk = nullptr;
        const size_t junk = (size_t)6 << 30;
        HIPCHK(hipMalloc(&d_junk, junk));
        for (uint32_t i = 0; i < iters; i++) {
            HIPCHK(hipMemsetAsync(d_junk, (int)i, junk, nullptr));
            HIPCHK(hipEventRecord(e0, nullptr));
            HIPCHK(cwk_bits_eval(nullptr, d_recs, d_cmds, steps, ring, cache, d_T, n_slots, n_groups, width, nullptr, 0, nullptr));
            HIPCHK(hipEventRecord(e1, nullptr));
            HIPCHK(hipEventSynchronize(e1));
            float ti = 0;
            HIPCHK(hipEventElapsedTime(&ti, e0, e1));
            t += ti;
        }
        hipFree(d_junk);
    } else {
        HIPCHK(hipEventRecord(e0, nullptr));
        for (uint32_t i = 0; i < iters; i++)
            HIPCHK(cwk_bits_eval(nullptr, d_recs, d_cmds, steps, ring, cache, d_T, n_slots, n_groups, width, nullptr, 0, nullptr));
        HIPCHK(hipEventRecord(e1, nullptr));
        HIPCHK(hipEventSynchronize(e1));
        HIPCHK(hipEventElapsedTime(&t, e0, e1));
    }
    *ms = t / iters;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d_recs);
    hipFree(d_cmds);
    hipFree(d_T);
    return CW_OK;
}

extern "C" int cw_fp_mul_bench(const uint8_t prime_le32[32], int device, uint32_t n, uint32_t iters, const uint8_t *a,
                               const uint8_t *b, uint8_t *out, float *ms) {
    if (!prime_le32 || !a || !b || !out || n == 0) return fail(CW_EINVAL, "bad argument");
    U256 q;
    memcpy(q.w, prime_le32, 32);
    if (!prime_supported(q)) return fail(CW_EINVAL, "unsupported prime (need an odd prime of 225..256 bits)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(CW_EDEVICE, "no HIP device available");
    HIPCHK(hipSetDevice(device));
    FpParams P = make_params(q);
    void *da, *db, *dout;
    size_t bytes = (size_t)n * 32;
    HIPCHK(hipMalloc(&da, bytes));
    HIPCHK(hipMalloc(&db, bytes));
    HIPCHK(hipMalloc(&dout, bytes));
    HIPCHK(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, b, bytes, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(cwk_mulbench(nullptr, da, db, dout, n, 1, P));   // warm-up
    HIPCHK(hipEventRecord(e0, nullptr));
    HIPCHK(cwk_mulbench(nullptr, da, db, dout, n, iters, P));
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, e0, e1));
    if (ms) *ms = t;
    HIPCHK(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(da);
    hipFree(db);
    hipFree(dout);
    return CW_OK;
}

extern "C" int cw_fp_op(const uint8_t prime_le32[32], int device, uint32_t dop, uint32_t n, const uint8_t *a,
                        const uint8_t *b, const uint8_t *c, uint8_t *out, uint32_t *status) {
    if (!prime_le32 || !a || !b || !c || !out || !status || n == 0) return fail(CW_EINVAL, "bad argument");
    U256 q;
    memcpy(q.w, prime_le32, 32);
    if (!prime_supported(q)) return fail(CW_EINVAL, "unsupported prime (need an odd prime of 225..256 bits)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(CW_EDEVICE, "no HIP device available");
    HIPCHK(hipSetDevice(device));
    FpParams P = make_params(q);
    void *da, *db, *dc, *dout;
    uint32_t *dst;
    size_t bytes = (size_t)n * 32;
    HIPCHK(hipMalloc(&da, bytes));
    HIPCHK(hipMalloc(&db, bytes));
    HIPCHK(hipMalloc(&dc, bytes));
    HIPCHK(hipMalloc(&dout, bytes));
    HIPCHK(hipMalloc((void **)&dst, (size_t)n * 4));
    HIPCHK(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, b, bytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dc, c, bytes, hipMemcpyHostToDevice));
    HIPCHK(cwk_fpop(nullptr, dop, da, db, dc, dout, dst, n, P));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(status, dst, (size_t)n * 4, hipMemcpyDeviceToHost));
    hipFree(da);
    hipFree(db);
    hipFree(dc);
    hipFree(dout);
    hipFree(dst);
    return CW_OK;
}
