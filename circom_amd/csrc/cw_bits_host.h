// cw_bits_host.h — host side of the bit-plane path (cw_bits.hip): the program section of the .cwt, its validation,
// and the R1CS check plan over the bit table.  Included by cw_host.cpp (uses its U256 helpers).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

namespace cwbits {

constexpr uint32_t SIG_BASE = 3;            // bit-table slot of signal 0 (slots 0,1,2 = constant 0, constant 1, reserved)
constexpr uint32_t K_GLOBAL = 0, K_RING = 1, K_PREV = 2;

struct Program {                             // hip_elements/bitsched.py::BitTape
    uint32_t ring = 0;                       // LDS ring entries (power of two)
    uint32_t n_vrows = 0;
    uint64_t n_slots = 0;                    // bit-table slots per group of 64 instances
    std::vector<uint32_t> recs;              // (n_vrows + 3) * 64 * 8 words, the last 3 vrows empty (kernel prefetch)
};

// Every offset a record carries is checked once at load time (files are untrusted input): operands and
// destinations inside the group's bit table / the ring / the wave, destinations never on the constant slots.
inline const char *validate(const Program &p, uint32_t n_signals) {
    if (p.ring < 2 || p.ring > 256 || (p.ring & (p.ring - 1))) return "bit program: ring size";
    if (p.n_slots < (uint64_t)SIG_BASE + n_signals || p.n_slots > (1ull << 27)) return "bit program: slot count";
    if (p.recs.size() != ((size_t)p.n_vrows + 3) * 64 * 8) return "bit program: record count";
    for (size_t v = 0; v < p.n_vrows; v++) {
        for (uint32_t lane = 0; lane < 64; lane++) {
            const uint32_t *r = &p.recs[(v * 64 + lane) * 8];
            for (int j = 0; j < 3; j++) {
                const uint32_t kind = r[j] >> 30, off = r[j] & 0x3FFFFFFFu;
                if (kind == K_GLOBAL) {
                    if ((off & 7) || off / 8 >= p.n_slots) return "bit program: operand slot out of range";
                } else if (kind == K_RING) {
                    if ((off & 7) || off >= p.ring * 512u) return "bit program: ring operand out of range";
                } else if (kind == K_PREV) {
                    if ((off & 3) || off >= 256 || v == 0) return "bit program: bad PREV operand";
                } else {
                    return "bit program: unknown operand kind";
                }
            }
            if (r[3] & ~0x1FFu) return "bit program: bad gate word";
            for (int j = 4; j < 8; j++) {
                const uint32_t d = r[j];
                if (d && ((d & 7) || d / 8 >= p.n_slots || d / 8 < SIG_BASE)) return "bit program: destination out of range";
            }
        }
    }
    return nullptr;
}

// ---- R1CS over the bit table --------------------------------------------------------------------------------------------
// Class E (<= 5 distinct non-constant wires, small coefficients): a 32-entry table "violated?" per constraint, 64
// constraints per vrow (cw_bits_r1cs_lut_kernel).  Class W: everything else, streamed term by term
// (cw_bits_r1cs_wide_kernel).  Constraints whose table is all zero (x*(x-1) = 0 on a bit) need no check at all.
struct R1Plan {
    std::vector<uint32_t> erecs;      // n_evrows * 64 * 8 words: 5 wire byte offsets, table, constraint index, 0
    uint32_t n_evrows = 0;
    std::vector<uint32_t> chunk;      // 4 words per chunk: first term, n terms, 0, first row (index into row_orig)
    std::vector<uint32_t> terms;      // 2 words per term
    std::vector<uint32_t> row_orig;   // W-class rows -> constraint index of the .r1cs
    std::vector<uint32_t> ctab;       // canonical coefficients, 8 words each
    uint32_t n_chunks = 0;
    uint64_t n_trivial = 0, n_lut = 0, n_wide = 0;
};

// small signed value of a canonical coefficient, if |val| < 2^40
inline bool small_coef(const uint64_t c[4], const uint64_t q[4], int64_t *out) {
    if (!(c[1] | c[2] | c[3]) && c[0] < (1ull << 40)) {
        *out = (int64_t)c[0];
        return true;
    }
    // q - c
    uint64_t d[4];
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; i++) {
        unsigned __int128 t = (unsigned __int128)q[i] - c[i] - (uint64_t)br;
        d[i] = (uint64_t)t;
        br = (t >> 64) & 1;
    }
    if (!(d[1] | d[2] | d[3]) && d[0] < (1ull << 40)) {
        *out = -(int64_t)d[0];
        return true;
    }
    return false;
}

// r_ptr: 3 * n_cons + 1 offsets (A, B, C parts of every row, processing order); r_slot: signal of each term;
// r_cc: canonical coefficient id of each term into cc (8 words each); r_orig: processing order -> constraint index.
inline R1Plan build_r1cs(const std::vector<uint32_t> &r_ptr, const std::vector<uint32_t> &r_slot, const std::vector<uint32_t> &r_cc,
                         const std::vector<uint32_t> &cc, const std::vector<uint32_t> &r_orig, const uint64_t q[4],
                         uint32_t terms_per_chunk) {
    R1Plan p;
    const size_t n_cons = r_orig.size();
    std::vector<int64_t> small(cc.size() / 8);
    std::vector<uint8_t> is_small(cc.size() / 8);
    for (size_t i = 0; i < small.size(); i++) {
        uint64_t c[4];
        memcpy(c, &cc[i * 8], 32);
        is_small[i] = small_coef(c, q, &small[i]);
    }
    p.ctab = cc;
    std::vector<uint32_t> erows;                  // flat 8-word records, packed into vrows afterwards
    uint32_t cur_terms = 0, cur_first_term = 0, cur_first_row = 0;
    auto close_chunk = [&]() {
        if (cur_terms) {
            p.chunk.push_back(cur_first_term);
            p.chunk.push_back(cur_terms);
            p.chunk.push_back(0);
            p.chunk.push_back(cur_first_row);
            cur_terms = 0;
        }
    };
    for (size_t j = 0; j < n_cons; j++) {
        const uint32_t orig = r_orig[j] & 0x7FFFFFFFu;
        const uint32_t t0 = r_ptr[3 * j], t3 = r_ptr[3 * j + 3];
        if (t0 == t3) { p.n_trivial++; continue; }
        // distinct non-constant wires
        uint32_t wires[6];
        int nw = 0;
        bool ok = true;
        for (uint32_t t = t0; t < t3 && ok; t++) {
            if (!is_small[r_cc[t]]) ok = false;
            const uint32_t s = r_slot[t];
            if (s == 0) continue;
            int k = 0;
            while (k < nw && wires[k] != s) k++;
            if (k == nw) {
                if (nw == 5) ok = false;
                else wires[nw++] = s;
            }
        }
        if (ok) {
            uint32_t tt = 0;
            for (uint32_t m = 0; m < (1u << nw); m++) {
                __int128 part[3] = {0, 0, 0};
                for (int pi = 0; pi < 3; pi++)
                    for (uint32_t t = r_ptr[3 * j + pi]; t < r_ptr[3 * j + pi + 1]; t++) {
                        const uint32_t s = r_slot[t];
                        int bit = 1;
                        if (s != 0) {
                            int k = 0;
                            while (wires[k] != s) k++;
                            bit = (m >> k) & 1;
                        }
                        if (bit) part[pi] += small[r_cc[t]];
                    }
                if (part[0] * part[1] - part[2] != 0) tt |= 1u << m;
            }
            // (wires beyond nw read the constant-0 slot: only the first 2^nw entries are ever selected)
            if (tt == 0) { p.n_trivial++; continue; }
            uint32_t rec[8] = {0, 0, 0, 0, 0, tt, orig, 0};
            for (int k = 0; k < nw; k++) rec[k] = (SIG_BASE + wires[k]) * 8;
            erows.insert(erows.end(), rec, rec + 8);
            p.n_lut++;
            continue;
        }
        // class W
        p.n_wide++;
        const uint32_t nt = t3 - t0;
        if (cur_terms && cur_terms + nt > terms_per_chunk) close_chunk();
        if (cur_terms == 0) {
            cur_first_term = (uint32_t)(p.terms.size() / 2);
            cur_first_row = (uint32_t)p.row_orig.size();
        }
        for (int pi = 0; pi < 3; pi++) {
            const uint32_t a = r_ptr[3 * j + pi], b = r_ptr[3 * j + pi + 1];
            for (uint32_t t = a; t < b; t++) {
                uint32_t w = (SIG_BASE + r_slot[t]) * 8;           // < 2^28 (n_slots <= 2^25 checked by the caller)
                w |= (uint32_t)pi << 28;
                if (t + 1 == b) w |= 1u << 31;                     // last term of its part
                if (t + 1 == t3) w |= 1u << 30;                    // last term of the row
                p.terms.push_back(w);
                p.terms.push_back(r_cc[t]);
            }
        }
        p.row_orig.push_back(orig);
        cur_terms += nt;
    }
    close_chunk();
    p.n_chunks = (uint32_t)(p.chunk.size() / 4);
    p.n_evrows = (uint32_t)((erows.size() / 8 + 63) / 64);
    p.erecs.assign((size_t)p.n_evrows * 64 * 8, 0);
    std::copy(erows.begin(), erows.end(), p.erecs.begin());
    if (p.terms.empty()) p.terms.assign(2, 0);
    if (p.row_orig.empty()) p.row_orig.assign(1, 0);
    if (p.chunk.empty()) p.chunk.assign(4, 0);
    return p;
}

}   // namespace cwbits
