// cw_bits_host.h — host side of the bit-plane path (cw_bits.hip): the program section of the .cwt, its validation,
// and the R1CS check plan over the bit table.  Included by cw_host.cpp (uses its U256 helpers).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <array>
#include <cstring>
#include <string>
#include <map>
#include <vector>

namespace cwbits {

constexpr uint32_t IN_BASE = 3;             // bit-table slot of main input 0 (slots 0,1,2 = constant 0, constant 1, reserved)

constexpr uint32_t BATCH = 8;               // vrows per batch (cw_bits.hip BITS_NB)
constexpr uint32_t MAX_LOADS = 4, MAX_FLUSH = 6, CMD_WORDS = 24;

struct Program {                             // hip_elements/bitsched.py::BitTape
    uint32_t ring = 0;                       // LDS ring rows (results that are no signal)
    uint32_t cache = 0;                      // LDS cache slots (rows of the bit table)
    uint32_t n_vrows = 0;                    // multiple of BATCH
    uint64_t n_slots = 0;                    // bit-table slots per group of 64 instances (multiple of 64)
    std::vector<uint32_t> recs;              // n_vrows * 64 * 2 words (record layout: bitsched.py / cw_bits.hip)
    std::vector<uint32_t> cmds;              // n_vrows / BATCH * CMD_WORDS words: row loads and flushes per batch
    std::vector<uint32_t> sig_slot;          // signal -> slot (signals that are copies of one another share a slot)
    std::vector<uint32_t> assert_slots;      // slots that must be zero in every instance (unproved `===`)
};

// The same network as emitted gfx950 code (hip_elements/bitjit.py): a code object with ONE kernel that runs one wave per chunk
// of 2 048 instances on the table layout T[chunk][slot][32 groups] (cw_bits.hip, sh = 5).  The section is machine code, as
// the reference's compiled <name>.cpp is: a .cwt is as trusted as an executable (CW_BITS_JIT=0 never loads it); what IS
// checked are the sizes and the slot map the host-side kernels index the table with.
struct JitProgram {
    uint64_t n_slots = 0;                    // rows (256 bytes) per chunk
    std::vector<uint32_t> sig_slot;          // signal -> row
    std::vector<uint8_t> code;               // ELF code object (hipModuleLoadData)
    std::vector<uint8_t> audit_code;         // ELF of the stand-alone audit of the table (same kernel name and arguments); may be empty
    uint32_t r1cs_crc = 0, r1cs_len = 0;     // constraint section of the .r1cs the fused check / the audit were built from (0, 0: unknown)
    bool check_complete = false;             // the fused R1CS check covers every constraint of the circuit
    uint32_t n_vgpr = 0, n_agpr = 0;
};
constexpr const char *JIT_KERNEL = "cw_bits_jit";
constexpr uint32_t JIT_MIN_BATCH = 1u << 18;
inline const char *validate_jit(const JitProgram &p, uint32_t n_signals, uint32_t n_inputs) {
    if (p.n_slots < (uint64_t)IN_BASE + n_inputs || p.n_slots * 256 >= (1ull << 32)) return "emitted program: slot count";
    if (p.sig_slot.size() != n_signals) return "emitted program: signal map size";
    for (uint32_t s : p.sig_slot)
        if (s >= p.n_slots || s == 2) return "emitted program: signal slot out of range";
    if (p.code.size() < 64 || memcmp(p.code.data(), "\177ELF", 4)) return "emitted program: not a code object";
    if (p.n_vgpr + p.n_agpr > 512 || p.n_vgpr < 8) return "emitted program: register counts";
    if (!p.audit_code.empty() && (p.audit_code.size() < 64 || memcmp(p.audit_code.data(), "\177ELF", 4))) return "emitted program: audit is not a code object";
    return nullptr;
}

// Every offset a record or a command carries is checked once at load time (files are untrusted input): LDS operands
// inside ring / cache / constants, results inside ring / cache, rows inside the group's bit table, flushes never onto
// the rows of the constants and main inputs.  (Races are the lowering's business: oracle/tape_eval.py replays the
// program with poisoned memory; a damaged program can only compute wrong bits inside its own group's table.)
inline const char *validate(const Program &p, uint32_t n_signals, uint32_t n_inputs) {
    if (p.ring < 8 || p.ring > 96 || p.ring % BATCH || p.cache < 8) return "bit program: ring / cache size";
    const uint32_t const_off = (p.ring + p.cache) * 512u;
    if (const_off + 16 > 0xFFF8u) return "bit program: LDS areas exceed the 16-bit offsets of a record";
    const uint64_t in_rows = ((uint64_t)IN_BASE + n_inputs + 63) / 64;
    if (p.n_slots % 64 || p.n_slots < in_rows * 64 || p.n_slots >= (1ull << 25)) return "bit program: slot count";
    if (p.n_vrows % BATCH) return "bit program: not a whole number of batches";
    if (p.recs.size() != (size_t)p.n_vrows * 64 * 2) return "bit program: record count";
    if (p.cmds.size() != (size_t)(p.n_vrows / BATCH) * CMD_WORDS) return "bit program: command block count";
    if (p.sig_slot.size() != n_signals) return "bit program: signal map size";
    for (uint32_t s : p.sig_slot)
        if (s >= p.n_slots || s == 2) return "bit program: signal slot out of range";
    for (uint32_t s : p.assert_slots)
        if (s >= p.n_slots) return "bit program: assertion slot out of range";
    for (size_t i = 0; i < (size_t)p.n_vrows * 64; i++) {
        const uint32_t w0 = p.recs[i * 2], w1 = p.recs[i * 2 + 1];
        const uint32_t a = w0 & 0xFFF8u, b = w0 >> 16, c = w1 & 0xFFFFu, d = w1 >> 16;
        if ((w0 & 4u) || (b & 7) || (c & 7) || (d & 7)) return "bit program: bad record word";
        if (a >= const_off + 16 || b >= const_off + 16 || c >= const_off + 16) return "bit program: operand outside the LDS areas";
        if (d >= const_off) return "bit program: result outside ring and cache";
    }
    const uint64_t tab_bytes = p.n_slots * 8;
    for (size_t b = 0; b < p.n_vrows / BATCH; b++) {
        const uint32_t *c = &p.cmds[b * CMD_WORDS];
        const uint32_t nl = c[0] & 0xFFu, nf = (c[0] >> 8) & 0xFFu;
        if (nl > MAX_LOADS || nf > MAX_FLUSH || (c[0] >> 16)) return "bit program: command counts";
        for (uint32_t j = 0; j < nl + nf; j++) {
            const uint32_t k = j < nl ? j : MAX_LOADS + (j - nl);
            const uint32_t goff = c[2 + 2 * k], loff = c[3 + 2 * k];
            if ((goff % 512) || (uint64_t)goff + 512 > tab_bytes) return "bit program: row outside the bit table";
            if ((loff % 512) || loff < p.ring * 512u || loff + 512 > const_off) return "bit program: cache slot outside the cache area";
            if (j >= nl && goff / 512 < in_rows) return "bit program: flush onto a constant / input row";
        }
    }
    return nullptr;
}

// Device stream: per batch 4 x 64 lanes x 16 bytes (load j of a lane carries the records of vrows 2j, 2j + 1 of the
// batch), padded to whole trips of TRIP batches plus AHEAD empty ones (the kernel requests records AHEAD batches ahead:
// cw_bits.hip BITS_AHEAD); idle lanes compute 0 ^ 0 ^ 0 into their own ring entry.  `cmds` gets zero blocks for the
// padding.  Returns the number of batches to execute (multiple of TRIP).
constexpr size_t AHEAD = 8, TRIP = 18;
inline uint32_t device_stream(const Program &p, std::vector<uint32_t> &dev, std::vector<uint32_t> &cmds) {
    const size_t batches = p.n_vrows / BATCH;
    const size_t run = (batches + TRIP - 1) / TRIP * TRIP;
    const uint32_t const_off = (p.ring + p.cache) * 512u;
    dev.assign((run + AHEAD) * 4 * 64 * 4, 0);
    for (size_t b = 0; b < run + AHEAD; b++)
        for (uint32_t k = 0; k < BATCH; k++) {
            const size_t v = b * BATCH + k;
            for (uint32_t lane = 0; lane < 64; lane++) {
                uint32_t w0, w1;
                if (v < p.n_vrows) {
                    w0 = p.recs[(v * 64 + lane) * 2];
                    w1 = p.recs[(v * 64 + lane) * 2 + 1];
                } else {
                    w0 = const_off | (const_off << 16);
                    w1 = const_off | ((uint32_t)((v % p.ring) * 512 + lane * 8) << 16);
                }
                const uint32_t d = 2 * k;                         // dword index of w0 among the lane's 16 of this batch
                uint32_t *q = &dev[((b * 4 + d / 4) * 64 + lane) * 4 + d % 4];
                q[0] = w0;
                q[1] = w1;
            }
        }
    cmds.assign((run + AHEAD) * CMD_WORDS, 0);
    std::copy(p.cmds.begin(), p.cmds.end(), cmds.begin());
    return (uint32_t)run;
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
    // class I: every coefficient is a small signed integer: exact 64-bit integer sums (cw_bits_r1cs_int_kernel).
    // The terms of a row are regrouped (a sum does not care about order): a GROUP = terms whose coefficients are
    // distinct powers of two with one sign inside one 32-bit half (the 32 bits of a word of a BinSum / Bits2Num row), so
    // that a term is "or the bit in at position k" (2 VALU); other coefficients form generic groups.
    // stream of u32: header {blocks:8 | sign:1 | hi:1 | generic:1 | part:2 | last of part:1 | end of row:1 | words:1} then
    // blocks x 8 words: power-of-two term = slot << 5 | k, generic term = two words (slot, coefficient id); padding
    // terms name slot 0 (the constant 0).  A WORD group (bit 15) carries `blocks` = n whole 32-bit words: n entries
    // first slot | half << 30 | sign << 31, padded to a multiple of 8 words.
    std::vector<uint32_t> ichunk, iwords, irow_orig;   // chunk = {first word, groups, 0, first row}
    std::vector<uint32_t> itab;       // signed 64-bit value of every coefficient id (2 words each; 0 if not small)
    uint32_t n_ichunks = 0;
    uint64_t n_trivial = 0, n_lut = 0, n_wide = 0, n_int = 0, n_int_blocks = 0, n_contig_blocks = 0, n_word_terms = 0;
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

// r_ptr: 3 * n_cons + 1 offsets (A, B, C parts of every row, processing order); r_sig: signal of each term;
// r_cc: canonical coefficient id of each term into cc (8 words each); r_orig: processing order -> constraint index;
// sig_slot: signal -> bit-table slot.  Wires are identified by their SLOT: signals that are copies of one another are
// one wire (an `a.in === b.out` wiring row then has table 0: it holds by construction and is not checked at run time),
// slot 1 is the constant 1 (like signal 0), slot 0 the constant 0 (its terms vanish).
inline R1Plan build_r1cs(const std::vector<uint32_t> &r_ptr, const std::vector<uint32_t> &r_sig, const std::vector<uint32_t> &r_cc,
                         const std::vector<uint32_t> &cc, const std::vector<uint32_t> &r_orig, const std::vector<uint32_t> &sig_slot,
                         const uint64_t q[4], uint32_t terms_per_chunk) {
    R1Plan p;
    std::vector<uint32_t> r_slot(r_sig.size());
    for (size_t i = 0; i < r_sig.size(); i++) r_slot[i] = sig_slot[r_sig[i]];
    const size_t n_cons = r_orig.size();
    std::vector<int64_t> small(cc.size() / 8);
    std::vector<uint8_t> is_small(cc.size() / 8);
    for (size_t i = 0; i < small.size(); i++) {
        uint64_t c[4];
        memcpy(c, &cc[i * 8], 32);
        is_small[i] = small_coef(c, q, &small[i]);
    }
    p.ctab = cc;
    p.itab.assign(small.size() * 2, 0);
    for (size_t i = 0; i < small.size(); i++)
        if (is_small[i]) {
            p.itab[2 * i] = (uint32_t)(uint64_t)small[i];
            p.itab[2 * i + 1] = (uint32_t)((uint64_t)small[i] >> 32);
        }
    std::vector<uint32_t> erows;                  // flat 8-word records, packed into vrows afterwards
    struct Stream {
        std::vector<uint32_t> *chunk, *terms, *rows;
        uint32_t cur_terms = 0, cur_first_term = 0, cur_first_row = 0;
        void close() {
            if (cur_terms) {
                chunk->push_back(cur_first_term);
                chunk->push_back(cur_terms);
                chunk->push_back(0);
                chunk->push_back(cur_first_row);
                cur_terms = 0;
            }
        }
    };
    Stream SW{&p.chunk, &p.terms, &p.row_orig};
    uint32_t i_groups = 0, i_first_word = 0, i_first_row = 0, i_terms = 0;
    auto iclose = [&]() {
        if (i_groups) {
            p.ichunk.push_back(i_first_word);
            p.ichunk.push_back(i_groups);
            p.ichunk.push_back(0);
            p.ichunk.push_back(i_first_row);
            i_groups = 0;
            i_terms = 0;
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
            if (s <= 1) continue;
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
                        int bit = (int)s;                       // slot 0: constant 0, slot 1: constant 1
                        if (s > 1) {
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
            for (int k = 0; k < nw; k++) rec[k] = wires[k] * 8;
            erows.insert(erows.end(), rec, rec + 8);
            p.n_lut++;
            continue;
        }
        // class I (all coefficients small: 64-bit integer sums) or W (field arithmetic)
        const uint32_t nt = t3 - t0;
        bool all_small = nt < (1u << 20);
        for (uint32_t t = t0; t < t3 && all_small; t++) all_small = is_small[r_cc[t]];
        if (all_small) {
            p.n_int++;
            if (i_groups && i_terms + nt > terms_per_chunk) iclose();
            if (i_groups == 0) {
                i_first_word = (uint32_t)p.iwords.size();
                i_first_row = (uint32_t)p.irow_orig.size();
            }
            struct G { uint32_t hdr; std::vector<uint32_t> w; uint32_t n_words = 0; };   // n_words != 0: a word group
            std::vector<G> groups;
            int last_part = -1;
            for (int pi = 0; pi < 3; pi++) {
                // power-of-two terms: key (sign, half) then slot order; generic terms afterwards
                std::vector<std::array<uint32_t, 3>> pw;          // key, slot, k
                std::vector<std::pair<uint32_t, uint32_t>> gen;    // slot, coefficient id
                for (uint32_t t = r_ptr[3 * j + pi]; t < r_ptr[3 * j + pi + 1]; t++) {
                    const uint32_t sl = r_slot[t];
                    if (sl == 0) continue;                          // the constant 0
                    const int64_t cv = small[r_cc[t]];
                    const uint64_t mag = cv < 0 ? (uint64_t)(-cv) : (uint64_t)cv;
                    if (mag && !(mag & (mag - 1))) {
                        const uint32_t k = (uint32_t)__builtin_ctzll(mag);
                        pw.push_back({(uint32_t)((cv < 0 ? 2u : 0u) | (k >= 32 ? 1u : 0u)), sl, k & 31u});
                    } else if (mag) {
                        gen.push_back({sl, r_cc[t]});
                    }
                }
                const size_t g0 = groups.size();
                std::map<uint32_t, std::vector<uint32_t>> word_slots;     // key (sign, half) -> first slots of whole words
                {   // whole words first, wherever their 32 terms sit in the part: per key, by slot, runs (s + t, 2^t), t = 0..31
                    std::vector<std::array<uint32_t, 3>> srt(pw);
                    std::sort(srt.begin(), srt.end());
                    std::vector<std::array<uint32_t, 3>> rest;
                    size_t a = 0;
                    while (a < srt.size()) {
                        bool word = a + 32 <= srt.size() && srt[a][2] == 0;
                        for (uint32_t t = 1; t < 32 && word; t++)
                            word = srt[a + t][0] == srt[a][0] && srt[a + t][1] == srt[a][1] + t && srt[a + t][2] == t;
                        if (word) {
                            word_slots[srt[a][0]].push_back(srt[a][1]);
                            p.n_word_terms += 32;
                            a += 32;
                        } else {
                            rest.push_back(srt[a]);
                            a++;
                        }
                    }
                    if (!word_slots.empty()) pw.swap(rest);
                }
                std::stable_sort(pw.begin(), pw.end(), [](const std::array<uint32_t, 3> &x, const std::array<uint32_t, 3> &y) { return x[0] < y[0]; });
                size_t i = 0;
                while (i < pw.size()) {
                    uint32_t used = 0;
                    G g;
                    g.hdr = ((pw[i][0] >> 1) << 8) | ((pw[i][0] & 1) << 9) | ((uint32_t)pi << 11);
                    const uint32_t key = pw[i][0];
                    while (i < pw.size() && pw[i][0] == key && !(used & (1u << pw[i][2]))) {
                        used |= 1u << pw[i][2];
                        g.w.push_back((pw[i][1] << 5) | pw[i][2]);
                        i++;
                    }
                    // blocks of 8 terms; a block whose slots are s, s+1, .., s+7 (the bits of a word usually are, see
                    // bitsched.py) is marked in bit 31 of its first word: the kernel then fetches the 8 masks with ONE
                    // 64-byte scalar load.  Runs of consecutive slots are aligned to block starts by padding.
                    std::sort(g.w.begin(), g.w.end());              // by slot
                    // (whole 32-bit words - 32 consecutive slots carrying 2^0 .. 2^31, the bits of a BinSum / Bits2Num operand,
                    // 87 % of the terms of SHA-256's adder rows - were taken out above: the kernel fetches their 32 masks with
                    // one coalesced vector load and transposes the 32 x 64 bit matrix across the lanes)
                    std::vector<uint32_t> out;
                    size_t a = 0;
                    while (a < g.w.size()) {
                        size_t run = 1;
                        while (a + run < g.w.size() && (g.w[a + run] >> 5) == (g.w[a] >> 5) + run) run++;
                        while (run >= 8) {
                            while (out.size() % 8) out.push_back(0);
                            const size_t at = out.size();
                            for (size_t k = 0; k < 8; k++) out.push_back(g.w[a + k]);
                            out[at] |= 1u << 31;
                            p.n_contig_blocks++;
                            a += 8;
                            run -= 8;
                        }
                        for (size_t k = 0; k < run; k++) out.push_back(g.w[a + k]);
                        a += run;
                    }
                    while (out.size() % 8) out.push_back(0);
                    p.n_int_blocks += out.size() / 8;
                    g.w.swap(out);
                    groups.push_back(std::move(g));
                }
                {   // ONE group for all whole words of the part: every entry carries its own sign (bit 31) and half (bit 30)
                    // next to the first slot (< 2^25), so that the kernel's four-at-a-time loop runs over ~6 words
                    // (a 195-term BinSum row) instead of over 1-3 per (sign, half) key
                    std::vector<uint32_t> all;
                    for (auto &kv : word_slots)
                        for (uint32_t sl : kv.second) all.push_back(sl | ((kv.first >> 1) << 31) | ((kv.first & 1u) << 30));
                    for (size_t k0 = 0; k0 < all.size(); k0 += 255) {
                        G g;
                        g.hdr = ((uint32_t)pi << 11) | (1u << 15);
                        g.n_words = (uint32_t)std::min<size_t>(255, all.size() - k0);
                        g.w.assign(all.begin() + k0, all.begin() + k0 + g.n_words);
                        while (g.w.size() % 8) g.w.push_back(0);     // the stream stays 32-byte aligned; entries past n are not read as words
                        groups.push_back(std::move(g));
                    }
                }
                for (size_t k0 = 0; k0 < gen.size(); k0 += 4 * 255) {
                    G g;
                    g.hdr = (1u << 10) | ((uint32_t)pi << 11);
                    for (size_t k = k0; k < gen.size() && k < k0 + 4 * 255; k++) {
                        g.w.push_back(gen[k].first);
                        g.w.push_back(gen[k].second);
                    }
                    while (g.w.size() % 8) { g.w.push_back(0); g.w.push_back(0); }
                    groups.push_back(std::move(g));
                }
                if (groups.size() > g0) {
                    groups.back().hdr |= 1u << 13;                 // last group of its part
                    last_part = pi;
                }
            }
            (void)last_part;
            if (groups.empty()) {                                   // 0 = 0
                p.n_int--;
                p.n_trivial++;
                continue;
            }
            groups.back().hdr |= 1u << 14;                          // end of the row
            for (auto &g : groups) {
                p.iwords.push_back(g.hdr | (g.n_words ? g.n_words : (uint32_t)(g.w.size() / 8)));
                p.iwords.insert(p.iwords.end(), g.w.begin(), g.w.end());
            }
            i_groups += (uint32_t)groups.size();
            i_terms += nt;
            p.irow_orig.push_back(orig);
            continue;
        }
        p.n_wide++;
        Stream &S = SW;
        if (S.cur_terms && S.cur_terms + nt > terms_per_chunk) S.close();
        if (S.cur_terms == 0) {
            S.cur_first_term = (uint32_t)(S.terms->size() / 2);
            S.cur_first_row = (uint32_t)S.rows->size();
        }
        for (int pi = 0; pi < 3; pi++) {
            const uint32_t a = r_ptr[3 * j + pi], b = r_ptr[3 * j + pi + 1];
            for (uint32_t t = a; t < b; t++) {
                uint32_t w = r_slot[t] * 8;                        // < 2^28 (n_slots < 2^25 checked by validate)
                w |= (uint32_t)pi << 28;
                if (t + 1 == b) w |= 1u << 31;                     // last term of its part
                if (t + 1 == t3) w |= 1u << 30;                    // last term of the row
                S.terms->push_back(w);
                S.terms->push_back(r_cc[t]);
            }
        }
        S.rows->push_back(orig);
        S.cur_terms += nt;
    }
    SW.close();
    iclose();
    p.n_chunks = (uint32_t)(p.chunk.size() / 4);
    p.n_evrows = (uint32_t)((erows.size() / 8 + 63) / 64);
    p.erecs.assign((size_t)p.n_evrows * 64 * 8, 0);
    {   // neighbouring lanes read neighbouring slots: records ordered by the youngest (highest) slot they read, so that the 64
        // lanes of a vrow - five scattered 8-byte loads each - fall into a few rows of the table instead of all over it
        const size_t n = erows.size() / 8;
        std::vector<std::pair<uint64_t, uint32_t>> key(n);
        for (size_t i = 0; i < n; i++) {
            uint32_t hi = 0, lo = 0xFFFFFFFFu;
            for (int k = 0; k < 5; k++) {
                hi = std::max(hi, erows[i * 8 + k]);
                if (erows[i * 8 + k]) lo = std::min(lo, erows[i * 8 + k]);
            }
            key[i] = {((uint64_t)hi << 32) | lo, (uint32_t)i};
        }
        std::sort(key.begin(), key.end());
        for (size_t i = 0; i < n; i++) std::copy(erows.begin() + (size_t)key[i].second * 8, erows.begin() + (size_t)key[i].second * 8 + 8, p.erecs.begin() + i * 8);
    }
    if (p.terms.empty()) p.terms.assign(2, 0);
    if (p.row_orig.empty()) p.row_orig.assign(1, 0);
    if (p.chunk.empty()) p.chunk.assign(4, 0);
    p.n_ichunks = (uint32_t)(p.ichunk.size() / 4);
    p.iwords.resize(p.iwords.size() + 16, 0);      // the kernel reads blocks of 8 words
    if (p.irow_orig.empty()) p.irow_orig.assign(1, 0);
    if (p.ichunk.empty()) p.ichunk.assign(4, 0);
    if (p.itab.empty()) p.itab.assign(2, 0);
    return p;
}

}   // namespace cwbits
