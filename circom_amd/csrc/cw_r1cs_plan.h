// cw_r1cs_plan.h — host-side plan for the staged R1CS check kernel (cw_r1cs_staged_kernel in cw_kernels.hip).
//
// The check reads every wire of a constraint row for 64 instances at a time (one wave = one instance group).
// Reading straight from the value table costs one HBM/L2 round trip per *term* (a wire appears in ~2.6 rows at
// --O0: once in the row that defines it, again in every row that wires it into a sub-component) and leaves one
// load in flight per wave.  The plan turns a chunk of rows into
//   * a LOAD list: each distinct wire is brought in once per chunk by an asynchronous global->LDS copy
//     (global_load_lds_dwordx4) into one of E 2-KiB LDS entries, issued DEPTH loads ahead of its first use;
//   * a TERM list that reads LDS entries only.
// Entry assignment is decided here, with the whole future known (Belady: evict the resident wire whose next
// use is farthest), under the hardware's hazard rule: load m is issued at the start of step m-DEPTH, so the
// previous content of its entry must not be read at any step >= m-DEPTH.  If no entry qualifies, filler loads
// (the previous load again) are inserted until one does.
//
// Kernel timeline for one chunk (step j consumes load j):
//     issue loads 0..DEPTH-1
//     for j in 0..n_loads-1:  issue load j+DEPTH ; s_waitcnt vmcnt(2*DEPTH) ; process the terms of step j
// The load list is padded with DEPTH copies of its last load so the wait count is a constant.
#pragma once
#include <stdint.h>
#include <algorithm>
#include <string>
#include <vector>

namespace cwplan {

constexpr uint32_t DEPTH = 4;              // loads in flight per wave (2 LDS-DMA instructions each)
constexpr uint32_t SLOT_BITS = 26;         // value slots < 2^26
constexpr uint32_t SLOT_MASK = (1u << SLOT_BITS) - 1;
// term word 0: [0,26) LDS entry | [27,29) accumulator 0=A 1=B 2=C |
//              [29,31) row end: 0 no, 1 quadratic row (A*B == C), 2 linear row (C == 0) | bit 31: last term of its part
//              (stream plan: [0,26) is the value slot itself; accumulator 3 = first wire of a pure equality row
//              x - y = 0, whose second term carries row end 3 = compare with the first)
// term word 1: coefficient id (0: +1, 1: -1, else index into ctab; bit 31: the wire is the constant 1 and
//              ctab holds the canonical coefficient, added without a multiplication)
constexpr uint32_t T_ACC_SH = 27, T_END_SH = 29;
constexpr uint32_t END_QUAD = 1, END_LIN = 2, END_EQ2 = 3, ACC_EQ2 = 3;
constexpr uint32_t COEF_CONST = 0x80000000u, T_PART_END = 0x80000000u;
constexpr uint32_t COEF_BITSEL = 0x40000000u;   // stream plan: the wire was checked "0 or 1" by the T_BOOL term in front (see build_stream)
// stream plan only: bit 26 of word 0 = the whole row is b * (b - 1) = 0 (or b * (1 - b) = 0): ONE term naming b, which must be 0 or 1 -
// one wire read and a comparison instead of three terms and a product.  (Num2Bits writes one such row per bit: 2.3 M of the 2.49 M
// rows of the ECDSA verifier.)
constexpr uint32_t T_BOOL = 1u << 26;

struct Plan {
    uint32_t entries = 0, n_chunks = 0;
    std::vector<uint32_t> chunk;    // 4 words per chunk: first record, n_loads, first term, first index into row_orig
    std::vector<uint32_t> rec;      // 2 words per record.  Per chunk: DEPTH prologue records {load word, 0}, then one per
                                    // step j: {load word of load j+DEPTH (or a copy of the last load), n terms of step j}
    std::vector<uint32_t> terms;    // 2 words per term (+ 2 terms of padding for the kernel's look-ahead)
    std::vector<uint32_t> row_orig; // .r1cs constraint index of every row that has terms, in plan order
    uint64_t n_loads = 0, n_terms = 0, n_filler = 0, n_unique = 0;
    uint64_t n_folded = 0, n_bitsel = 0;   // stream plan: boolean rows that ride in another row / products replaced by a select
};

// rows in processing order; ptr has 3 ranges per row (A, B, C) into slot[]/coef[].
inline Plan build(const std::vector<uint32_t> &ptr, const std::vector<uint32_t> &slot, const std::vector<uint32_t> &coef,
                  const std::vector<uint32_t> &orig, uint32_t n_slots, uint32_t want_chunks, uint32_t entries) {
    Plan p;
    const uint32_t n_rows = (uint32_t)(ptr.size() / 3);
    entries = std::max<uint32_t>(entries, DEPTH + 2);
    p.entries = entries;
    const uint64_t total_terms = slot.size();
    want_chunks = std::max<uint32_t>(1, std::min<uint32_t>(want_chunks, std::max<uint32_t>(1, n_rows / 16)));
    const uint64_t per_chunk = (total_terms + want_chunks - 1) / want_chunks;

    std::vector<int64_t> last_seen(n_slots, -1);        // scratch for next-use computation
    std::vector<int32_t> where(n_slots, -1);            // wire -> entry (or -1)
    std::vector<uint32_t> nxt;                          // next use (term index within chunk) per term
    struct Ent { int64_t slot = -1; int64_t last_read = -1000000; uint32_t next_use = 0xFFFFFFFFu; };
    std::vector<Ent> ent(entries);
    std::vector<uint8_t> seen_unique(n_slots, 0);

    uint32_t r = 0;
    while (r < n_rows) {
        // ---- chunk = rows [r, r1) holding about per_chunk terms -------------------------------------------
        uint32_t r1 = r;
        uint64_t tcount = 0;
        while (r1 < n_rows && (tcount < per_chunk || r1 == r)) {
            tcount += ptr[3 * r1 + 3] - ptr[3 * r1];
            r1++;
        }
        const uint32_t t0 = ptr[3 * r], t1 = ptr[3 * r1];
        if (t1 == t0) {           // rows without terms (0 = 0): nothing to check
            r = r1;
            continue;
        }
        const uint32_t nt = t1 - t0;
        nxt.assign(nt, 0xFFFFFFFFu);
        for (uint32_t u = nt; u-- > 0;) {
            uint32_t s = slot[t0 + u];
            if (last_seen[s] >= 0) nxt[u] = (uint32_t)last_seen[s];
            last_seen[s] = u;
        }
        for (uint32_t u = 0; u < nt; u++) last_seen[slot[t0 + u]] = -1;
        for (auto &e : ent) e = Ent();

        const uint32_t rec0 = (uint32_t)(p.rec.size() / 2);
        const uint32_t term0 = (uint32_t)(p.terms.size() / 2);
        const uint32_t orig0 = (uint32_t)p.row_orig.size();
        std::vector<uint32_t> loads;          // load words of this chunk
        std::vector<uint32_t> cnt;            // terms per step
        int64_t step = -1;
        // per-term row bookkeeping
        uint32_t row = r, part = 0;
        auto advance_row = [&](uint32_t t) {  // position (row, part) such that ptr[3row+part] <= t < ptr[3row+part+1]
            while (t >= ptr[3 * row + part + 1]) {
                part++;
                if (part == 3) { part = 0; row++; }
            }
        };
        for (uint32_t u = 0; u < nt; u++) {
            const uint32_t t = t0 + u, s = slot[t];
            advance_row(t);
            const bool row_end = (t + 1 == ptr[3 * row + 3]);
            uint32_t endk = 0;
            const bool eq2 = (orig[row] >> 31) != 0;              // x - y = 0: compare instead of subtracting
            if (row_end) {
                p.row_orig.push_back(orig[row] & 0x7FFFFFFFu);
                const bool a_empty = ptr[3 * row] == ptr[3 * row + 1], b_empty = ptr[3 * row + 1] == ptr[3 * row + 2];
                endk = eq2 ? END_EQ2 : ((a_empty || b_empty) ? END_LIN : END_QUAD);
            }
            // rows whose A or B is empty reduce to C == 0; their (non-empty) A or B terms do not matter, but they
            // are rare enough to keep in the stream (they are accumulated and ignored).
            if (!seen_unique[s]) { seen_unique[s] = 1; p.n_unique++; }
            int32_t e = where[s];
            if (e < 0) {
                int32_t best = -1;
                for (;;) {
                    const int64_t m = (int64_t)loads.size();      // this miss would be load m, consumed at step m
                    for (uint32_t k = 0; k < entries; k++) {
                        if (ent[k].last_read > m - (int64_t)DEPTH - 1) continue;   // still readable at issue time
                        if (ent[k].slot < 0) { best = (int32_t)k; break; }           // never used in this chunk
                        if (best < 0 || ent[k].next_use > ent[best].next_use) best = (int32_t)k;
                    }
                    if (best >= 0) break;
                    // every entry was read within the last DEPTH steps: let time pass with a filler load (the
                    // previous load again, same wire into the same entry: identical bytes, an L2 hit)
                    loads.push_back(loads.back());
                    cnt.push_back(0);
                    p.n_filler++;
                }
                const int64_t m = (int64_t)loads.size();
                if (ent[best].slot >= 0) where[ent[best].slot] = -1;
                ent[best].slot = s;
                where[s] = best;
                loads.push_back(s | ((uint32_t)best << SLOT_BITS));
                cnt.push_back(0);
                step = m;
                e = best;
            }
            ent[e].last_read = step;
            ent[e].next_use = nxt[u];
            uint32_t w0 = (uint32_t)e;
            w0 |= ((eq2 && !row_end) ? ACC_EQ2 : part) << T_ACC_SH;
            if (t + 1 == ptr[3 * row + part + 1]) w0 |= T_PART_END;
            w0 |= endk << T_END_SH;
            p.terms.push_back(w0);
            p.terms.push_back(coef[t]);
            cnt[(size_t)step]++;       // step >= 0: the first term of a chunk always misses into an empty entry
        }
        for (uint32_t u = 0; u < nt; u++) where[slot[t0 + u]] = -1;
        const uint32_t nl = (uint32_t)loads.size();
        for (uint32_t d = 0; d < DEPTH; d++) {
            p.rec.push_back(loads[std::min(d, nl - 1)]);
            p.rec.push_back(0);
        }
        for (uint32_t j = 0; j < nl; j++) {
            p.rec.push_back(loads[std::min(j + DEPTH, nl - 1)]);
            p.rec.push_back(cnt[j]);
        }
        p.chunk.push_back(rec0);
        p.chunk.push_back(nl);
        p.chunk.push_back(term0);
        p.chunk.push_back(orig0);
        p.n_loads += nl;
        p.n_terms += nt;
        p.n_chunks++;
        r = r1;
    }
    for (int k = 0; k < 4; k++) p.terms.push_back(0);
    return p;
}

// Stream plan (default kernel, cw_r1cs_stream_kernel): no LDS; the terms name value slots and are read from the
// value table two terms ahead of their use.  chunk = {first term, n terms, 0, first index into row_orig}.
// skip: optional bitmap over the constraints' indices in the .r1cs file - rows the emitted evaluation code has already checked
//
// FOLDED boolean rows (round 6).  Num2Bits writes `b * (b - 1) = 0` per bit AND one sum `sum 2^k b_k - in = 0` over the same bits:
// checked as separate rows every bit is read twice (the ECDSA verifier's check moved 171 GB for a table of 81 GB) and every
// term of the sum costs a Montgomery product.  With `fold`, the boolean row of b travels INSIDE the first other row that reads b,
// directly in front of that row's term on b: the kernel reads b once (a term that names the slot of the term before it reuses
// the registers), files the boolean row's verdict under its own constraint index (the T_BOOL term is a complete row: it ends,
// it advances the row counter, it leaves the accumulators alone), and - term flagged COEF_BITSEL - adds `b ? c : 0` instead of
// multiplying when every lane of the wave has just been seen to hold 0 or 1 (else the product, so a row's verdict never depends
// on another row's).  A COEF_BITSEL coefficient id names a PAIR of table entries: id = the multiplier's operand (c R'), id + 1 =
// what the term is worth when b = 1 (c on a canonical table, c R' on a table of Montgomery forms).
inline Plan build_stream(const std::vector<uint32_t> &ptr, const std::vector<uint32_t> &slot,
                         const std::vector<uint32_t> &coef, const std::vector<uint32_t> &orig, uint32_t terms_per_chunk,
                         const std::vector<uint32_t> *skip = nullptr, const std::vector<uint8_t> *boolrow = nullptr,
                         bool fold = true) {
    Plan p;
    const uint32_t n_rows = (uint32_t)(ptr.size() / 3);
    auto skipped = [&](uint32_t row) {
        if (!skip) return false;
        const uint32_t o = orig[row] & 0x7FFFFFFFu;
        return (o >> 5) < skip->size() && (((*skip)[o >> 5] >> (o & 31)) & 1u);
    };
    auto is_bool = [&](uint32_t row) { return boolrow && row < boolrow->size() && (*boolrow)[row]; };
    auto bool_slot = [&](uint32_t row) { return slot[(ptr[3 * row + 1] - ptr[3 * row] == 1) ? ptr[3 * row] : ptr[3 * row + 1]]; };
    // which boolean row rides in which other row: guest[s] = the boolean row of slot s that some later-emitted row will carry
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    std::vector<uint32_t> guest;                 // per slot: boolean row (NONE: none); bit 31 set once a host row was found
    if (fold && boolrow) {
        uint32_t mx = 0;
        for (uint32_t s : slot) mx = std::max(mx, s);
        guest.assign((size_t)mx + 1, NONE);
        for (uint32_t row = 0; row < n_rows; row++)
            if (ptr[3 * row + 3] != ptr[3 * row] && !skipped(row) && is_bool(row) && guest[bool_slot(row)] == NONE) guest[bool_slot(row)] = row;
        std::vector<uint8_t> hosted(guest.size(), 0);
        for (uint32_t row = 0; row < n_rows; row++) {
            if (ptr[3 * row + 3] == ptr[3 * row] || skipped(row) || is_bool(row) || (orig[row] >> 31)) continue;
            for (uint32_t t = ptr[3 * row]; t < ptr[3 * row + 3]; t++)
                if (slot[t] && guest[slot[t]] != NONE) hosted[slot[t]] = 1;
        }
        for (size_t s = 0; s < guest.size(); s++)
            if (!hosted[s]) guest[s] = NONE;
    }
    auto bool_term = [&](uint32_t row) {
        p.terms.push_back(bool_slot(row) | T_BOOL | (END_QUAD << T_END_SH) | T_PART_END);
        p.terms.push_back(0);
        p.row_orig.push_back(orig[row] & 0x7FFFFFFFu);
    };
    uint32_t chunk_t0 = 0, chunk_row0 = 0;
    for (uint32_t row = 0; row < n_rows; row++) {
        const uint32_t pa = ptr[3 * row], pb = ptr[3 * row + 1], pc = ptr[3 * row + 2], pe = ptr[3 * row + 3];
        if (pe == pa) continue;
        if (skipped(row)) continue;
        const bool lin = (pa == pb) || (pb == pc);
        const bool eq2 = (orig[row] >> 31) != 0;
        if (is_bool(row)) {
            // (the loader recognised the row: one of A / B is the single term +b, the other is b - 1 or 1 - b, C is empty)
            const uint32_t bs = bool_slot(row);
            if (bs < guest.size() && guest[bs] == row) continue;          // rides in another row
            bool_term(row);
        } else {
            for (uint32_t t = pa; t < pe; t++) {
                uint32_t part = t < pb ? 0 : (t < pc ? 1 : 2), endk = 0, ci = coef[t];
                if (t + 1 == pe) endk = eq2 ? END_EQ2 : (lin ? END_LIN : END_QUAD);
                else if (eq2) part = ACC_EQ2;
                if (!eq2 && slot[t] < guest.size() && guest[slot[t]] != NONE) {
                    bool_term(guest[slot[t]]);
                    guest[slot[t]] = NONE;                                 // once
                    if (ci >= 2 && !(ci & COEF_CONST)) { ci |= COEF_BITSEL; p.n_bitsel++; }
                    p.n_folded++;
                }
                const uint32_t pend = (t + 1 == pb || t + 1 == pc || t + 1 == pe) ? T_PART_END : 0;
                p.terms.push_back(slot[t] | (part << T_ACC_SH) | (endk << T_END_SH) | pend);
                p.terms.push_back(ci);
            }
            p.row_orig.push_back(orig[row] & 0x7FFFFFFFu);
        }
        const uint32_t nt = (uint32_t)(p.terms.size() / 2);
        if (nt - chunk_t0 >= terms_per_chunk) {
            p.chunk.insert(p.chunk.end(), {chunk_t0, nt - chunk_t0, 0u, chunk_row0});
            p.n_chunks++;
            chunk_t0 = nt;
            chunk_row0 = (uint32_t)p.row_orig.size();
        }
    }
    const uint32_t nt = (uint32_t)(p.terms.size() / 2);
    if (nt > chunk_t0) {
        p.chunk.insert(p.chunk.end(), {chunk_t0, nt - chunk_t0, 0u, chunk_row0});
        p.n_chunks++;
    }
    p.n_terms = p.n_loads = nt;
    for (int k = 0; k < 8; k++) p.terms.push_back(0);          // look-ahead padding: slot 0, coefficient +1
    return p;
}

// Replays a plan against the hazard rule with the loosest timing the hardware allows (a load may land anywhere
// between its issue and the wait that precedes its first use) and checks every term reads the wire the row
// names.  Returns an empty string if the plan is safe.
inline std::string verify(const Plan &p, const std::vector<uint32_t> &slot) {
    uint32_t t = 0;                                               // plan terms are the row terms, in order
    for (uint32_t c = 0; c < p.n_chunks; c++) {
        const uint32_t rec0 = p.chunk[4 * c], nl = p.chunk[4 * c + 1], term0 = p.chunk[4 * c + 2];
        if (term0 != t) return "chunk does not start at the next term";
        // cur[e] = wire readable in entry e; -2 while a load of another wire is in flight (content undefined)
        std::vector<int64_t> cur(p.entries, -1);
        auto issue = [&](uint32_t lw) {
            uint32_t e = lw >> SLOT_BITS;
            if (cur[e] != (int64_t)(lw & SLOT_MASK)) cur[e] = -2;
        };
        std::vector<uint32_t> inflight;                           // load words of issued, not yet waited loads (FIFO)
        for (uint32_t d = 0; d < DEPTH; d++) {
            uint32_t lw = p.rec[2 * (rec0 + d)];
            if (lw >> SLOT_BITS >= p.entries) return "entry out of range";
            issue(lw);
            inflight.push_back(lw);
        }
        uint32_t tp = term0;
        for (uint32_t j = 0; j < nl; j++) {
            uint32_t lw = p.rec[2 * (rec0 + DEPTH + j)], n = p.rec[2 * (rec0 + DEPTH + j) + 1];
            if (lw >> SLOT_BITS >= p.entries) return "entry out of range";
            issue(lw);
            inflight.push_back(lw);
            // vmcnt(2*DEPTH): everything but the youngest DEPTH loads has landed
            while (inflight.size() > DEPTH) {
                const uint32_t old = inflight.front(), e = old >> SLOT_BITS;
                inflight.erase(inflight.begin());
                bool clobbered = false;                            // a younger load of ANOTHER wire targets the entry
                for (uint32_t x : inflight) clobbered |= ((x >> SLOT_BITS) == e && x != old);
                if (!clobbered) cur[e] = old & SLOT_MASK;
            }
            for (uint32_t k = 0; k < n; k++, tp++, t++) {
                uint32_t w0 = p.terms[2 * tp];
                uint32_t want = slot[t];
                uint32_t e = w0 & SLOT_MASK;
                if (e >= p.entries) return "term entry out of range";
                if (cur[e] != (int64_t)want)
                    return "hazard: chunk " + std::to_string(c) + " step " + std::to_string(j) + " term " +
                           std::to_string(tp) + " reads entry " + std::to_string(e) + " holding " +
                           std::to_string(cur[e]) + ", wants " + std::to_string(want);
            }
        }
    }
    if (t != slot.size()) return "plan does not cover every term";
    return "";
}

}  // namespace cwplan
