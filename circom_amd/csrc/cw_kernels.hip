// cw_kernels.hip — HIP kernels of the batched witness calculator (gfx950 / CDNA4).
//
// Data layout in HBM: the value table V holds every signal (and spill temps) of every
// instance as structure-of-arrays:  element (slot s, instance i) = two 16-byte halves
//     lo: V[(2*s + 0) * Bp + i]      hi: V[(2*s + 1) * Bp + i]            (uint4 units)
// so a wave's 64 lanes read/write 1 KiB contiguous per half (global_load/store_dwordx4, fully
// coalesced).  One witness instance per lane; the schedule (tape) is wave-uniform and is fetched
// with scalar loads; constants are wave-uniform too (SGPRs).
#include <hip/hip_runtime.h>
#include "cw_kernels.h"
#include "fp256.hip.h"
#include "cw_rowops.hip.h"

#define CW_BLOCK 256

// ---- value-table access --------------------------------------------------------------------------
__device__ __forceinline__ fe v_load(const uint4 *__restrict__ V, uint32_t slot, uint32_t Bp, uint32_t i) {
    size_t base = (size_t)slot * 2 * Bp + i;
    uint4 lo = V[base];
    uint4 hi = V[base + Bp];
    fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void v_store(uint4 *__restrict__ V, uint32_t slot, uint32_t Bp, uint32_t i, const fe &x) {
    size_t base = (size_t)slot * 2 * Bp + i;
    V[base] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    V[base + Bp] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}
__device__ __forceinline__ fe aos_load(const uint4 *p, uint32_t i) {
    uint4 lo = p[2 * (size_t)i], hi = p[2 * (size_t)i + 1];
    fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}

// ---- init: slot 0 = 1 (calcwit.cpp:34), status = 0 -----------------------------------------------
// `one` = 1, or R' mod q when the table holds Montgomery forms (lower.py pass A6)
__global__ void __launch_bounds__(CW_BLOCK) cw_init_kernel(uint4 *V, uint32_t Bp, uint32_t *status, uint32_t *first_bad, fe one) {
    uint32_t i = blockIdx.x * CW_BLOCK + threadIdx.x;
    if (i >= Bp) return;
    v_store(V, 0, Bp, i, one);
    status[i] = 0;
    first_bad[i] = 0xFFFFFFFFu;
}

// ---- ingest: AoS canonical inputs [batch][n_in][32 B] -> SoA input slots ----------------------------
// (setInputSignal's `signalValues[si] = val`, calcwit.cpp:93, for the whole batch)
// MONT: the table holds Montgomery forms: x -> x R' = mmul(x, R'^2)
template <bool MONT>
__global__ void __launch_bounds__(CW_BLOCK) cw_ingest_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ V,
                                                              uint32_t input_start, uint32_t n_in, uint32_t batch,
                                                              uint32_t Bp, FpParams P) {
    uint32_t i = blockIdx.x * CW_BLOCK + threadIdx.x;
    if (i >= batch) return;
    for (uint32_t k = blockIdx.y; k < n_in; k += gridDim.y) {      // grid.y is capped at 65535 input signals per pass
        size_t src = ((size_t)i * n_in + k) * 2;
        size_t dst = (size_t)(input_start + k) * 2 * Bp + i;
        if (MONT) {
            const fe x = fe_mmul(aos_load(in, (uint32_t)((size_t)i * n_in + k)), fe_from(P.r2), P);
            V[dst] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
            V[dst + Bp] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
        } else {
            V[dst] = in[src];
            V[dst + Bp] = in[src + 1];
        }
    }
}

// ---- schedule evaluation (the hot path) ---------------------------------------------------------------
// One lane = one instance.  A workgroup = S waves ("strands") that all work on the SAME 64 instances,
// each walking its own row stream of the schedule; strands hand values to each other through LDS slots
// (or, when the LDS pool is exhausted, the value table) and meet at BARRIER rows
// (hip_elements/lower.py passes C/D).  S = 1 for large batches (instance parallelism alone fills the
// chip), S up to 16 for the small batches of the BASELINE configs.
// Latency hiding inside a strand:
//  (1) operands of row r+1 are fetched before row r executes (one-row-ahead software prefetch; legal
//      because the lowering encodes any operand produced by the preceding row as kind PREV = register
//      forwarding).  The fetch is branch-free — one pointer select, two unconditional 16-byte loads per
//      operand — so the loaded registers ARE the loop-carried registers and nothing forces an early wait;
//  (2) pure copies never load: they are extra destinations of the row that produced the value, taken
//      from the strand's extra-destination table (scalar loads issued before the row's arithmetic);
//  (3) two barrier flavours are encoded: LIGHT (workgroup-scope release/acquire fences around s_barrier) where the
//      strands only exchanged LDS slots, FULL (__syncthreads) where a value crosses strands through the value
//      table.  On gfx950 both compile to `s_waitcnt lgkmcnt(0); s_barrier` (checked in the ISA): the waves of a
//      workgroup share the CU's vector L1, which orders one wave's earlier store before another wave's later
//      load, so neither drains vmcnt and global stores stay in flight across every barrier.
// FULL_OPS selects the variant that also carries the slow-path operators (INV/IDIV/MOD/POW).
// -DCW_PROFILE (tools/profile_ops.sh, never the product build): workgroup 0 accumulates the shader-clock time of
// every interpreter step per (strand, opcode), operand waits included.
#ifdef CW_PROFILE
__device__ unsigned long long cw_prof[16 * 64 * 2];
// strand 0 of workgroup 0, per opcode: clocks until the operands are in registers | arithmetic | destinations + extras
__device__ unsigned long long cw_prof_seg[64 * 4];
// workgroup 0: shader clock at which strand s ARRIVES at its k-th barrier (k < CW_PROF_LEVELS): who is last, level by level
#define CW_PROF_LEVELS 4096
__device__ unsigned long long cw_prof_arrive[CW_PROF_LEVELS * 16];
#define CW_PROF_SEG(k, opc, t_from)                                                                    \
    do {                                                                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0)                                                      \
            atomicAdd(&cw_prof_seg[((opc) & 63u) * 4 + (k)], (unsigned long long)(__builtin_readcyclecounter() - (t_from))); \
    } while (0)
#define CW_PROF_BEGIN() prof_t0 = __builtin_readcyclecounter()
#define CW_PROF_END(row)                                                                              \
    do {                                                                                              \
        if (blockIdx.x == 0 && lane == 0 && wave < 16) {                                              \
            const uint64_t prof_t1 = __builtin_readcyclecounter();                                    \
            const uint32_t prof_k = (wave * 64 + ((row).w0 & 63u)) * 2;                               \
            atomicAdd(&cw_prof[prof_k], (unsigned long long)(prof_t1 - prof_t0));                     \
            atomicAdd(&cw_prof[prof_k + 1], 1ull);                                                    \
        }                                                                                             \
    } while (0)
extern "C" int cw_debug_profile_arrive(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cw_prof_arrive), sizeof(cw_prof_arrive)) == hipSuccess ? 0 : -1;
}
extern "C" int cw_debug_profile_seg(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cw_prof_seg), sizeof(cw_prof_seg)) == hipSuccess ? 0 : -1;
}
extern "C" int cw_debug_profile(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(cw_prof), sizeof(cw_prof)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long zero[16 * 64 * 2];
        if (hipMemcpyToSymbol(HIP_SYMBOL(cw_prof), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define CW_PROF_BEGIN()
#define CW_PROF_END(row)
#define CW_PROF_SEG(k, opc, t_from)
#endif
extern __shared__ uint4 cw_lds[];       // [slot][2 halves][64 lanes] x 16 B

// Status word of an instance = bits | index << 8, index = the flat operation of the failing check (cw_tape.h).  When several
// checks fail - in any row order, on any strand - the smallest index wins: the check the reference's sequential program
// stops at (assert_bucket.rs:75-77, calcwit.cpp:104-114).
#include "cw_call.hip.h"        // c_load, EvalCtx, the tier-2 interpreter (eval_call_body)

__device__ __forceinline__ fe lds_load_off(uint32_t slot_off, const EvalCtx &c) {
    const char *p = (const char *)cw_lds + slot_off + c.lane16;
    const uint4 lo = *(const uint4 *)p, hi = *(const uint4 *)(p + c.lds_hi);
    fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void lds_store_off(uint32_t slot_off, const EvalCtx &c, const fe &x) {
    char *p = (char *)cw_lds + slot_off + c.lane16;
    *(uint4 *)p = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    *(uint4 *)(p + c.lds_hi) = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}
// branch-free operand fetch: wave-uniform base (SGPR pair) + per-lane 32-bit offset (saddr addressing).
// Constants: every lane reads the same 32 B.  Kinds resolved at execution time (PREV, LDS) read slot 0.
__device__ __forceinline__ fe fetch_off(uint32_t kind, uint64_t off, const EvalCtx &c) {
    const bool is_const = kind == K_CONST;
    const char *base = (is_const ? c.Cb : c.Vb) + off;
    const uint32_t o_lo = is_const ? 0u : c.vlo, o_hi = is_const ? 16u : c.vhi;
    const uint4 lo = *(const uint4 *)(base + o_lo), hi = *(const uint4 *)(base + o_hi);
    fe r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void store_off(uint64_t off, const EvalCtx &c, const fe &x) {
#ifdef CW_EXPERIMENT_NOSTORE       // timing experiment only (tools/profile_ops.sh): results are garbage
    return;
#endif
    char *base = (char *)c.Vb + off;
    *(uint4 *)(base + c.vlo) = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    *(uint4 *)(base + c.vhi) = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}

// ---- D_LINSUM: d = c0 + sum_i coef_i * x_i, small signed integer coefficients ------------------------------
// The common case (every lane's x_i is a small non-negative integer: bits, partial sums) accumulates the
// 64x64-bit products in two unsigned 192-bit accumulators (positive / negative terms) with no modular
// reduction at all; a lane holding anything else sends the wave through the generic field path for that
// term.  Result = g + P - N reduced once.  Terms are read at execution time, one ahead of their use.
// LDS_ONLY: the pipelined kernel has no value-table operands (and must not contain a single vector-memory load: hipcc
// would guard the merged registers with s_waitcnt vmcnt(small), which also drains the stores and LDS-DMA loads in flight)
template <bool LDS_ONLY = false>
__device__ __forceinline__ fe term_load(uint64_t t0, const fe &prev, const EvalCtx &c) {
    const uint32_t kind = (uint32_t)(t0 >> 61);
    const uint64_t off = t0 & 0x1FFFFFFFFFFFFFFFull;
    if (kind == K_PREV) return prev;
    if (LDS_ONLY || kind == K_LDS) return lds_load_off((uint32_t)off, c);
    return fetch_off(K_SIG, off, c);
}
// Terms are processed W at a time (W = 2 or 4): one scalar load brings the table entries, the W operand loads are in
// flight together (memory-level parallelism), then the products are accumulated.  Entries past the row's last
// term (the table is padded by 4) are neutralised by a zero coefficient.  W = 4 costs 16 more live registers: it is
// used for schedules dominated by long small-coefficient sums (bit-level circuits), W = 2 otherwise.
template <int W, bool LDS_ONLY = false>
__device__ __forceinline__ fe eval_linsum(uint32_t n, const fe &c0, const fe &prev, EvalCtx &c, const FpParams &P) {
    fe g = c0;
    Acc192 pos = {0, 0, 0}, neg = {0, 0, 0};
    const uint64_t *tt = c.terms + (size_t)c.tp * 2;
    for (uint32_t k = 0; k < n; k += W) {
        const uint64_t a0 = tt[2 * k], c0_ = tt[2 * k + 1], a1 = tt[2 * k + 2], c1 = tt[2 * k + 3];
        const fe x0 = term_load<LDS_ONLY>(a0, prev, c);
        const fe x1 = term_load<LDS_ONLY>(a1, prev, c);
        if (W == 4) {
            const uint64_t a2 = tt[2 * k + 4], c2 = tt[2 * k + 5], a3 = tt[2 * k + 6], c3 = tt[2 * k + 7];
            const fe x2 = term_load<LDS_ONLY>(a2, prev, c);
            const fe x3 = term_load<LDS_ONLY>(a3, prev, c);
            linsum_term(x0, c0_, g, pos, neg, P);
            linsum_term(x1, k + 1 < n ? c1 : 0, g, pos, neg, P);
            linsum_term(x2, k + 2 < n ? c2 : 0, g, pos, neg, P);
            linsum_term(x3, k + 3 < n ? c3 : 0, g, pos, neg, P);
        } else {
            linsum_term(x0, c0_, g, pos, neg, P);
            linsum_term(x1, k + 1 < n ? c1 : 0, g, pos, neg, P);
        }
    }
    c.tp += n;
    g = fe_add(g, acc192_to_fe(pos), P);
    return fe_sub(g, acc192_to_fe(neg), P);
}

// ---- D_DOTC: d = c0 + sum_i coef_i * x_i with field-sized coefficients --------------------------------------------
// Up to four products x_i * (coef_i R') are accumulated as unreduced 29-bit-limb columns and reduced ONCE
// (81 multiply-adds per term + 90 for the reduction, instead of a full Montgomery product and a modular
// addition per term).  The constants come from the limb-form table through scalar loads.
template <bool LDS_ONLY = false>
__device__ __forceinline__ fe eval_dotc(uint32_t n, const fe &c0, const fe &prev, EvalCtx &c, const FpParams &P) {
    fe res = c0;
    const uint64_t *tt = c.terms + (size_t)c.tp * 2;
    uint64_t acc[18];
    for (int j = 0; j < 18; j++) acc[j] = 0;
    fe x = term_load<LDS_ONLY>(tt[0], prev, c);
    for (uint32_t k = 0; k < n; k++) {
        const fe29 xc = fe_to29(x);
        const uint32_t ci = (uint32_t)tt[2 * k + 1];
        if (k + 1 < n) x = term_load<LDS_ONLY>(tt[2 * k + 2], prev, c);       // next operand in flight during the 81 MACs
        fe29_mac(acc, xc, c.Lb + (size_t)ci * 12);
        if ((k & 3) == 3 || k + 1 == n) {                            // at most 4 products per reduction (column bound)
            res = fe_add(res, fe_from29(fe29_reduce(acc, P)), P);
            for (int j = 0; j < 18; j++) acc[j] = 0;
        }
    }
    c.tp += n;
    return res;
}

// Out of line in the strand kernels (launch bound 1024 = 128 VGPRs per wave: inlined, the interpreter would squeeze the
// row loop); inlined in the single-wave kernels of circuits with functions (launch bound 64: up to 512 VGPRs; BigMultModP
// x 65 536: 13.8 -> 19.1 M witnesses/s).
__device__ __noinline__ void eval_call(uint32_t fn, uint64_t reg_off, uint32_t row_id, uint32_t &st, const EvalCtx &c, const FpParams &P) {
    eval_call_body(fn, reg_off, row_id, st, c, P);
}

// One interpreter step: executes `row` with operands (xa, xb) while the operands of `nrow` are requested into
// (ya, yb).  The loop calls it twice per iteration with the two register sets swapped (no rotation moves).
template <bool FULL_OPS, int LW, bool CALL_INLINE = false>
__device__ __forceinline__ void eval_step(const CwDRow &row, const fe &xa, const fe &xb, const CwDRow &nrow, fe &ya,
                                          fe &yb, fe &prev, uint64_t &selmask, uint32_t &st, uint32_t r,
                                          const uint64_t *__restrict__ extras, uint32_t &xp, EvalCtx &c,
                                          const FpParams &P) {
    const uint32_t op = row.w0 & 0xFF;
    const uint32_t dk = (row.w0 >> SH_DK) & 7, ak = (row.w0 >> SH_AK) & 7, bk = (row.w0 >> SH_BK) & 7;
    const uint32_t nx = (row.w0 >> SH_NX) & 0xFFF;
    if (op == D_BARRIER) {                                           // nothing is prefetched across a barrier
#ifdef CW_PROFILE
        if (blockIdx.x == 0 && (threadIdx.x & 63u) == 0 && c.prof_level < CW_PROF_LEVELS)
            cw_prof_arrive[c.prof_level * 16 + (threadIdx.x >> 6)] = __builtin_readcyclecounter();
        c.prof_level++;
#endif
        if (row.aux) {
            __syncthreads();                                         // FULL: values cross strands through the value table
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS writes of this wave are done ...
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // ... before any wave reads them
        }
        ya = fetch_off((nrow.w0 >> SH_AK) & 7, nrow.a_off, c);
        yb = fetch_off((nrow.w0 >> SH_BK) & 7, nrow.b_off, c);
        return;
    }
    // request the next row's operands first (two rows' loads in flight)
    ya = fetch_off((nrow.w0 >> SH_AK) & 7, nrow.a_off, c);
    yb = fetch_off((nrow.w0 >> SH_BK) & 7, nrow.b_off, c);
    // extra destinations: scalar loads issued now, consumed after the arithmetic (table padded by 4 entries)
    const uint64_t x0 = extras[xp], x1 = extras[xp + 1], x2 = extras[xp + 2], x3 = extras[xp + 3];
    fe a = xa, b = xb;
    if (ak == K_PREV) a = prev;
    else if (ak == K_LDS) a = lds_load_off((uint32_t)row.a_off, c);
    if (bk == K_PREV) b = prev;
    else if (bk == K_LDS) b = lds_load_off((uint32_t)row.b_off, c);
    fe d;
    bool has_d = true;
#ifdef CW_PROFILE
    const uint64_t seg_t0 = __builtin_readcyclecounter();
    asm volatile("" ::"v"(a.v[0]), "v"(a.v[7]), "v"(b.v[0]), "v"(b.v[7]));      // the operands are in registers here
    uint64_t seg_t1 = __builtin_readcyclecounter();
    CW_PROF_SEG(0, op, seg_t0);
#endif
    switch (op) {
    case D_COPY: d = a; break;
    case D_ADD: d = fe_add(a, b, P); break;
    case D_SUB: d = fe_sub(a, b, P); break;
    case D_NEG: d = fe_neg(a, P); break;
    case D_MMUL: d = fe_mmul(a, b, P); break;
    case D_MUL2: d = fe_mul2_auto(a, b, P); break;
    case D_MADD: d = fe_add(fe_mmul(a, b, P), prev, P); break;
    case D_MULC:
    case D_MADDC: {
        // operand b = c*R'; the constant after it = |val(c)| when the lowering flagged c as a small integer
        const uint32_t cs = (row.w0 >> SH_FLAG) & 3;
        uint64_t cmag = 0;
        if (cs) cmag = *(const uint64_t *)(c.Cb + row.b_off + 32);
        d = fe_mulc_auto(a, b, cs != 0, cmag, cs == 2, P);
        if (op == D_MADDC) d = fe_add(d, prev, P);
        break;
    }
    case D_LINSUM: d = eval_linsum<LW>(row.aux, bk == K_CONST ? b : fe_zero(), prev, c, P); break;
    case D_DOTC: d = eval_dotc(row.aux, bk == K_CONST ? b : fe_zero(), prev, c, P); break;
    case D_BIT: {                                                    // (a >> k) & 1, k = row.aux (wave-uniform)
        const uint32_t k = row.aux, w = k >> 5;
        const uint32_t limb = w == 0 ? a.v[0] : w == 1 ? a.v[1] : w == 2 ? a.v[2] : w == 3 ? a.v[3] : w == 4 ? a.v[4]
                              : w == 5 ? a.v[5] : w == 6 ? a.v[6] : a.v[7];
        d = fe_small(k < 256 ? (limb >> (k & 31)) & 1u : 0u);
        break;
    }
    case D_BITS: {
        // consecutive bits of a, from bit row.aux: every entry of the row's extra-destination list is one store of the current
        // bit; an entry flagged X_NEXT_DEV moves on to the next bit first (cw_tape.h).  One row for a whole Num2Bits.  Sixteen
        // entries per trip: their scalar loads are in flight together (the device table is padded by sixteen, entries past the row's end
        // are skipped), so a 64-bit range check waits for the table four times, not sixteen.
        uint32_t k = row.aux;
        for (uint32_t e = 0; e < nx; e += 16) {
            uint64_t ys[16];
            const uint64_t *ex = extras + xp + e;
            FE_UNROLL for (int t = 0; t < 16; t++) ys[t] = ex[t];
            FE_UNROLL for (int t = 0; t < 16; t++) {
                if (e + t < nx) {
                    k += (uint32_t)(ys[t] >> 62) & 1u;               // wave-uniform
                    const uint32_t w = k >> 5;
                    const uint32_t limb = w == 0 ? a.v[0] : w == 1 ? a.v[1] : w == 2 ? a.v[2] : w == 3 ? a.v[3] : w == 4 ? a.v[4]
                                          : w == 5 ? a.v[5] : w == 6 ? a.v[6] : a.v[7];
                    const uint32_t bit = (limb >> (k & 31)) & 1u;
                    const uint64_t off = ys[t] & ~(X_NEXT_DEV | X_LO_DEV);
                    if (ys[t] & X_LO_DEV) *(uint4 *)((char *)c.Vb + off + c.vlo) = make_uint4(bit, 0u, 0u, 0u);   // (wave-uniform choice)
                    else store_off(off, c, fe_small(bit));
                }
            }
        }
        has_d = false;
        break;
    }
    case D_SHL: d = fe_shl(a, b, P); break;
    case D_SHR: d = fe_shr(a, b, P); break;
    case D_BAND: d = fe_band(a, b, P); break;
    case D_BOR: d = fe_bor(a, b, P); break;
    case D_BXOR: d = fe_bxor(a, b, P); break;
    case D_BNOT: d = fe_bnot(a, P); break;
    case D_LT: d = fe_small(fe_lt(a, b, P)); break;
    case D_GT: d = fe_small(fe_lt(b, a, P)); break;
    case D_LEQ: d = fe_small(!fe_lt(b, a, P)); break;
    case D_GEQ: d = fe_small(!fe_lt(a, b, P)); break;
    case D_EQ: d = fe_small(fe_eq(a, b)); break;
    case D_NEQ: d = fe_small(!fe_eq(a, b)); break;
    case D_LAND: d = fe_small(!fe_is_zero(a) & !fe_is_zero(b)); break;
    case D_LOR: d = fe_small(!fe_is_zero(a) | !fe_is_zero(b)); break;
    case D_LNOT: d = fe_small(fe_is_zero(a)); break;
    case D_SELECT:                                                   // latch cond != 0; the EXT row selects
        selmask = __ballot(!fe_is_zero(a));
        has_d = false;
        break;
    case D_EXT: {
        const bool t = (selmask >> (c.lane16 >> 4)) & 1;
        for (int k = 0; k < 8; k++) d.v[k] = t ? a.v[k] : b.v[k];
        break;
    }
    case D_ASSERT_EQ:
        if (!fe_eq(a, b)) cw_fail(st, CW_ST_ASSERT_FAILED, row.aux);
        has_d = false;
        break;
    case D_ASSERT_NZ:
        if (fe_is_zero(a)) cw_fail(st, CW_ST_ASSERT_FAILED, row.aux);
        has_d = false;
        break;
    default:
        if (FULL_OPS) {
            switch (op) {
            case D_INV: d = fe_inv(a, P); break;
            case D_CALL:
                if (CALL_INLINE) eval_call_body(row.aux, row.b_off, (uint32_t)row.dst_off, st, c, P);
                else eval_call(row.aux, row.b_off, (uint32_t)row.dst_off, st, c, P);
                has_d = false;
                break;
            case D_POW: d = fe_pow(a, b, P); break;
            case D_IDIV:
            case D_MOD: {
                fe qq, rr;
                if (fe_is_zero(b)) {
                    cw_fail(st, CW_ST_ARITH, row.aux);
                    d = fe_zero();
                } else {
                    fe_divmod(a, b, &qq, &rr);
                    d = (op == D_IDIV) ? qq : rr;
                }
                break;
            }
            default: has_d = false; break;
            }
        } else {
            has_d = false;
        }
        break;
    }
#ifdef CW_PROFILE
    if (has_d) asm volatile("" ::"v"(d.v[0]), "v"(d.v[7]));
    CW_PROF_SEG(1, op, seg_t1);
    seg_t1 = __builtin_readcyclecounter();
#endif
    if (has_d) {
        prev = d;
        if (dk == K_LDS) lds_store_off((uint32_t)row.dst_off, c, d);
        else if (dk != KD_NONE) store_off(row.dst_off, c, d);
        // the first four table entries were requested before the arithmetic; longer fan-outs (a bit wired into dozens
        // of sub-components) fetch four entries per scalar load instead of waiting for one entry at a time
        uint64_t y0 = x0, y1 = x1, y2 = x2, y3 = x3;
        for (uint32_t e = 0; e < nx; e += 4) {
            if (e) { y0 = extras[xp + e]; y1 = extras[xp + e + 1]; y2 = extras[xp + e + 2]; y3 = extras[xp + e + 3]; }
            if (y0 >> 63) lds_store_off((uint32_t)y0, c, d); else store_off(y0, c, d);
            if (e + 1 < nx) { if (y1 >> 63) lds_store_off((uint32_t)y1, c, d); else store_off(y1, c, d); }
            if (e + 2 < nx) { if (y2 >> 63) lds_store_off((uint32_t)y2, c, d); else store_off(y2, c, d); }
            if (e + 3 < nx) { if (y3 >> 63) lds_store_off((uint32_t)y3, c, d); else store_off(y3, c, d); }
        }
    }
    xp += nx;
#ifdef CW_PROFILE
    CW_PROF_SEG(2, op, seg_t1);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&cw_prof_seg[(op & 63u) * 4 + 3], 1ull);
#endif
}

// ---- schedule evaluation (the hot path) ---------------------------------------------------------------
// One lane = one instance.  A workgroup = S waves ("strands") that all work on the SAME 64 instances,
// each walking its own row stream of the schedule; strands hand values to each other through LDS slots
// (or, when the LDS pool is exhausted, the value table) and meet at BARRIER rows
// (hip_elements/lower.py passes C/D).  S = 1 for large batches (instance parallelism alone fills the
// chip), S up to 16 for the small batches of the BASELINE configs.
// The kernel is VALU/SALU-issue bound (profiles/), so the interpreter is kept lean:
//  * rows are pre-resolved on the host for this batch (CwDRow: byte offsets instead of slot numbers), so an
//    operand address is one scalar 64-bit add and the load uses scalar-base + lane-offset addressing;
//  * operands of row r+1 are requested before row r executes, into the OTHER of two register sets (the loop
//    body is two interpreter steps with the sets swapped: no rotation moves).  Legal because the lowering
//    encodes any operand produced by the preceding row as kind PREV = register forwarding;
//  * pure copies never load: they are extra destinations of the row that produced the value;
//  * barriers never drain vmcnt (see (3) above): global stores stay in flight across them;
//  * products of small signed values take the per-wave short path (fp256.hip.h).
// FULL_OPS selects the variant that also carries the slow-path operators (INV/IDIV/MOD/POW).
// MAXT = the largest workgroup the instantiation is launched with: 1024 (16 strands: 128 VGPRs per wave) for every
// strand schedule; 64 for the single-strand schedules of circuits with run-time functions, where the interpreter is
// inlined (see eval_call).
template <bool FULL_OPS, int LW, int MAXT>
__global__ void __launch_bounds__(MAXT)
cw_eval_kernel(const CwDRow *__restrict__ rows, const uint32_t *__restrict__ stream_off,
               const uint64_t *__restrict__ extras, const uint32_t *__restrict__ extra_off,
               const uint64_t *__restrict__ terms, const uint32_t *__restrict__ term_off, uint4 *V,
               const uint32_t *__restrict__ consts, const uint32_t *__restrict__ lconsts, const uint4 *__restrict__ fcode,
               const uint4 *__restrict__ ftab, uint64_t slot_stride, uint32_t Bp, uint32_t batch,
               uint32_t lanes, uint32_t prio_mask, uint32_t *status, FpParams P) {
    // strand executed by this wave: rotated by the workgroup index, so that the strand carrying the critical chain
    // (the same one in every workgroup) does not land on the same SIMD of the CU in all co-resident workgroups
    const uint32_t nstr = blockDim.x >> 6;
    const uint32_t wave = (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + blockIdx.x) % nstr;
    const uint32_t lane = threadIdx.x & 63;
    // `lanes` (64, 32 or 16) instances per workgroup: small batches of long schedules are spread over more
    // workgroups (= more CUs, each with its own path to memory) by leaving the upper lanes of every wave idle.
    // The idle lanes are masked off for the whole kernel; every wave still reaches every barrier.
    // the strand(s) that carry the longest share of the schedule (the serial S-box chain of Poseidon) win the issue
    // arbitration of their SIMD against the waves of other workgroups: measured 1.03 -> 0.94 ms on Poseidon(2) x 65 536
    // (round 5: priority by wave AGE - the youngest of the four waves that share a SIMD arrives last at 74 % of the barriers of the
    // ECDSA verifier's levels - was measured: 294.6 / 298.1 ms against 294.2 / 296.1, no effect; removed)
    if ((prio_mask >> wave) & 1u) __builtin_amdgcn_s_setprio(3);
    if (lane < lanes) {
    const uint32_t i = blockIdx.x * lanes + lane;                  // < Bp (Bp is a multiple of 256 >= batch)
    EvalCtx c;
    c.Vb = (const char *)V;
    c.Cb = (const char *)consts;
    c.vlo = i * 16u;
    c.vhi = i * 16u + Bp * 16u;
    c.lane16 = lane * 16u;
    c.lds_hi = lanes * 16u;
    c.Lb = lconsts;
    c.fcode = fcode;
    c.ftab = ftab;
    c.slot_stride = slot_stride;
#ifdef CW_PROFILE
    c.prof_level = 0;
#endif
    c.terms = terms;
    c.tp = term_off[wave];
    // every stream is padded with 3 NOP rows, so rows[r+1..r+3] are always readable
    uint32_t r = stream_off[2 * wave];
    const uint32_t end = stream_off[2 * wave + 1];
    uint32_t xp = extra_off[wave];
    uint32_t st = 0;
    uint64_t selmask = 0;
    fe prev = fe_zero();
    CwDRow r0 = rows[r], r1 = rows[r + 1];
    fe a0 = fetch_off((r0.w0 >> SH_AK) & 7, r0.a_off, c), b0 = fetch_off((r0.w0 >> SH_BK) & 7, r0.b_off, c);
    fe a1, b1;
    uint64_t prof_t0 = 0;
    (void)prof_t0;
    while (r < end) {
        CwDRow r2 = rows[r + 2];
        CW_PROF_BEGIN();
        eval_step<FULL_OPS, LW, MAXT == 64>(r0, a0, b0, r1, a1, b1, prev, selmask, st, r, extras, xp, c, P);
        CW_PROF_END(r0);
        r0 = rows[r + 3];
        CW_PROF_BEGIN();
        eval_step<FULL_OPS, LW, MAXT == 64>(r1, a1, b1, r2, a0, b0, prev, selmask, st, r + 1, extras, xp, c, P);
        CW_PROF_END(r1);
        r1 = r0;
        r0 = r2;
        r += 2;
    }
    if (st && i < batch) cw_publish_status(status, i, st);
    }
}

// ---- pipelined single-wave schedule (hip_elements/pipe.py) -----------------------------------------------------------
// One wave = 64 (or 32 / 16) instances, rows in batches of NB.  No row waits for the value table:
//   * the far operands of batch k (value-table slots, constants) are listed in the batch's load list and copied
//     global -> LDS by the LDS-DMA path (global_load_lds_dwordx4: no VGPRs, no VALU) a whole batch ahead: L(k+1) is issued
//     when batch k starts, into the staging half (k+1) % 2;
//   * every result goes to a ring of 2*NB LDS entries (entry = row position mod 2*NB) besides the forwarding register, so
//     consumers up to 2*NB rows behind read LDS; the a/b operands of row r+1 are requested from LDS before row r executes;
//   * every row issues exactly two value-table stores (4 buffer_store_dwordx4; a NONE target is a zero-sized buffer, dropped
//     by the bounds check) and every batch exactly NLD loads, so 4*NB vector-memory instructions separate the issue of L(k)
//     from the start of batch k: `s_waitcnt vmcnt(4*NB)` there means "L(k) is in LDS" (hipcc does not count asm memory
//     operations, and the stores never need a wait).
// LDS: [ring 2*NB][staging 2*NLD] entries of 2 KiB ([2 halves][64 lanes] x 16 B).
struct CwPRow {      // 32 bytes, one scalar dwordx8 load
    uint32_t w0;     // op[0:8) | kind of a [8:11) | kind of b [11:14) | const flags [29:31); kinds: 0 none, K_PREV, K_LDS
    uint32_t aux;    // bit index / number of terms / row reported in the status word
    uint32_t abd;    // LDS entry of a | of b << 8 | ring entry of the result << 16 (0xFF = no value)
    uint32_t st0, st1;   // value-table slots the result is stored to (0xFFFFFFFF = none)
    uint32_t cm_lo, cm_hi;   // D_MULC / D_MADDC: |val(c)| of a small constant
    uint32_t pad;
};
static_assert(sizeof(CwPRow) == 32, "CwPRow is loaded with one s_load_dwordx8");

template <int NLD>
__device__ __forceinline__ void pipe_issue(const uint32_t *__restrict__ loads, uint32_t k, uint32_t ring_entries, uint64_t vbase,
                                           uint64_t cbase, uint64_t slot_bytes, uint64_t half_bytes, uint32_t voff, uint32_t lds_base) {
#ifdef CW_PEXP_NOLOAD          // timing experiment only (tools/pipe_exp.sh): results are garbage
    return;
#endif
    // the whole list with one scalar load (NLD words, 16-byte aligned)
    const uint4 *lp = (const uint4 *)(loads + (size_t)k * NLD);
    uint32_t lws[NLD];
    {
        const uint4 q0 = lp[0];
        lws[0] = q0.x; lws[1] = q0.y; lws[2] = q0.z; lws[3] = q0.w;
        if (NLD == 8) {
            const uint4 q1 = lp[1];
            lws[NLD - 4] = q1.x; lws[NLD - 3] = q1.y; lws[NLD - 2] = q1.z; lws[NLD - 1] = q1.w;
        }
    }
#pragma unroll
    for (int j = 0; j < NLD; j++) {
        const uint32_t lw = lws[j];
        const bool none = lw == 0xFFFFFFFFu, is_const = !none && (lw >> 31);
        const uint32_t idx = none ? 0u : (lw & 0x7FFFFFFFu);
        const uint64_t lo = is_const ? cbase + (uint64_t)idx * 32u : vbase + (uint64_t)idx * slot_bytes;
        const uint64_t hi = lo + (is_const ? 16u : half_bytes);
        const uint32_t vo = is_const ? 0u : voff;                   // constants: every lane copies the same 16 bytes
        const uint32_t dlo = lds_base + (ring_entries + (k & 1u) * NLD + j) * 2048u, dhi = dlo + 1024u;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
                     "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(vo), "s"(dlo), "s"(dhi), "s"(lo), "s"(hi)
                     : "memory");
    }
}

typedef uint32_t pipe_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pipe_store(uint32_t slot, const fe &x, const char *Vb, uint64_t slot_bytes, uint32_t vlo,
                                           uint32_t vhi) {
#ifdef CW_PEXP_NOSTORE
    return;
#endif
    const bool none = slot == 0xFFFFFFFFu;
    const char *base = Vb + (uint64_t)(none ? 0u : slot) * slot_bytes;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, none ? 0 : (int)(uint32_t)slot_bytes, 0x00020000);
    pipe_u32x4 lo = {x.v[0], x.v[1], x.v[2], x.v[3]}, hi = {x.v[4], x.v[5], x.v[6], x.v[7]};
    __builtin_amdgcn_raw_buffer_store_b128(lo, rs, (int)vlo, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(hi, rs, (int)vhi, 0, 0);
}

template <bool FULL_OPS, int LW>
__device__ __forceinline__ void pipe_step(const CwPRow &row, const fe &xa, const fe &xb, const CwPRow &nrow, fe &ya, fe &yb,
                                          fe &prev, uint64_t &selmask, uint32_t &st, EvalCtx &c, uint64_t slot_bytes,
                                          const FpParams &P) {
    const uint32_t op = row.w0 & 0xFF, ak = (row.w0 >> 8) & 7, bk = (row.w0 >> 11) & 7;
    // the next row's LDS operands first (its producer is never THIS row: that would be kind PREV)
    ya = lds_load_off((nrow.abd & 0xFFu) << 11, c);
    yb = lds_load_off(((nrow.abd >> 8) & 0xFFu) << 11, c);
    fe a = xa, b = xb;
    if (ak == K_PREV) a = prev;
    if (bk == K_PREV) b = prev;
    fe d = prev;
    bool has_d = true;
    switch (op) {
    case D_COPY: d = a; break;
    case D_ADD: d = fe_add(a, b, P); break;
    case D_SUB: d = fe_sub(a, b, P); break;
    case D_NEG: d = fe_neg(a, P); break;
    case D_MMUL: d = fe_mmul(a, b, P); break;
    case D_MUL2: d = fe_mul2_auto(a, b, P); break;
    case D_MADD: d = fe_add(fe_mmul(a, b, P), prev, P); break;
    case D_MULC:
    case D_MADDC: {
        const uint32_t cs = (row.w0 >> SH_FLAG) & 3;
        d = fe_mulc_auto(a, b, cs != 0, ((uint64_t)row.cm_hi << 32) | row.cm_lo, cs == 2, P);
        if (op == D_MADDC) d = fe_add(d, prev, P);
        break;
    }
    case D_LINSUM: d = eval_linsum<LW, true>(row.aux, bk ? b : fe_zero(), prev, c, P); break;
    case D_DOTC: d = eval_dotc<true>(row.aux, bk ? b : fe_zero(), prev, c, P); break;
    case D_BIT: {
        const uint32_t k = row.aux, w = k >> 5;
        const uint32_t limb = w == 0 ? a.v[0] : w == 1 ? a.v[1] : w == 2 ? a.v[2] : w == 3 ? a.v[3] : w == 4 ? a.v[4]
                              : w == 5 ? a.v[5] : w == 6 ? a.v[6] : a.v[7];
        d = fe_small(k < 256 ? (limb >> (k & 31)) & 1u : 0u);
        break;
    }
    case D_SHL: d = fe_shl(a, b, P); break;
    case D_SHR: d = fe_shr(a, b, P); break;
    case D_BAND: d = fe_band(a, b, P); break;
    case D_BOR: d = fe_bor(a, b, P); break;
    case D_BXOR: d = fe_bxor(a, b, P); break;
    case D_BNOT: d = fe_bnot(a, P); break;
    case D_LT: d = fe_small(fe_lt(a, b, P)); break;
    case D_GT: d = fe_small(fe_lt(b, a, P)); break;
    case D_LEQ: d = fe_small(!fe_lt(b, a, P)); break;
    case D_GEQ: d = fe_small(!fe_lt(a, b, P)); break;
    case D_EQ: d = fe_small(fe_eq(a, b)); break;
    case D_NEQ: d = fe_small(!fe_eq(a, b)); break;
    case D_LAND: d = fe_small(!fe_is_zero(a) & !fe_is_zero(b)); break;
    case D_LOR: d = fe_small(!fe_is_zero(a) | !fe_is_zero(b)); break;
    case D_LNOT: d = fe_small(fe_is_zero(a)); break;
    case D_SELECT:
        selmask = __ballot(!fe_is_zero(a));
        has_d = false;
        break;
    case D_EXT: {
        const bool t = (selmask >> (c.lane16 >> 4)) & 1;
        for (int k = 0; k < 8; k++) d.v[k] = t ? a.v[k] : b.v[k];
        break;
    }
    case D_ASSERT_EQ:
        if (!fe_eq(a, b)) cw_fail(st, CW_ST_ASSERT_FAILED, row.aux);
        has_d = false;
        break;
    case D_ASSERT_NZ:
        if (fe_is_zero(a)) cw_fail(st, CW_ST_ASSERT_FAILED, row.aux);
        has_d = false;
        break;
    default:
        has_d = false;
        if (FULL_OPS) {
            has_d = true;
            switch (op) {
            case D_INV: d = fe_inv(a, P); break;
            case D_POW: d = fe_pow(a, b, P); break;
            case D_IDIV:
            case D_MOD: {
                fe qq, rr;
                if (fe_is_zero(b)) {
                    cw_fail(st, CW_ST_ARITH, row.aux);
                    d = fe_zero();
                } else {
                    fe_divmod(a, b, &qq, &rr);
                    d = (op == D_IDIV) ? qq : rr;
                }
                break;
            }
            default: has_d = false; break;
            }
        }
        break;
    }
    if (has_d) {
        prev = d;
        const uint32_t de = (row.abd >> 16) & 0xFFu;
        if (de != 0xFFu) lds_store_off(de << 11, c, d);
    }
    // always two stores = four vector-memory instructions (rows without a value carry no targets)
    pipe_store(row.st0, d, c.Vb, slot_bytes, c.vlo, c.vhi);
    pipe_store(row.st1, d, c.Vb, slot_bytes, c.vlo, c.vhi);
}

template <bool FULL_OPS, int LW, int NB, int NLD>
__global__ void __launch_bounds__(64)
cw_pipe_kernel(const CwPRow *__restrict__ rows, uint32_t n_rows, const uint32_t *__restrict__ loads,
               const uint64_t *__restrict__ terms, uint4 *V, const uint32_t *__restrict__ consts,
               const uint32_t *__restrict__ lconsts, uint64_t slot_stride, uint32_t Bp, uint32_t batch, uint32_t lanes,
               uint32_t *status, FpParams P) {
    static_assert(NB == 4 || NB == 8, "the wait below is written for these");
    const uint32_t lane = threadIdx.x;
    if (lane < lanes) {
        const uint32_t i = blockIdx.x * lanes + lane;              // < Bp
        EvalCtx c;
        c.Vb = (const char *)V;
        c.Cb = (const char *)consts;
        c.vlo = i * 16u;
        c.vhi = i * 16u + Bp * 16u;
        c.lane16 = lane * 16u;
        c.lds_hi = 1024u;                                            // the pipelined variant's entries are sized for 64 lanes
        c.Lb = lconsts;
        c.fcode = nullptr;
        c.ftab = nullptr;
        c.slot_stride = slot_stride;
        c.terms = terms;
        c.tp = 0;
        const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)cw_lds;
        const uint64_t vbase = (uint64_t)V, cbase = (uint64_t)consts, half_bytes = (uint64_t)Bp * 16u;
        pipe_issue<NLD>(loads, 0, 2 * NB, vbase, cbase, slot_stride, half_bytes, c.vlo, lds_base);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pipe_issue<NLD>(loads, 1, 2 * NB, vbase, cbase, slot_stride, half_bytes, c.vlo, lds_base);
        uint32_t st = 0;
        uint64_t selmask = 0;
        fe prev = fe_zero();
#ifdef CW_PROFILE                  // per-opcode clocks of workgroup 0 in LDS behind the value entries (tools/profile_ops.sh)
        unsigned long long *prof_tab = (unsigned long long *)((char *)cw_lds + (2 * NB + 2 * NLD) * 2048);
        unsigned long long prof_t0 = 0;
        if (lane == 0)
            for (int k = 0; k < 128; k++) prof_tab[k] = 0;
#define PIPE_PROF_BEGIN() prof_t0 = __builtin_readcyclecounter()
#define PIPE_PROF_END(row)                                                                   \
    do {                                                                                     \
        if (blockIdx.x == 0 && lane == 0) {                                                  \
            prof_tab[((row).w0 & 63u) * 2] += __builtin_readcyclecounter() - prof_t0;       \
            prof_tab[((row).w0 & 63u) * 2 + 1] += 1;                                         \
        }                                                                                    \
    } while (0)
#else
#define PIPE_PROF_BEGIN()
#define PIPE_PROF_END(row)
#endif
        // the row table is padded by 3 NOP rows
        CwPRow r0 = rows[0], r1 = rows[1];
        fe a0 = fe_zero(), b0 = fe_zero(), a1, b1;
        for (uint32_t r = 0; r < n_rows; r += 2) {
            const CwPRow r2 = rows[r + 2];
            if ((r & (NB - 1)) == 0) {
                if (r) {
                    // 4*NB stores were issued since L(r / NB): everything older has completed
#ifndef CW_PEXP_NOLOAD
                    if (NB == 8) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#endif
                    pipe_issue<NLD>(loads, r / NB + 1, 2 * NB, vbase, cbase, slot_stride, half_bytes, c.vlo, lds_base);
                }
                // first row of a batch: its staged operands have only just landed
                a0 = lds_load_off((r0.abd & 0xFFu) << 11, c);
                b0 = lds_load_off(((r0.abd >> 8) & 0xFFu) << 11, c);
            }
            PIPE_PROF_BEGIN();
            pipe_step<FULL_OPS, LW>(r0, a0, b0, r1, a1, b1, prev, selmask, st, c, slot_stride, P);
            PIPE_PROF_END(r0);
            r0 = rows[r + 3];
            PIPE_PROF_BEGIN();
            pipe_step<FULL_OPS, LW>(r1, a1, b1, r2, a0, b0, prev, selmask, st, c, slot_stride, P);
            PIPE_PROF_END(r1);
            r1 = r0;
            r0 = r2;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the last load list must not outlive the wave's LDS
#ifdef CW_PROFILE
        if (blockIdx.x == 0 && lane == 0)
            for (int k = 0; k < 128; k++) atomicAdd(&cw_prof[k], prof_tab[k]);
#endif
        if (st && i < batch) cw_publish_status(status, i, st);
    }
}

// ---- R1CS check:  (A.w) * (B.w) == C.w  for every constraint row and instance ---------------------------
// One single-wave workgroup = 64 instances x one chunk of constraint rows (uniform control flow).  The host
// flattens the rows into a term stream (cw_r1cs_plan.h): word 0 = value slot | accumulator (A/B/C) | row-end
// kind, word 1 = coefficient id.  Coefficient ids 0/1 mean +1/-1 (add/sub); other ids index ctab, which holds
// c*R' mod q so that one MMUL gives w*c on canonical w; a term on the constant-1 wire carries the canonical
// coefficient instead and costs no multiplication.  The host orders the rows by the time their youngest wire
// is produced by the schedule (temporal locality: a wire is re-read while still in L2) and marks pure
// equalities x - y = 0 (component wiring, ~75 % of the rows at --O0), which are checked by comparison.
// Terms are scalar-loaded and their wires fetched one term ahead of use, so a wave keeps two value loads in
// flight.  `row_orig` maps the processing order back to the constraint index of the .r1cs file.
// `cur` accumulates the part (A, B or C) being read; the last term of A or B (bit 31) files it under its name,
// C stays in `cur`.  Written with selects so that A/B stay in registers (an `if (part == ..)` ladder over
// three accumulators is turned into an indexed stack array).
struct R1State {
    fe A, B, cur;
    uint32_t row, bad;
    bool allbool;           // the T_BOOL term just processed found 0 or 1 in every lane of the wave (cw_r1cs_plan.h COEF_BITSEL)
};
__device__ __forceinline__ fe fe_pick(bool c, const fe &a, const fe &b) {
    fe r;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = c ? a.v[k] : b.v[k];
    return r;
}
template <bool MONT>
__device__ __forceinline__ void r1_term(const fe &wv, uint32_t w0, uint32_t ci, R1State &s,
                                        const uint32_t *__restrict__ ctab, const uint32_t *__restrict__ ctab29,
                                        const uint32_t *__restrict__ row_orig, const FpParams &P) {
    const uint32_t acc = (w0 >> 27) & 3u, endk = (w0 >> 29) & 3u;
    bool ok = true;
    if (w0 & (1u << 26)) {                                          // (cwplan::T_BOOL) the row b * (b - 1) = 0: b is 0 or 1
        fe one_ = fe_small(1);
        if (MONT) { FE_UNROLL for (int k = 0; k < 8; k++) one_.v[k] = P.one_m[k]; }
        ok = fe_is_zero(wv) | fe_eq(wv, one_);
        s.allbool = __all(ok);
    } else if (endk == 3) ok = fe_eq(s.cur, wv);                    // second wire of a pure equality row: x == y
    else if (acc == 3) s.cur = wv;                                  // its first wire
    else {
        fe w = wv;
        if (ci >> 31) w = c_load(ctab, ci & 0x7FFFFFFFu);           // coefficient on the constant-1 wire
        else if ((ci & (1u << 30)) && s.allbool) {                  // (cwplan::COEF_BITSEL) the wire is 0 or 1 in every lane: b ? c : 0
            const fe c1 = c_load(ctab, (ci & 0x3FFFFFFFu) + 1);
            const bool nz = !fe_is_zero(w);
            FE_UNROLL for (int k = 0; k < 8; k++) w.v[k] = nz ? c1.v[k] : 0u;
            ci = 0;
        } else if (ci >= 2) {                                       // w * c: ctab29 holds c*R' as 9 x 29-bit limbs
            fe29 cc;
            FE_UNROLL for (int k = 0; k < 9; k++) cc.l[k] = ctab29[(size_t)(ci & 0x3FFFFFFFu) * 9 + k];
            w = fe_from29(fe29_mmul(fe_to29(w), cc, P));
        }
        s.cur = (ci == 1) ? fe_sub(s.cur, w, P) : fe_add(s.cur, w, P);
        if ((w0 >> 31) && acc != 2) {                               // last term of part A or B
            s.A = fe_pick(acc == 0, s.cur, s.A);
            s.B = fe_pick(acc == 1, s.cur, s.B);
            s.cur = fe_zero();
        }
        if (endk == 2) ok = fe_is_zero(s.cur);                      // A or B empty: linear row, C must vanish
        else if (endk == 1) {
            if (MONT) {                                             // wires in Montgomery form: mmul(A~, B~) = (AB)~ == C~
                ok = fe_eq(fe_from29(fe29_mmul(fe_to29(s.A), fe_to29(s.B), P)), s.cur);
            } else {                                                // A*B == C  <=>  (A*B + (q - C)) / R' == 0 (mod q)
                const fe29 z = fe29_mmul_add(fe_to29(s.A), fe_to29(s.B), fe_to29(fe_neg(s.cur, P)), P);
                uint32_t o = 0;
                FE_UNROLL for (int k = 0; k < 9; k++) o |= z.l[k];
                ok = (o == 0);
            }
        }
    }
    if (endk) {
        if (__any(!ok)) {
            const uint32_t oc = row_orig[s.row];
            if (!ok && oc < s.bad) s.bad = oc;
        }
        s.row++;
        // a boolean row may ride inside another row, in front of that row's term on the same wire: the accumulators are its host's
        if (!(w0 & (1u << 26))) { s.A = fe_zero(); s.B = fe_zero(); s.cur = fe_zero(); }
    }
}
__device__ __forceinline__ void r1_finish(const R1State &s, uint32_t i, uint32_t batch, uint32_t *status, uint32_t *first_bad) {
    if (s.bad != 0xFFFFFFFFu && i < batch) {
        atomicMin(&first_bad[i], s.bad);
        atomicOr(&status[i], CW_ST_R1CS_FAILED);
    }
}

template <bool MONT>
__global__ void __launch_bounds__(64)
cw_r1cs_stream_kernel(const uint4 *__restrict__ chunk, uint32_t n_chunks, const uint2 *__restrict__ terms, const uint32_t *__restrict__ ctab,
                      const uint32_t *__restrict__ ctab29, const uint32_t *__restrict__ row_orig, const uint4 *__restrict__ V, uint32_t Bp, uint32_t batch,
                      uint32_t *status, uint32_t *first_bad, FpParams P) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;               // < Bp
    R1State s;
    s.bad = 0xFFFFFFFFu;
    s.allbool = false;
    for (uint32_t cix = blockIdx.y; cix < n_chunks; cix += gridDim.y) {   // grid.y is capped at 65535: large systems loop
        const uint4 ch = chunk[cix];                                // first term, n terms, -, first row
        const uint2 *tp = terms + ch.x;                             // the stream is padded: tp[n] is readable
        s.A = fe_zero(); s.B = fe_zero(); s.cur = fe_zero();
        s.row = ch.w;
        uint2 t0 = tp[0];
        fe w0 = v_load(V, t0.x & 0x3FFFFFFu, Bp, i);
        // the next term's wire is in flight while this one is accumulated (two terms ahead costs 16 more VGPRs, which
        // drops a wave per SIMD, and measured no faster)
        for (uint32_t k = 0; k < ch.y; k++) {
            const uint2 t1 = tp[k + 1];
            fe w1 = w0;                                             // a term on the wire of the term before it (a folded boolean row
            if (((t1.x ^ t0.x) & 0x3FFFFFFu) != 0) w1 = v_load(V, t1.x & 0x3FFFFFFu, Bp, i);   // and its host's term): one read
            r1_term<MONT>(w0, t0.x, t0.y, s, ctab, ctab29, row_orig, P);
            t0 = t1; w0 = w1;
        }
    }
    r1_finish(s, i, batch, status, first_bad);
}

// ---- R1CS check, staged through LDS (plan: cw_r1cs_plan.h) -------------------------------------------------
// One single-wave workgroup = 64 instances x one chunk of rows.  Every distinct wire of the chunk is copied
// global -> LDS once by the LDS-DMA path (global_load_lds_dwordx4: no VGPRs, no VALU, lane-linear 1 KiB per
// half), R1_DEPTH wires ahead of its first use; terms read LDS only.  The host chose the LDS entry of every
// load (Belady under the in-flight hazard rule) so the kernel has no tags and no misses.  hipcc does not count
// asm memory operations, so the wait is ours: after issuing load j+DEPTH, vmcnt(2*DEPTH) means load j landed.
#define R1_DEPTH 4
static_assert(R1_DEPTH == 4, "keep in sync with cwplan::DEPTH and the s_waitcnt immediate below");

__device__ __forceinline__ void r1_issue(uint32_t lw, uint64_t vbase, uint64_t slot_bytes, uint64_t half_bytes,
                                         uint32_t voff, uint32_t lds_base) {
    const uint32_t slot = lw & 0x3FFFFFFu, e = lw >> 26;
    const uint64_t lo = vbase + (uint64_t)slot * slot_bytes, hi = lo + half_bytes;
    const uint32_t dlo = lds_base + e * 2048u, dhi = dlo + 1024u;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
                 "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(dlo), "s"(dhi), "s"(lo), "s"(hi)
                 : "memory");
}

template <bool MONT>
__global__ void __launch_bounds__(64)
cw_r1cs_staged_kernel(const uint4 *__restrict__ chunk, const uint2 *__restrict__ rec, const uint2 *__restrict__ terms,
                      const uint32_t *__restrict__ ctab, const uint32_t *__restrict__ ctab29,
                      const uint32_t *__restrict__ row_orig, const uint4 *__restrict__ V, uint32_t Bp, uint32_t batch,
                      uint32_t *status, uint32_t *first_bad, FpParams P) {
    extern __shared__ uint4 r1_lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t i = blockIdx.x * 64 + lane;                      // < Bp
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)r1_lds;
    const uint4 ch = chunk[blockIdx.y];                             // first record, n loads, first term, first row
    const uint64_t vbase = (uint64_t)V + (uint64_t)blockIdx.x * 1024u;   // this group's 64 x 16 B window of a half-row
    const uint64_t half_bytes = (uint64_t)Bp * 16u, slot_bytes = 2 * half_bytes;
    const uint32_t voff = lane * 16u;
    const uint2 *rp = rec + ch.x;
#pragma unroll
    for (int d = 0; d < R1_DEPTH; d++) r1_issue(rp[d].x, vbase, slot_bytes, half_bytes, voff, lds_base);
    rp += R1_DEPTH;
    const uint2 *tp = terms + ch.z;
    R1State s;
    s.A = fe_zero(); s.B = fe_zero(); s.cur = fe_zero();
    s.row = ch.w; s.bad = 0xFFFFFFFFu;
    s.allbool = false;
    uint2 tw = tp[0];
    for (uint32_t j = 0; j < ch.y; j++) {
        const uint2 r = rp[j];
        r1_issue(r.x, vbase, slot_bytes, half_bytes, voff, lds_base);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // 2 * R1_DEPTH: load j is in LDS
        for (uint32_t k = 0; k < r.y; k++) {
            const uint2 nx = *++tp;                                 // next term's words while this one computes
            const uint32_t e = tw.x & 0x3FFFFFFu;
            const uint4 lo = r1_lds[e * 128u + lane], hi = r1_lds[e * 128u + 64u + lane];
            fe w;
            w.v[0] = lo.x; w.v[1] = lo.y; w.v[2] = lo.z; w.v[3] = lo.w;
            w.v[4] = hi.x; w.v[5] = hi.y; w.v[6] = hi.z; w.v[7] = hi.w;
            r1_term<MONT>(w, tw.x, tw.y, s, ctab, ctab29, row_orig, P);
            tw = nx;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // padding loads must not outlive the wave's LDS
    r1_finish(s, i, batch, status, first_bad);
}

// ---- egress: one instance's witness as [n_witness][32 B] (getWitness + Fr_toLongNormal, main.cpp:326-332) ----
// MONT: the table holds Montgomery forms: x~ -> x = mmul(x~, 1)   (the role of Fr_toLongNormal, generic/fr.cpp)
template <bool MONT>
__global__ void __launch_bounds__(CW_BLOCK)
cw_gather_kernel(const uint4 *__restrict__ V, const uint32_t *__restrict__ w2s, uint32_t n_wit, uint32_t Bp,
                 uint32_t instance, uint4 *__restrict__ out, FpParams P) {
    uint32_t k = blockIdx.x * CW_BLOCK + threadIdx.x;
    if (k >= n_wit) return;
    size_t base = (size_t)w2s[k] * 2 * Bp + instance;
    uint4 lo = V[base], hi = V[base + Bp];
    if (MONT) {
        fe x;
        x.v[0] = lo.x; x.v[1] = lo.y; x.v[2] = lo.z; x.v[3] = lo.w; x.v[4] = hi.x; x.v[5] = hi.y; x.v[6] = hi.z; x.v[7] = hi.w;
        x = fe_mmul(x, fe_small(1), P);
        lo = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
        hi = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    }
    out[2 * (size_t)k] = lo;
    out[2 * (size_t)k + 1] = hi;
}

// ---- bulk egress: witnesses of `count` consecutive instances as [count][n_witness][32 B] ----------------------
// SoA -> AoS transpose through LDS: a 256-thread block moves a tile of 64 instances x GM_TILE witness elements.
// Reads are coalesced along instances (the table's layout), writes along the witness index (the output's).
#ifndef GM_TILE
#define GM_TILE 16        // witness elements per block: 33 KB of LDS, four blocks per CU.  Poseidon(2) x 65 536 (2.3 GB image, Montgomery
                          // -> canonical on the way): 16 -> 0.86 ms = 2.69 TB/s written + as much read (the copy roof of
                          // tools/ubench_isa is 2.4 + 2.4); 8 -> 1.05 ms; 32 (66 KB, two blocks per CU) -> 1.19 ms; round 2's
                          // kernel (32, default-policy stores) 1.28 ms
#endif
template <bool MONT>
__global__ void __launch_bounds__(256)
cw_gather_many_kernel(const uint4 *__restrict__ V, const uint32_t *__restrict__ w2s, uint32_t n_wit, uint32_t Bp,
                      uint32_t first, uint32_t count, uint4 *__restrict__ out, FpParams P) {
    __shared__ uint4 tile[GM_TILE][2][65];                          // [element][half][instance] (+1: bank spread)
    const uint32_t k0 = blockIdx.x * GM_TILE, i0 = blockIdx.y * 64;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;  // 4 waves
    for (uint32_t e = wv; e < GM_TILE; e += 4) {
        const uint32_t k = k0 + e;
        if (k < n_wit && i0 + lane < count) {
            const size_t base = (size_t)w2s[k] * 2 * Bp + first + i0 + lane;
            uint4 lo = V[base], hi = V[base + Bp];
            if (MONT) {
                fe x;
                x.v[0] = lo.x; x.v[1] = lo.y; x.v[2] = lo.z; x.v[3] = lo.w; x.v[4] = hi.x; x.v[5] = hi.y; x.v[6] = hi.z; x.v[7] = hi.w;
                x = fe_mmul(x, fe_small(1), P);
                lo = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
                hi = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
            }
            tile[e][0][lane] = lo;
            tile[e][1][lane] = hi;
        }
    }
    __syncthreads();
    // 2 * GM_TILE consecutive 16-byte pieces of one instance's row = GM_TILE elements x 2 halves; a wave writes 64 / (2 *
    // GM_TILE) instances per store instruction, streaming (nothing reads the image back on this device)
    constexpr uint32_t PIECES = 2 * GM_TILE, PER = 64 / PIECES;
    const uint32_t piece = lane % PIECES, sub = lane / PIECES, e = piece >> 1, h = piece & 1;
    for (uint32_t ii = wv * PER + sub; ii < 64; ii += 4 * PER) {
        const uint32_t k = k0 + e;
        if (k < n_wit && i0 + ii < count) {
            const uint4 v = tile[e][h][ii];
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 w = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w, (u32x4 *)&out[((size_t)(i0 + ii) * n_wit + k) * 2 + h]);
        }
    }
}

// ---- Fp multiplication micro-benchmark: iters dependent Montgomery products per lane --------------------
__global__ void __launch_bounds__(CW_BLOCK)
cw_mulbench_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b, uint4 *__restrict__ out, uint32_t n,
                   uint32_t iters, FpParams P) {
    uint32_t i = blockIdx.x * CW_BLOCK + threadIdx.x;
    if (i >= n) return;
    fe x = aos_load(a, i), y = aos_load(b, i);   // AoS operands: element i = uint4[2i], uint4[2i+1]
    fe29 x29 = fe_to29(x);
    const fe29 y29 = fe_to29(y);
    for (uint32_t k = 0; k < iters; k++) x29 = fe29_mmul(x29, y29, P);
    x = fe_from29(x29);
    out[2 * (size_t)i] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    out[2 * (size_t)i + 1] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
}

// ---- element-wise single-operator kernel (unit-test hook for the device field functions) ------------------
__global__ void __launch_bounds__(CW_BLOCK)
cw_fpop_kernel(uint32_t op, const uint4 *a_, const uint4 *b_, const uint4 *c_, uint4 *out, uint32_t *status, uint32_t n,
               FpParams P) {
    uint32_t i = blockIdx.x * CW_BLOCK + threadIdx.x;
    if (i >= n) return;
    fe a = aos_load(a_, i), b = aos_load(b_, i), c = aos_load(c_, i);
    fe d = fe_zero();
    uint32_t st = 0;
    switch (op) {
    case D_COPY: d = a; break;
    case D_ADD: d = fe_add(a, b, P); break;
    case D_SUB: d = fe_sub(a, b, P); break;
    case D_NEG: d = fe_neg(a, P); break;
    case D_MMUL: d = fe_mmul(a, b, P); break;
    case D_MUL2: d = fe_mul2_auto(a, b, P); break;       // includes the per-wave short path
    case D_MADD: d = fe_add(fe_mmul(a, b, P), c, P); break;
    case D_INV: d = fe_inv(a, P); break;
    case D_POW: d = fe_pow(a, b, P); break;
    case D_IDIV:
    case D_MOD: {
        fe qq, rr;
        if (fe_is_zero(b)) { st = CW_ST_ARITH; }
        else { fe_divmod(a, b, &qq, &rr); d = (op == D_IDIV) ? qq : rr; }
        break;
    }
    case D_SHL: d = fe_shl(a, b, P); break;
    case D_SHR: d = fe_shr(a, b, P); break;
    case D_BAND: d = fe_band(a, b, P); break;
    case D_BOR: d = fe_bor(a, b, P); break;
    case D_BXOR: d = fe_bxor(a, b, P); break;
    case D_BNOT: d = fe_bnot(a, P); break;
    case D_LT: d = fe_small(fe_lt(a, b, P)); break;
    case D_GT: d = fe_small(fe_lt(b, a, P)); break;
    case D_LEQ: d = fe_small(!fe_lt(b, a, P)); break;
    case D_GEQ: d = fe_small(!fe_lt(a, b, P)); break;
    case D_EQ: d = fe_small(fe_eq(a, b)); break;
    case D_NEQ: d = fe_small(!fe_eq(a, b)); break;
    case D_LAND: d = fe_small(!fe_is_zero(a) & !fe_is_zero(b)); break;
    case D_LOR: d = fe_small(!fe_is_zero(a) | !fe_is_zero(b)); break;
    case D_LNOT: d = fe_small(fe_is_zero(a)); break;
    case D_SELECT: { bool t = !fe_is_zero(a); for (int k = 0; k < 8; k++) d.v[k] = t ? b.v[k] : c.v[k]; break; }
    case D_ASSERT_EQ: if (!fe_eq(a, b)) st = CW_ST_ASSERT_FAILED; break;
    case D_ASSERT_NZ: if (fe_is_zero(a)) st = CW_ST_ASSERT_FAILED; break;
    default: break;
    }
    out[2 * (size_t)i] = make_uint4(d.v[0], d.v[1], d.v[2], d.v[3]);
    out[2 * (size_t)i + 1] = make_uint4(d.v[4], d.v[5], d.v[6], d.v[7]);
    status[i] = st;
}

// ---- launch wrappers -----------------------------------------------------------------------------------------
static inline dim3 blocks_for(uint32_t n) { return dim3((n + CW_BLOCK - 1) / CW_BLOCK); }

hipError_t cwk_init(hipStream_t s, void *V, uint32_t Bp, uint32_t *status, uint32_t *first_bad, bool mont, const FpParams &P) {
    fe one;
    for (int k = 0; k < 8; k++) one.v[k] = mont ? P.one_m[k] : (k == 0 ? 1u : 0u);
    hipLaunchKernelGGL(cw_init_kernel, blocks_for(Bp), dim3(CW_BLOCK), 0, s, (uint4 *)V, Bp, status, first_bad, one);
    return hipGetLastError();
}
hipError_t cwk_ingest(hipStream_t s, const void *in, void *V, uint32_t input_start, uint32_t n_in, uint32_t batch,
                      uint32_t Bp, bool mont, const FpParams &P) {
    if (n_in == 0) return hipSuccess;
    dim3 g((batch + CW_BLOCK - 1) / CW_BLOCK, n_in < 65535u ? n_in : 65535u);
    if (mont)
        hipLaunchKernelGGL(cw_ingest_kernel<true>, g, dim3(CW_BLOCK), 0, s, (const uint4 *)in, (uint4 *)V, input_start, n_in, batch, Bp, P);
    else
        hipLaunchKernelGGL(cw_ingest_kernel<false>, g, dim3(CW_BLOCK), 0, s, (const uint4 *)in, (uint4 *)V, input_start, n_in, batch, Bp, P);
    return hipGetLastError();
}
hipError_t cwk_eval(hipStream_t s, bool full, bool wide_linsum, const CwDRow *rows, const uint32_t *stream_off,
                    const uint64_t *extras, const uint32_t *extra_off, const uint64_t *terms, const uint32_t *term_off,
                    uint32_t n_strands, uint32_t n_lds, void *V, const uint32_t *consts, const uint32_t *lconsts,
                    const uint32_t *fncode, const uint32_t *fntab, uint64_t slot_stride,
                    uint32_t Bp, uint32_t batch, uint32_t lanes, uint32_t prio_mask, uint32_t *status, const FpParams &P) {
    dim3 grid((batch + lanes - 1) / lanes), block(64 * n_strands);
    const size_t lds_bytes = (size_t)n_lds * lanes * 32;           // hand-over slots hold the lanes in use
    typedef void (*kern_t)(const CwDRow *, const uint32_t *, const uint64_t *, const uint32_t *, const uint64_t *,
                           const uint32_t *, uint4 *, const uint32_t *, const uint32_t *, const uint4 *, const uint4 *, uint64_t,
                           uint32_t, uint32_t, uint32_t, uint32_t, uint32_t *, FpParams);
    kern_t k = full ? (wide_linsum ? (kern_t)cw_eval_kernel<true, 4, 1024> : (kern_t)cw_eval_kernel<true, 2, 1024>)
                    : (wide_linsum ? (kern_t)cw_eval_kernel<false, 4, 1024> : (kern_t)cw_eval_kernel<false, 2, 1024>);
    if (full && n_strands == 1 && fncode)       // circuits with run-time functions: one wave per workgroup, no register ceiling
        k = wide_linsum ? (kern_t)cw_eval_kernel<true, 4, 64> : (kern_t)cw_eval_kernel<true, 2, 64>;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, grid, block, lds_bytes, s, rows, stream_off, extras, extra_off, terms, term_off, (uint4 *)V,
                       consts, lconsts, (const uint4 *)fncode, (const uint4 *)fntab, slot_stride, Bp, batch, lanes, prio_mask,
                       status, P);
    return hipGetLastError();
}
hipError_t cwk_eval_pipe(hipStream_t s, bool full, bool wide_linsum, uint32_t nb, uint32_t nld, const void *rows, uint32_t n_rows,
                         const uint32_t *loads, const uint64_t *terms, void *V, const uint32_t *consts, const uint32_t *lconsts,
                         uint64_t slot_stride, uint32_t Bp, uint32_t batch, uint32_t lanes, uint32_t *status, const FpParams &P) {
    typedef void (*kern_t)(const CwPRow *, uint32_t, const uint32_t *, const uint64_t *, uint4 *, const uint32_t *, const uint32_t *,
                           uint64_t, uint32_t, uint32_t, uint32_t, uint32_t *, FpParams);
    kern_t k = nullptr;
#define PIPE_PICK(F, W)                                                                   \
    (nb == 8 && nld == 8   ? (kern_t)cw_pipe_kernel<F, W, 8, 8>                            \
     : nb == 8 && nld == 4 ? (kern_t)cw_pipe_kernel<F, W, 8, 4>                            \
     : nb == 4 && nld == 4 ? (kern_t)cw_pipe_kernel<F, W, 4, 4>                            \
                           : (kern_t) nullptr)
    (void)wide_linsum;          // long small-coefficient sums belong to bit-level circuits, which have their own engine
    k = full ? PIPE_PICK(true, 2) : PIPE_PICK(false, 2);
#undef PIPE_PICK
    if (!k) return hipErrorInvalidValue;
    size_t lds_bytes = (size_t)(2 * nb + 2 * nld) * 2048;
#ifdef CW_PROFILE
    lds_bytes += 1024;
#endif
    if (lds_bytes >= 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    dim3 grid((batch + lanes - 1) / lanes), block(64);
    hipLaunchKernelGGL(k, grid, block, lds_bytes, s, (const CwPRow *)rows, n_rows, loads, terms, (uint4 *)V, consts, lconsts,
                       slot_stride, Bp, batch, lanes, status, P);
    return hipGetLastError();
}
// findings of the fused R1CS check of the emitted code (second half of the status array: smallest violated constraint per
// instance, 0xFFFFFFFF = none) -> the first_bad / status words the stand-alone kernels report through
__global__ void __launch_bounds__(CW_BLOCK) cw_fused_merge_kernel(const uint32_t *__restrict__ found, uint32_t batch, uint32_t *status, uint32_t *first_bad) {
    const uint32_t i = blockIdx.x * CW_BLOCK + threadIdx.x;
    if (i >= batch) return;
    const uint32_t f = found[i];
    if (f != 0xFFFFFFFFu) {
        atomicMin(&first_bad[i], f);
        atomicOr(&status[i], CW_ST_R1CS_FAILED);
    }
}
// 32-bit fill as a KERNEL: cw_run / cw_check_r1cs are also recorded into HIP graphs (cw_run_check), and a memset node of the captured
// graph was seen to write garbage on the second replay (the finding words of the fused check held host pointers: GPU suite of round
// 6, profiles/r06ad_*) - the captured paths launch kernels only
__global__ void __launch_bounds__(CW_BLOCK) cw_fill32_kernel(uint32_t *__restrict__ p, uint32_t v, size_t n) {
    const size_t i = (size_t)blockIdx.x * CW_BLOCK + threadIdx.x;
    if (i < n) p[i] = v;
}
hipError_t cwk_fill32(hipStream_t s, uint32_t *p, uint32_t v, size_t n) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(cw_fill32_kernel, dim3((unsigned)((n + CW_BLOCK - 1) / CW_BLOCK)), dim3(CW_BLOCK), 0, s, p, v, n);
    return hipGetLastError();
}
hipError_t cwk_fused_merge(hipStream_t s, const uint32_t *found, uint32_t batch, uint32_t *status, uint32_t *first_bad) {
    hipLaunchKernelGGL(cw_fused_merge_kernel, blocks_for(batch), dim3(CW_BLOCK), 0, s, found, batch, status, first_bad);
    return hipGetLastError();
}
hipError_t cwk_r1cs(hipStream_t s, const uint32_t *chunk, uint32_t n_chunks, const uint32_t *terms, const uint32_t *ctab,
                    const uint32_t *ctab29, const uint32_t *row_orig, const void *V, uint32_t Bp, uint32_t batch, uint32_t *status,
                    uint32_t *first_bad, bool mont, const FpParams &P) {
    if (n_chunks == 0) return hipSuccess;
    dim3 g((batch + 63) / 64, n_chunks < 65535u ? n_chunks : 65535u);
    if (mont)
        hipLaunchKernelGGL(cw_r1cs_stream_kernel<true>, g, dim3(64), 0, s, (const uint4 *)chunk, n_chunks, (const uint2 *)terms, ctab, ctab29,
                           row_orig, (const uint4 *)V, Bp, batch, status, first_bad, P);
    else
        hipLaunchKernelGGL(cw_r1cs_stream_kernel<false>, g, dim3(64), 0, s, (const uint4 *)chunk, n_chunks, (const uint2 *)terms, ctab, ctab29,
                           row_orig, (const uint4 *)V, Bp, batch, status, first_bad, P);
    return hipGetLastError();
}
hipError_t cwk_r1cs_staged(hipStream_t s, const uint32_t *chunk, uint32_t n_chunks, const uint32_t *rec, const uint32_t *terms,
                           const uint32_t *ctab, const uint32_t *ctab29, const uint32_t *row_orig, uint32_t entries, const void *V, uint32_t Bp,
                           uint32_t batch, uint32_t *status, uint32_t *first_bad, bool mont, const FpParams &P) {
    if (n_chunks == 0) return hipSuccess;
    const uint32_t lds_bytes = entries * 2048u;
    typedef void (*kern_t)(const uint4 *, const uint2 *, const uint2 *, const uint32_t *, const uint32_t *, const uint32_t *, const uint4 *,
                           uint32_t, uint32_t, uint32_t *, uint32_t *, FpParams);
    kern_t k = mont ? (kern_t)cw_r1cs_staged_kernel<true> : (kern_t)cw_r1cs_staged_kernel<false>;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    dim3 g((batch + 63) / 64, n_chunks);
    hipLaunchKernelGGL(k, g, dim3(64), lds_bytes, s, (const uint4 *)chunk, (const uint2 *)rec, (const uint2 *)terms, ctab, ctab29, row_orig,
                       (const uint4 *)V, Bp, batch, status, first_bad, P);
    return hipGetLastError();
}
hipError_t cwk_gather(hipStream_t s, const void *V, const uint32_t *w2s, uint32_t n_wit, uint32_t Bp, uint32_t instance,
                      void *out, bool mont, const FpParams &P) {
    if (mont)
        hipLaunchKernelGGL(cw_gather_kernel<true>, blocks_for(n_wit), dim3(CW_BLOCK), 0, s, (const uint4 *)V, w2s, n_wit, Bp, instance,
                           (uint4 *)out, P);
    else
        hipLaunchKernelGGL(cw_gather_kernel<false>, blocks_for(n_wit), dim3(CW_BLOCK), 0, s, (const uint4 *)V, w2s, n_wit, Bp, instance,
                           (uint4 *)out, P);
    return hipGetLastError();
}
hipError_t cwk_gather_many(hipStream_t s, const void *V, const uint32_t *w2s, uint32_t n_wit, uint32_t Bp, uint32_t first,
                           uint32_t count, void *out, bool mont, const FpParams &P) {
    if (!count || !n_wit) return hipSuccess;
    dim3 g((n_wit + GM_TILE - 1) / GM_TILE, (count + 63) / 64);
    if (mont)
        hipLaunchKernelGGL(cw_gather_many_kernel<true>, g, dim3(256), 0, s, (const uint4 *)V, w2s, n_wit, Bp, first, count, (uint4 *)out, P);
    else
        hipLaunchKernelGGL(cw_gather_many_kernel<false>, g, dim3(256), 0, s, (const uint4 *)V, w2s, n_wit, Bp, first, count, (uint4 *)out, P);
    return hipGetLastError();
}
hipError_t cwk_mulbench(hipStream_t s, const void *a, const void *b, void *out, uint32_t n, uint32_t iters,
                        const FpParams &P) {
    hipLaunchKernelGGL(cw_mulbench_kernel, blocks_for(n), dim3(CW_BLOCK), 0, s, (const uint4 *)a, (const uint4 *)b,
                       (uint4 *)out, n, iters, P);
    return hipGetLastError();
}
hipError_t cwk_fpop(hipStream_t s, uint32_t op, const void *a, const void *b, const void *c, void *out, uint32_t *status,
                    uint32_t n, const FpParams &P) {
    hipLaunchKernelGGL(cw_fpop_kernel, blocks_for(n), dim3(CW_BLOCK), 0, s, op, (const uint4 *)a, (const uint4 *)b,
                       (const uint4 *)c, (uint4 *)out, status, n, P);
    return hipGetLastError();
}
