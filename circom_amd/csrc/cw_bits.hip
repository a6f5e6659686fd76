// cw_bits.hip — bit-plane kernels (gfx950 / CDNA4) for circuits whose signals are all provably boolean.
//
// Reference counterpart: the emitted calculator runs such circuits (SHA-256, Num2Bits gadgets) through the short-int
// paths of the tagged field library (generic/fr.cpp:416-439 mul_s1s2, 696-701, 900-917 add_s1s2, 1799-1988 Fr_band)
// on 40-byte FrElements, one instance at a time.  Here a boolean signal costs ONE BIT per instance:
//
//   bit table   T[group][slot] : uint64, bit i = value of the signal in instance group*64 + i      (HBM)
//               slot 0 = constant 0, slot 1 = constant 1 (all ones), slot 2 reserved, signal s = 3 + s, then temps
//
// and the witness program is a sequence of VROWS (hip_elements/bitsched.py): one wave = one group of 64 instances,
// in a vrow every LANE evaluates one 3-input gate of the network on 64-bit masks (64 gates x 64 instances per ~60
// VALU instructions).  Results go to the wave's LDS ring (entry vrow mod R) and to up to four bit-table slots;
// operands and results are entries of the wave's LDS (result ring + cached rows of the bit table); vector memory moves
// whole rows at batch boundaries.  No barriers: a single wave, in-order LDS and in-order vector memory.
//
// The assumption "main inputs are 0/1" is checked by the ingest kernel; instances that violate it, or trip an
// assertion gate, are flagged in fbmask[group] and re-evaluated by the 256-bit schedule (cw_host.cpp), so every
// result is the reference's for every input.
#include <hip/hip_runtime.h>
#include "cw_kernels.h"
#include "fp256.hip.h"


typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- table layout ---------------------------------------------------------------------------------------------------------
// Two layouts share every kernel below through the shift `sh` (a launch parameter):
//   sh = 0   T[group][slot]                      the interpreter's table: a group's slots are consecutive uint64 (rows of 64 slots)
//   sh = 5   T[chunk][slot][group in chunk]      the emitted code's table (hip_elements/bitjit.py): a chunk = 32 groups = 2 048
//                                                instances, a slot of a chunk = one 256-byte row (lane l of the emitting wave
//                                                holds dword l: instances 32 l .. 32 l + 31 of the chunk)
// element (group g, slot s) = T[(((g >> sh) * slots + s) << sh) + (g & ((1 << sh) - 1))]: the slots of a group are
// (1 << sh) uint64 apart, starting at bits_group(T, slots, sh, g).
__device__ __forceinline__ uint64_t *bits_group(uint64_t *T, uint64_t slots, uint32_t sh, uint32_t g) {
    return T + (((size_t)(g >> sh) * slots) << sh) + (g & ((1u << sh) - 1u));
}
__device__ __forceinline__ const uint64_t *bits_group(const uint64_t *T, uint64_t slots, uint32_t sh, uint32_t g) {
    return T + (((size_t)(g >> sh) * slots) << sh) + (g & ((1u << sh) - 1u));
}

// ---- init: constant slots, flags -------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cw_bits_init_kernel(uint64_t *T, uint64_t slots, uint32_t sh, uint32_t n_groups, uint64_t *fbmask,
                                                            uint64_t *r1flag, uint32_t *status, uint32_t *first_bad, uint32_t Bp) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_groups) {
        uint64_t *Tg = bits_group(T, slots, sh, i);
        Tg[(size_t)0 << sh] = 0;
        Tg[(size_t)1 << sh] = ~0ull;
        Tg[(size_t)2 << sh] = 0;
        fbmask[i] = 0;
        if (r1flag) r1flag[i] = 0;
    }
    if (i < Bp) {
        status[i] = 0;
        first_bad[i] = 0xFFFFFFFFu;
    }
}

// ---- ingest: AoS canonical inputs [batch][n_in][32 B] -> one mask per (group, input) -----------------------------------
// (setInputSignal's `signalValues[si] = val`, calcwit.cpp:93, for 64 instances at a time).  A workgroup of four waves handles
// 256 consecutive inputs of one group: thread t owns input c * 256 + t and walks the 64 instances of the group, so every pair
// of load instructions of a wave reads 2 KiB contiguous, the workgroup 8 KiB of one instance's 32 n_in-byte record; the thread
// shifts the value's low bit into its own mask and keeps its own "not a bit" mask (values other than 0/1 flag their instance:
// one atomicOr per offending thread, none in the loop); a wave's 64 masks leave as one store instruction.
// Order matters more than the loop body (tools/ubench_ingest.hip, profiles/r05n_ubench_ingest*.txt, 137 GB of inputs; a
// read-only uint4 sum of the same buffer reaches 6.45 TB/s on the same box):
//   one wave per workgroup, groups fastest (rounds 3-5)                        24.2 ms   5.69 TB/s
//   the same with 4 / 8 / 16 loads in flight per wave (unrolled, lane flags)    24.3-24.5 (round 3: streaming loads 26.4; round 5:
//                                                                               whole-line loads 24.3-24.7)
//   chunks of one group fastest                                                 28.3 ms: the waves that run together march through
//                                                                               the 64 KB records of their groups in step
//   chunks fastest + every group starts at its own instance (hash(g) & 63)      23.4 ms
//   that, four waves per workgroup on consecutive chunks, unrolled x 4          22.2 ms   6.19 TB/s   <- this kernel
//   (two waves 22.6, eight 23.2, sixteen 23.3; unrolled x 2 22.9, x 8 22.5; no rotation 24.0-24.4)
#define BITS_INGEST_WAVES 4
__global__ void __launch_bounds__(64 * BITS_INGEST_WAVES)
cw_bits_ingest_kernel(const uint4 *__restrict__ in, uint64_t *__restrict__ T, uint64_t slots, uint32_t sh, uint32_t input_slot0, uint32_t n_in,
                      uint32_t batch, uint64_t *fbmask, uint32_t n_chunks) {
    const uint32_t c = blockIdx.x % n_chunks, g = blockIdx.x / n_chunks;
    const uint32_t k = c * (64 * BITS_INGEST_WAVES) + threadIdx.x;
    if (k >= n_in) return;                                           // no cross-lane operation below
    const uint32_t i0 = g * 64, ni = min(64u, batch - i0);
    uint64_t mine = 0, bad = 0;
    const uint4 *base = in + ((size_t)i0 * n_in + k) * 2;
    const size_t step = (size_t)n_in * 2;
    if (ni == 64) {
        const uint32_t r = (g * 0x9E3779B1u) >> 26;                  // the group's first instance
#pragma unroll 4
        for (uint32_t ii = 0; ii < 64; ii++) {
            const uint32_t inst = (ii + r) & 63u;
            const uint4 *p = base + inst * step;
            const uint4 lo = p[0], hi = p[1];
            const uint32_t rest = lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w;
            mine |= (uint64_t)(lo.x & 1u) << inst;
            bad |= (uint64_t)((lo.x > 1u) | (rest != 0u)) << inst;
        }
    } else {
        const uint4 *p = base;
        for (uint32_t ii = 0; ii < ni; ii++) {
            const uint4 lo = p[0], hi = p[1];
            p += step;
            const uint32_t rest = lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w;
            mine |= (uint64_t)(lo.x & 1u) << ii;
            bad |= (uint64_t)((lo.x > 1u) | (rest != 0u)) << ii;
        }
    }
    bits_group(T, slots, sh, g)[(size_t)(input_slot0 + k) << sh] = mine;
    if (bad) atomicOr((unsigned long long *)&fbmask[g], (unsigned long long)bad);
}

// ---- the gate program ------------------------------------------------------------------------------------------------------
// Program model: hip_elements/bitsched.py (round 3).  The wave's LDS holds a RING of R result rows, a CACHE of C rows of
// the group's bit table and the constants 0 / ~0; every operand and every result of a lane is ONE LDS
// entry named in its 8-byte record:
//   w0 = a_off | K1 | K2 << 1 | b_off << 16        LDS byte offsets (multiples of 8); K1: stage 1 is AND (else XOR),
//   w1 = c_off | dst_off << 16                                                         K2: stage 2 is OR (else XOR)
//   result = K2 ? (u | c) : (u ^ c),  u = K1 ? (a & b) : (a ^ b)          two v_bitop3_b32 per 32 instances
// (bitmap.py maps every 3-input gate of the network onto this primitive; round 2 evaluated a lane-specific 8-bit truth
// table with 8 v_bfe + 14 v_bfi per vrow).  Vector memory moves whole 512-byte ROWS only, at batch boundaries, named in a
// per-batch command block that the wave reads through the scalar cache: row LOADS (bit table -> cache slot: main
// inputs, values that left the cache) and row FLUSHES (cache slot -> bit table: every row exactly once, when complete).
// Round 2's per-lane scattered stores and loads cost 23..48 clocks of the CU's memory pipeline per instruction with
// four waves per CU (tools/ubench_isa) - 2.5 of them per vrow: that was the bound of the kernel.
// W = instances per wave: 64 (one wave per group), or 32 / 16 (2 / 4 independent waves per group, each on its slice of
// every mask: small batches then spread over more CUs).
#define BITS_NB 8
#define BITS_MAX_LOADS 4
#define BITS_MAX_FLUSH 6
#define BITS_CMD_WORDS 24
#define BITS_AHEAD 8          // records are requested this many batches ahead (BITS_AHEAD + 1 register sets of 16 dwords).  vmcnt
                              // retires loads AND stores in order, so a record wait also waits for every row flush issued before the
                              // newest 4 * BITS_AHEAD operations: behind cold caches / TLBs (tools/bits_shape_bench.py with
                              // CW_BENCH_COLD=1) four batches of slack cost 1.31 ms, eight 1.18 (0.90 warm either way)
#ifndef BITS_STORE_AUX
#define BITS_STORE_AUX 2      // cache policy of the row flushes: nt (streaming).  A flushed row is dead for this kernel; with the default
                              // policy the 1.2 GB of rows a launch writes thrashed the 4 MB L2s and the in-order vmcnt made the record
                              // loads wait behind slow stores: 1.24 -> 0.91 ms for Sha256(2048) x 65 536 (tools/bits_shape_bench.py)
#endif
#define BITS_OOR 0xFFFFFFF0u

// v_bitop3_b32 table of a function of (s0, s1, s2): bit (s0 << 2 | s1 << 1 | s2)
constexpr uint32_t bitop3_stage1() {           // (a, b, K) -> K ? a & b : a ^ b
    uint32_t t = 0;
    for (int i = 0; i < 8; i++) {
        const int a = (i >> 2) & 1, b = (i >> 1) & 1, k = i & 1;
        t |= (uint32_t)(k ? (a & b) : (a ^ b)) << i;
    }
    return t;
}
constexpr uint32_t bitop3_stage2() {           // (u, c, K) -> K ? u | c : u ^ c
    uint32_t t = 0;
    for (int i = 0; i < 8; i++) {
        const int u = (i >> 2) & 1, c = (i >> 1) & 1, k = i & 1;
        t |= (uint32_t)(k ? (u | c) : (u ^ c)) << i;
    }
    return t;
}
static_assert(bitop3_stage1() == 0x94u && bitop3_stage2() == 0xBCu, "v_bitop3_b32 tables of the two primitive stages");
__device__ __forceinline__ uint32_t prim32(uint32_t a, uint32_t b, uint32_t c, uint32_t k1, uint32_t k2) {
    const uint32_t u = __builtin_amdgcn_bitop3_b32(a, b, k1, 0x94);
    return __builtin_amdgcn_bitop3_b32(u, c, k2, 0xBC);
}
// 3-input lookup on 64 instances with a wave-uniform or per-lane table (R1CS LUT class): index bit 0 = a, 1 = b, 2 = c
__device__ __forceinline__ uint32_t bfi32(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }   // v_bfi_b32
__device__ __forceinline__ uint32_t lut3_32(uint32_t a, uint32_t b, uint32_t c, const uint32_t t[8]) {
    const uint32_t x0 = bfi32(a, t[1], t[0]), x1 = bfi32(a, t[3], t[2]), x2 = bfi32(a, t[5], t[4]), x3 = bfi32(a, t[7], t[6]);
    const uint32_t y0 = bfi32(b, x1, x0), y1 = bfi32(b, x3, x2);
    return bfi32(c, y1, y0);
}
__device__ __forceinline__ uint64_t lut3(uint64_t a, uint64_t b, uint64_t c, uint32_t tt) {
    uint32_t t[8];
#pragma unroll
    for (int m = 0; m < 8; m++) t[m] = (uint32_t)((int32_t)(tt << (31 - m)) >> 31);      // v_bfe_i32: 0 or ~0
    const uint32_t lo = lut3_32((uint32_t)a, (uint32_t)b, (uint32_t)c, t);
    const uint32_t hi = lut3_32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), t);
    return ((uint64_t)hi << 32) | lo;
}

template <int W> struct BitsMask;
template <> struct BitsMask<64> {
    typedef uint64_t T;
    static __device__ __forceinline__ T lds(uint32_t off) { return *(const __attribute__((address_space(3))) uint64_t *)(uintptr_t)off; }
    static __device__ __forceinline__ void lds_st(uint32_t off, T v) { *(__attribute__((address_space(3))) uint64_t *)(uintptr_t)off = v; }
    static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
        return ((uint64_t)v.y << 32) | v.x;
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, uint32_t off, T v) {
        const u32x2 x = {(uint32_t)v, (uint32_t)(v >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(x, r, (int)off, 0, BITS_STORE_AUX);
    }
    static __device__ __forceinline__ T prim(T a, T b, T c, uint32_t k1, uint32_t k2) {
        // one block: the compiler then waits ONCE for the three LDS operands instead of once per stage (a wave alone on
        // its SIMD issues one instruction of any kind per ~4 clocks: s_waitcnt instructions count)
        uint32_t lo, hi;
        asm("v_bitop3_b32 %0, %2, %4, %8 bitop3:0x94\n\t"
            "v_bitop3_b32 %1, %3, %5, %8 bitop3:0x94\n\t"
            "v_bitop3_b32 %0, %0, %6, %9 bitop3:0xbc\n\t"
            "v_bitop3_b32 %1, %1, %7, %9 bitop3:0xbc"
            : "=&v"(lo), "=&v"(hi)
            : "v"((uint32_t)a), "v"((uint32_t)(a >> 32)), "v"((uint32_t)b), "v"((uint32_t)(b >> 32)), "v"((uint32_t)c),
              "v"((uint32_t)(c >> 32)), "v"(k1), "v"(k2));
        return ((uint64_t)hi << 32) | lo;
    }
};
template <> struct BitsMask<32> {
    typedef uint32_t T;
    static __device__ __forceinline__ T lds(uint32_t off) { return *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)off; }
    static __device__ __forceinline__ void lds_st(uint32_t off, T v) { *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)off = v; }
    static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        return __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0);
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, uint32_t off, T v) {
        __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)off, 0, 0);
    }
    static __device__ __forceinline__ T prim(T a, T b, T c, uint32_t k1, uint32_t k2) { return prim32(a, b, c, k1, k2); }
};
template <> struct BitsMask<16> {
    typedef uint32_t T;
    static __device__ __forceinline__ T lds(uint32_t off) { return *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)off; }
    static __device__ __forceinline__ void lds_st(uint32_t off, T v) { *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)off = v; }
    static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r, (int)off, 0, 0);
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, uint32_t off, T v) {
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, r, (int)off, 0, 0);
    }
    static __device__ __forceinline__ T prim(T a, T b, T c, uint32_t k1, uint32_t k2) { return prim32(a, b, c, k1, k2); }
};

extern __shared__ uint64_t cw_bits_lds[];        // ring rows | cache rows | constants 0, ~0

// One batch = BITS_NB vrows.  When batch b starts the wave requests the records of batch b + BITS_AHEAD (device stream: 4 x 16
// bytes per lane and batch) and the row loads of its command block; the BITS_NB steps then run on registers and LDS
// only (the operands of step k + 1 are read while step k computes: a consumer sits two vrows behind its producer); the
// rows requested by batch b - 1 are written to their cache slots before the last step reads the operands of the next
// batch's first vrow; the flushes of the command block copy cache slots to the bit table when the batch ends.
template <int W>
struct BitsEval {
    typedef BitsMask<W> M;
    typedef typename M::T mask_t;
    __amdgpu_buffer_rsrc_t rsrc, rrecs, rcmds;   // the group's bit table; the record stream; the command blocks
    uint32_t lane8;
    mask_t a, b, c;
    // command blocks travel in ONE VGPR each (lane j = word j, a 96-byte vector load a batch ahead) and are read with
    // v_readlane where a wave-uniform value is needed.  Scalar loads would be the natural fit, but they share the
    // LGKM counter with the LDS and return out of order: with one in flight hipcc turns every LDS wait into
    // lgkmcnt(0) - the pipelined operand reads of the next steps then serialise behind a ~200-clock scalar load.
    uint32_t cprev;

    static __device__ __forceinline__ uint32_t word(const uint4 (&r)[4], int k, int i) {
        const int d = 2 * k + i;
        const uint4 &q = r[d >> 2];
        return (d & 3) == 0 ? q.x : (d & 3) == 1 ? q.y : (d & 3) == 2 ? q.z : q.w;
    }
#ifdef CW_EXP_NOCMD       /* timing experiment only (tools/bits_exp.sh): every command block reads as empty */
    static __device__ __forceinline__ uint32_t cw(uint32_t, int) { return 0; }
#else
    static __device__ __forceinline__ uint32_t cw(uint32_t blk, int j) { return (uint32_t)__builtin_amdgcn_readlane((int)blk, j); }
#endif

    // records of batch b, load j: 1 KiB at (b * 4 + j) * 1024, lane l takes 16 bytes at l * 16 (32-bit lane offset +
    // scalar batch offset)
    __device__ __forceinline__ uint4 rec_load(uint32_t b, int j) const {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rrecs, (int)(lane8 * 2), (int)((b * 4 + j) * 1024u), 0);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    __device__ __forceinline__ uint32_t cmd_load(uint32_t b) const {
        return __builtin_amdgcn_raw_buffer_load_b32(rcmds, (int)(lane8 >> 1), (int)(b * (BITS_CMD_WORDS * 4u)), 0);
    }

    __device__ __forceinline__ void batch(uint32_t bi, const uint4 (&cur)[4], const uint4 (&nxt)[4], uint4 (&fill)[4],
                                          const uint32_t ccur, uint32_t &cfill,
                                          const mask_t (&lprev)[BITS_MAX_LOADS], mask_t (&lfill)[BITS_MAX_LOADS]) {
        // the rows the PREVIOUS batch completed leave now: the first two (a batch completes 1.4 rows on average) are read
        // from the LDS before anything of this batch is written and stored a step later, without a wait on the spot
        const uint32_t pcounts = cw(cprev, 0), pn = pcounts & 0xFFu, fn = (pcounts >> 8) & 0xFFu;
        const mask_t fv0 = M::lds(cw(cprev, 3 + 2 * BITS_MAX_LOADS) + lane8);          // unused entries are 0: ring row 0
        const mask_t fv1 = M::lds(cw(cprev, 5 + 2 * BITS_MAX_LOADS) + lane8);
        const uint32_t nl = cw(ccur, 0) & 0xFFu;
        // row loads first: when they are awaited (a batch later) the unconditional record loads issued behind them stand
        // between, so the compiler's vmcnt never waits for a request younger than a batch
        if (nl) {                                   // rare (a few row loads per hundred batches): one branch in the common case
#pragma unroll
            for (int j = 0; j < BITS_MAX_LOADS; j++)
                if (j < (int)nl) lfill[j] = M::load(rsrc, cw(ccur, 2 + 2 * j) + lane8);
        }
#ifdef CW_EXP_NORECS      /* timing experiment only (tools/bits_exp.sh): no record fetch, every batch replays the first */
#pragma unroll
        for (int j = 0; j < 4; j++) fill[j] = cur[j];
#else
#pragma unroll
        for (int j = 0; j < 4; j++) fill[j] = rec_load(bi + BITS_AHEAD, j);
#endif
        // the command block travels with the records of its batch (same age in the in-order request queue: the wait for it
        // leaves the newer BITS_AHEAD - 1 batches of requests in flight.  Requested one batch ahead, as the first version of
        // this kernel did, its wait was a vmcnt(0) at the top of every batch: the queue drained 280 times a launch.)
        cfill = cmd_load(bi + BITS_AHEAD);
#pragma unroll
        for (int k = 0; k < BITS_NB; k++) {
            if (k == 1) {
                asm volatile("" ::"v"(fv0), "v"(fv1));          // the two reads are complete HERE on every path
                if (fn) {
                    M::store(rsrc, cw(cprev, 2 + 2 * BITS_MAX_LOADS) + lane8, fv0);
                    if (fn > 1) M::store(rsrc, cw(cprev, 4 + 2 * BITS_MAX_LOADS) + lane8, fv1);
                    if (fn > 2) {
#pragma unroll
                        for (int j = 2; j < BITS_MAX_FLUSH; j++)
                            if (j < (int)fn) M::store(rsrc, cw(cprev, 2 + 2 * BITS_MAX_LOADS + 2 * j) + lane8, M::lds(cw(cprev, 3 + 2 * BITS_MAX_LOADS + 2 * j) + lane8));
                    }
                }
            }
            if (k == BITS_NB - 1 && pn) {
#pragma unroll
                for (int j = 0; j < BITS_MAX_LOADS; j++)
                    if (j < (int)pn) M::lds_st(cw(cprev, 3 + 2 * j) + lane8, lprev[j]);
            }
            const uint32_t n0 = k + 1 < BITS_NB ? word(cur, k + 1 < BITS_NB ? k + 1 : 0, 0) : word(nxt, 0, 0);
            const uint32_t n1 = k + 1 < BITS_NB ? word(cur, k + 1 < BITS_NB ? k + 1 : 0, 1) : word(nxt, 0, 1);
            const mask_t na = M::lds(n0 & 0xFFF8u), nb = M::lds(n0 >> 16), nc = M::lds(n1 & 0xFFFFu);
            // the three reads of step k + 1 are ISSUED before step k computes (left to itself the scheduler put them behind
            // the four v_bitop3: 8 instructions between a read and its wait, ~35 clocks of a lone wave against an LDS latency
            // of ~100; in front there are 17)
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t w0 = word(cur, k, 0), w1 = word(cur, k, 1);
            const uint32_t k1 = (uint32_t)((int32_t)(w0 << 31) >> 31), k2 = (uint32_t)((int32_t)(w0 << 30) >> 31);
            M::lds_st(w1 >> 16, M::prim(a, b, c, k1, k2));
            a = na; b = nb; c = nc;
        }
        cprev = ccur;
    }

    // after the last batch: what it completed, and the row loads it may still hold are dropped (nothing reads them)
    __device__ __forceinline__ void drain() {
        const uint32_t fn = (cw(cprev, 0) >> 8) & 0xFFu;
#pragma unroll
        for (int j = 0; j < BITS_MAX_FLUSH; j++)
            if (j < (int)fn) M::store(rsrc, cw(cprev, 2 + 2 * BITS_MAX_LOADS + 2 * j) + lane8, M::lds(cw(cprev, 3 + 2 * BITS_MAX_LOADS + 2 * j) + lane8));
    }
};

template <int W>
__global__ void __launch_bounds__(64)
cw_bits_eval_kernel(const uint4 *__restrict__ recs, const uint32_t *__restrict__ cmds, uint32_t n_batches, uint32_t const_off,
                    uint64_t *T, uint64_t slots) {
    typedef BitsMask<W> M;
    typedef typename M::T mask_t;
    const uint32_t lane = threadIdx.x, g = blockIdx.x, slice = blockIdx.y;
    char *Tg = (char *)(T + (size_t)g * slots) + slice * (W / 8);
    if (n_batches == 0) return;
    BitsEval<W> E;
    // buffer descriptors (wave-uniform by construction: kernel arguments and blockIdx only); a wave of a narrower slice
    // addresses its bytes of every 8-byte mask through the shifted base
    E.rsrc = __builtin_amdgcn_make_buffer_rsrc(Tg, 0, (int)(uint32_t)(slots * 8 - slice * (W / 8)), 0x00020000);
    E.rrecs = __builtin_amdgcn_make_buffer_rsrc((void *)recs, 0, (int)((n_batches + BITS_AHEAD) * 4096u), 0x00020000);
    E.rcmds = __builtin_amdgcn_make_buffer_rsrc((void *)cmds, 0, (int)((n_batches + BITS_AHEAD) * (BITS_CMD_WORDS * 4u)), 0x00020000);
    E.lane8 = lane * 8;
    // the dynamic LDS of this kernel starts at LDS address 0 (it declares no static LDS): records carry raw LDS offsets
    if (lane == 0) {
        M::lds_st(const_off, (mask_t)0);
        M::lds_st(const_off + 8, (mask_t)~(mask_t)0);
    }
    // every lane's ring entries start as 0 (idle lanes of the first vrows read constants only; a defined value keeps
    // the kernel deterministic when a damaged program names an entry that was never written)
    for (uint32_t o = E.lane8; o < const_off; o += 512) M::lds_st(o, (mask_t)0);
    // the host pads the stream to whole trips of 18 batches and appends BITS_AHEAD empty ones (records are requested
    // BITS_AHEAD batches ahead); nine record sets and two loaded-row sets rotate by NAME through the 18 expansions of
    // the batch body (no register moves)
    uint4 R0[4], R1[4], R2[4], R3[4], R4[4], R5[4], R6[4], R7[4], R8[4];
    mask_t L0[BITS_MAX_LOADS], L1[BITS_MAX_LOADS];
#pragma unroll
    for (int j = 0; j < BITS_MAX_LOADS; j++) L0[j] = L1[j] = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        R0[j] = E.rec_load(0, j);
        R1[j] = E.rec_load(1, j);
        R2[j] = E.rec_load(2, j);
        R3[j] = E.rec_load(3, j);
        R4[j] = E.rec_load(4, j);
        R5[j] = E.rec_load(5, j);
        R6[j] = E.rec_load(6, j);
        R7[j] = E.rec_load(7, j);
    }
    E.cprev = 0;
    uint32_t C0 = E.cmd_load(0), C1 = E.cmd_load(1), C2 = E.cmd_load(2), C3 = E.cmd_load(3), C4 = E.cmd_load(4),
             C5 = E.cmd_load(5), C6 = E.cmd_load(6), C7 = E.cmd_load(7), C8 = 0;
    E.a = M::lds(R0[0].x & 0xFFF8u);
    E.b = M::lds(R0[0].x >> 16);
    E.c = M::lds(R0[0].y & 0xFFFFu);
    static_assert(BITS_AHEAD == 8, "the rotation below is written for 9 record sets");
    for (uint32_t bi = 0; bi < n_batches; bi += 18) {
        E.batch(bi + 0, R0, R1, R8, C0, C8, L0, L1);
        E.batch(bi + 1, R1, R2, R0, C1, C0, L1, L0);
        E.batch(bi + 2, R2, R3, R1, C2, C1, L0, L1);
        E.batch(bi + 3, R3, R4, R2, C3, C2, L1, L0);
        E.batch(bi + 4, R4, R5, R3, C4, C3, L0, L1);
        E.batch(bi + 5, R5, R6, R4, C5, C4, L1, L0);
        E.batch(bi + 6, R6, R7, R5, C6, C5, L0, L1);
        E.batch(bi + 7, R7, R8, R6, C7, C6, L1, L0);
        E.batch(bi + 8, R8, R0, R7, C8, C7, L0, L1);
        E.batch(bi + 9, R0, R1, R8, C0, C8, L1, L0);
        E.batch(bi + 10, R1, R2, R0, C1, C0, L0, L1);
        E.batch(bi + 11, R2, R3, R1, C2, C1, L1, L0);
        E.batch(bi + 12, R3, R4, R2, C3, C2, L0, L1);
        E.batch(bi + 13, R4, R5, R3, C4, C3, L1, L0);
        E.batch(bi + 14, R5, R6, R4, C5, C4, L0, L1);
        E.batch(bi + 15, R6, R7, R5, C6, C5, L1, L0);
        E.batch(bi + 16, R7, R8, R6, C7, C6, L0, L1);
        E.batch(bi + 17, R8, R0, R7, C8, C7, L1, L0);
    }
    E.drain();
}

// instances that tripped an assertion gate: the program gives every assertion value a bit-table slot; a mask that is not
// zero names the instances to be re-run by the 256-bit schedule
__global__ void __launch_bounds__(256)
cw_bits_assert_kernel(const uint64_t *__restrict__ T, uint64_t slots, const uint32_t *__restrict__ aslots, uint32_t n_asserts,
                      uint32_t n_groups, uint64_t *fbmask) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n_groups) return;
    uint64_t v = 0;
    for (uint32_t i = 0; i < n_asserts; i++) v |= T[(size_t)g * slots + aslots[i]];
    if (v) fbmask[g] |= v;
}

// ---- egress: canonical 32-byte values from the bit table (getWitness + Fr_toLongNormal, main.cpp:326-332) ----------------
// out[j][k] (32 bytes) = value of witness element k in instance first + j.  The kernel is a pure HBM writer: a wave owns
// 32 * BITS_GATHER_RUN consecutive elements and walks 64 instances; lane l holds the mask of element k0 + l / 2 (ONE 8-byte read
// per 2 KiB written, through wslot = sig_slot o w2s composed on the host) and every store instruction writes the 1 KiB that 32
// elements of one instance occupy (global_store_dwordx4, consecutive lanes -> consecutive 16-byte halves).  Round 2's
// kernel ran one thread per element with two dependent index loads per 32 bytes: 0.53 of the HBM peak.
// Round 5 (tools/ubench_egress.hip on the --O1 witness of the default line, 156 809 wires, profiles/r05n_ubench_egress*.txt; a
// one-pass 16-byte fill of the same buffers writes 6.9 TB/s, hipMemsetAsync 6.7): streaming (nt) stores 4.29 TB/s -> plain
// stores 4.7 - 4.8; 8 KiB instead of 4 KiB runs per wave 4.8 - 4.9; with 8 or more groups of 64 instances per launch the grid
// walks the GROUPS fastest (the workgroups that run together then write the same elements of different instances): 5.1 - 5.8
// for launches of 1 024 instances, 5.4 - 5.5 for 2 048 (elements fastest: 4.5 - 4.9); a sweep in pure address order (one
// instance per workgroup, the mask words re-read from L2) 3.0, rotated starts 4.2 - 5.1, other workgroup sizes 4.6 - 4.8.
#define BITS_GATHER_RUN 8      // consecutive 1 KiB pieces (32 elements each) a wave writes per instance: 8 KiB runs, 32 KiB per block
__global__ void __launch_bounds__(256)
cw_bits_gather_kernel(const uint64_t *__restrict__ T, uint64_t slots, uint32_t lsh, const uint32_t *__restrict__ wslot, uint32_t n_wit,
                      uint32_t first, uint32_t count, uint4 *__restrict__ out, uint32_t groups_fastest) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t bk = groups_fastest ? blockIdx.y : blockIdx.x, bj = groups_fastest ? blockIdx.x : blockIdx.y;
    const uint32_t k0 = (bk * 4 + wv) * 32 * BITS_GATHER_RUN;               // this wave's elements
    const uint32_t j0 = bj * 64;                                             // its 64 output instances
    if (k0 >= n_wit) return;
    const uint32_t i0 = first + j0, g = i0 >> 6, sh = i0 & 63u;
    const bool two = sh && (uint64_t)(i0 + 64 - sh) < (uint64_t)first + count;
    uint64_t win[BITS_GATHER_RUN];                                           // bit jj = the value in instance i0 + jj
    bool have[BITS_GATHER_RUN];
#pragma unroll
    for (int r = 0; r < BITS_GATHER_RUN; r++) {
        const uint32_t k = k0 + r * 32 + (lane >> 1);
        have[r] = k < n_wit;
        win[r] = 0;
        if (have[r]) {
            const uint32_t sl = wslot[k];
            win[r] = bits_group(T, slots, lsh, g)[(size_t)sl << lsh] >> sh;
            if (two) win[r] |= bits_group(T, slots, lsh, g + 1)[(size_t)sl << lsh] << (64 - sh);
        }
    }
    const uint32_t nj = min(64u, count - j0);
    const bool low_half = !(lane & 1);
    uint4 *o = out + ((size_t)j0 * n_wit + k0) * 2 + lane;
    for (uint32_t jj = 0; jj < nj; jj++) {
#pragma unroll
        for (int r = 0; r < BITS_GATHER_RUN; r++) {
            const uint32_t bit = low_half ? (uint32_t)(win[r] >> jj) & 1u : 0u;
            if (have[r]) o[r * 64] = make_uint4(bit, 0u, 0u, 0u);
        }
        o += (size_t)n_wit * 2;
    }
}

// packed main inputs (cw_set_inputs_bits*): masks[group][k] -> bit-table slot input_slot0 + k
__global__ void __launch_bounds__(256)
cw_bits_ingest_packed_kernel(const uint64_t *__restrict__ masks, uint64_t *__restrict__ T, uint64_t slots, uint32_t sh, uint32_t input_slot0,
                             uint32_t n_in, uint32_t batch) {
    const uint32_t k = blockIdx.y * 256 + threadIdx.x, g = blockIdx.x;   // (groups in x: 65 536 of them at 2^22 instances)
    if (k >= n_in) return;
    uint64_t m = masks[(size_t)g * n_in + k];
    const uint32_t live = batch - g * 64;                            // instances of the last group beyond the batch read as 0
    if (live < 64) m &= (1ull << live) - 1;
    bits_group(T, slots, sh, g)[(size_t)(input_slot0 + k) << sh] = m;
}
// canonical inputs of the listed instances from packed masks / from the AoS input image (rows of the side batch that
// re-runs them on the 256-bit schedule): out[j][k] = input k of instance inst[j]
__global__ void __launch_bounds__(256)
cw_bits_collect_inputs_kernel(const uint64_t *__restrict__ masks, const uint4 *__restrict__ aos, const uint32_t *__restrict__ inst,
                              uint32_t n_inst, uint32_t n_in, uint4 *__restrict__ out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (k >= n_in || j >= n_inst) return;
    const uint32_t i = inst[j];
    uint4 lo, hi = make_uint4(0, 0, 0, 0);
    if (masks) {
        lo = make_uint4((uint32_t)(masks[(size_t)(i >> 6) * n_in + k] >> (i & 63u)) & 1u, 0, 0, 0);
    } else {
        lo = aos[((size_t)i * n_in + k) * 2];
        hi = aos[((size_t)i * n_in + k) * 2 + 1];
    }
    out[((size_t)j * n_in + k) * 2] = lo;
    out[((size_t)j * n_in + k) * 2 + 1] = hi;
}

// ---- R1CS check on the bit table ---------------------------------------------------------------------------------------------
// Class E: constraints over <= 5 distinct wires.  The host enumerated A*B - C over the 2^k assignments of the wires
// (cw_host.cpp, exact integer arithmetic): the constraint is a 32-entry truth table "violated?".  A vrow checks 64
// constraints for 64 instances: 5 mask loads per lane, four 3-input lookups + three selects.
struct ERec { uint32_t w[5]; uint32_t tt, row, pad; };
__global__ void __launch_bounds__(64)
cw_bits_r1cs_lut_kernel(const uint4 *__restrict__ recs, uint32_t n_vrows, uint32_t vrows_per_chunk, const uint64_t *__restrict__ T,
                        uint64_t slots, uint32_t sh, const uint64_t *__restrict__ only, uint32_t n_groups, uint32_t batch, uint32_t *status,
                        uint32_t *first_bad) {
    const uint32_t lane = threadIdx.x;
    // audit of flagged groups only (the emitted code checked every constraint itself): a few workgroups walk the flag words
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    if (only && only[g] == 0) continue;
    const char *Tg = (const char *)bits_group(T, slots, sh, g);
    const uint32_t v0 = blockIdx.y * vrows_per_chunk, v1 = min(n_vrows, v0 + vrows_per_chunk);
    for (uint32_t v = v0; v < v1; v++) {
        const uint4 x = recs[((size_t)v * 64 + lane) * 2], y = recs[((size_t)v * 64 + lane) * 2 + 1];
        const uint64_t w0 = *(const uint64_t *)(Tg + ((size_t)x.x << sh)), w1 = *(const uint64_t *)(Tg + ((size_t)x.y << sh)),
                       w2 = *(const uint64_t *)(Tg + ((size_t)x.z << sh)), w3 = *(const uint64_t *)(Tg + ((size_t)x.w << sh)),
                       w4 = *(const uint64_t *)(Tg + ((size_t)y.x << sh));
        const uint32_t tt = y.y;
        const uint64_t f00 = lut3(w0, w1, w2, tt & 0xFFu), f01 = lut3(w0, w1, w2, (tt >> 8) & 0xFFu),
                       f10 = lut3(w0, w1, w2, (tt >> 16) & 0xFFu), f11 = lut3(w0, w1, w2, tt >> 24);
        const uint64_t lo = (f01 & w3) | (f00 & ~w3), hi = (f11 & w3) | (f10 & ~w3);
        uint64_t viol = (hi & w4) | (lo & ~w4);
        if (__any(viol != 0)) {                                    // rare: report the first bad row of each instance
            const uint32_t row = y.z;
            while (viol) {
                const uint32_t i = g * 64 + (uint32_t)__builtin_ctzll(viol);
                viol &= viol - 1;
                if (i < batch) {
                    atomicMin(&first_bad[i], row);
                    atomicOr(&status[i], CW_ST_R1CS_FAILED);
                }
            }
        }
    }
    }
}

// Class W: any other constraint (long linear rows of BinSum / Bits2Num shape, field-sized coefficients).  One lane =
// one instance; terms are wave-uniform (scalar loads), a wire's mask is ONE 8-byte scalar-cache read for the whole
// wave and each lane picks its bit.  term = {slot byte offset | part (2 bits) << 30, coefficient id}; coefficient
// table = canonical residues; row ends as in cw_r1cs_stream_kernel: (A*B + (q - C)) * R'^-1 == 0.
__global__ void __launch_bounds__(64)
cw_bits_r1cs_wide_kernel(const uint4 *__restrict__ chunk, uint32_t n_chunks, const uint2 *__restrict__ terms,
                         const uint32_t *__restrict__ ctab, const uint32_t *__restrict__ row_orig,
                         const uint64_t *__restrict__ T, uint64_t slots, uint32_t sh, const uint64_t *__restrict__ only,
                         uint32_t n_groups, uint32_t batch, uint32_t *status, uint32_t *first_bad, FpParams P) {
    const uint32_t lane = threadIdx.x;
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const uint32_t i = g * 64 + lane;
    if (only && only[g] == 0) continue;
    const char *Tg = (const char *)bits_group(T, slots, sh, g);
    uint32_t bad = 0xFFFFFFFFu;
    for (uint32_t cix = blockIdx.y; cix < n_chunks; cix += gridDim.y) {
        const uint4 ch = chunk[cix];                                // first term, n terms, -, first row
        const uint2 *tp = terms + ch.x;
        fe A = fe_zero(), B = fe_zero(), cur = fe_zero();
        uint32_t row = ch.w;
        for (uint32_t k = 0; k < ch.y; k++) {
            const uint2 t = tp[k];
            const uint32_t off = t.x & 0x0FFFFFFFu, part = (t.x >> 28) & 3u, last = t.x >> 31, endrow = (t.x >> 30) & 1u;
            const uint64_t m = *(const uint64_t *)(Tg + ((size_t)off << sh));      // wave-uniform address
            const bool bit = (m >> lane) & 1ull;
            const fe cf = fe_from(ctab + (size_t)t.y * 8);
            fe w;
#pragma unroll
            for (int j = 0; j < 8; j++) w.v[j] = bit ? cf.v[j] : 0u;
            cur = fe_add(cur, w, P);
            if (last) {                                             // last term of its part
                if (part == 0) { A = cur; cur = fe_zero(); }
                else if (part == 1) { B = cur; cur = fe_zero(); }
            }
            if (endrow) {
                const fe29 z = fe29_mmul_add(fe_to29(A), fe_to29(B), fe_to29(fe_neg(cur, P)), P);
                uint32_t o = 0;
#pragma unroll
                for (int j = 0; j < 9; j++) o |= z.l[j];
                if (o != 0) {
                    const uint32_t oc = row_orig[row];
                    if (oc < bad) bad = oc;
                }
                row++;
                A = fe_zero(); B = fe_zero(); cur = fe_zero();
            }
        }
    }
    if (bad != 0xFFFFFFFFu && i < batch) {
        atomicMin(&first_bad[i], bad);
        atomicOr(&status[i], CW_ST_R1CS_FAILED);
    }
    }
}

// Class I: every coefficient is a small signed integer (|c| < 2^40, < 2^20 terms per row): the three parts are exact
// 64-bit integer sums and the row holds iff A * B - C == 0 over the integers (|A*B - C| < 2^125 < q, so this IS the test
// modulo q).  SHA-256's `lin === lout` rows (up to 195 terms) live here.  One lane = one instance; the stream is
// wave-uniform (scalar loads).  The host regrouped the terms (cw_bits_host.h): inside a GROUP the coefficients are
// distinct powers of two of one sign within one 32-bit half, so a term costs two VALU instructions — select 0/1 with the
// wire's 64-instance mask as the condition, shift-or it into the group's word — and a group one 64-bit add.
__device__ __forceinline__ uint32_t bits_lane_bit(uint64_t mask) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(mask));      // lane i gets bit i of the (wave-uniform) mask
    return r;
}
// 32 x 32 bit-matrix transpose across the 32 lanes of a half-wave: lane k enters with row k (bit j = element (k, j)) and
// leaves with column k.  Five butterfly stages; stage d exchanges the off-diagonal d x d blocks between lanes k and k ^ d:
// the partner's word arrives through ds_swizzle (xor mask inside groups of 32 lanes: no index register), is ROTATED so
// that the wanted blocks land on the positions this lane gives up (left by d in the lower lane of a pair, right by d in the
// upper one; what the rotation drags into the kept positions is masked out) and merged with one v_bfi: 3 instructions
// per stage (round 2: ~12 - index arithmetic for ds_bpermute, two masked shifts, compare + select).
struct BitsTr {
    uint32_t rot[5], keep[5];                 // per stage: v_alignbit shift (32 - rotate-left amount), mask of the bits this lane keeps
};
__device__ __forceinline__ BitsTr bits_tr_setup(uint32_t lane) {
    BitsTr t;
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const uint32_t d = 16u >> s;
        const uint32_t m = s == 0 ? 0x0000FFFFu : s == 1 ? 0x00FF00FFu : s == 2 ? 0x0F0F0F0Fu : s == 3 ? 0x33333333u : 0x55555555u;
        const bool upper = lane & d;
        t.keep[s] = upper ? ~m : m;
        t.rot[s] = upper ? d : 32u - d;       // v_alignbit_b32(p, p, n) = rotate right by n
    }
    return t;
}
template <int S>
__device__ __forceinline__ uint32_t bits_tr_stage(uint32_t x, const BitsTr &t) {
    constexpr int d = 16 >> S;
    // partner lane ^ d: hipcc turns the small distances into DPP moves (VALU speed), the large ones into LDS-crossbar
    // permutes; five ds_swizzle per word were measured slower (1.73 vs 1.51 ms for the check of Sha256(2048) x 65 536)
    const uint32_t p = (uint32_t)__shfl_xor((int)x, d);
    const uint32_t r = __builtin_amdgcn_alignbit(p, p, t.rot[S]);
    return (x & t.keep[S]) | (r & ~t.keep[S]);                                                // v_bfi_b32
}
__device__ __forceinline__ uint32_t bits_transpose32(uint32_t x, const BitsTr &t) {
    x = bits_tr_stage<0>(x, t);
    x = bits_tr_stage<1>(x, t);
    x = bits_tr_stage<2>(x, t);
    x = bits_tr_stage<3>(x, t);
    return bits_tr_stage<4>(x, t);
}
// the word of every instance from the 32 masks of a whole word (slots s .. s + 31 = bits 0 .. 31): lane l loads the
// (l >> 5)-th dword of mask l & 31 - one coalesced 256-byte load, lanes 0..31 then hold the rows of the instances 0..31
// and lanes 32..63 those of the instances 32..63 - and the transpose hands lane i the word of instance i
__device__ __forceinline__ uint32_t bits_word_load(const uint64_t *__restrict__ Tg, uint32_t sh, uint32_t slot, uint32_t lane) {
    return ((const uint32_t *)(Tg + ((size_t)(slot + (lane & 31u)) << sh)))[lane >> 5];
}
__global__ void __launch_bounds__(64)
cw_bits_r1cs_int_kernel(const uint4 *__restrict__ chunk, uint32_t n_chunks, const uint32_t *__restrict__ words,
                        const uint2 *__restrict__ itab, const uint32_t *__restrict__ row_orig,
                        const uint64_t *__restrict__ T, uint64_t slots, uint32_t sh, const uint64_t *__restrict__ only,
                        uint32_t n_groups, uint32_t batch, uint32_t *status, uint32_t *first_bad) {
    const uint32_t lane = threadIdx.x;
    const BitsTr TR = bits_tr_setup(lane);
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const uint32_t i = g * 64 + lane;
    if (only && only[g] == 0) continue;
    const uint64_t *Tg = bits_group(T, slots, sh, g);
    uint32_t bad = 0xFFFFFFFFu;
    for (uint32_t cix = blockIdx.y; cix < n_chunks; cix += gridDim.y) {
        const uint4 ch = chunk[cix];                                // first word, groups, -, first row
        const uint32_t *wp = words + ch.x;
        int64_t A = 0, B = 0, cur = 0;
        uint32_t row = ch.w;
        for (uint32_t gi = 0; gi < ch.y; gi++) {
            const uint32_t hdr = wp[0];
            const uint32_t nb = hdr & 0xFFu;
            wp++;
            if (hdr & (1u << 15)) {                                 // whole words: nb entries slot | half << 30 | sign << 31 (list padded to 8)
                const uint32_t padded = (nb + 7u) & ~7u;
                uint64_t pos = 0, neg = 0;                          // sums of the words with + / - sign, as 64-bit integers
                for (uint32_t j = 0; j < nb; j += 4) {
                    uint32_t e[4], x[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) e[k] = wp[j + k];
#pragma unroll
                    for (int k = 0; k < 4; k++) x[k] = bits_word_load(Tg, sh, e[k] & 0x3FFFFFFFu, lane);   // padding entries name slot 0: valid memory, unused
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t y = j + k < nb ? bits_transpose32(x[k], TR) : 0u;
                        const uint64_t v = (e[k] & (1u << 30)) ? ((uint64_t)y << 32) : (uint64_t)y;       // wave-uniform choices
                        if (e[k] >> 31) neg += v;
                        else pos += v;
                    }
                }
                cur += (int64_t)(pos - neg);
                wp += padded;
            } else if (!(hdr & (1u << 10))) {
                uint32_t acc = 0;
                for (uint32_t blk = 0; blk < nb; blk++, wp += 8) {
                    uint32_t w[8];
                    uint64_t m[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) w[k] = wp[k];
                    if (w[0] >> 31) {                                          // 8 consecutive slots: one 64-byte scalar load
                        const uint64_t *mp = Tg + ((size_t)((w[0] & 0x7FFFFFFFu) >> 5) << sh);
#pragma unroll
                        for (int k = 0; k < 8; k++) m[k] = mp[(size_t)k << sh];
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; k++) m[k] = Tg[(size_t)(w[k] >> 5) << sh];       // wave-uniform addresses: scalar loads
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) acc |= bits_lane_bit(m[k]) << (w[k] & 31u);
                }
                const int64_t val = (int64_t)((hdr & (1u << 9)) ? ((uint64_t)acc << 32) : (uint64_t)acc);
                cur += (hdr & (1u << 8)) ? -val : val;
            } else {
                for (uint32_t blk = 0; blk < nb; blk++, wp += 8) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint64_t m = Tg[(size_t)wp[2 * k] << sh];
                        const uint2 cw = itab[wp[2 * k + 1]];
                        const int64_t cf = (int64_t)(((uint64_t)cw.y << 32) | cw.x);
                        cur += bits_lane_bit(m) ? cf : 0;
                    }
                }
            }
            if (hdr & (1u << 13)) {                                 // last group of its part
                const uint32_t part = (hdr >> 11) & 3u;
                if (part == 0) { A = cur; cur = 0; }
                else if (part == 1) { B = cur; cur = 0; }
            }
            if (hdr & (1u << 14)) {                                 // end of the row
                const __int128 z = (__int128)A * (__int128)B - (__int128)cur;
                if (z != 0) {
                    const uint32_t oc = row_orig[row];
                    if (oc < bad) bad = oc;
                }
                row++;
                A = 0; B = 0; cur = 0;
            }
        }
    }
    if (bad != 0xFFFFFFFFu && i < batch) {
        atomicMin(&first_bad[i], bad);
        atomicOr(&status[i], CW_ST_R1CS_FAILED);
    }
    }
}

// ---- launch wrappers ---------------------------------------------------------------------------------------------------
hipError_t cwk_bits_init(hipStream_t s, void *T, uint64_t slots, uint32_t sh, uint32_t n_groups, void *fbmask, void *r1flag,
                         uint32_t *status, uint32_t *first_bad, uint32_t Bp) {
    const uint32_t n = n_groups > Bp ? n_groups : Bp;
    hipLaunchKernelGGL(cw_bits_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (uint64_t *)T, slots, sh, n_groups,
                       (uint64_t *)fbmask, (uint64_t *)r1flag, status, first_bad, Bp);
    return hipGetLastError();
}
hipError_t cwk_bits_ingest(hipStream_t s, const void *in, void *T, uint64_t slots, uint32_t sh, uint32_t input_slot0, uint32_t n_in,
                           uint32_t batch, void *fbmask) {
    if (n_in == 0 || batch == 0) return hipSuccess;
    const uint32_t per = 64 * BITS_INGEST_WAVES, n_chunks = (n_in + per - 1) / per;
    const uint64_t blocks = (uint64_t)((batch + 63) / 64) * n_chunks;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cw_bits_ingest_kernel, dim3((uint32_t)blocks), dim3(per), 0, s, (const uint4 *)in, (uint64_t *)T, slots, sh, input_slot0, n_in,
                       batch, (uint64_t *)fbmask, n_chunks);
    return hipGetLastError();
}
hipError_t cwk_bits_eval(hipStream_t s, const void *recs, const uint32_t *cmds, uint32_t n_batches, uint32_t ring, uint32_t cache,
                         void *T, uint64_t slots, uint32_t n_groups, uint32_t width, const uint32_t *aslots, uint32_t n_asserts,
                         void *fbmask) {
    const uint32_t const_off = (ring + cache) * 512u;
    const size_t lds = (size_t)const_off + 16;
    typedef void (*kern_t)(const uint4 *, const uint32_t *, uint32_t, uint32_t, uint64_t *, uint64_t);
    kern_t k = width == 16 ? (kern_t)cw_bits_eval_kernel<16> : width == 32 ? (kern_t)cw_bits_eval_kernel<32> : (kern_t)cw_bits_eval_kernel<64>;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k, dim3(n_groups, 64 / width), dim3(64), lds, s, (const uint4 *)recs, cmds, n_batches, const_off, (uint64_t *)T, slots);
    if (n_asserts)
        hipLaunchKernelGGL(cw_bits_assert_kernel, dim3((n_groups + 255) / 256), dim3(256), 0, s, (const uint64_t *)T, slots, aslots, n_asserts,
                           n_groups, (uint64_t *)fbmask);
    return hipGetLastError();
}
hipError_t cwk_bits_gather(hipStream_t s, const void *T, uint64_t slots, uint32_t sh, const uint32_t *wslot, uint32_t n_wit, uint32_t first,
                           uint32_t count, void *out) {
    if (!count || !n_wit) return hipSuccess;
    const uint32_t kblocks = (n_wit + 128 * BITS_GATHER_RUN - 1) / (128 * BITS_GATHER_RUN);
    for (uint32_t done = 0; done < count; done += 65535u * 64u) {       // grid.y limit
        const uint32_t n = count - done < 65535u * 64u ? count - done : 65535u * 64u, groups = (n + 63) / 64;
        const uint32_t gf = groups >= 8 && kblocks <= 65535u;           // see the kernel's comment
        hipLaunchKernelGGL(cw_bits_gather_kernel, gf ? dim3(groups, kblocks) : dim3(kblocks, groups), dim3(256), 0, s, (const uint64_t *)T, slots,
                           sh, wslot, n_wit, first + done, n, (uint4 *)out + (size_t)done * n_wit * 2, gf);
    }
    return hipGetLastError();
}
hipError_t cwk_bits_ingest_packed(hipStream_t s, const void *masks, void *T, uint64_t slots, uint32_t sh, uint32_t input_slot0, uint32_t n_in,
                                  uint32_t batch) {
    if (n_in == 0) return hipSuccess;
    dim3 g((batch + 63) / 64, (n_in + 255) / 256);
    if (g.y > 65535u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cw_bits_ingest_packed_kernel, g, dim3(256), 0, s, (const uint64_t *)masks, (uint64_t *)T, slots, sh, input_slot0, n_in,
                       batch);
    return hipGetLastError();
}
hipError_t cwk_bits_collect_inputs(hipStream_t s, const void *masks, const void *aos, const uint32_t *inst, uint32_t n_inst,
                                   uint32_t n_in, void *out) {
    if (!n_inst || !n_in) return hipSuccess;
    for (uint32_t done = 0; done < n_inst; done += 65535u) {
        const uint32_t n = n_inst - done < 65535u ? n_inst - done : 65535u;
        hipLaunchKernelGGL(cw_bits_collect_inputs_kernel, dim3((n_in + 255) / 256, n), dim3(256), 0, s, (const uint64_t *)masks,
                           (const uint4 *)aos, inst + done, n, n_in, (uint4 *)out + (size_t)done * n_in * 2);
    }
    return hipGetLastError();
}
hipError_t cwk_bits_r1cs(hipStream_t s, const void *erecs, uint32_t n_evrows, const uint32_t *chunk, uint32_t n_chunks,
                         const uint32_t *terms, const uint32_t *ctab, const uint32_t *row_orig, const uint32_t *ichunk,
                         uint32_t n_ichunks, const uint32_t *iterms, const uint32_t *itab, const uint32_t *irow_orig, const void *T,
                         uint64_t slots, uint32_t sh, const void *only, uint32_t n_groups, uint32_t batch, uint32_t *status, uint32_t *first_bad,
                         const FpParams &P) {
    const uint64_t *onl = (const uint64_t *)only;
    // audit of flagged groups: 1 024 workgroups walk the flag words (a launch of one workgroup per group costs ~14 ns each:
    // 1.4 ms of nothing at 2 M instances)
    const uint32_t gx = onl && n_groups > 1024u ? 1024u : n_groups;
    if (n_evrows) {
        // the kernel is bound by the latency of its 5 mask loads per lane: ~8 waves per SIMD (8192 on the chip) hide it
        uint32_t chunks = onl ? 4u : (8192 + n_groups - 1) / n_groups;
        if (chunks > n_evrows) chunks = n_evrows;
        if (chunks > 65535u) chunks = 65535u;
        if (chunks < 1) chunks = 1;
        const uint32_t per = (n_evrows + chunks - 1) / chunks;
        chunks = (n_evrows + per - 1) / per;
        hipLaunchKernelGGL(cw_bits_r1cs_lut_kernel, dim3(gx, chunks), dim3(64), 0, s, (const uint4 *)erecs, n_evrows, per,
                           (const uint64_t *)T, slots, sh, onl, n_groups, batch, status, first_bad);
    }
    if (n_ichunks) {
        dim3 g(gx, onl ? (n_ichunks < 4u ? n_ichunks : 4u) : n_ichunks < 65535u ? n_ichunks : 65535u);
        hipLaunchKernelGGL(cw_bits_r1cs_int_kernel, g, dim3(64), 0, s, (const uint4 *)ichunk, n_ichunks, iterms,
                           (const uint2 *)itab, irow_orig, (const uint64_t *)T, slots, sh, onl, n_groups, batch, status, first_bad);
    }
    if (n_chunks) {
        dim3 g(gx, onl ? (n_chunks < 4u ? n_chunks : 4u) : n_chunks < 65535u ? n_chunks : 65535u);
        hipLaunchKernelGGL(cw_bits_r1cs_wide_kernel, g, dim3(64), 0, s, (const uint4 *)chunk, n_chunks, (const uint2 *)terms, ctab,
                           row_orig, (const uint64_t *)T, slots, sh, onl, n_groups, batch, status, first_bad, P);
    }
    return hipGetLastError();
}
