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
// operands come from the previous vrow (ds_bpermute), the ring, or the bit table.  No barriers: a single wave,
// in-order LDS and in-order vector memory.
//
// The assumption "main inputs are 0/1" is checked by the ingest kernel; instances that violate it, or trip an
// assertion gate, are flagged in fbmask[group] and re-evaluated by the 256-bit schedule (cw_host.cpp), so every
// result is the reference's for every input.
#include <hip/hip_runtime.h>
#include "cw_kernels.h"
#include "fp256.hip.h"


// ---- init: constant slots, flags -------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cw_bits_init_kernel(uint64_t *T, uint64_t slots, uint32_t n_groups, uint64_t *fbmask,
                                                            uint32_t *status, uint32_t *first_bad, uint32_t Bp) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_groups) {
        T[(size_t)i * slots + 0] = 0;
        T[(size_t)i * slots + 1] = ~0ull;
        T[(size_t)i * slots + 2] = 0;
        fbmask[i] = 0;
    }
    if (i < Bp) {
        status[i] = 0;
        first_bad[i] = 0xFFFFFFFFu;
    }
}

// ---- ingest: AoS canonical inputs [batch][n_in][32 B] -> one mask per (group, input) -----------------------------------
// (setInputSignal's `signalValues[si] = val`, calcwit.cpp:93, for 64 instances at a time).  A wave handles 64 inputs
// of one group: lane k owns input k0 + k and walks the 64 instances of the group, so every load instruction reads
// 2 KiB contiguous (64 consecutive inputs of one instance) and the lane shifts the value's low bit into its own mask;
// the 64 masks leave as one coalesced 512-byte store.  Values other than 0/1 flag their instance.
__global__ void __launch_bounds__(64) cw_bits_ingest_kernel(const uint4 *__restrict__ in, uint64_t *__restrict__ T,
                                                             uint64_t slots, uint32_t input_slot0, uint32_t n_in,
                                                             uint32_t batch, uint64_t *fbmask) {
    const uint32_t lane = threadIdx.x, g = blockIdx.x, k = blockIdx.y * 64 + lane;
    const bool have = k < n_in;
    const uint32_t i0 = g * 64, ni = min(64u, batch - i0);
    uint64_t mine = 0, badmask = 0;
    for (uint32_t ii = 0; ii < ni; ii++) {
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (have) {
            const size_t src = ((size_t)(i0 + ii) * n_in + k) * 2;
            lo = in[src];
            hi = in[src + 1];
        }
        const bool isbit = (lo.x <= 1u) & ((lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) == 0u);
        mine |= (uint64_t)(lo.x & 1u) << ii;
        if (__any(!isbit)) badmask |= 1ull << ii;                   // wave-uniform
    }
    if (have) T[(size_t)g * slots + input_slot0 + k] = mine;
    if (lane == 0 && badmask) atomicOr((unsigned long long *)&fbmask[g], (unsigned long long)badmask);
}

// ---- the gate program ------------------------------------------------------------------------------------------------------
// Record (hip_elements/bitsched.py), 4 words per lane:
//   w0 = a_off | b_off << 16      LDS byte offsets of ring entries (8-byte entries, < R * 512)
//   w1 = c_off | tt << 16 | flags << 24
//   w2 = g_off                    LOAD lanes: byte offset of a bit-table slot; gate lanes: NONE
//   w3 = d_off                    byte offset of the slot this value is stored to, NONE = 0xFFFFFFFF
// Signals that are copies of one another share a slot (sig_slot[] maps signal -> slot), so a value is stored once.
// result = LUT(a, b, c) | loaded value: gate lanes load nothing (NONE is out of the buffer's range: the hardware
// returns 0), load lanes carry table 0.  The destination goes through the same buffer descriptor, so a NONE store is
// dropped: no branches, and hipcc counts every memory operation exactly (its waits for later loads then do not
// drain the stores — vmcnt counts loads and stores alike on gfx9).
// W = instances per wave: 64 (one wave per group), or 32 / 16 (2 / 4 independent waves per group, each on its slice
// of every mask: small batches then spread over more CUs and the lookups run on 32-bit halves).
#define BITS_NONE 0xFFFFFFFFu
#define BITS_F_ASSERT 1u
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t bfi32(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }   // v_bfi_b32
// 3-input lookup on 32 instances: index bit 0 = a, bit 1 = b, bit 2 = c; t[m] = 0 or ~0
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
    static __device__ __forceinline__ T lds(const char *p) { return *(const uint64_t *)p; }
    static __device__ __forceinline__ void lds_st(char *p, T v) { *(uint64_t *)p = v; }
    static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
        return ((uint64_t)v.y << 32) | v.x;
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, uint32_t off, T v) {
        const u32x2 x = {(uint32_t)v, (uint32_t)(v >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(x, r, (int)off, 0, 0);
    }
    static __device__ __forceinline__ T lut(T a, T b, T c, uint32_t w1) { return lut3(a, b, c, w1 >> 16); }
};
template <> struct BitsMask<32> {
    typedef uint32_t T;
    static __device__ __forceinline__ T lds(const char *p) { return *(const uint32_t *)p; }
    static __device__ __forceinline__ void lds_st(char *p, T v) { *(uint32_t *)p = v; }
    static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        return __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0);
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, uint32_t off, T v) {
        __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)off, 0, 0);
    }
    static __device__ __forceinline__ T lut(T a, T b, T c, uint32_t w1) {
        uint32_t t[8];
#pragma unroll
        for (int m = 0; m < 8; m++) t[m] = (uint32_t)((int32_t)(w1 << (15 - m)) >> 31);
        return lut3_32(a, b, c, t);
    }
};
template <> struct BitsMask<16> {
    typedef uint32_t T;
    static __device__ __forceinline__ T lds(const char *p) { return *(const uint32_t *)p; }
    static __device__ __forceinline__ void lds_st(char *p, T v) { *(uint32_t *)p = v; }
    static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r, (int)off, 0, 0);
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, uint32_t off, T v) {
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, r, (int)off, 0, 0);
    }
    static __device__ __forceinline__ T lut(T a, T b, T c, uint32_t w1) { return BitsMask<32>::lut(a, b, c, w1); }
};

extern __shared__ uint64_t cw_bits_ring[];       // [R][64 lanes] x 8 B

// timing experiments only (tools/bits_exp.sh; results are garbage): drop the stores / the bit-table loads
#ifdef CW_EXP_NOSTORE
#define BITS_EXP_STORE(x)
#else
#define BITS_EXP_STORE(x) x
#endif
#ifdef CW_EXP_NOLOAD
#define BITS_EXP_LOAD(x) 0
#else
#define BITS_EXP_LOAD(x) x
#endif

// The program runs in BATCHES of BITS_NB vrows so that vector-memory traffic never sits on the critical path:
// at the start of batch b the wave requests, in one burst, the records of batch b + 2 and the bit-table values of the
// LOAD lanes of batch b + 1 (their addresses are in the records of batch b + 1, resident since the previous burst);
// then the BITS_NB steps run on registers and LDS only (ring operands of step k + 1 are read while step k computes);
// at the end of the batch its BITS_NB results are stored in one burst.  Everything a batch consumes was requested a
// whole batch (~1.5 K clocks) earlier, so the waits hipcc inserts (vmcnt counts loads and stores alike on gfx9 and
// retires in order) find their operations long complete.  The three record sets and two loaded-value sets rotate
// by NAME through six expansions of the batch body (no register moves).
// Scheduler contract (bitsched.py): a LOAD lane of batch b reads a value stored by batch b - 2 or older; LOAD lanes only
// sit in even vrows (the kernel issues no bit-table request for odd ones: the vector-memory path - ~23 TA cycles per
// instruction whatever its width - is the busiest unit of this kernel, 79 % at 1 024 waves).
#define BITS_NB 8

template <int W>
struct BitsEval {
    typedef BitsMask<W> M;
    typedef typename M::T mask_t;
    const uint4 *__restrict__ recs;
    char *ring;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t lane, ring_mask;
    mask_t a, b, c, viol;

    // cur: records of this batch, nxt: of the next one, fill: receives batch + 2; gcur: loaded values of this batch,
    // gfill: receives those of the next one
    __device__ __forceinline__ void batch(uint32_t v0, const uint4 (&cur)[BITS_NB], const uint4 (&nxt)[BITS_NB], uint4 (&fill)[BITS_NB],
                                          const mask_t (&gcur)[BITS_NB], mask_t (&gfill)[BITS_NB]) {
#pragma unroll
        for (int k = 0; k < BITS_NB; k++) fill[k] = recs[(size_t)(v0 + 2 * BITS_NB + k) * 64 + lane];
        // LOAD lanes sit in even vrows only (bitsched.py LOAD_EVERY = 2): half the bit-table requests
#pragma unroll
        for (int k = 0; k < BITS_NB; k += 2) gfill[k] = BITS_EXP_LOAD(M::load(rsrc, nxt[k].z));
        mask_t res[BITS_NB];
#pragma unroll
        for (int k = 0; k < BITS_NB; k++) {
            const uint4 rn = k + 1 < BITS_NB ? cur[k + 1 < BITS_NB ? k + 1 : 0] : nxt[0];
            const mask_t na = M::lds(ring + (rn.x & 0xFFFFu)), nb = M::lds(ring + (rn.x >> 16)), nc = M::lds(ring + (rn.y & 0xFFFFu));
            const mask_t r = (k & 1) ? M::lut(a, b, c, cur[k].y) : (M::lut(a, b, c, cur[k].y) | gcur[k]);
            if ((cur[k].y >> 24) & BITS_F_ASSERT) viol |= r;
            M::lds_st(ring + (((v0 + k) & ring_mask) << 9) + lane * 8, r);
            res[k] = r;
            a = na; b = nb; c = nc;
        }
#pragma unroll
        for (int k = 0; k < BITS_NB; k++) BITS_EXP_STORE(M::store(rsrc, cur[k].w, res[k]));
    }
};

template <int W>
__global__ void __launch_bounds__(64)
cw_bits_eval_kernel(const uint4 *__restrict__ recs, uint32_t n_batches, uint32_t ring_mask, uint64_t *T, uint64_t slots,
                    uint64_t *fbmask) {
    typedef BitsMask<W> M;
    typedef typename M::T mask_t;
    const uint32_t lane = threadIdx.x, g = blockIdx.x, slice = blockIdx.y;
    char *Tg = (char *)(T + (size_t)g * slots) + slice * (W / 8);
    if (n_batches == 0) return;
    BitsEval<W> E;
    E.recs = recs;
    E.ring = (char *)cw_bits_ring;
    // buffer descriptor of this group's table (wave-uniform by construction: kernel arguments and blockIdx only);
    // a wave of a narrower slice addresses its bytes of every 8-byte mask through the shifted base
    E.rsrc = __builtin_amdgcn_make_buffer_rsrc(Tg, 0, (int)(uint32_t)(slots * 8 - slice * (W / 8)), 0x00020000);
    E.lane = lane;
    E.ring_mask = ring_mask;
    E.viol = 0;
    // the host pads the program to whole batches and appends three empty ones (records are requested two batches ahead,
    // the loop runs in trips of six batches)
    uint4 R0[BITS_NB], R1[BITS_NB], R2[BITS_NB];
    mask_t G0[BITS_NB], G1[BITS_NB];
#pragma unroll
    for (int k = 0; k < BITS_NB; k++) {
        R0[k] = recs[(size_t)k * 64 + lane];
        R1[k] = recs[(size_t)(BITS_NB + k) * 64 + lane];
    }
#pragma unroll
    for (int k = 0; k < BITS_NB; k += 2) G0[k] = BITS_EXP_LOAD(M::load(E.rsrc, R0[k].z));
    E.a = M::lds(E.ring + (R0[0].x & 0xFFFFu));
    E.b = M::lds(E.ring + (R0[0].x >> 16));
    E.c = M::lds(E.ring + (R0[0].y & 0xFFFFu));
    uint32_t v = 0;
    const uint32_t n_steps = n_batches * BITS_NB;
    while (v < n_steps) {
        E.batch(v, R0, R1, R2, G0, G1); v += BITS_NB;
        E.batch(v, R1, R2, R0, G1, G0); v += BITS_NB;
        E.batch(v, R2, R0, R1, G0, G1); v += BITS_NB;
        E.batch(v, R0, R1, R2, G1, G0); v += BITS_NB;
        E.batch(v, R1, R2, R0, G0, G1); v += BITS_NB;
        E.batch(v, R2, R0, R1, G1, G0); v += BITS_NB;
    }
    const mask_t viol = E.viol;
    // instances that tripped an assertion gate: OR over the lanes (gates), then into the group's fallback mask
    uint64_t vz = (uint64_t)viol;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)vz, off), hi = (uint32_t)__shfl_xor((int)(uint32_t)(vz >> 32), off);
        vz |= ((uint64_t)hi << 32) | lo;
    }
    if (W < 64) vz = (vz & ((1ull << (W & 63)) - 1)) << (slice * W);
    if (lane == 0 && vz) atomicOr((unsigned long long *)&fbmask[g], (unsigned long long)vz);
}

// ---- egress: canonical 32-byte values from the bit table (getWitness + Fr_toLongNormal, main.cpp:326-332) ----------------
// element k of instance `first + blockIdx.y` -> out[(blockIdx.y * n_wit + k)]; w2s = witness -> signal map
__global__ void __launch_bounds__(256)
cw_bits_gather_kernel(const uint64_t *__restrict__ T, uint64_t slots, const uint32_t *__restrict__ w2s,
                      const uint32_t *__restrict__ sig_slot, uint32_t n_wit, uint32_t first, uint4 *__restrict__ out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_wit) return;
    const uint32_t i = first + blockIdx.y;
    const uint64_t m = T[(size_t)(i >> 6) * slots + sig_slot[w2s[k]]];
    const uint32_t bit = (uint32_t)(m >> (i & 63u)) & 1u;
    const size_t o = ((size_t)blockIdx.y * n_wit + k) * 2;
    out[o] = make_uint4(bit, 0, 0, 0);
    out[o + 1] = make_uint4(0, 0, 0, 0);
}

// ---- R1CS check on the bit table ---------------------------------------------------------------------------------------------
// Class E: constraints over <= 5 distinct wires.  The host enumerated A*B - C over the 2^k assignments of the wires
// (cw_host.cpp, exact integer arithmetic): the constraint is a 32-entry truth table "violated?".  A vrow checks 64
// constraints for 64 instances: 5 mask loads per lane, four 3-input lookups + three selects.
struct ERec { uint32_t w[5]; uint32_t tt, row, pad; };
__global__ void __launch_bounds__(64)
cw_bits_r1cs_lut_kernel(const uint4 *__restrict__ recs, uint32_t n_vrows, uint32_t vrows_per_chunk, const uint64_t *__restrict__ T,
                        uint64_t slots, uint32_t batch, uint32_t *status, uint32_t *first_bad) {
    const uint32_t lane = threadIdx.x, g = blockIdx.x;
    const char *Tg = (const char *)(T + (size_t)g * slots);
    const uint32_t v0 = blockIdx.y * vrows_per_chunk, v1 = min(n_vrows, v0 + vrows_per_chunk);
    for (uint32_t v = v0; v < v1; v++) {
        const uint4 x = recs[((size_t)v * 64 + lane) * 2], y = recs[((size_t)v * 64 + lane) * 2 + 1];
        const uint64_t w0 = *(const uint64_t *)(Tg + x.x), w1 = *(const uint64_t *)(Tg + x.y), w2 = *(const uint64_t *)(Tg + x.z),
                       w3 = *(const uint64_t *)(Tg + x.w), w4 = *(const uint64_t *)(Tg + y.x);
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

// Class W: any other constraint (long linear rows of BinSum / Bits2Num shape, field-sized coefficients).  One lane =
// one instance; terms are wave-uniform (scalar loads), a wire's mask is ONE 8-byte scalar-cache read for the whole
// wave and each lane picks its bit.  term = {slot byte offset | part (2 bits) << 30, coefficient id}; coefficient
// table = canonical residues; row ends as in cw_r1cs_stream_kernel: (A*B + (q - C)) * R'^-1 == 0.
__global__ void __launch_bounds__(64)
cw_bits_r1cs_wide_kernel(const uint4 *__restrict__ chunk, uint32_t n_chunks, const uint2 *__restrict__ terms,
                         const uint32_t *__restrict__ ctab, const uint32_t *__restrict__ row_orig,
                         const uint64_t *__restrict__ T, uint64_t slots, uint32_t batch, uint32_t *status,
                         uint32_t *first_bad, FpParams P) {
    const uint32_t lane = threadIdx.x, g = blockIdx.x;
    const uint32_t i = g * 64 + lane;
    const char *Tg = (const char *)(T + (size_t)g * slots);
    uint32_t bad = 0xFFFFFFFFu;
    for (uint32_t cix = blockIdx.y; cix < n_chunks; cix += gridDim.y) {
        const uint4 ch = chunk[cix];                                // first term, n terms, -, first row
        const uint2 *tp = terms + ch.x;
        fe A = fe_zero(), B = fe_zero(), cur = fe_zero();
        uint32_t row = ch.w;
        for (uint32_t k = 0; k < ch.y; k++) {
            const uint2 t = tp[k];
            const uint32_t off = t.x & 0x0FFFFFFFu, part = (t.x >> 28) & 3u, last = t.x >> 31, endrow = (t.x >> 30) & 1u;
            const uint64_t m = *(const uint64_t *)(Tg + off);      // wave-uniform address
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
// leaves with column k.  Five butterfly stages; stage d exchanges the off-diagonal d x d blocks between lanes k and k ^ d.
__device__ __forceinline__ uint32_t bits_transpose32(uint32_t x, uint32_t lane) {
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const uint32_t d = 16u >> s;
        const uint32_t m = s == 0 ? 0x0000FFFFu : s == 1 ? 0x00FF00FFu : s == 2 ? 0x0F0F0F0Fu : s == 3 ? 0x33333333u : 0x55555555u;
        const uint32_t p = (uint32_t)__shfl_xor((int)x, (int)d);
        x = (lane & d) ? (((p >> d) & m) | (x & ~m)) : ((x & m) | ((p & m) << d));
    }
    return x;
}
// the word of every instance from the 32 masks of a whole word (slots s .. s + 31 = bits 0 .. 31): lane l loads the
// (l >> 5)-th dword of mask l & 31 - one coalesced 256-byte load, lanes 0..31 then hold the rows of the instances 0..31
// and lanes 32..63 those of the instances 32..63 - and the transpose hands lane i the word of instance i
__device__ __forceinline__ uint32_t bits_word_load(const uint64_t *__restrict__ Tg, uint32_t slot, uint32_t lane) {
    return ((const uint32_t *)(Tg + slot))[(lane & 31u) * 2u + (lane >> 5)];
}
__global__ void __launch_bounds__(64)
cw_bits_r1cs_int_kernel(const uint4 *__restrict__ chunk, uint32_t n_chunks, const uint32_t *__restrict__ words,
                        const uint2 *__restrict__ itab, const uint32_t *__restrict__ row_orig,
                        const uint64_t *__restrict__ T, uint64_t slots, uint32_t batch, uint32_t *status,
                        uint32_t *first_bad) {
    const uint32_t lane = threadIdx.x, g = blockIdx.x;
    const uint32_t i = g * 64 + lane;
    const uint64_t *Tg = T + (size_t)g * slots;
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
            if (hdr & (1u << 15)) {                                 // whole words: nb first slots, four loads in flight
                const uint32_t padded = (nb + 7u) & ~7u;
                for (uint32_t j = 0; j < nb; j += 4) {
                    uint32_t sl[4], x[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) sl[k] = wp[j + k];                 // the list is padded to 8 entries (zeros)
#pragma unroll
                    for (int k = 0; k < 4; k++) x[k] = bits_word_load(Tg, j + k < nb ? sl[k] : sl[0], lane);   // padding: a valid word, unused
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t y = j + k < nb ? bits_transpose32(x[k], lane) : 0u;
                        const int64_t val = (int64_t)((hdr & (1u << 9)) ? ((uint64_t)y << 32) : (uint64_t)y);
                        cur += (hdr & (1u << 8)) ? -val : val;
                    }
                }
                wp += padded;
            } else if (!(hdr & (1u << 10))) {
                uint32_t acc = 0;
                for (uint32_t blk = 0; blk < nb; blk++, wp += 8) {
                    uint32_t w[8];
                    uint64_t m[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) w[k] = wp[k];
                    if (w[0] >> 31) {                                          // 8 consecutive slots: one 64-byte scalar load
                        const uint64_t *mp = Tg + ((w[0] & 0x7FFFFFFFu) >> 5);
#pragma unroll
                        for (int k = 0; k < 8; k++) m[k] = mp[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; k++) m[k] = Tg[w[k] >> 5];       // wave-uniform addresses: scalar loads
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
                        const uint64_t m = Tg[wp[2 * k]];
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

// ---- launch wrappers ---------------------------------------------------------------------------------------------------
hipError_t cwk_bits_init(hipStream_t s, void *T, uint64_t slots, uint32_t n_groups, void *fbmask, uint32_t *status,
                         uint32_t *first_bad, uint32_t Bp) {
    const uint32_t n = n_groups > Bp ? n_groups : Bp;
    hipLaunchKernelGGL(cw_bits_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (uint64_t *)T, slots, n_groups,
                       (uint64_t *)fbmask, status, first_bad, Bp);
    return hipGetLastError();
}
hipError_t cwk_bits_ingest(hipStream_t s, const void *in, void *T, uint64_t slots, uint32_t input_slot0, uint32_t n_in,
                           uint32_t batch, void *fbmask) {
    if (n_in == 0) return hipSuccess;
    dim3 g((batch + 63) / 64, (n_in + 63) / 64);
    if (g.y > 65535u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cw_bits_ingest_kernel, g, dim3(64), 0, s, (const uint4 *)in, (uint64_t *)T, slots, input_slot0, n_in,
                       batch, (uint64_t *)fbmask);
    return hipGetLastError();
}
hipError_t cwk_bits_eval(hipStream_t s, const void *recs, uint32_t n_vrows, uint32_t ring, void *T, uint64_t slots,
                         uint32_t n_groups, uint32_t width, void *fbmask) {
    const size_t lds = (size_t)ring * 512;
    typedef void (*kern_t)(const uint4 *, uint32_t, uint32_t, uint64_t *, uint64_t, uint64_t *);
    kern_t k = width == 16 ? (kern_t)cw_bits_eval_kernel<16> : width == 32 ? (kern_t)cw_bits_eval_kernel<32> : (kern_t)cw_bits_eval_kernel<64>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(n_groups, 64 / width), dim3(64), lds, s, (const uint4 *)recs, n_vrows, ring - 1, (uint64_t *)T, slots,
                       (uint64_t *)fbmask);
    return hipGetLastError();
}
hipError_t cwk_bits_gather(hipStream_t s, const void *T, uint64_t slots, const uint32_t *w2s, const uint32_t *sig_slot,
                           uint32_t n_wit, uint32_t first, uint32_t count, void *out) {
    if (!count || !n_wit) return hipSuccess;
    for (uint32_t done = 0; done < count; done += 65535u) {       // grid.y limit
        const uint32_t n = count - done < 65535u ? count - done : 65535u;
        hipLaunchKernelGGL(cw_bits_gather_kernel, dim3((n_wit + 255) / 256, n), dim3(256), 0, s, (const uint64_t *)T, slots, w2s,
                           sig_slot, n_wit, first + done, (uint4 *)out + (size_t)done * n_wit * 2);
    }
    return hipGetLastError();
}
hipError_t cwk_bits_r1cs(hipStream_t s, const void *erecs, uint32_t n_evrows, const uint32_t *chunk, uint32_t n_chunks,
                         const uint32_t *terms, const uint32_t *ctab, const uint32_t *row_orig, const uint32_t *ichunk,
                         uint32_t n_ichunks, const uint32_t *iterms, const uint32_t *itab, const uint32_t *irow_orig, const void *T,
                         uint64_t slots, uint32_t n_groups, uint32_t batch, uint32_t *status, uint32_t *first_bad, const FpParams &P) {
    if (n_evrows) {
        // the kernel is bound by the latency of its 5 mask loads per lane: ~8 waves per SIMD (8192 on the chip) hide it
        uint32_t chunks = (8192 + n_groups - 1) / n_groups;
        if (chunks > n_evrows) chunks = n_evrows;
        if (chunks > 65535u) chunks = 65535u;
        if (chunks < 1) chunks = 1;
        const uint32_t per = (n_evrows + chunks - 1) / chunks;
        chunks = (n_evrows + per - 1) / per;
        hipLaunchKernelGGL(cw_bits_r1cs_lut_kernel, dim3(n_groups, chunks), dim3(64), 0, s, (const uint4 *)erecs, n_evrows, per,
                           (const uint64_t *)T, slots, batch, status, first_bad);
    }
    if (n_ichunks) {
        dim3 g(n_groups, n_ichunks < 65535u ? n_ichunks : 65535u);
        hipLaunchKernelGGL(cw_bits_r1cs_int_kernel, g, dim3(64), 0, s, (const uint4 *)ichunk, n_ichunks, iterms,
                           (const uint2 *)itab, irow_orig, (const uint64_t *)T, slots, batch, status, first_bad);
    }
    if (n_chunks) {
        dim3 g(n_groups, n_chunks < 65535u ? n_chunks : 65535u);
        hipLaunchKernelGGL(cw_bits_r1cs_wide_kernel, g, dim3(64), 0, s, (const uint4 *)chunk, n_chunks, (const uint2 *)terms, ctab,
                           row_orig, (const uint64_t *)T, slots, batch, status, first_bad, P);
    }
    return hipGetLastError();
}
