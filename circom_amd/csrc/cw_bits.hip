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

#define BK_GLOBAL 0u
#define BK_RING 1u
#define BK_PREV 2u
#define BF_ASSERT 0x100u

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
// of one group: lane i reads instance i's value, the ballot over the wave is the mask; lane k keeps the mask of
// input k, so the 64 masks leave as one coalesced 512-byte store.  Values other than 0/1 flag their instance.
__global__ void __launch_bounds__(64) cw_bits_ingest_kernel(const uint4 *__restrict__ in, uint64_t *__restrict__ T,
                                                             uint64_t slots, uint32_t input_slot0, uint32_t n_in,
                                                             uint32_t batch, uint64_t *fbmask) {
    const uint32_t lane = threadIdx.x, g = blockIdx.x, k0 = blockIdx.y * 64;
    const uint32_t i = g * 64 + lane;
    const bool valid = i < batch;
    uint64_t mine = 0;
    bool bad = false;
    const uint32_t kn = min(64u, n_in - k0);
    for (uint32_t kk = 0; kk < kn; kk++) {
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (valid) {
            const size_t src = ((size_t)i * n_in + k0 + kk) * 2;
            lo = in[src];
            hi = in[src + 1];
        }
        const bool isbit = (lo.x <= 1u) & ((lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) == 0u);
        bad |= !isbit;
        const uint64_t m = __ballot(valid && (lo.x & 1u));
        if (lane == kk) mine = m;
    }
    if (lane < kn) T[(size_t)g * slots + input_slot0 + k0 + lane] = mine;
    const uint64_t bm = __ballot(valid && bad);
    if (lane == 0 && bm) atomicOr((unsigned long long *)&fbmask[g], (unsigned long long)bm);
}

// ---- the gate program ------------------------------------------------------------------------------------------------------
struct BRec {
    uint32_t a, b, c, t;      // operands (kind << 30 | offset), truth table | flags
    uint32_t d0, d1, d2, d3;  // destination byte offsets in the group's bit table (0 = none)
};
__device__ __forceinline__ BRec brec_load(const uint4 *__restrict__ recs, size_t idx) {
    const uint4 x = recs[idx * 2], y = recs[idx * 2 + 1];
    BRec r;
    r.a = x.x; r.b = x.y; r.c = x.z; r.t = x.w;
    r.d0 = y.x; r.d1 = y.y; r.d2 = y.z; r.d3 = y.w;
    return r;
}
__device__ __forceinline__ uint32_t bfi32(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }   // v_bfi_b32
// 3-input lookup on 32 instances: index bit 0 = a, bit 1 = b, bit 2 = c
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
// early operand fetch (ring / bit table), branch-free: lanes of the other kinds read entry / slot 0
__device__ __forceinline__ uint64_t bits_early(uint32_t w, const char *Tg, const char *ring) {
    const uint32_t kind = w >> 30, off = w & 0x3FFFFFFFu;
    const uint64_t gv = *(const uint64_t *)(Tg + (kind == BK_GLOBAL ? off : 0u));
    const uint64_t rv = *(const uint64_t *)(ring + (kind == BK_RING ? off : 0u));
    return kind == BK_RING ? rv : gv;
}
__device__ __forceinline__ uint64_t bits_prev(uint32_t w, uint64_t early, uint64_t res) {
    const uint32_t kind = w >> 30;
    const int addr = (int)(kind == BK_PREV ? (w & 0xFCu) : 0u);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)res);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)(res >> 32));
    const uint64_t p = ((uint64_t)hi << 32) | lo;
    return kind == BK_PREV ? p : early;
}

extern __shared__ uint64_t cw_bits_ring[];       // [R][64 lanes]
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(64)
cw_bits_eval_kernel(const uint4 *__restrict__ recs, uint32_t n_vrows, uint32_t ring_mask, uint64_t *T, uint64_t slots,
                    uint64_t *fbmask) {
    const uint32_t lane = threadIdx.x, g = blockIdx.x;
    char *Tg = (char *)(T + (size_t)g * slots);
    char *ring = (char *)cw_bits_ring;
    // buffer descriptor of this group's table (wave-uniform by construction: kernel arguments and blockIdx only)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(Tg, 0, (int)(uint32_t)(slots * 8), 0x00020000);
    // entry 0 of the ring is read by lanes that have no ring operand: keep it defined
    cw_bits_ring[lane] = 0;
    if (n_vrows == 0) return;
    // records are streamed three vrows ahead (2 KiB per vrow, coalesced); the stream is padded with 3 empty vrows
    BRec r0 = brec_load(recs, lane), r1 = brec_load(recs, 64 + lane), r2 = brec_load(recs, 128 + lane);
    uint64_t a = bits_early(r0.a, Tg, ring), b = bits_early(r0.b, Tg, ring), c = bits_early(r0.c, Tg, ring);
    uint64_t viol = 0;
    for (uint32_t v = 0; v < n_vrows; v++) {
        const BRec r3 = brec_load(recs, (size_t)(v + 3) * 64 + lane);
        // ring / bit-table operands of the NEXT vrow, requested before this vrow's results are written
        const uint64_t na = bits_early(r1.a, Tg, ring), nb = bits_early(r1.b, Tg, ring), nc = bits_early(r1.c, Tg, ring);
        const uint64_t res = lut3(a, b, c, r0.t);
        if (r0.t & BF_ASSERT) viol |= res;
        *(uint64_t *)(ring + (size_t)(v & ring_mask) * 512 + lane * 8) = res;
        // destinations: buffer stores through the group's descriptor; "none" (0) becomes an out-of-range offset, which
        // the hardware drops.  No branches: hipcc then counts the stores exactly and its waits for later loads do not
        // drain them (vmcnt counts loads and stores alike on gfx9).
        const u32x2 rv = {(uint32_t)res, (uint32_t)(res >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(rv, rsrc, (int)(r0.d0 ? r0.d0 : 0x80000000u), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(rv, rsrc, (int)(r0.d1 ? r0.d1 : 0x80000000u), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(rv, rsrc, (int)(r0.d2 ? r0.d2 : 0x80000000u), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(rv, rsrc, (int)(r0.d3 ? r0.d3 : 0x80000000u), 0, 0);
        // PREV operands of the next vrow: lanes of this vrow's result
        a = bits_prev(r1.a, na, res);
        b = bits_prev(r1.b, nb, res);
        c = bits_prev(r1.c, nc, res);
        r0 = r1; r1 = r2; r2 = r3;
    }
    // instances that tripped an assertion gate: OR over the lanes (gates), then into the group's fallback mask
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)viol, off), hi = (uint32_t)__shfl_xor((int)(uint32_t)(viol >> 32), off);
        viol |= ((uint64_t)hi << 32) | lo;
    }
    if (lane == 0 && viol) atomicOr((unsigned long long *)&fbmask[g], (unsigned long long)viol);
}

// ---- egress: canonical 32-byte values from the bit table (getWitness + Fr_toLongNormal, main.cpp:326-332) ----------------
// element k of instance `first + blockIdx.y` -> out[(blockIdx.y * n_wit + k)]; w2s = witness -> signal map
__global__ void __launch_bounds__(256)
cw_bits_gather_kernel(const uint64_t *__restrict__ T, uint64_t slots, const uint32_t *__restrict__ w2s, uint32_t n_wit,
                      uint32_t first, uint4 *__restrict__ out) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_wit) return;
    const uint32_t i = first + blockIdx.y;
    const uint64_t m = T[(size_t)(i >> 6) * slots + 3u + w2s[k]];
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
                         uint32_t n_groups, void *fbmask) {
    const size_t lds = (size_t)ring * 512;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)cw_bits_eval_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(cw_bits_eval_kernel, dim3(n_groups), dim3(64), lds, s, (const uint4 *)recs, n_vrows, ring - 1,
                       (uint64_t *)T, slots, (uint64_t *)fbmask);
    return hipGetLastError();
}
hipError_t cwk_bits_gather(hipStream_t s, const void *T, uint64_t slots, const uint32_t *w2s, uint32_t n_wit, uint32_t first,
                           uint32_t count, void *out) {
    if (!count || !n_wit) return hipSuccess;
    for (uint32_t done = 0; done < count; done += 65535u) {       // grid.y limit
        const uint32_t n = count - done < 65535u ? count - done : 65535u;
        hipLaunchKernelGGL(cw_bits_gather_kernel, dim3((n_wit + 255) / 256, n), dim3(256), 0, s, (const uint64_t *)T, slots, w2s,
                           n_wit, first + done, (uint4 *)out + (size_t)done * n_wit * 2);
    }
    return hipGetLastError();
}
hipError_t cwk_bits_r1cs(hipStream_t s, const void *erecs, uint32_t n_evrows, const uint32_t *chunk, uint32_t n_chunks,
                         const uint32_t *terms, const uint32_t *ctab, const uint32_t *row_orig, const void *T, uint64_t slots,
                         uint32_t n_groups, uint32_t batch, uint32_t *status, uint32_t *first_bad, const FpParams &P) {
    if (n_evrows) {
        // enough workgroups to fill the chip: groups x chunks >= ~2048 waves
        uint32_t chunks = (2048 + n_groups - 1) / n_groups;
        if (chunks > n_evrows) chunks = n_evrows;
        if (chunks < 1) chunks = 1;
        const uint32_t per = (n_evrows + chunks - 1) / chunks;
        chunks = (n_evrows + per - 1) / per;
        hipLaunchKernelGGL(cw_bits_r1cs_lut_kernel, dim3(n_groups, chunks), dim3(64), 0, s, (const uint4 *)erecs, n_evrows, per,
                           (const uint64_t *)T, slots, batch, status, first_bad);
    }
    if (n_chunks) {
        dim3 g(n_groups, n_chunks < 65535u ? n_chunks : 65535u);
        hipLaunchKernelGGL(cw_bits_r1cs_wide_kernel, g, dim3(64), 0, s, (const uint4 *)chunk, n_chunks, (const uint2 *)terms, ctab,
                           row_orig, (const uint64_t *)T, slots, batch, status, first_bad, P);
    }
    return hipGetLastError();
}
