// cw_kernels.h — launch wrappers implemented in cw_kernels.hip
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include "cw_tape.h"

// `mont`: the value table holds Montgomery forms x R' (lower.py pass A6): init writes R' into the constant-one slot, ingest
// multiplies by R'^2, the gathers by 1, the R1CS check compares mmul(A~, B~) with C~
hipError_t cwk_init(hipStream_t s, void *V, uint32_t Bp, uint32_t *status, uint32_t *first_bad, bool mont, const FpParams &P);
hipError_t cwk_ingest(hipStream_t s, const void *in, void *V, uint32_t input_start, uint32_t n_in, uint32_t batch,
                      uint32_t Bp, bool mont, const FpParams &P);
hipError_t cwk_eval(hipStream_t s, bool full, bool wide_linsum, const CwDRow *rows, const uint32_t *stream_off,
                    const uint64_t *extras, const uint32_t *extra_off, const uint64_t *terms, const uint32_t *term_off,
                    uint32_t n_strands, uint32_t n_lds, void *V, const uint32_t *consts, const uint32_t *lconsts,
                    const uint32_t *fncode, const uint32_t *fntab, uint64_t slot_stride,
                    uint32_t Bp, uint32_t batch, uint32_t lanes, uint32_t prio_mask, uint32_t *status, const FpParams &P);
// pipelined single-wave schedule: rows = device rows (CwPRow, padded by 3 NOPs), loads = (n_rows / nb + 2) * nld words
hipError_t cwk_eval_pipe(hipStream_t s, bool full, bool wide_linsum, uint32_t nb, uint32_t nld, const void *rows, uint32_t n_rows,
                         const uint32_t *loads, const uint64_t *terms, void *V, const uint32_t *consts, const uint32_t *lconsts,
                         uint64_t slot_stride, uint32_t Bp, uint32_t batch, uint32_t lanes, uint32_t *status, const FpParams &P);
hipError_t cwk_fill32(hipStream_t s, uint32_t *p, uint32_t v, size_t n);     // p[0..n) = v, as a kernel (graph-safe)
hipError_t cwk_fused_merge(hipStream_t s, const uint32_t *found, uint32_t batch, uint32_t *status, uint32_t *first_bad);
hipError_t cwk_r1cs(hipStream_t s, const uint32_t *chunk, uint32_t n_chunks, const uint32_t *terms, const uint32_t *ctab,
                    const uint32_t *ctab29, const uint32_t *row_orig, const void *V, uint32_t Bp, uint32_t batch, uint32_t *status,
                    uint32_t *first_bad, bool mont, const FpParams &P);
hipError_t cwk_r1cs_staged(hipStream_t s, const uint32_t *chunk, uint32_t n_chunks, const uint32_t *rec, const uint32_t *terms,
                           const uint32_t *ctab, const uint32_t *ctab29, const uint32_t *row_orig, uint32_t entries, const void *V, uint32_t Bp,
                           uint32_t batch, uint32_t *status, uint32_t *first_bad, bool mont, const FpParams &P);
hipError_t cwk_gather(hipStream_t s, const void *V, const uint32_t *w2s, uint32_t n_wit, uint32_t Bp, uint32_t instance,
                      void *out, bool mont, const FpParams &P);
hipError_t cwk_gather_many(hipStream_t s, const void *V, const uint32_t *w2s, uint32_t n_wit, uint32_t Bp, uint32_t first,
                           uint32_t count, void *out, bool mont, const FpParams &P);
hipError_t cwk_mulbench(hipStream_t s, const void *a, const void *b, void *out, uint32_t n, uint32_t iters,
                        const FpParams &P);
hipError_t cwk_fpop(hipStream_t s, uint32_t op, const void *a, const void *b, const void *c, void *out, uint32_t *status,
                    uint32_t n, const FpParams &P);

// ---- bit-plane path (cw_bits.hip) ----
// `sh` = layout shift of the bit table (cw_bits.hip: 0 = T[group][slot] of the interpreter, 5 = T[chunk][slot][group in chunk]
// of the emitted code); `only` (R1CS audit): groups whose flag word is zero are skipped (nullptr = audit every group)
hipError_t cwk_bits_init(hipStream_t s, void *T, uint64_t slots, uint32_t sh, uint32_t n_groups, void *fbmask, void *r1flag,
                         uint32_t *status, uint32_t *first_bad, uint32_t Bp);
hipError_t cwk_bits_ingest(hipStream_t s, const void *in, void *T, uint64_t slots, uint32_t sh, uint32_t input_slot0, uint32_t n_in,
                           uint32_t batch, void *fbmask);
hipError_t cwk_bits_eval(hipStream_t s, const void *recs, const uint32_t *cmds, uint32_t n_batches, uint32_t ring, uint32_t cache,
                         void *T, uint64_t slots, uint32_t n_groups, uint32_t width, const uint32_t *aslots, uint32_t n_asserts,
                         void *fbmask);
hipError_t cwk_bits_gather(hipStream_t s, const void *T, uint64_t slots, uint32_t sh, const uint32_t *wslot, uint32_t n_wit, uint32_t first,
                           uint32_t count, void *out);
hipError_t cwk_bits_ingest_packed(hipStream_t s, const void *masks, void *T, uint64_t slots, uint32_t sh, uint32_t input_slot0, uint32_t n_in,
                                  uint32_t batch);
hipError_t cwk_bits_collect_inputs(hipStream_t s, const void *masks, const void *aos, const uint32_t *inst, uint32_t n_inst,
                                   uint32_t n_in, void *out);
hipError_t cwk_bits_r1cs(hipStream_t s, const void *erecs, uint32_t n_evrows, const uint32_t *chunk, uint32_t n_chunks,
                         const uint32_t *terms, const uint32_t *ctab, const uint32_t *row_orig, const uint32_t *ichunk,
                         uint32_t n_ichunks, const uint32_t *iterms, const uint32_t *itab, const uint32_t *irow_orig, const void *T,
                         uint64_t slots, uint32_t sh, const void *only, uint32_t n_groups, uint32_t batch, uint32_t *status, uint32_t *first_bad,
                         const FpParams &P);

// ---- 64-bit runtime (cw64.hip: --prime goldilocks) ----
hipError_t cwk64_init(hipStream_t s, void *V, uint32_t Bp, uint32_t *status, uint32_t *first_bad);
hipError_t cwk64_ingest(hipStream_t s, const void *in, void *V, uint32_t input_start, uint32_t n_in, uint32_t batch, uint32_t Bp);
hipError_t cwk64_eval(hipStream_t s, const void *rows, uint32_t n_rows, const void *consts, void *V, uint32_t Bp, uint32_t batch,
                      uint32_t *status);
hipError_t cwk64_r1cs(hipStream_t s, const void *chunks, uint32_t n_chunks, const void *terms, const void *V, uint32_t Bp, uint32_t batch,
                      uint32_t *status, uint32_t *first_bad);
hipError_t cwk64_gather(hipStream_t s, const void *V, const uint32_t *w2s, uint32_t n_wit, uint32_t Bp, uint32_t first, uint32_t count,
                        void *out);
