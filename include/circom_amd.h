/* circom_amd.h — C ABI of the MI355X-native batched witness calculator for circom circuits.
 *
 * Drop-in boundary.  circom has no plugin registry; the emitted C++ calculator is
 * reached through four concrete seams.  Each entry point below names the reference interface it
 * replaces (paths relative to the iden3/circom tree, v2.2.3):
 *
 *   seam 3 (runtime)   code_producers/src/c_elements/common/calcwit.hpp:17-66  class Circom_CalcWit
 *                      code_producers/src/c_elements/common/circom.hpp:36-43,79-87  Circom_Circuit + get_*()
 *   seam 4 (process)   code_producers/src/c_elements/common/main.cpp:22-124 loadCircuit,
 *                      :243-286 loadJson, :288-334 writeBinWitness, :336-373 main
 *
 * Differences forced by batching: one `cw_batch` holds B independent instances (the reference runs one
 * process per input); a failed `===`/assert does not abort (assert_bucket.rs:75-77) but is reported in
 * the per-instance status word; everything is plain pointers + sizes, caller owns host buffers, the
 * library owns device memory.  All functions return 0 on success or a negative CW_E* code;
 * cw_last_error() gives the message (thread-local).  No function is thread-safe on the same handle.
 */
#ifndef CIRCOM_AMD_H
#define CIRCOM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CW_OK 0
#define CW_EIO (-1)        /* file missing / unreadable / malformed */
#define CW_EINVAL (-2)     /* bad argument */
#define CW_EINPUT (-3)     /* input error: signal not found / assigned twice / wrong count (calcwit.cpp:51-97) */
#define CW_EDEVICE (-4)    /* HIP error */
#define CW_ESTATE (-5)     /* call out of order (e.g. run before all inputs are set) */

typedef struct cw_circuit cw_circuit; /* replaces Circom_Circuit (circom.hpp:36-43) + compiled-in tables */
typedef struct cw_batch cw_batch;     /* replaces Circom_CalcWit (calcwit.hpp:17-66), x B instances */

/* instance status word (cw_get_status) */
#define CW_ST_OK 0u
#define CW_ST_ASSERT_FAILED 1u /* a `===` / assert() did not hold */
#define CW_ST_ARITH 2u         /* `\` or `%` by zero (reference: GMP abort, generic/fr.cpp:2835-2875), run-away function */
/* bits 8..31 of a word with bit 0 or 1 set: index of the failing operation in the circuit's flat witness program.  When
   several checks of an instance fail, the one with the smallest index is reported whatever the schedule's row order and
   strand count - the check the reference's sequential program stops at (assert_bucket.rs:75-77, calcwit.cpp:104-114). */
#define CW_ST_R1CS_FAILED 4u   /* set by cw_check_r1cs */

const char *cw_last_error(void);
const char *cw_version(void);

/* ---- circuit (loadCircuit, main.cpp:22-124; the nine get_*() of circom.hpp:79-87) -------------- */
/* tape_path: <name>.cwt (hip_elements schedule); dat_path: <name>.dat (reference layout, hash map +
 * witness list are read from it; may be NULL -> taken from the tape); r1cs_path: <name>.r1cs or NULL. */
int cw_load(const char *tape_path, const char *dat_path, const char *r1cs_path, cw_circuit **out);
void cw_free(cw_circuit *c);
uint32_t cw_n_signals(const cw_circuit *c);          /* get_total_signal_no() */
/* get_size_of_io_map(): template instances of Mixed component clusters described in the .dat's io-map section
 * (c_code_generator.rs:681-738; Circom_Circuit::templateInsId2IOSignalInfo, circom.hpp:41), and the local offset of io
 * signal `signal_code` of one of them (-1: unknown).  The evaluator resolves every access at trace time; the table is
 * read, validated and exposed because the file format carries it. */
uint32_t cw_io_map_size(const cw_circuit *c);
int64_t cw_io_map_offset(const cw_circuit *c, uint32_t template_id, uint32_t signal_code);
/* get_size_of_bus_field_map(): bus instances described in the .dat's bus-field map (c_code_generator.rs:740-794;
 * Circom_Circuit::busInsId2FieldInfo, circom.hpp:42; reader main.cpp:95-121), and one field of one of them: its offset inside
 * the bus, the size of one element, the id of the field's own bus (0: a signal, as the reference writes it) and the number of
 * dimensions behind the first.  Read, validated and exposed because the file format carries it (CW_EINVAL: no such field). */
uint32_t cw_bus_map_size(const cw_circuit *c);
int cw_bus_field(const cw_circuit *c, uint32_t bus_id, uint32_t field, uint32_t *offset, uint32_t *size, uint32_t *field_bus_id,
                 uint32_t *n_lengths);
uint32_t cw_n_witness(const cw_circuit *c);          /* get_size_of_witness() */
uint32_t cw_n_inputs(const cw_circuit *c);           /* get_main_input_signal_no() */
/* The witness of a SIMPLIFIED constraint system (the reference's default --O1, constraint_list/src/constraint_simplification.rs,
 * keeps a subset of the signals; the emitted calculator writes those: witness2signal of the .dat, calcwit.hpp:54-56,
 * main.cpp:288-334).  `signals`: n strictly increasing signal ids starting with 0; the first 1 + cw_n_public entries stay as
 * they are.  Evaluation and R1CS check keep working on the full system cw_load was given; every egress of batches created
 * AFTER this call (cw_get_witness(es)(_device), cw_write_wtns(_many), cw_write_wtnsb) hands out these entries, and
 * cw_n_witness() returns n.  CW_EINVAL while any batch of the circuit is alive: a batch sizes its device images of the list
 * when it is created, so the list is fixed before the first cw_batch_create (or after the last cw_batch_free). */
int cw_set_witness_list(cw_circuit *c, const uint32_t *signals, uint32_t n);
/* 1 unless the .r1cs given to cw_load is NOT the constraint system the tape's emitted checks were built from (the tape records
 * CRC-32 and length of that file's constraint section): then the fused checks / covered rows / audit program are not used and
 * the stand-alone kernels check every row of the file that was loaded.  (No reference counterpart: its runtime never checks.) */
int cw_emitted_checks_match(const cw_circuit *c);
uint32_t cw_input_start(const cw_circuit *c);        /* get_main_input_signal_start() */
uint32_t cw_n_constraints(const cw_circuit *c);      /* from the .r1cs header, 0 if none loaded */
uint32_t cw_n_public(const cw_circuit *c);     /* nPubOut + nPubIn of the r1cs header */
uint64_t cw_n_rows(const cw_circuit *c);             /* schedule length */
uint64_t cw_n_mmul(const cw_circuit *c);             /* Montgomery multiplications per instance in the schedule */
void cw_prime(const cw_circuit *c, uint8_t le32[32]); /* Fr_q (fr.hpp) */
/* getInputSignalSize(h) (calcwit.cpp:99-102): size of input `name`, or -1 */
int64_t cw_input_size(const cw_circuit *c, const char *name, uint32_t *start_slot);

/* ---- batch (Circom_CalcWit ctor calcwit.cpp:26-45) -------------------------------------------------- */
/* device: HIP device ordinal; batch: number of instances; stream: a hipStream_t (as void*) all work of
 * this batch is enqueued on, NULL = the null stream.  device < 0 creates a host-only batch: inputs can be
 * staged and validated, every computing call fails with CW_EDEVICE (there is no CPU fallback). */
int cw_batch_create(cw_circuit *c, int device, uint32_t batch, void *stream, cw_batch **out);
void cw_batch_free(cw_batch *b);
uint32_t cw_batch_size(const cw_batch *b);
/* strands (waves cooperating on the same 64 instances) of the schedule variant picked for this batch */
uint32_t cw_batch_strands(const cw_batch *b);
/* 0, or (rows per batch | loads per batch << 8) when the batch runs the pipelined single-wave variant of the schedule
   (LDS result ring + load lists issued one batch ahead: hip_elements/pipe.py; CW_PIPE=0/1 overrides the choice) */
uint32_t cw_batch_pipelined(const cw_batch *b);
/* 1 when the rows of the picked variant run as EMITTED gfx950 code (hip_elements/fpjit.py: the counterpart of the reference's
   per-template C++, compiler/src/circuit_design/template.rs:174-474 - operand loads, a call of the operator's body and the
   stores of every row as straight-line code), 2 when that code also carries the R1CS check of the rows it covers (recomputed
   from the stored wires behind the rows that produce them; chosen for throughput-bound batches, CW_FP_FUSED=0/1 overrides;
   cw_check_r1cs then merges its findings and streams only the remaining rows), 0 when cw_eval_kernel interprets the rows.
   CW_FP_JIT=0 forces the interpreter. */
uint32_t cw_batch_emitted(const cw_batch *b);
/* instances per workgroup (64, 32 or 16) the evaluation kernel uses for this batch */
uint32_t cw_batch_lanes(const cw_batch *b);
/* 1 if this batch runs the bit-plane program (circuits whose signals are all boolean for 0/1 inputs: one bit per
 * signal per instance, replaces the short-int paths of the tagged FrElement, generic/fr.cpp:416-439,900-917);
 * instances whose inputs are not 0/1 are transparently re-run by the 256-bit schedule.  CW_BITS=0 disables it. */
int cw_batch_bitmode(const cw_batch *b);
/* shape of the circuit's bit-plane program: out = {present, vrows, bit-table slots per group of 64 instances, LDS ring
 * rows, gate lanes, row loads, row flushes, LDS cache rows} (all zero when the circuit has none) */
int cw_bits_info(const cw_circuit *c, uint64_t out[8]);
/* host-only: shape of the R1CS check plan over the bit table: out = {rows that hold by construction, LUT-class rows,
 * integer-class rows, terms read as whole 32-bit words, blocks of 8 single-bit terms, of which on 8 consecutive slots,
 * field-class rows, words of the integer-class stream} */
int cw_bits_r1cs_plan_stats(const cw_circuit *c, uint64_t out[8]);

/* setInputSignal(h, i, val) (calcwit.cpp:77-97) for one instance; `name` is hashed with FNV-1a
 * (calcwit.cpp:17-24).  val = canonical 32-byte little-endian value, reduced mod q by the caller. */
int cw_set_input_signal(cw_batch *b, uint32_t instance, const char *name, uint32_t idx, const uint8_t val[32]);
/* loadJson (main.cpp:243-286, value grammar json2FrElements :144-188, nesting qualify_input :221-241)
 * for one instance from JSON text. */
int cw_set_inputs_json(cw_batch *b, uint32_t instance, const char *json_text);
/* Bulk: all instances at once, [batch][n_inputs][32] canonical LE values in main-input slot order
 * (slot = cw_input_start() + k).  Marks every input of every instance as set. */
int cw_set_inputs(cw_batch *b, const uint8_t *le32);
/* Same, but `d_le32` is a DEVICE pointer (HBM-resident inputs; no host copy).  The buffer is read by cw_run (stream
 * order) and, in bit-plane batches, once more by the first cw_sync / getter after it - instances whose inputs are not 0/1
 * are re-run by the 256-bit schedule from these bytes - so it must stay unmodified until then. */
int cw_set_inputs_device(cw_batch *b, const void *d_le32);
/* Bit-plane batches (cw_batch_bitmode) only: PACKED boolean inputs, uint64 masks[groups][n_inputs], groups =
 * ceil(batch / 64), bit i of masks[g][k] = main input k of instance 64 g + i.  The reference reads one JSON number per
 * bit (main.cpp:243-286); the 32-byte-per-value bulk form moves 256 bytes per input BIT of a SHA-256 circuit, this form
 * one bit.  cw_set_inputs_bits copies from host memory; the _device form keeps reading the caller's device buffer
 * (same lifetime rule as cw_set_inputs_device).  CW_ESTATE for batches on the 256-bit schedule. */
int cw_set_inputs_bits(cw_batch *b, const uint64_t *masks);
int cw_set_inputs_bits_device(cw_batch *b, const void *d_masks);
/* read back input k (slot cw_input_start()+k) of one instance as staged by the two per-signal setters */
int cw_get_staged_input(cw_batch *b, uint32_t instance, uint32_t k, uint8_t out[32]);
/* getRemaingInputsToBeSet() (calcwit.hpp:50-52) of one instance */
int64_t cw_remaining_inputs(const cw_batch *b, uint32_t instance);

/* run(ctx) (calcwit.cpp:6,71-75) for all instances: ingest + schedule evaluation, asynchronous on the
 * batch's stream.  Fails with CW_ESTATE if some instance still has unset inputs (main.cpp:352-355). */
int cw_run(cw_batch *b);
/* A*w o B*w = C*w over the loaded .r1cs for all instances (second kernel of the north-star). */
int cw_check_r1cs(cw_batch *b);
/* cw_run + cw_check_r1cs as one launch: the second call with unchanged input pointers captures the launches of both into a HIP graph
 * on the batch's stream, later calls replay it (a step of a small batch is a dozen kernel launches of microseconds each - the host
 * sets the pace otherwise).  Same results, same errors, same asynchrony as the two calls; falls back to them while timing marks are
 * on, while host-set inputs await their copy, with CW_NO_GRAPH set, or if the capture fails.  The graph is dropped when the
 * batch's input pointer changes or the caller takes the raw table pointer.  cw_batch_graph_captured: 1 once replaying.
 * (No reference counterpart: the reference's run + snarkjs' `wtns check` are two processes per witness.) */
int cw_run_check(cw_batch *b);
int cw_batch_graph_captured(const cw_batch *b);
/* Event timing of the parts of cw_run / cw_check_r1cs on the batch's stream (HIP events recorded where the kernels are launched;
 * no reference counterpart: the reference's runtime is one process per witness - this is what bench.py's roofline figures read).
 * cw_batch_kernel_ms drains the stream and returns, for the LAST run / check: ms[0] = table init + input ingest, ms[1] = the
 * evaluation kernel(s), ms[2] = the R1CS check; -1 for a part that has not run since timing was switched on. */
int cw_batch_set_timing(cw_batch *b, int on);
int cw_batch_kernel_ms(cw_batch *b, float ms[3]);
/* on = 2 keeps the marks of the last 64 runs of the batch; cw_batch_kernel_ms_mean averages each part over every run recorded
 * since (counts[k] = how many): the kernels' durations INSIDE a timed region, other batches in flight beside them - the figure
 * a rocprofv3 kernel-trace average of the same region reproduces. */
int cw_batch_kernel_ms_mean(cw_batch *b, float ms[3], int counts[3]);
int cw_sync(cw_batch *b);

/* results (synchronise the stream first) */
int cw_get_status(cw_batch *b, uint32_t *status /* [batch] */);
/* getWitness(i) + Fr_toLongNormal for all witness positions (main.cpp:326-332): [n_witness][32] */
int cw_get_witness(cw_batch *b, uint32_t instance, uint8_t *out);
/* bulk form for provers: `count` instances from `first`, [count][n_witness][32], one device-side transpose */
int cw_get_witnesses(cw_batch *b, uint32_t first, uint32_t count, uint8_t *out);
/* the same to DEVICE memory (no host copy): what a GPU prover consumes.  Asynchronous on the batch's stream, except that
 * the first egress after a cw_run of a bit-plane batch waits for the evaluation (it has to know which instances were
 * re-run by the 256-bit schedule; every egress serves those from there). */
int cw_get_witnesses_device(cw_batch *b, uint32_t first, uint32_t count, void *d_out);
/* chunked form: `chunk` instances at a time into d_buf0 / d_buf1 in turn ([chunk][n_witness][32] each); `consume` is called
 * after each chunk's transpose has been enqueued on the batch's stream (passed as `stream`, a hipStream_t): work enqueued
 * there sees the chunk complete and orders the library's next write to that buffer behind itself.  Non-zero return of
 * `consume` aborts with CW_ESTATE.  (writeBinWitness's loop over getWitness(i), main.cpp:326-332, for a consumer that
 * cannot hold B x 32 MB.)  Size the chunk generously where memory allows: on the bit-plane path a launch of 8 or more groups of
 * 64 instances writes faster (4.8 -> 5.2-6.3 TB/s on the 156 809-wire --O1 witness of the metric circuit). */
typedef int (*cw_chunk_fn)(void *user, uint32_t first, uint32_t count, void *d_chunk, void *stream);
int cw_stream_witnesses_device(cw_batch *b, uint32_t first, uint32_t count, uint32_t chunk, void *d_buf0, void *d_buf1,
                               cw_chunk_fn consume, void *user);
/* public signals (main's outputs, then its public inputs = witness positions 1..cw_n_public) of EVERY instance,
 * [batch][n_public][32]: to host memory, or to device memory for a multi-GPU gather */
int cw_get_public(cw_batch *b, uint8_t *out);
int cw_get_public_device(cw_batch *b, void *d_out);
/* one signal of one instance (signalValues[slot]) */
int cw_get_signal(cw_batch *b, uint32_t instance, uint32_t slot, uint8_t out[32]);
/* writeBinWitness (main.cpp:288-334) */
int cw_write_wtns(cw_batch *b, uint32_t instance, const char *path);
/* `count` .wtns files from one bulk device transpose; `pattern` = printf pattern with one %u (instance number) */
int cw_write_wtns_many(cw_batch *b, uint32_t first, uint32_t count, const char *pattern);
/* The whole batch as ONE compact container (<name>.wtnsb): field elements for 256-bit batches, the BIT TABLE (1 bit per
 * distinct signal value and instance + the slot of every witness element + full values of the instances the 256-bit schedule
 * re-ran) for bit-plane batches.  Format: csrc/cw_host.cpp at cw_write_wtnsb; reader / expander circom_amd/wtnsb.py
 * (`expand(i)` = the bytes cw_write_wtns writes for instance i, main.cpp:288-334). */
int cw_write_wtnsb(cw_batch *b, const char *path);
/* human-readable trace of one instance into out[out_len]: decoded status word and, for a violated constraint, its index
 * and every wire with its name from <name>.sym (may be NULL) and value — the batch counterpart of the trace the
 * reference prints before aborting (c_code_generator.rs:461-468, calcwit.cpp:104-114) */
int cw_explain(cw_batch *b, uint32_t instance, const char *sym_path, char *out, size_t out_len);
/* log(...) statements of the circuit (LogBucket, compiler/src/intermediate_representation/log_bucket.rs:105-162: the
 * emitted calculator prints every argument with printf - values through Fr_element2str - one blank between arguments
 * and a newline per statement).  A batched run has no console per instance: the schedule keeps every logged value in
 * the table (hidden signals behind the circuit's own; cw_n_signals does not count them) and cw_get_log formats what the
 * reference binary prints on stdout for ONE instance, up to its first failed run-time check (where the reference process
 * exits).  Returns the length of the text or a negative CW_E* code; stores at most out_len - 1 characters + NUL. */
uint32_t cw_n_log_statements(const cw_circuit *c);
int64_t cw_get_log(cw_batch *b, uint32_t instance, char *out, size_t out_len);
/* first violated constraint per instance after cw_check_r1cs: [batch], 0xFFFFFFFF = none */
int cw_get_r1cs_first_bad(cw_batch *b, uint32_t *row);
/* host-only: build and hazard-check the LDS staging plan of the R1CS check kernel for `chunks` row chunks
   and `entries` 2-KiB LDS entries per wave (0 = the defaults cw_batch_create would pick for `batch`).
   out[8] = {chunks, loads, terms, filler loads, distinct wires, entries, prefetch depth, 0}. */
int cw_r1cs_plan_stats(const cw_circuit *c, uint32_t batch, uint32_t chunks, uint32_t entries, uint64_t out[8]);
/* host-only: the term stream of the default check kernel (csrc/cw_r1cs_plan.h build_stream), as cw_batch_create uploads it, so
   that a test can replay it on the CPU.  flags: 1 = boolean rows b (b - 1) = 0 stay three terms and a product, 2 = boolean rows
   do not ride inside the sum that reads the same bit.  sizes[8] = {words of chunk[], words of terms[], words of row_orig[],
   words of ctab[], terms, boolean rows folded into another row, products replaced by a select, chunks}; a null buffer is not
   written (call once for the sizes).  No reference counterpart (snarkjs `wtns check` is the reference-side check). */
int cw_r1cs_stream_plan(const cw_circuit *c, uint32_t terms_per_chunk, uint32_t flags, uint64_t sizes[8], uint32_t *chunk,
                        uint32_t *terms, uint32_t *row_orig, uint32_t *ctab);

/* raw device pointers for zero-copy consumers (provers): value table, layout in DESIGN.md.  When cw_circuit_montgomery()
   is 1 the table holds x * 2^261 mod q (the schedule of an arithmetic circuit keeps its signals in Montgomery form, so that
   a product of two signals is one Montgomery product); every other egress (cw_get_witness*, cw_get_signal, cw_write_wtns*,
   cw_get_witnesses_device) returns canonical values, as the reference's Fr_toLongNormal does (main.cpp:326-332). */
void *cw_device_values(cw_batch *b, uint64_t *n_bytes, uint32_t *padded_batch);
int cw_circuit_montgomery(const cw_circuit *c);
/* bit-plane batches: the bit table T[group][slot] (uint64, bit i = instance group*64+i).  Slot 0 / 1 = the constants
 * 0 / 1, main input k = slot 3 + k; every other signal sits in the slot cw_signal_slots() names (signals that are copies
 * of one another share a slot; slots follow the order in which the program produces the values, 32-bit words of
 * consecutive signals keep 32 consecutive slots).  NULL for 256-bit batches (cw_device_values is NULL for bit-plane ones). */
void *cw_device_bits(cw_batch *b, uint64_t *n_bytes, uint64_t *slots_per_group);
/* signal -> bit-table slot, n_signals entries (host memory owned by the circuit), NULL when there is no bit program */
const uint32_t *cw_signal_slots(const cw_circuit *c);
/* A circuit may carry its bit-plane program twice: interpreted (cw_bits_eval_kernel, T[group][slot]) and as EMITTED gfx950
 * code (hip_elements/bitjit.py: the counterpart of the reference's per-circuit <name>.cpp, compiler/src/circuit_design/
 * template.rs:174-474), which batches of >= 2^18 instances run (CW_BITS_JIT=0/1 overrides).  The emitted code has its
 * own table layout and slot map, so a caller of cw_device_bits asks the BATCH:
 *   out = {slots per group (sh = 0) or rows per chunk (sh = 5), sh, groups allocated, 1 if the emitted code runs}
 *   element (group g, slot s) = T[(((g >> sh) * slots + s) << sh) + (g & ((1 << sh) - 1))]   (uint64, bit i = instance 64 g + i)
 * cw_batch_signal_slots: signal -> slot for THIS batch (cw_signal_slots names the interpreter's map).
 * Taking the raw pointer (cw_device_bits) makes the next cw_check_r1cs audit every group from the table instead of
 * trusting the check the emitted code fused into the evaluation. */
int cw_batch_bits_layout(const cw_batch *b, uint64_t out[4]);
const uint32_t *cw_batch_signal_slots(const cw_batch *b);

/* ---- field micro-benchmark + unit-test hooks (Fr_* seam 2: bn128/fr.hpp:28-81) --------------------- */
/* n lanes x iters dependent Montgomery multiplications on the device; out[i] = a[i]*b[i]^iters (raw
 * Montgomery domain).  a,b,out: host [n][32].  ms: kernel time from HIP events. */
int cw_fp_mul_bench(const uint8_t prime_le32[32], int device, uint32_t n, uint32_t iters, const uint8_t *a,
                    const uint8_t *b, uint8_t *out, float *ms);
/* Time the bit-plane evaluation kernel on an arbitrary program (records / command blocks as in the .cwt, validated like
 * a loaded one) for n_groups groups of 64 instances on a zero-filled table: average ms over `iters` launches.  A
 * measurement hook (tools/bits_shape_bench.py), not part of the witness path. */
int cw_bits_eval_bench(int device, uint32_t ring, uint32_t cache, uint32_t n_vrows, uint64_t n_slots, const uint32_t *recs,
                       const uint32_t *cmds, uint32_t n_groups, uint32_t width, uint32_t iters, float *ms);
/* Element-wise device evaluation of one schedule opcode (D_* numbering of cw_tape.h) on n operand
 * pairs — the unit-test hook for the device field functions.  status[n] receives CW_ST_* bits. */
int cw_fp_op(const uint8_t prime_le32[32], int device, uint32_t dop, uint32_t n, const uint8_t *a, const uint8_t *b,
             const uint8_t *c, uint8_t *out, uint32_t *status);

#ifdef __cplusplus
}
#endif
#endif /* CIRCOM_AMD_H */
