"""ctypes binding of the C ABI (include/circom_amd.h -> circom_amd/lib/libcircom_amd.so).

Host-side mirror of the reference's runtime objects: `Circuit` ~ Circom_Circuit (circom.hpp:36-43),
`Batch` ~ Circom_CalcWit (calcwit.hpp:17-66) for B instances.  There is no CPU fallback: if the HIP
library is missing or no GPU is present, computing calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["CW_LIB"]).resolve() if os.environ.get("CW_LIB") else _HERE / "lib" / "libcircom_amd.so"   # CW_LIB: diagnostics builds

ST_OK, ST_ASSERT_FAILED, ST_ARITH, ST_R1CS_FAILED = 0, 1, 2, 4


class CwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("circom_amd error %d: %s" % (code, msg))
        self.code = code


CLI_PATH = _HERE / "bin" / "cw_witness"     # process-level drop-in (csrc/cw_cli.cpp), built with the library


def build_library(force: bool = False) -> Path:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [_HERE / "csrc" / n for n in ("cw_kernels.hip", "cw_bits.hip", "cw64.hip", "cw_host.cpp", "cw_cli.cpp", "cw_kernels.h",
                                         "cw_tape.h", "cw_r1cs_plan.h", "cw_bits_host.h", "fp256.hip.h", "cw_rowops.hip.h", "cw_call.hip.h")]
    outs = [LIB_PATH, CLI_PATH]
    if not force and all(o.exists() and all(o.stat().st_mtime >= s.stat().st_mtime for s in srcs) for o in outs):
        return LIB_PATH
    subprocess.run(["make", "-j4", "-C", str(_HERE / "csrc")], check=True, capture_output=True)
    return LIB_PATH


_lib = None

_SIGS = {
    "cw_last_error": (C.c_char_p, []),
    "cw_version": (C.c_char_p, []),
    "cw_load": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "cw_free": (None, [C.c_void_p]),
    "cw_n_signals": (C.c_uint32, [C.c_void_p]),
    "cw_io_map_size": (C.c_uint32, [C.c_void_p]),
    "cw_io_map_offset": (C.c_int64, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "cw_bus_map_size": (C.c_uint32, [C.c_void_p]),
    "cw_bus_field": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32] + [C.POINTER(C.c_uint32)] * 4),
    "cw_n_witness": (C.c_uint32, [C.c_void_p]),
    "cw_n_inputs": (C.c_uint32, [C.c_void_p]),
    "cw_set_witness_list": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]),
    "cw_input_start": (C.c_uint32, [C.c_void_p]),
    "cw_n_constraints": (C.c_uint32, [C.c_void_p]),
    "cw_n_public": (C.c_uint32, [C.c_void_p]),
    "cw_n_rows": (C.c_uint64, [C.c_void_p]),
    "cw_n_mmul": (C.c_uint64, [C.c_void_p]),
    "cw_prime": (None, [C.c_void_p, C.c_char_p]),
    "cw_input_size": (C.c_int64, [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32)]),
    "cw_batch_create": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "cw_batch_free": (None, [C.c_void_p]),
    "cw_batch_size": (C.c_uint32, [C.c_void_p]),
    "cw_batch_strands": (C.c_uint32, [C.c_void_p]),
    "cw_circuit_montgomery": (C.c_int, [C.c_void_p]),
    "cw_batch_pipelined": (C.c_uint32, [C.c_void_p]),
    "cw_batch_emitted": (C.c_uint32, [C.c_void_p]),
    "cw_batch_lanes": (C.c_uint32, [C.c_void_p]),
    "cw_batch_bitmode": (C.c_int, [C.c_void_p]),
    "cw_bits_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cw_bits_r1cs_plan_stats": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cw_device_bits": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cw_set_input_signal": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p]),
    "cw_set_inputs_json": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p]),
    "cw_set_inputs": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cw_set_inputs_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cw_set_inputs_bits": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cw_set_inputs_bits_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cw_stream_witnesses_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cw_signal_slots": (C.POINTER(C.c_uint32), [C.c_void_p]),
    "cw_batch_signal_slots": (C.POINTER(C.c_uint32), [C.c_void_p]),
    "cw_batch_bits_layout": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cw_get_staged_input": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p]),
    "cw_remaining_inputs": (C.c_int64, [C.c_void_p, C.c_uint32]),
    "cw_run": (C.c_int, [C.c_void_p]),
    "cw_check_r1cs": (C.c_int, [C.c_void_p]),
    "cw_run_check": (C.c_int, [C.c_void_p]),
    "cw_batch_graph_captured": (C.c_int, [C.c_void_p]),
    "cw_emitted_checks_match": (C.c_int, [C.c_void_p]),
    "cw_batch_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "cw_batch_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "cw_batch_kernel_ms_mean": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "cw_sync": (C.c_int, [C.c_void_p]),
    "cw_get_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cw_get_witness": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "cw_get_witnesses": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "cw_get_witnesses_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "cw_get_public": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cw_get_public_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cw_get_signal": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p]),
    "cw_write_wtns": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p]),
    "cw_get_r1cs_first_bad": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cw_write_wtns_many": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p]),
    "cw_write_wtnsb": (C.c_int, [C.c_void_p, C.c_char_p]),
    "cw_explain": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p, C.c_size_t]),
    "cw_n_log_statements": (C.c_uint32, [C.c_void_p]),
    "cw_get_log": (C.c_int64, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]),
    "cw_r1cs_plan_stats": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "cw_r1cs_stream_plan": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cw_device_values": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "cw_fp_mul_bench": (C.c_int, [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.POINTER(C.c_float)]),
    "cw_bits_eval_bench": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32,
                                     C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]),
    "cw_fp_op": (C.c_int, [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_void_p]),
}


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise CwError(-4, "HIP extension %s is not built (run __graft_entry__.build()); "
                              "there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise CwError(rc, lib().cw_last_error().decode())


def fe_to_bytes(values, n=None) -> bytes:
    return b"".join(int(v).to_bytes(32, "little") for v in values)


def bytes_to_ints(buf: bytes):
    return [int.from_bytes(buf[i:i + 32], "little") for i in range(0, len(buf), 32)]


class Circuit:
    def __init__(self, tape_path, dat_path=None, r1cs_path=None):
        h = C.c_void_p()
        enc = lambda p: None if p is None else os.fsencode(str(p))
        _chk(lib().cw_load(enc(tape_path), enc(dat_path), enc(r1cs_path), C.byref(h)))
        self.h = h
        L = lib()
        self.n_signals = L.cw_n_signals(h)
        self.emitted_checks_match = bool(L.cw_emitted_checks_match(h))    # False: the .r1cs is not the one the emitted checks were built from
        self.n_witness = L.cw_n_witness(h)
        self.n_inputs = L.cw_n_inputs(h)
        self.input_start = L.cw_input_start(h)
        self.n_constraints = L.cw_n_constraints(h)
        self.n_public = L.cw_n_public(h)
        self.n_rows = L.cw_n_rows(h)
        self.n_mmul = L.cw_n_mmul(h)
        self.montgomery = bool(L.cw_circuit_montgomery(h))     # the device value table holds x * 2^261 mod q
        buf = C.create_string_buffer(32)
        L.cw_prime(h, buf)
        self.q = int.from_bytes(buf.raw, "little")

    def set_witness_list(self, signals):
        """egress of every batch created from now on hands out these signals (the witness of a simplified system: `--O1`,
        frontend/circom_simplify.py); evaluation and R1CS check stay on the full system"""
        arr = np.ascontiguousarray(signals, dtype=np.uint32)
        _chk(lib().cw_set_witness_list(self.h, arr.ctypes.data_as(C.POINTER(C.c_uint32)), len(arr)))
        self.n_witness = lib().cw_n_witness(self.h)

    def bits_info(self) -> dict:
        """shape of the bit-plane program ({} when the circuit has none)"""
        out = (C.c_uint64 * 8)()
        _chk(lib().cw_bits_info(self.h, out))
        if not out[0]:
            return {}
        return dict(zip(("vrows", "slots_per_group", "ring", "gate_lanes", "row_loads", "row_flushes", "cache"), [int(x) for x in out[1:8]]))

    def bits_r1cs_plan_stats(self) -> dict:
        out = (C.c_uint64 * 8)()
        _chk(lib().cw_bits_r1cs_plan_stats(self.h, out))
        return dict(zip(("trivial_rows", "lut_rows", "int_rows", "word_terms", "bit_blocks", "contiguous_blocks", "field_rows", "int_stream_words"),
                        [int(x) for x in out]))

    def input_size(self, name: str):
        start = C.c_uint32()
        n = lib().cw_input_size(self.h, name.encode(), C.byref(start))
        return (None, None) if n < 0 else (start.value, n)

    def close(self):
        if self.h:
            lib().cw_free(self.h)
            self.h = None

    def r1cs_plan_stats(self, batch: int = 65536, chunks: int = 0, entries: int = 0) -> dict:
        """Build + hazard-check the R1CS kernel's LDS staging plan on the host (no GPU needed)."""
        out = (C.c_uint64 * 8)()
        _chk(lib().cw_r1cs_plan_stats(self.h, batch, chunks, entries, out))
        keys = ("chunks", "loads", "terms", "filler_loads", "distinct_wires", "entries", "depth")
        return dict(zip(keys, [int(x) for x in out]))

    def r1cs_stream_plan(self, terms_per_chunk: int = 192, no_bool: bool = False, no_fold: bool = False) -> dict:
        """The term stream of the default R1CS check kernel, as the batch uploads it (host only; csrc/cw_r1cs_plan.h)."""
        import numpy as np
        flags = (1 if no_bool else 0) | (2 if no_fold else 0)
        sizes = (C.c_uint64 * 8)()
        _chk(lib().cw_r1cs_stream_plan(self.h, terms_per_chunk, flags, sizes, None, None, None, None))
        arrs = [np.zeros(int(sizes[k]), dtype=np.uint32) for k in range(4)]
        _chk(lib().cw_r1cs_stream_plan(self.h, terms_per_chunk, flags, sizes, *[a.ctypes.data_as(C.c_void_p) for a in arrs]))
        return {"chunk": arrs[0].reshape(-1, 4), "terms": arrs[1].reshape(-1, 2), "row_orig": arrs[2], "ctab": arrs[3].reshape(-1, 8),
                "n_terms": int(sizes[4]), "n_folded": int(sizes[5]), "n_bitsel": int(sizes[6]), "n_chunks": int(sizes[7])}

    def batch(self, batch: int, device: int = 0, stream=None) -> "Batch":
        return Batch(self, batch, device, stream)


class Batch:
    def __init__(self, circuit: Circuit, batch: int, device: int = 0, stream=None):
        self.circuit = circuit
        self.n = batch
        h = C.c_void_p()
        _chk(lib().cw_batch_create(circuit.h, device, batch, C.c_void_p(stream or 0), C.byref(h)))
        self.h = h
        self.strands = lib().cw_batch_strands(h)
        pp = lib().cw_batch_pipelined(h)
        self.pipelined = (pp & 0xFF, pp >> 8) if pp else None     # (rows per batch, loads per batch) of the pipelined variant
        self.lanes = lib().cw_batch_lanes(h)
        em = lib().cw_batch_emitted(h)
        self.emitted = bool(em)              # the variant's rows run as emitted code (hip_elements/fpjit.py)
        self.fused_check = em == 2           # ... which also recomputes the R1CS rows it covers
        self.bitmode = bool(lib().cw_batch_bitmode(h))
        lay = (C.c_uint64 * 4)()
        _chk(lib().cw_batch_bits_layout(h, lay))
        # bit table of this batch: element (group g, slot s) = T[(((g >> sh) * slots + s) << sh) + (g & ((1 << sh) - 1))]
        self.bits_slots, self.bits_sh, self.bits_groups, self.jit = int(lay[0]), int(lay[1]), int(lay[2]), bool(lay[3])

    def bits_index(self, group: int, slot: int) -> int:
        """uint64 index of (group, slot) in the table behind cw_device_bits"""
        sh = self.bits_sh
        return (((group >> sh) * self.bits_slots + slot) << sh) + (group & ((1 << sh) - 1))

    def device_bits(self):
        """(device pointer, bytes, slots per group) of the bit table (cw_device_bits); taking it makes the next check_r1cs audit
        the whole table - the caller may have changed it"""
        nb, spg = C.c_uint64(), C.c_uint64()
        p = lib().cw_device_bits(self.h, C.byref(nb), C.byref(spg))
        return (int(p) if p else 0), int(nb.value), int(spg.value)

    def signal_slots(self) -> np.ndarray:
        p = lib().cw_batch_signal_slots(self.h)
        return np.ctypeslib.as_array(p, shape=(self.circuit.n_signals,)).copy() if p else None

    def close(self):
        if self.h:
            lib().cw_batch_free(self.h)
            self.h = None

    # -- inputs -------------------------------------------------------------------------------------
    def set_input_signal(self, instance: int, name: str, idx: int, value: int):
        _chk(lib().cw_set_input_signal(self.h, instance, name.encode(), idx, int(value).to_bytes(32, "little")))

    def set_inputs_json(self, instance: int, text: str):
        _chk(lib().cw_set_inputs_json(self.h, instance, text.encode()))

    def set_inputs(self, arr):
        """arr: uint8 array [batch, n_inputs, 32] (canonical LE) or a list of lists of ints."""
        if not isinstance(arr, np.ndarray):
            arr = np.frombuffer(b"".join(fe_to_bytes(row) for row in arr), dtype=np.uint8)
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        assert arr.size == self.n * self.circuit.n_inputs * 32, "input array has the wrong size"
        _chk(lib().cw_set_inputs(self.h, arr.ctypes.data_as(C.c_void_p)))

    def set_inputs_bits(self, masks):
        """packed boolean inputs: uint64 [groups][n_inputs], bit i of masks[g][k] = input k of instance 64 g + i"""
        arr = np.ascontiguousarray(masks, dtype=np.uint64)
        assert arr.shape == ((self.n + 63) // 64, self.circuit.n_inputs), arr.shape
        _chk(lib().cw_set_inputs_bits(self.h, arr.ctypes.data_as(C.c_void_p)))

    def set_inputs_bits_device(self, dptr: int):
        _chk(lib().cw_set_inputs_bits_device(self.h, C.c_void_p(dptr)))

    def stream_witnesses_device(self, first: int, count: int, chunk: int, d_buf0: int, d_buf1: int, consume):
        """consume(first, count, d_chunk, stream) -> int, called once per chunk (cw_stream_witnesses_device)"""
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p)
        cb = CB(lambda user, f, n, ptr, st: int(consume(f, n, ptr, st) or 0))
        _chk(lib().cw_stream_witnesses_device(self.h, first, count, chunk, C.c_void_p(d_buf0), C.c_void_p(d_buf1), cb, None))

    def set_inputs_device(self, dptr: int):
        _chk(lib().cw_set_inputs_device(self.h, C.c_void_p(dptr)))

    def staged_input(self, instance: int, k: int) -> int:
        buf = C.create_string_buffer(32)
        _chk(lib().cw_get_staged_input(self.h, instance, k, buf))
        return int.from_bytes(buf.raw, "little")

    def remaining_inputs(self, instance: int) -> int:
        return lib().cw_remaining_inputs(self.h, instance)

    # -- compute ------------------------------------------------------------------------------------
    def run(self):
        _chk(lib().cw_run(self.h))

    def check_r1cs(self):
        _chk(lib().cw_check_r1cs(self.h))

    def run_check(self):
        """run() + check_r1cs() as one launch: captured as a HIP graph on the second call, replayed afterwards (cw_run_check)"""
        _chk(lib().cw_run_check(self.h))

    @property
    def graph_captured(self) -> bool:
        return bool(lib().cw_batch_graph_captured(self.h))

    def sync(self):
        _chk(lib().cw_sync(self.h))

    def set_timing(self, on=True):
        """HIP events around the parts of run() / check_r1cs() on this batch's stream (cw_batch_set_timing)"""
        _chk(lib().cw_batch_set_timing(self.h, 2 if on == "history" else 1 if on else 0))

    def kernel_ms(self):
        """{"ingest": ms, "eval": ms, "check": ms} of the last run / check (None: that part has not run); drains the stream"""
        out = (C.c_float * 3)()
        _chk(lib().cw_batch_kernel_ms(self.h, out))
        return {k: (float(v) if v >= 0 else None) for k, v in zip(("ingest", "eval", "check"), out)}

    def kernel_ms_mean(self):
        """the same parts averaged over every run since set_timing("history") (cw_batch_kernel_ms_mean): ({part: ms | None},
        {part: runs averaged over})"""
        out, cnt = (C.c_float * 3)(), (C.c_int * 3)()
        _chk(lib().cw_batch_kernel_ms_mean(self.h, out, cnt))
        names = ("ingest", "eval", "check")
        return {k: (float(v) if v >= 0 else None) for k, v in zip(names, out)}, {k: int(n) for k, n in zip(names, cnt)}

    # -- results ------------------------------------------------------------------------------------
    def status(self) -> np.ndarray:
        out = np.zeros(self.n, dtype=np.uint32)
        _chk(lib().cw_get_status(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def r1cs_first_bad(self) -> np.ndarray:
        out = np.zeros(self.n, dtype=np.uint32)
        _chk(lib().cw_get_r1cs_first_bad(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def witness_bytes(self, instance: int) -> bytes:
        out = np.zeros(self.circuit.n_witness * 32, dtype=np.uint8)
        _chk(lib().cw_get_witness(self.h, instance, out.ctypes.data_as(C.c_void_p)))
        return out.tobytes()

    def witness(self, instance: int):
        return bytes_to_ints(self.witness_bytes(instance))

    def public_signals(self) -> np.ndarray:
        """[batch][n_public][32] uint8: outputs then public inputs of main, for every instance."""
        out = np.zeros((self.n, self.circuit.n_public, 32), dtype=np.uint8)
        _chk(lib().cw_get_public(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def public_signals_device(self, d_ptr: int) -> None:
        """Same, written to device memory at `d_ptr` (batch * n_public * 32 bytes)."""
        _chk(lib().cw_get_public_device(self.h, C.c_void_p(d_ptr)))

    def witnesses(self, first: int = 0, count: int | None = None) -> np.ndarray:
        """[count][n_witness][32] uint8, canonical little-endian values (bulk egress, one device transpose)."""
        count = self.n - first if count is None else count
        out = np.zeros((count, self.circuit.n_witness, 32), dtype=np.uint8)
        _chk(lib().cw_get_witnesses(self.h, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def witnesses_device(self, first: int, count: int, d_ptr: int) -> None:
        """canonical values of `count` instances written to device memory at d_ptr ([count][n_witness][32])"""
        _chk(lib().cw_get_witnesses_device(self.h, first, count, C.c_void_p(d_ptr)))

    def signal(self, instance: int, slot: int) -> int:
        buf = C.create_string_buffer(32)
        _chk(lib().cw_get_signal(self.h, instance, slot, buf))
        return int.from_bytes(buf.raw, "little")

    def write_wtns_many(self, first: int, count: int, pattern: str):
        _chk(lib().cw_write_wtns_many(self.h, first, count, os.fsencode(str(pattern))))

    def write_wtnsb(self, path):
        """the whole batch as one compact container (circom_amd/wtnsb.py reads / expands it)"""
        _chk(lib().cw_write_wtnsb(self.h, os.fsencode(str(path))))

    def explain(self, instance: int, sym_path=None) -> str:
        buf = C.create_string_buffer(1 << 16)
        _chk(lib().cw_explain(self.h, instance, None if sym_path is None else os.fsencode(str(sym_path)), buf, len(buf)))
        return buf.value.decode()

    def write_wtns(self, instance: int, path, witness2signal=None):
        """witness2signal: the kept signals of a simplified constraint system (`--O1`: frontend/circom_simplify.py, the
        `<name>.w2s` file of the driver): the file then holds the witness of THAT system - the entries of the full witness at
        these positions, which is what the reference binary writes when its `.dat` carries the list (main.cpp:288-334)"""
        _chk(lib().cw_write_wtns(self.h, instance, os.fsencode(str(path))))
        if witness2signal is not None:
            from .frontend.circom_simplify import reduce_wtns
            with open(path, "rb") as f:
                full = f.read()
            with open(path, "wb") as f:
                f.write(reduce_wtns(full, witness2signal))

    def log(self, instance: int) -> str:
        """what the reference binary prints on stdout for this instance (its log(...) statements)"""
        n = lib().cw_get_log(self.h, instance, None, 0)
        if n < 0:
            _chk(int(n))
        buf = C.create_string_buffer(int(n) + 1)
        n = lib().cw_get_log(self.h, instance, buf, len(buf))
        if n < 0:
            _chk(int(n))
        return buf.value.decode()


def fp_mul_bench(q: int, a: np.ndarray, b: np.ndarray, iters: int, device: int = 0):
    """a, b: uint8 [n,32].  Returns (out uint8 [n,32], milliseconds)."""
    n = a.shape[0]
    out = np.zeros_like(a)
    ms = C.c_float()
    _chk(lib().cw_fp_mul_bench(q.to_bytes(32, "little"), device, n, iters, a.ctypes.data_as(C.c_void_p),
                               b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.byref(ms)))
    return out, ms.value


def fp_op(q: int, dop: int, a, b, c, device: int = 0):
    """Element-wise device op on lists of ints; returns (list of ints, status array)."""
    n = len(a)
    A = np.frombuffer(fe_to_bytes(a), dtype=np.uint8)
    B = np.frombuffer(fe_to_bytes(b), dtype=np.uint8)
    Cc = np.frombuffer(fe_to_bytes(c), dtype=np.uint8)
    out = np.zeros(n * 32, dtype=np.uint8)
    st = np.zeros(n, dtype=np.uint32)
    _chk(lib().cw_fp_op(q.to_bytes(32, "little"), device, dop, n, A.ctypes.data_as(C.c_void_p),
                        B.ctypes.data_as(C.c_void_p), Cc.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                        st.ctypes.data_as(C.c_void_p)))
    return bytes_to_ints(out.tobytes()), st
