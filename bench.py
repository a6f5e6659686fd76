#!/usr/bin/env python3
"""bench.py — witnesses/sec of the batched HIP witness calculator (BASELINE.json's metric).

Default workload = the metric's: `sha256_2048`, a SHA-256 circuit of 5 compression blocks = 1 020 832 constraints at
`--O0` (>= 1M, asserted), batch 2 097 152 per MI355X through the circuit's emitted code (DEFAULT_BATCH; `--batch 65536` = the
interpreting kernels).  Other workloads (parity-test configurations of BASELINE.json): poseidon2 (configs[1], batch 65536),
sha256_512 (configs[2], batch 4096), semaphore20 / 20p / 20w (configs[3]'s relation), bigmultmodp and ecdsa_verify (configs[4]).

A "step" = one pass of the hot path over one batch of synthetic inputs that are already resident in HBM:
ingest (AoS -> value-table input slots) + schedule evaluation (witness generation) + R1CS check.
Multi-GPU: one process per GPU (torch.distributed / RCCL), instances are sharded; rank 0 compiles the circuit once
and the other ranks load the artefacts; the only collective is the final gather of status words + public signals.
`--total-batch N` fixes the whole job's batch (strong scaling, BASELINE config 4 = 8192 over 8 GPUs); the default is
a fixed per-GPU batch (weak scaling).

After the timed region, sampled instances AT THE BENCHMARK BATCH are compared with the oracle (reference C++ runtime
built under oracle/_ref when present, else the Python restatement) byte for byte, and — for SHA-256 workloads — every
instance's digest with hashlib; the JSON reports `parity_checked`.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel of the run, chosen by measured time), per-kernel
rooflines and `cpu_baseline` (reference C++ runtime timed on the host cores, core count stated).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable
N_SIMD = 1024             # 256 CUs x 4 SIMD-32
NOMINAL_CLOCK_HZ = 2.4e9  # max shader clock; profiles carry the clock the kernel actually ran at (GRBM_GUI_ACTIVE / 8 / duration)
VALU_CLK_PER_WAVE_INST = 2.0       # a SIMD-32 issues a wave64 VALU instruction over 2 clocks (guide; tools/ubench_isa: 2.05 from 2 waves/SIMD)
LONE_WAVE_CLK_PER_INST = 4.1       # ONE wave issues at most one instruction of any kind per ~4.1 clocks (tools/ubench_isa)
BITS_VALU_PER_VROW = 10            # cw_bits_eval_kernel<64>, from the disassembly: 3 operand offsets, 2 mask expansions, 1 result
BITS_INSTS_PER_VROW = 15           # offset, 4 v_bitop3  + 3 ds_read_b64, 1 ds_write_b64, 1 s_waitcnt
# one Montgomery product of the 256-bit engine (fp256.hip.h fe_mmul, 9 x 29-bit limbs): 266 VALU instructions, 162 of them
# v_mad_u64_u32; chip-wide issue rates measured by tools/ubench_isa (profiles/r03_ubench_isa.json, 8 waves per SIMD)
FPMUL_INSTS, FPMUL_MADS = 266, 162
MAD_U64_WAVE_INSTS_PER_S = 5.4796e11      # v_mad_u64_u32 (half rate)
SIMPLE_VALU_WAVE_INSTS_PER_S = 9.8689e11  # v_and_b32 / v_add_u32 class (full rate)
JIT_BATCH = 1 << 21                # the emitted bit-plane code runs one wave per 2 048 instances: 1 024 waves = one per SIMD
DEFAULT_BATCH = {"bigmultmodp": 8192, "ecdsa_verify": 1024, "sha256_2048": JIT_BATCH, "sha256_27008": 1 << 19, "sha256_512": 4096, "poseidon2": 65536, "semaphore20": 8192, "semaphore20p": 8192,
                 "semaphore20w": 8192}


def _semaphore_shape(name: str):
    """semaphore<levels>[p|w]: p = the witness hints walk the scalar multiplications on a projective ladder; w = circomlib's
    structure (windowed EscalarMulFix, Montgomery-form EscalarMulAny: circuits/escalarmul.py)"""
    tail = name[len("semaphore"):]
    proj = "window" if tail.endswith("w") else tail.endswith("p")
    return int(tail.rstrip("pw") or 20), proj


def make_program(name: str):
    from circom_amd.frontend.dsl import Program
    if name == "poseidon2":
        from circom_amd.circuits.poseidon import Poseidon
        return Program(Poseidon(2))
    if name.startswith("sha256_"):
        from circom_amd.circuits.sha256 import Sha256
        return Program(Sha256(int(name.split("_")[1])))
    if name.startswith("semaphore"):
        from circom_amd.circuits.eddsa import SemaphoreStyle
        levels, proj = _semaphore_shape(name)
        return Program(SemaphoreStyle(levels, proj))
    if name == "ecdsa_verify":
        # BASELINE config 5: secp256k1 ECDSA verification (circom-ecdsa's shape) over the BLS12-381 scalar field: 2.47 M signals,
        # 2.49 M constraints; the witness hints are circom functions - long_div interpreted per lane, the three that contain a
        # modular inverse through their native device routines
        from circom_amd.circuits import secp256k1 as S
        return Program(S.ECDSAVerifyNoPubkeyCheck(64, 4, S.SECP256K1, 8), prime="bls12381")
    if name.startswith("bigmultmodp"):
        # circom-ecdsa's field multiplication (a * b mod p on k limbs of n bits; the witness comes from the run-time
        # functions long_div / short_div with value-dependent branches = tier 2) on the BLS12-381 scalar field:
        # BASELINE config 5's building block.  bigmultmodp = 3 limbs of 32 bits; bigmultmodp_<n>_<k> picks the shape
        from circom_amd.circuits.bigint import BigMultModP
        n, k = ([int(x) for x in name.split("_")[1:3]] if "_" in name else (32, 3))
        return Program(BigMultModP(n, k), prime="bls12381")
    raise SystemExit("unknown workload " + name)


def source_fingerprint() -> str:
    """Hash of everything that shapes the kernels and the schedule: cached artefacts and committed counter figures
    (profiles/traffic.json) are only used when they were produced by this exact source."""
    h = hashlib.sha256()
    files = sorted((ROOT / "circom_amd" / "csrc").glob("*")) + sorted((ROOT / "circom_amd" / "hip_elements").glob("*.py")) \
        + sorted((ROOT / "circom_amd" / "frontend").glob("*.py")) + sorted((ROOT / "circom_amd" / "circuits").glob("*.py"))
    for f in files:
        if f.is_file() and f.suffix in (".hip", ".h", ".cpp", ".py"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def files_fingerprint(files) -> str:
    """hash of the named files (paths relative to the repository): what profiles/traffic.json entries carry for the sources
    that shape one workload's kernels"""
    h = hashlib.sha256()
    for rel in files:
        f = ROOT / rel
        h.update(rel.encode())
        h.update(f.read_bytes() if f.is_file() else b"<missing>")
    return h.hexdigest()[:16]


def artefact_fingerprint() -> str:
    """Hash of what shapes the compiled artefacts (.cwt/.dat/.r1cs): the Python front-end and lowering, the circuit
    library and the tape format - NOT the kernels, so that a kernel change does not invalidate a cached schedule."""
    if os.environ.get("CW_ARTEFACT_FP"):          # experiments: artefacts prebuilt under this key (tools/), whatever the tree says now
        return os.environ["CW_ARTEFACT_FP"]
    h = hashlib.sha256()
    # (frontend/circom_*.py, the front-end for circom SOURCE TEXT, is left out: the bench's circuits are traced from the eDSL)
    files = sorted((ROOT / "circom_amd" / "hip_elements").glob("*.py")) \
        + [f for f in sorted((ROOT / "circom_amd" / "frontend").glob("*.py")) if not f.name.startswith("circom_")] \
        + sorted((ROOT / "circom_amd" / "circuits").glob("*.py")) + [ROOT / "circom_amd" / "csrc" / "cw_tape.h",
                                                                    # (the emitted code's row bodies are compiled from these two)
                                                                    ROOT / "circom_amd" / "csrc" / "fp256.hip.h", ROOT / "circom_amd" / "csrc" / "cw_rowops.hip.h",
                                                                    ROOT / "circom_amd" / "compiler.py", ROOT / "circom_amd" / "opcodes.py",
                                                                    ROOT / "circom_amd" / "field.py"]
    for f in files:
        if f.is_file():
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def _o1_size(fc):
    """What the circuit shrinks to at `--O1`, the level the reference applies when no flag is given (constant and renaming
    substitutions, frontend/circom_simplify.py): reported beside the --O0 sizes the device evaluates and checks row by row.
    Never lets the bench fail: None when the count could not be taken."""
    try:
        if os.environ.get("CW_BENCH_NO_O1") or fc.n_signals > 3_000_000:
            return None
        from circom_amd.frontend.circom_simplify import simplify_o1
        sm = simplify_o1(fc)
        _o1_size.witness2signal = np_asarray_u32(sm.witness2signal)      # kept for the O1 egress measurement below
        return {"wires": sm.n_wires, "constraints": len(sm.constraints),
                "note": "the device generates and checks the --O0 system; cw_set_witness_list hands out these wires"}
    except Exception as ex:                                       # noqa: BLE001
        return {"error": repr(ex)[:200]}


_o1_size.witness2signal = None


def np_asarray_u32(x):
    import numpy as np
    return np.ascontiguousarray(x, dtype=np.uint32)


def get_compiled(name: str, batch: int, cache_root: str, rank: int, dist):
    """Rank 0 traces + lowers the circuit once (or finds the artefacts of this exact source in the cache); the other
    ranks wait and load the files.  Every rank traces + flattens (seconds) to have the flat code for the oracle."""
    from circom_amd import compiler
    from circom_amd.frontend.flatten import flatten
    from circom_amd.hip_elements import writers
    from circom_amd.hip_elements.lower import lower
    fp = artefact_fingerprint()
    strands = compiler.strands_for(batch)
    if (batch + 63) // 64 >= 1024 and 1 not in strands:
        # a batch that fills every SIMD with one strand: the runtime prefers the single-strand emitted program with the fused
        # check (cw_batch_create) - lower that variant as well (bit-level circuits lower one strand anyway)
        strands = (1,) + tuple(strands)
    if os.environ.get("CW_BENCH_STRANDS"):          # experiments: lower (only) this strand count
        strands = (int(os.environ["CW_BENCH_STRANDS"]),)
    d = os.path.join(cache_root, "%s_s%s_b%s_m%s_%s" % (name, "-".join(map(str, strands)), os.environ.get("CW_BITS", "1"),
                                                       os.environ.get("CW_MONT", "a"), fp))
    p = lambda ext: os.path.join(d, name + ext)
    t0 = time.perf_counter()
    fc = flatten(make_program(name))
    done = os.path.join(d, "done")
    lock = None
    if rank == 0:
        # several processes of one box may want the same artefacts (pytest-xdist workers): one lowers, the others wait and load
        import fcntl
        os.makedirs(cache_root, exist_ok=True)
        lock = open(d + ".lock", "w")
        fcntl.flock(lock, fcntl.LOCK_EX)
    cached = os.path.exists(done)
    if cached and rank == 0:
        # large artefacts travel compressed (tools/prebuild_cache.py: the .r1cs of the ECDSA verifier is 407 MB, 34 MB gzipped)
        # (.xz for the 53-block SHA-256, tools/r06_prebuild_27008.sh: 1.7 GB of tables in 93 MB - gzip leaves 254 MB, and the gpurun
        # snapshot has a size limit)
        import gzip
        import lzma
        import shutil
        for ext in (".cwt", ".dat", ".r1cs"):
            for suffix, opener in ((".gz", gzip.open), (".xz", lzma.open)):
                if not os.path.exists(p(ext)) and os.path.exists(p(ext) + suffix):
                    with opener(p(ext) + suffix, "rb") as fi, open(p(ext) + ".tmp", "wb") as fo:
                        shutil.copyfileobj(fi, fo, 1 << 24)
                    os.replace(p(ext) + ".tmp", p(ext))
    if rank == 0 and not cached:
        os.makedirs(d, exist_ok=True)
        bittape, bitnet = (None, None) if os.environ.get("CW_BITS", "1") == "0" else compiler.lower_bitplane_net(fc)
        mont = compiler.choose_mont(fc)     # arithmetic circuits keep their signals in Montgomery form on the device
        if os.environ.get("CW_MONT"):
            mont = os.environ["CW_MONT"] != "0"
        if bittape is not None:
            strands = (1,)                  # the 256-bit schedule only serves instances re-run with non-boolean inputs
            mont = False
        tapes = [lower(fc, n_strands=s, mont=mont) for s in strands]
        if tapes[0].functions and tapes[0].n_strands > 1 and 1 not in strands and compiler.parallel_speedup(tapes[0]) < 2.0:
            # strands that would mostly wait for one chain of calls (BigMultModP = one long_div + a few rows): the single-strand
            # variant - which has an emitted form - goes into the tape too, and cw_batch_create prefers it
            tapes.insert(0, lower(fc, n_strands=1, mont=mont))
        jp = compiler.emit_jit(bitnet, fc) if bittape is not None else None     # the same network as emitted code
        del bitnet
        # arithmetic circuits: the rows of every strand variant as emitted code as well (hip_elements/fpjit.py); schedules with
        # run-time functions on several strands / with the native long_div (config 5's verifier) run on the interpreting kernel
        fps = compiler.emit_fpjit(tapes, fc, False if bittape is not None else "auto")
        rid = writers.write_r1cs(p(".r1cs"), fc)
        writers.write_tape(p(".cwt"), tapes, bittape, jp, fps, r1cs_id=rid)
        json.dump(dict(jp.stats, code_bytes=len(jp.code), audit_code_bytes=len(jp.audit_code or b"")) if jp is not None else {}, open(p(".jit.json"), "w"))
        json.dump([dict(fp_.stats, n_strands=fp_.n_strands, code_bytes=len(fp_.code)) for fp_ in fps], open(p(".fpjit.json"), "w"))
        writers.write_dat(p(".dat"), fc)
        open(done, "w").write("%s\n%.1f\n" % (fp, time.perf_counter() - t0))     # fingerprint, seconds of flatten + lowering + emission
    if lock is not None:
        lock.close()                    # (releases the flock)
    if dist:
        dist.barrier()
    cp = compiler.Compiled(name, d, p(".cwt"), p(".dat"), p(".r1cs"), p(".sym"), fc, None)
    try:
        cp.jit_stats = json.load(open(p(".jit.json")))
    except Exception:
        cp.jit_stats = {}
    try:
        cp.fpjit_stats = json.load(open(p(".fpjit.json")))
    except Exception:
        cp.fpjit_stats = []
    try:
        cp.compile_s_cold = float(open(done).read().split()[1])          # what the cached artefacts cost when they were made
    except Exception:
        cp.compile_s_cold = None
    return cp, time.perf_counter() - t0, cached


def synth_inputs(name: str, q: int, batch: int, n_inputs: int, seed: int):
    import numpy as np
    rng = np.random.default_rng(seed)
    if name.startswith("sha256_"):
        bits = rng.integers(0, 2, size=(batch, n_inputs), dtype=np.uint8)
        arr = np.zeros((batch, n_inputs, 32), dtype=np.uint8)
        arr[:, :, 0] = bits
        return arr
    if name == "ecdsa_verify":
        # valid (r, s, msghash, pubkey) tuples synthesised on the host: 32 distinct ones, tiled
        import random
        from circom_amd.circuits import secp256k1 as S
        rnd = random.Random(seed)
        pool = [np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in S.sign(S.SECP256K1, 64, 4, rnd)), dtype=np.uint8).reshape(n_inputs, 32)
                for _ in range(min(batch, 32))]
        return np.ascontiguousarray(np.stack([pool[i % len(pool)] for i in range(batch)]))
    if name.startswith("bigmultmodp"):
        n, k = ([int(x) for x in name.split("_")[1:3]] if "_" in name else (32, 3))
        import random
        rnd = random.Random(seed)
        out = np.zeros((batch, n_inputs, 32), dtype=np.uint8)
        pool = []
        for it in range(min(batch, 512)):                  # 512 distinct (a, b, p): every instance takes its own branches
            p = rnd.randrange(1 << (n * k - 1), 1 << (n * k))
            a, b = rnd.randrange(p), rnd.randrange(p)
            vals = [(x >> (n * i)) & ((1 << n) - 1) for x in (a, b, p) for i in range(k)]
            pool.append(np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(n_inputs, 32))
        perm = rng.integers(0, len(pool), size=batch)
        for i in range(batch):
            out[i] = pool[perm[i]]
        return out
    if name.startswith("semaphore"):
        # valid EdDSA signatures + Merkle paths must be synthesised on the host: 64 distinct
        # (key, message, signature, path) vectors, tiled over the batch (the schedule is data-independent)
        import random
        from circom_amd.circuits import eddsa_host as H
        r = random.Random(seed)
        levels = _semaphore_shape(name)[0]
        pool = [H.semaphore_inputs(q, levels, r)[0] for _ in range(min(batch, 64))]
        one = np.frombuffer(b"".join(v.to_bytes(32, "little") for row in pool for v in row),
                            dtype=np.uint8).reshape(len(pool), n_inputs, 32)
        return np.ascontiguousarray(np.tile(one, ((batch + len(pool) - 1) // len(pool), 1, 1))[:batch])
    # uniform field elements: 256 random bits reduced mod q
    raw = rng.integers(0, 256, size=(batch, n_inputs, 32), dtype=np.uint8)
    vals = [int.from_bytes(raw[i, k].tobytes(), "little") % q for i in range(batch) for k in range(n_inputs)]
    return np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(batch, n_inputs, 32).copy()


class BoolInputs:
    """Boolean inputs held as one byte each ([B][n_inputs] of 0/1) that index like the canonical [B][n_inputs][32] image
    (`--packed-inputs`: a 53-block SHA-256 batch of 65 536 is 57 GB in canonical form, 221 MB as packed masks)."""

    def __init__(self, bits):
        self.bits = bits
        self.nbytes = int(bits.shape[0]) * int(bits.shape[1]) * 32

    def __getitem__(self, key):
        import numpy as np
        if isinstance(key, tuple) and len(key) == 3 and key[2] == 0:
            return self.bits[key[0], key[1]]
        if isinstance(key, tuple):
            return self[key[0]][key[1]]
        row = np.zeros((self.bits.shape[1], 32), dtype=np.uint8)
        row[:, 0] = self.bits[key]
        return row

    def masks(self):
        import numpy as np
        B, n = self.bits.shape
        ng = (B + 63) // 64
        out = np.empty((ng, n), dtype=np.uint64)
        for g0 in range(0, ng, 64):                          # 64 groups at a time: the padded copy stays small
            g1 = min(ng, g0 + 64)
            blk = np.zeros(((g1 - g0) * 64, n), dtype=np.uint8)
            part = self.bits[g0 * 64:min(B, g1 * 64)]
            blk[:part.shape[0]] = part
            m = np.packbits(blk.reshape(g1 - g0, 64, n), axis=1, bitorder="little")
            out[g0:g1] = np.ascontiguousarray(m.transpose(0, 2, 1)).view(np.uint64).reshape(g1 - g0, n)
        return out


XGMI_LINK_GBS = 153.0            # per point-to-point link (MI355X_MICROARCH.md); every peer reaches rank 0 over its own link


def xgmi_gather_ms(public_bytes_per_instance: int, rows_per_peer: int) -> float:
    """the job's one exchange, predicted: each peer's status words + public signals to rank 0, links in parallel"""
    return 0.02 + (4 + public_bytes_per_instance) * rows_per_peer / (XGMI_LINK_GBS * 1e9) * 1e3


def batch_witness_bytes(b, i, n_wit):
    """the witness of instance i as the per-instance egress hands it out (cw_get_witness): n_wit x 32 bytes"""
    wb = b.witness_bytes(i)
    assert len(wb) == n_wit * 32
    return wb


def in_step_view(roof: dict, alg: float, peak: float, iso_ms, in_ms, runs) -> dict:
    """VERDICT r5 #1a: the headline figures of a roofline object are those of the kernel INSIDE the timed region (mean over
    its steps, other batches in flight beside it); the same kernel running alone stays beside them under `isolated`."""
    if not roof or not iso_ms:
        return roof
    use = in_ms or iso_ms
    roof["isolated"] = {"kernel_ms": iso_ms, "achieved": alg / (iso_ms * 1e-3) / roof.get("_scale", 1e9),
                        "frac": alg / (iso_ms * 1e-3) / roof.get("_scale", 1e9) / peak, "is": "one step running alone on the GPU (one HIP-event pair)"}
    roof["kernel_ms"] = use
    roof["achieved"] = alg / (use * 1e-3) / roof.get("_scale", 1e9)
    roof["frac"] = roof["achieved"] / peak
    roof["in_step_ms"] = in_ms
    roof["kernel_ms_is"] = ("mean over the %d launches of the timed region (cw_batch_kernel_ms_mean: HIP events on each batch's own stream), "
                            "other batches in flight beside it" % runs) if in_ms else "one step alone (no in-step marks)"
    roof.pop("_scale", None)
    return roof


def dominant_roofline(roof_eval, roof_r1cs, roof_ingest, gen_ms, chk_ms, jit, packed):
    """the roofline object of the kernel with the longest measured duration in this run"""
    if jit:
        cands = [roof_eval] + ([roof_ingest] if roof_ingest and not packed else [])
        return max(cands, key=lambda r: r.get("kernel_ms") or 0.0)
    return roof_eval if gen_ms >= chk_ms else roof_r1cs


def auto_in_flight(bitmode: bool, B: int, lanes: int) -> int:
    """Batches in flight for `--in-flight 0`.  Two keep the check of one step next to the evaluation of the next; a batch
    that covers a fraction of the chip gets more, so that the rest of the chip works through the length of its dependency
    chain."""
    if bitmode:             # one wave per SIMD and group slice: 1 024 of them fill the chip (Sha256(512) x 4 096 = 256 waves:
        waves = ((B + 63) // 64) * (64 // max(1, lanes))     # 2 in flight -> 20.0 M, 4 -> 34.1 M, 8 -> 33.0 M witnesses/s)
        # (16 hardware queues, steps as HIP graphs: 4 in flight 38.0 M, 16 in flight 41.2-41.7 M, profiles/r06s_*, r06w_*)
        if os.environ.get("GPU_MAX_HW_QUEUES") == "16":
            return max(2, min(16, 4096 // max(1, waves)))
        return max(2, min(4, 1024 // max(1, waves)))
    # 256-bit engine: twice the batches that fill the 256 CUs, at most eight: the dispatcher does not always put the
    # workgroups of four 64-workgroup batches on four disjoint quarters of the chip (tools/sema_inflight.py: 4 in flight gave
    # 62.8 K in one process and 119 K in another, 8 gave 118.9 K and 118.5 K)
    wgs = (B + lanes - 1) // max(1, lanes)
    return max(2, min(16 if os.environ.get("GPU_MAX_HW_QUEUES") == "16" else 8, 512 // max(1, wgs)))


def in_flight_for(batch) -> int:
    """... and a single-strand emitted program on a chip-filling batch (one wave per SIMD per batch) wants a third batch in
    flight: Poseidon(2) x 65 536: 2 -> 59.2 M, 3 -> 62.1 M, 4 -> 58.4 M witnesses/s (tools/fpjit_poseidon_inflight.sh)"""
    if batch.bitmode and getattr(batch, "jit", False) and (batch.n + 2047) // 2048 >= 1024:
        # emitted bit-plane code on a chip-filling batch: the ingest (HBM reads at the read roof) and the emitted kernel (HBM
        # writes) of two batches in flight slow each other down exactly as much as they overlap - 35.4 ms per step either way
        # (profiles/r05r_*, r06a_*: 35.38 in flight vs 22.17 + 13.15 alone) - so the steps run one after the other: every
        # kernel's in-step duration is then its own, and the step is the sum of its kernels
        return 1
    n = auto_in_flight(batch.bitmode, batch.n, batch.lanes)
    if not batch.bitmode and getattr(batch, "emitted", False) and batch.strands == 1 and (batch.n + 63) // 64 >= 1024:
        n = max(n, 3)
    return n


def golden_messages(workload: str):
    """[(instance position rule, message bytes, wtns sha256, wtns length)] of tests/golden/reference_wtns_<workload>.json: vectors
    the REFERENCE runtime wrote in the container that has the reference tree (make_golden_27008.py), for workloads whose
    reference binary and tables are too large to travel with the snapshot.  [] when there is no such file."""
    f = ROOT / "tests" / "golden" / ("reference_wtns_%s.json" % workload)
    if not f.is_file():
        return []
    g = json.load(open(f))
    return [(bytes.fromhex(v["message_hex"]), v["wtns_sha256"], int(v["wtns_len"])) for v in g.get("vectors", ()) if "message_hex" in v]


def parity_check(cp, circ, batch, h_in, workload: str, n_sample: int = 4, digest_bits=None, golden_at=None):
    """Oracle comparison at the benchmark batch (after the timed region).  Returns a dict for the JSON line; raises
    AssertionError on any mismatch (a fast wrong answer is not a result)."""
    import numpy as np
    B = batch.n
    picks = sorted({0, 1, B // 2, B - 1} | {(B * k) // max(n_sample, 1) for k in range(n_sample)})[:max(n_sample, 1)]
    picks = [i for i in picks if i < B]
    out = {"instances": picks, "oracle": None, "digests_checked": 0}
    fc = cp.flat
    n_in = circ.n_inputs
    from circom_amd.hip_elements.writers import wtns_bytes
    td = tempfile.mkdtemp(prefix="cw_parity_")
    got = {}
    if golden_at:
        # instances that carry the golden messages: the `.wtns` the C ABI writes against the hash of the REFERENCE runtime's file
        for pos, (msg, sha, length) in golden_at.items():
            fn = os.path.join(td, "gold_%d.wtns" % pos)
            batch.write_wtns(pos, fn)
            h = hashlib.sha256()
            n = 0
            with open(fn, "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
                    n += len(blk)
            os.unlink(fn)
            assert n == length and h.hexdigest() == sha, "PARITY FAILURE: .wtns of instance %d differs from the reference runtime's golden" % pos
        out["golden_instances"] = sorted(golden_at)
        out["oracle"] = "reference C++ runtime, golden .wtns hashes (tests/golden/reference_wtns_%s.json), full files" % workload
        picks = []
        out["instances"] = sorted(golden_at)
    for i in picks:                                       # through the C ABI's writeBinWitness (cw_write_wtns)
        batch.write_wtns(i, os.path.join(td, "g_%d.wtns" % i))
        got[i] = open(os.path.join(td, "g_%d.wtns" % i), "rb").read()
    want = None
    try:
        if not picks:
            raise RuntimeError("golden vectors were compared")
        from oracle import ref_build
        cli, loop = ref_build.build_circuit(cp)          # cached binary if its fingerprint matches this circuit
        if loop.exists():
            raw = b"".join(h_in[i].tobytes() for i in picks)
            ref_build.run_loop(cp, raw, len(picks), 1, wtns_prefix=os.path.join(td, "r_"))
            want = {i: open(os.path.join(td, "r_%d.wtns" % k), "rb").read() for k, i in enumerate(picks)}
            out["oracle"] = "reference C++ runtime (oracle/_ref/%s/%s_loop), full .wtns bytes" % (fc.prime, cp.name)
    except Exception as e:      # fall back to the Python restatement below
        out["oracle_note"] = "reference binary unusable: %s" % str(e)[:120]
    if want is None and picks:
        from oracle.tape_eval import eval_flat
        want = {}
        for i in picks:
            inp = {fc.main_input_start + k: int.from_bytes(h_in[i, k].tobytes(), "little") for k in range(n_in)}
            sig, failed = eval_flat(circ.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp, getattr(fc, "functions", ()))
            assert failed is None, "oracle reports a failed assert for instance %d" % i
            want[i] = wtns_bytes(circ.q, sig)
        out["oracle"] = "oracle/tape_eval.eval_flat (Python restatement of the emitted calculator), full .wtns bytes"
    for i in picks:
        assert got[i] == want[i], "PARITY FAILURE: .wtns of instance %d differs from the oracle's" % i
    if picks:
        out["wtns_sha256_first"] = hashlib.sha256(got[picks[0]]).hexdigest()
    else:
        out.pop("oracle_note", None)                      # (nothing left for the sampled comparison: the goldens were it)
    if workload.startswith("sha256_"):
        # every instance's digest against hashlib (output bit k = bit 7-(k%8) of digest byte k/8, msb first)
        if digest_bits is None:
            pub = batch.public_signals()                     # [B][n_public][32]
            assert not pub[:, :256, 1:].any(), "digest signals are not bits"
            digest_bits = pub[:, :256, 0]
        nb = int(workload.split("_")[1])
        msgs = np.packbits(h_in[:, :nb, 0], axis=1)          # [B][nb / 8] message bytes
        sha = hashlib.sha256
        want = np.frombuffer(b"".join(sha(msgs[i].tobytes()).digest() for i in range(B)), dtype=np.uint8).reshape(B, 32)
        got = np.packbits(digest_bits, axis=1)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, "PARITY FAILURE: digest of instance %d differs from hashlib" % int(bad[0])
        out["digests_checked"] = B
    return out


def cpu_baseline(cp, name, seconds_budget=15.0):
    """Time the CPU checker on a bounded sample of the same workload (rank 0, N=1 only)."""
    try:
        from oracle import ref_build
    except Exception:
        return None
    try:
        return ref_build.time_reference(cp, name, seconds_budget)
    except Exception as e:   # the baseline is a report, never a reason to fail the bench
        return {"value": None, "unit": "witnesses/s", "cores": 0, "kind": "reference", "sample": "failed: %s" % e}


def host_only_rehearsal(args, world, rank):
    """The N-rank launch without GPUs (gloo): rank 0 compiles once, every rank loads the artefacts, stages the inputs of
    ITS shard in a host-only batch through the C ABI (validated, nothing computes), and the one exchange of the job -
    the gather of per-instance words and public-signal rows on rank 0 - runs over gloo.  Prints the JSON line with
    `value: null`."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from circom_amd import runtime as rt
    from circom_amd.sharding import shard_range, gather_status, gather_rows
    if world > 1:
        dist.init_process_group("gloo")
    else:
        dist = None
    B = args.batch or 64
    scaling = "weak"
    if args.total_batch:
        scaling = "strong"
        lo, hi = shard_range(args.total_batch, rank, world)
        B = hi - lo
    cache_root = args.cache_dir or os.path.join(tempfile.gettempdir(), "cw_bench_cache_%d" % os.getuid())
    cp, compile_s, cached = get_compiled(args.workload, B, cache_root, rank, dist)
    circ = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    batch = circ.batch(B, device=-1)
    h_in = synth_inputs(args.workload, circ.q, B, circ.n_inputs, seed=1 + rank)
    batch.set_inputs(h_in)
    assert all(batch.remaining_inputs(k) == 0 for k in (0, B - 1))
    try:
        batch.run()
        raise AssertionError("a host-only batch must not compute")
    except rt.CwError:
        pass
    # stand-ins for the status words / public signals of this shard: derived from the staged inputs
    words = torch.tensor([batch.staged_input(k, 0) % (1 << 20) for k in range(B)], dtype=torch.int32)
    got = gather_status(words, dist, rank, world)
    rows = gather_rows(torch.from_numpy(np.ascontiguousarray(h_in[:, :1, :])), dist, rank, world)
    shard_sizes = gather_status(torch.tensor([B], dtype=torch.int32), dist, rank, world)       # what every rank owns
    if rank == 0:
        print(json.dumps({"metric": "witnesses/sec (batched inputs)", "value": None, "unit": "witnesses/s", "n_gpus": world,
                          "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": True, "scaling": scaling,
                          "vs_baseline": None, "dtype": "none (host-only rehearsal)", "data": "synthetic", "host_only": True,
                          "config": {"workload": "%s, batch=%d per rank (host-only rehearsal of the launch path)" % (args.workload, B),
                                     "compile_s": compile_s, "compile_cached": cached},
                          "gathered": {"status_words": int(got.numel()), "public_signal_rows": int(rows.shape[0])},
                          "shards": [int(x) for x in shard_sizes.tolist()],
                          # the job's one exchange on the GPU path: status word + public signals of every instance, peer -> rank 0
                          "gather_bytes_per_peer": (4 + 32 * circ.n_public) * B,
                          "predicted_gather_ms": xgmi_gather_ms(32 * circ.n_public, B) if world > 1 else 0.0}))
    batch.close()
    circ.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def goldilocks_bench(args):
    """`--workload poseidon2_goldilocks`: SURVEY row f4 measured - Poseidon(2) on the reference's 64-bit runtime prime (`--prime
    goldilocks`: goldilocks/fr.hpp + common64/), its own small engine on the device (csrc/cw64.hip: one 8-byte value per signal
    and instance, one row per operation).  One JSON line of the usual shape; parity = full n8 = 8 `.wtns` files against the
    reference's 64-bit runtime (oracle/_ref/goldilocks/poseidon2, built by build()) or, where that binary is absent, against the
    Python oracle; cpu_baseline = that runtime's CLI, one process per witness (it has no in-process loop build)."""
    import numpy as np
    import torch
    from circom_amd import runtime as rt, compiler
    from circom_amd.frontend.dsl import Program
    from circom_amd.circuits.poseidon import Poseidon
    from circom_amd.hip_elements.writers import wtns_bytes
    name, B = "poseidon2", args.batch or 65536
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    d = tempfile.mkdtemp(prefix="cw_gl_")
    t0 = time.perf_counter()
    cp = compiler.compile_program(Program(Poseidon(2), prime="goldilocks"), d, name, sym=False)
    compile_s = time.perf_counter() - t0
    circ = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    q, n_in, n_wit = circ.q, circ.n_inputs, circ.n_witness
    rng = np.random.default_rng(7)
    vals = rng.integers(0, q, size=(B, n_in), dtype=np.uint64)
    h_in = np.zeros((B, n_in, 32), dtype=np.uint8)
    h_in[:, :, :8] = vals.view(np.uint8).reshape(B, n_in, 8)
    d_in = torch.from_numpy(h_in).to(dev)
    # (16 hardware queues: 1 in flight 51.6 M, 2 -> 68.5 M, 3 -> 78.8 M, 6 -> 82.8 M witnesses/s, profiles/r06ai_goldilocks_in_flight.txt)
    n_fl = max(1, args.in_flight or (6 if os.environ.get("GPU_MAX_HW_QUEUES") == "16" else 2))
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_fl)]
    batches = [circ.batch(B, device=0, stream=s_.cuda_stream) for s_ in streams]
    for b_ in batches:
        b_.set_inputs_device(d_in.data_ptr())
        b_.set_timing(True)

    def step(i):
        b_ = batches[i % n_fl]
        b_.run()
        b_.check_r1cs()
    step(0); torch.cuda.synchronize()
    step(0); torch.cuda.synchronize()
    iso = batches[0].kernel_ms()
    for i in range(max(args.warmup, n_fl)):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st = batches[0].status()
    assert (st == 0).all(), "failed instances on the goldilocks workload"
    # parity: full .wtns bytes of a few instances
    picks = sorted({0, 1, B // 2, B - 1})
    cli = ROOT / "oracle" / "_ref" / "goldilocks" / name
    par = {"instances": picks}
    for i in picks:
        p_ = os.path.join(d, "g%d.wtns" % i)
        batches[0].write_wtns(i, p_)
        got = open(p_, "rb").read()
        row = [int(v) for v in vals[i]]
        if cli.exists():
            from oracle import ref_build
            out_w = Path(d) / ("r%d.wtns" % i)
            r_ = ref_build.run_cli64(cli, json.dumps({"inputs": [str(x) for x in row]}), out_w)
            assert r_.returncode == 0, r_.stderr[-300:]
            want = out_w.read_bytes()
            par["oracle"] = "reference 64-bit runtime (oracle/_ref/goldilocks/%s: goldilocks/fr.hpp + common64/), full .wtns bytes" % name
        else:
            from oracle.tape_eval import eval_flat
            fc = cp.flat
            sig, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {fc.main_input_start + k: v for k, v in enumerate(row)})
            assert failed is None
            want = wtns_bytes(q, sig)
            par["oracle"] = "oracle/tape_eval.eval_flat on the Goldilocks prime (pinned to the reference's 64-bit runtime by tests/test_goldilocks_oracle.py)"
        assert got == want, "PARITY FAILURE: goldilocks .wtns of instance %d" % i
    par["parity_checked"] = len(picks)
    # the kernels' own intervals and their byte roof: 8 bytes per value (SURVEY 8d with n8 = 8)
    alg_gen, alg_chk = 8.0 * (n_in + n_wit) * B, 8.0 * n_wit * B
    roof = {"bound": "hbm", "kernel": "cw64_eval_kernel (one row per operation, 8-byte values)", "unit": "GB/s",
            "achieved": alg_gen / (iso["eval"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "frac": alg_gen / (iso["eval"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": None, "traffic_measured_in_run": False, "algorithmic_bytes_per_launch": alg_gen, "kernel_ms": iso["eval"],
            "kernel_ms_source": "HIP events around the kernel on its stream (cw_batch_kernel_ms), one step alone",
            "rows_per_witness": int(circ.n_rows), "field_ops_per_s": circ.n_rows * B / (iso["eval"] * 1e-3)}
    roof_chk = {"bound": "hbm", "kernel": "cw64_r1cs_kernel", "unit": "GB/s", "achieved": alg_chk / (iso["check"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "frac": alg_chk / (iso["check"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": iso["check"], "algorithmic_bytes_per_launch": alg_chk}
    cpu = None
    if not args.no_cpu_baseline and cli.exists():
        from oracle import ref_build
        import concurrent.futures as cf
        cores = ref_build.host_cores()[0]
        n = 8 * cores

        def one(i):
            row = [int(v) for v in vals[i % B]]
            return ref_build.run_cli64(cli, json.dumps({"inputs": [str(x) for x in row]}), Path(d) / ("c%d.wtns" % i)).returncode
        t1 = time.perf_counter()
        with cf.ThreadPoolExecutor(max_workers=cores) as ex:
            rcs = list(ex.map(one, range(n)))
        wall = time.perf_counter() - t1
        cpu = {"value": n / wall, "unit": "witnesses/s", "cores": cores, "kind": "reference",
               "sample": "%d runs of `./%s input.json out.wtns` (the reference's 64-bit runtime), %d at a time, wall %.1f s; process start "
                         "included: this runtime has no in-process loop build" % (n, name, cores, wall), "failed_runs": sum(1 for r_ in rcs if r_)}
    total = B * args.steps
    print(json.dumps({
        "metric": "witnesses/sec (batched inputs)", "value": total / elapsed, "unit": "witnesses/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64 (Goldilocks: 2^64 = 2^32 - 1 reduction)", "data": "synthetic",
        "config": {"workload": "poseidon2 goldilocks --O0 (%d constraints), batch=%d" % (circ.n_constraints, B), "n_signals": circ.n_signals,
                   "n_witness": n_wit, "engine": "64-bit runtime on the device (csrc/cw64.hip)", "in_flight": n_fl, "compile_s": compile_s},
        "roofline": roof, "roofline_r1cs": roof_chk, "isolated": {"kernels_ms": iso}, "parity": par, "parity_checked": par["parity_checked"],
        "failed_instances": 0, "cpu_baseline": cpu}))
    for b_ in batches:
        b_.close()
    circ.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("CW_WORKLOAD", "sha256_2048"))
    ap.add_argument("--batch", type=int, default=int(os.environ.get("CW_BATCH", "0")), help="instances per GPU")
    ap.add_argument("--total-batch", type=int, default=0, help="instances of the whole job (strong scaling)")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="with --total-batch on ONE GPU: run the shard rank 0 of a job of this many GPUs would own "
                         "(BASELINE config 4 = --workload semaphore20p --total-batch 8192 --shard-of 8 -> 1024 instances)")
    ap.add_argument("--host-only", action="store_true",
                    help="CPU rehearsal of the N-rank launch (gloo, host-only batches: inputs are staged and validated through "
                         "the C ABI, nothing computes - there is no CPU fallback); used by tests/test_bench_spawn.py")
    ap.add_argument("--cache-dir", default=os.environ.get("CW_CACHE", ""))
    ap.add_argument("--packed-inputs", action="store_true",
                    help="boolean circuits: the batch's inputs enter as packed masks (cw_set_inputs_bits_device, one bit per "
                         "input and instance) in the timed steps too; the canonical 32-byte image is never materialised")
    ap.add_argument("--parity-instances", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default=os.environ.get("CW_BENCH_GRAPH", "auto"),
                    help="steps as one HIP-graph launch each (cw_run_check); auto: when a step alone is shorter than 1 ms")
    ap.add_argument("--cu-partitions", type=int, default=int(os.environ.get("CW_BENCH_CU_PARTITIONS", "0")),
                    help="streams of the batches in flight are created with CU masks (hipExtStreamCreateWithCUMask): partition j = CUs "
                         "[j * 256 / N, (j + 1) * 256 / N) - every batch in flight gets its own part of the chip and its own hardware queue "
                         "(an experiment, measured WORSE on the Semaphore shard: 509 K -> 357 K witnesses/s with 16 parts - the wide check "
                         "kernel is confined to 16 CUs as well; profiles/r06x_sema_shard_cu_masked_streams.txt)")
    ap.add_argument("--no-small", action="store_true", help="skip the batch-4096 side measurement (profiling runs)")
    ap.add_argument("--fp-bench-lanes", type=int, default=1 << 24)
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("CW_IN_FLIGHT", "0")),
                    help="batches in flight on separate HIP streams (consecutive steps alternate between them, so the R1CS "
                         "check of one step overlaps the evaluation of the next); 1 = strictly sequential steps")
    args = ap.parse_args()

    # `python bench.py --gpus N` started by hand (no launcher environment): start the N ranks ourselves, exactly the way
    # the driver does (one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1), and pass their output on
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    # Small batches of the 256-bit engine are run many at a time (each a dependency chain of milliseconds on a few CUs).  The HIP
    # runtime maps its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with 4, any number of batches in flight gives 4
    # launches side by side (profiles/r06p_*: 8 .. 32 in flight, 16 / 32 / 64 lanes per wave - all 3.9 ms per 1 024-instance
    # Semaphore batch); with 16 queues, full waves and 32 batches in flight the same shard runs at 2.0 ms per batch
    # (profiles/r06q_*).  Must be in the environment before the runtime initialises; the metric's own line keeps the default.
    if args.workload != "sha256_2048" and "GPU_MAX_HW_QUEUES" not in os.environ:
        os.environ["GPU_MAX_HW_QUEUES"] = "16"

    import numpy as np
    import torch
    from circom_amd import runtime as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not os.environ.get("CW_FORCE_DIST"):
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to report a line whose "
                         "n_gpus would not be what was asked for" % (args.gpus, world))
    if args.host_only:
        return host_only_rehearsal(args, world, rank)
    if args.workload == "poseidon2_goldilocks":
        assert world == 1, "the goldilocks line is a single-GPU measurement"
        return goldilocks_bench(args)
    dist = None
    if world > 1 or os.environ.get("CW_FORCE_DIST"):      # CW_FORCE_DIST: exercise the RCCL path on a single GPU
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    scaling = "weak"
    B = args.batch or DEFAULT_BATCH.get(args.workload, 4096)
    if args.total_batch:
        scaling = "strong"
        from circom_amd.sharding import shard_range
        if args.shard_of:
            assert world == 1, "--shard-of describes a single-GPU run of one shard of a larger job"
            lo, hi = shard_range(args.total_batch, 0, args.shard_of)
        else:
            lo, hi = shard_range(args.total_batch, rank, world)
        B = hi - lo
    cache_root = args.cache_dir or (str(ROOT / "gpurun_in" / "cache") if (ROOT / "gpurun_in" / "cache").is_dir()
                                    else os.path.join(tempfile.gettempdir(), "cw_bench_cache_%d" % os.getuid()))
    cp, compile_s, compile_cached = get_compiled(args.workload, B, cache_root, rank, dist)
    circ = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    if args.workload == "sha256_2048":
        assert circ.n_constraints >= 1_000_000, "the metric's workload must have >= 1M constraints"
    stream = torch.cuda.current_stream()
    try:
        batch = circ.batch(B, device=local_rank, stream=stream.cuda_stream)
    except rt.CwError:
        if args.batch or args.total_batch or B <= 1024:
            raise
        B //= 2                                              # the value table did not fit: halve the batch once
        batch = circ.batch(B, device=local_rank, stream=stream.cuda_stream)
    # A shard on the 256-bit engine whose tables fit the HBM many times over: the library spreads ONE such batch over the chip by
    # leaving lanes idle (16 instances per workgroup); with many batches in flight full waves are the better use of a CU
    many_small = False
    if (not batch.bitmode and args.in_flight <= 0 and "CW_LANES" not in os.environ and batch.lanes < 64
            and os.environ.get("GPU_MAX_HW_QUEUES") == "16"):
        est_ = 32.0 * circ.n_signals * ((B + 255) // 256 * 256) * 1.1
        wgs64_ = (B + 63) // 64
        n_want_ = max(2, min(32, 512 // wgs64_))               # batches in flight that cover the 256 CUs twice with full waves
        # (1 024 instances: 32 in flight, 171 K -> 489 K witnesses/s; 8 192: 4 in flight, 455 K -> 673 K, profiles/r06ag_*.  The ECDSA
        # verifier's 81 GB tables fit three times: it keeps 16 lanes per wave.)
        if 0.8 * torch.cuda.get_device_properties(dev).total_memory / est_ >= n_want_:
            batch.close()
            os.environ["CW_LANES"] = "64"
            batch = circ.batch(B, device=local_rank, stream=stream.cuda_stream)
            many_small = n_want_
    golden_at = {}
    # synthetic inputs, resident in HBM before the timed region (different seed per rank = different shard)
    big_bool = batch.bitmode and args.workload.startswith("sha256_") and B * circ.n_inputs * 32 > (8 << 30)
    if big_bool and B * circ.n_inputs * 32 > 0.5 * torch.cuda.get_device_properties(dev).total_memory:
        args.packed_inputs = True      # (the canonical image does not fit beside the table: 453 GB for 2^19 x 27 008 inputs)
    if args.packed_inputs or big_bool:
        assert batch.bitmode and args.workload.startswith("sha256_"), "--packed-inputs needs a bit-plane circuit"
        # random message bits from the device's generator (2 M x 2 048 bits: seconds on the host); the host keeps one byte per
        # bit (BoolInputs indexes like the 32-byte image), the 32-byte image itself - 137 GB for the default batch - only
        # ever exists in HBM
        gen = torch.Generator(device=dev)
        gen.manual_seed(1 + rank)
        d_bits = torch.randint(0, 2, (B, circ.n_inputs), dtype=torch.uint8, device=dev, generator=gen)
        gm = golden_messages(args.workload) if rank == 0 else []
        for k_, (msg_, sha_, len_) in enumerate(gm):
            pos_ = 0 if k_ == 0 else B - k_                 # the first golden at instance 0, the others at the end of the batch
            d_bits[pos_] = torch.from_numpy(np.unpackbits(np.frombuffer(msg_, dtype=np.uint8))).to(dev)
            golden_at[pos_] = (msg_, sha_, len_)
        h_in = BoolInputs(d_bits.cpu().numpy())
        if args.packed_inputs:
            main_masks = h_in.masks()
            d_in = torch.from_numpy(main_masks.view(np.int64)).to(dev)
            set_in = lambda b_: b_.set_inputs_bits_device(d_in.data_ptr())
        else:
            d_in = torch.zeros((B, circ.n_inputs, 32), dtype=torch.uint8, device=dev)
            d_in[:, :, 0] = d_bits
            set_in = lambda b_: b_.set_inputs_device(d_in.data_ptr())
        del d_bits
    else:
        h_in = synth_inputs(args.workload, circ.q, B, circ.n_inputs, seed=1 + rank)
        d_in = torch.from_numpy(h_in).to(dev)
        set_in = lambda b_: b_.set_inputs_device(d_in.data_ptr())
    set_in(batch)
    # Steps are independent batches, so consecutive steps may overlap: `in_flight` batch objects (own tables, own HIP
    # stream each) take the steps in turn; the evaluation of step k+1 (vector-memory / issue bound) runs while the R1CS check
    # of step k (scalar-load / latency bound) is still going.  Every step is a complete pass: ingest + evaluation + check of
    # all B instances.  The first warm-up step runs alone on the first batch: the `isolated` timings of the JSON line.
    # Default (0): two; a batch whose workgroups cover a quarter of the chip or less (a small shard on the 256-bit engine: one
    # workgroup per 16 instances, each a dependency chain of tens of milliseconds on ONE CU) gets as many batches in flight
    # as fill the 256 CUs twice, at most eight - measured on the 1 024-instance Semaphore shard: 2 -> 63.6 K, 4 -> 112.9 K,
    # 8 -> 113.2 K witnesses/s.
    n_fl = args.in_flight
    if n_fl <= 0:
        n_fl = in_flight_for(batch)
    if many_small:
        n_fl = many_small
    n_fl = max(1, n_fl)
    if not batch.bitmode:
        # value tables of a million-signal circuit are tens of GB each (ECDSA verifier x 1 024: 81 GB): as many batches in
        # flight as fit the HBM
        est = 32.0 * circ.n_signals * ((B + 255) // 256 * 256) * 1.1
        # (a batch that does not get its table anymore is caught below: fewer in flight.  ECDSA verifier x 1 024: three tables
        # of 81 GB fit the 288 GB; measured 174 witnesses/s with one batch, 268 with two, 389 with three in flight)
        cap = float(os.environ.get("CW_BENCH_HBM_CAP", "0")) or 0.95 * torch.cuda.get_device_properties(dev).total_memory
        n_fl = max(1, min(n_fl, int(cap // max(est, 1.0))))
    streams, batches = [stream], [batch]
    masked = None
    if args.cu_partitions > 0:
        # CU-masked streams through the HIP runtime this process already uses (torch's copy of libamdhip64)
        import ctypes as C_
        hip_path = sorted({l.split()[-1] for l in open("/proc/self/maps").read().splitlines() if "libamdhip64" in l})[0]
        hip_ = C_.CDLL(hip_path)
        hip_.hipExtStreamCreateWithCUMask.argtypes = [C_.POINTER(C_.c_void_p), C_.c_uint32, C_.POINTER(C_.c_uint32)]
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        per = n_cu // args.cu_partitions

        def masked(j):
            words = (C_.c_uint32 * ((n_cu + 31) // 32))()
            for cu in range((j % args.cu_partitions) * per, (j % args.cu_partitions + 1) * per):
                words[cu // 32] |= 1 << (cu % 32)
            h_ = C_.c_void_p()
            rc_ = hip_.hipExtStreamCreateWithCUMask(C_.byref(h_), len(words), words)
            assert rc_ == 0, "hipExtStreamCreateWithCUMask failed: %d" % rc_
            return torch.cuda.ExternalStream(h_.value, device=dev)
        # (the first batch was created on the current stream: it moves to a masked stream too)
        batch.close()
        streams[0] = stream = masked(0)
        batches[0] = batch = circ.batch(B, device=local_rank, stream=stream.cuda_stream)
        set_in(batch)
    for j_ in range(1, n_fl):
        try:
            st_ = masked(j_) if masked else torch.cuda.Stream(device=dev)
            b_ = circ.batch(B, device=local_rank, stream=st_.cuda_stream)
        except rt.CwError:
            break                                            # no room for another table: fewer batches in flight
        set_in(b_)
        streams.append(st_)
        batches.append(b_)
    n_fl = len(batches)
    # HIP events around the parts of cw_run / cw_check_r1cs ON the stream each batch launches its kernels on (the C ABI records
    # them: cw_batch_set_timing / cw_batch_kernel_ms) - the kernels' own intervals, alone and inside the timed region
    for b_ in batches:
        b_.set_timing(True)

    def step(i, ev=None):
        b, s_ = batches[i % n_fl], streams[i % n_fl]
        if ev:
            ev[0].record(s_)
        b.run()
        if ev:
            ev[1].record(s_)
        b.check_r1cs()
        if ev:
            ev[2].record(s_)

    iso = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    step(0, None)
    torch.cuda.synchronize()
    step(0, iso)
    torch.cuda.synchronize()
    isolated = {"eval_ms": iso[0].elapsed_time(iso[1]), "r1cs_check_ms": iso[1].elapsed_time(iso[2]),
                "ms_per_step": iso[0].elapsed_time(iso[2])}
    isolated["kernels_ms"] = batch.kernel_ms()           # {ingest (+ table init), eval, check}: each part's own event pair
    for i in range(max(args.warmup, n_fl)):
        step(i)
    torch.cuda.synchronize()
    # A step of a small batch is ~10 launches of microseconds each + the event marks (Sha256(512) x 4 096: 0.12 ms per step with 4, 8 or
    # 16 batches in flight, profiles/r06q_*; as graph steps 0.10 ms - the rest is the device's).  Such steps run as ONE HIP-graph launch each
    # (cw_run_check: captured on the batch's second call, replayed afterwards; GPU tests: tests/test_run_check_graph.py).  Event
    # marks cannot ride in a graph: the kernels' in-step durations then come from a second, plain pass behind the timed region.
    use_graph = args.graph == "on" or (args.graph == "auto" and isolated["ms_per_step"] < 1.0 and circ.n_constraints > 0)
    if use_graph:
        for b_ in batches:
            b_.set_timing(False)
        for _ in range(3):                                    # plain, capture, first replay
            for b_ in batches:
                b_.run_check()
        torch.cuda.synchronize()
        use_graph = all(b_.graph_captured for b_ in batches)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    if use_graph:
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(args.steps):
            batches[s % n_fl].run_check()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        for b_ in batches:
            b_.set_timing("history")
        for s in range(min(args.steps, 64 * n_fl)):           # the plain pass: in-step kernel durations, generation / check split
            step(s, evs[s])
        torch.cuda.synchronize()
        evs = evs[:min(args.steps, 64 * n_fl)]
    else:
        # from here on every run of every batch keeps its own marks (cw_batch_set_timing(b, 2), a ring of 64): the kernels' durations
        # INSIDE the timed region are averages over all its steps - what a rocprofv3 kernel trace of the region averages to
        for b_ in batches:
            b_.set_timing("history")
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(args.steps):
            step(s, evs[s])
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the parts' intervals averaged over EVERY timed step (each batch in flight keeps the marks of its last 64 runs; the steps
    # ran beside each other)
    in_step_k = [b_.kernel_ms_mean() for b_ in batches[:min(n_fl, args.steps)]]
    in_step, in_step_runs = {}, {}
    for k in ("ingest", "eval", "check"):
        tot = sum(m_[k] * n_[k] for m_, n_ in in_step_k if m_[k] is not None)
        cnt = sum(n_[k] for m_, n_ in in_step_k if m_[k] is not None)
        in_step[k] = tot / cnt if cnt else None
        in_step_runs[k] = cnt
    for b_ in batches:
        b_.set_timing(True)                                  # the side measurements below read the LAST run again
    for b_ in batches[1:]:                                   # every batch in flight computed the same instances
        assert (b_.status() == batch.status()).all()

    # correctness gate + the one data-path collective: gather per-instance status words on rank 0
    # (status words + public signals of every instance; full witnesses stay on the GPU that computed them)
    from circom_amd.sharding import gather_status, gather_public
    status = gather_status(torch.from_numpy(batch.status().astype(np.int32)).to(dev), dist, rank, world)
    n_bad = int((status != 0).sum().item()) if rank == 0 else 0
    n_total = int(status.numel()) if rank == 0 else 0
    pub = torch.empty((B, circ.n_public, 32), dtype=torch.uint8, device=dev)
    if circ.n_public:
        batch.public_signals_device(pub.data_ptr())
        torch.cuda.synchronize()
    pub_local = pub
    # (bit-level circuits send one bit per public signal: 2^21 SHA-256 digests are 64 MB per rank, not 17 GB)
    pub, pub_form = gather_public(pub, dist, rank, world)
    n_pub_gathered = int(pub.shape[0]) if rank == 0 else 0
    pub_bytes_per_instance = (int(pub.shape[1]) if pub_form == "bits" else 32 * circ.n_public) if rank == 0 else 0

    gen_ms = sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs)
    chk_ms = sum(e[1].elapsed_time(e[2]) for e in evs) / len(evs)

    parity = None
    if not args.no_parity:
        dg_bits = None
        if args.workload.startswith("sha256_") and circ.n_public >= 256:
            assert not bool(pub_local[:, :256, 1:].any().item()), "digest signals are not bits"
            dg_bits = pub_local[:, :256, 0].contiguous().cpu().numpy()          # reduced on the device: 256 bytes per instance
        parity = parity_check(cp, circ, batch, h_in, args.workload, args.parity_instances, dg_bits, golden_at)   # every rank checks its own shard
        parity["parity_checked"] = len(parity["instances"])

    # canonical egress: the 32-byte-per-element image a prover reads (SURVEY 8d's B_gen: what the reference's
    # writeBinWitness produces), through the chunked device-side API into two rotating buffers; a pure HBM writer.
    # Timed on a slice of the batch (the image of a 1M-signal circuit is 32 MB per instance: 2.1 TB for 65 536) alone,
    # and while the OTHER batch in flight keeps evaluating + checking on its own stream (the overlapped pipeline)
    egress = None
    if rank == 0:
        row_bytes = circ.n_witness * 32
        chunk = max(1, min(B, (1 << 30) // row_bytes))
        n_e = min(B, chunk * 8)
        bufs = [torch.empty((chunk, circ.n_witness, 32), dtype=torch.uint8, device=dev) for _ in range(2)]
        noop = lambda f, n, ptr, st: 0
        batch.stream_witnesses_device(0, min(n_e, chunk), chunk, bufs[0].data_ptr(), bufs[1].data_ptr(), noop)   # warm-up (+ resolve)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(streams[0])
        batch.stream_witnesses_device(0, n_e, chunk, bufs[0].data_ptr(), bufs[1].data_ptr(), noop)
        e1.record(streams[0])
        torch.cuda.synchronize()
        e_ms = e0.elapsed_time(e1)
        e_gbs = n_e * row_bytes / (e_ms * 1e-3) / 1e9
        step_ms = elapsed / args.steps * 1e3
        whole_ms = e_ms * B / n_e
        egress = {"instances": n_e, "extrapolated_from": n_e if n_e < B else None, "chunk_instances": chunk, "ms": e_ms, "GB/s": e_gbs, "frac_of_hbm_peak": e_gbs / HBM_PEAK_GBS,
                  "whole_batch_ms": whole_ms, "witnesses_per_s_with_egress": B / ((step_ms + whole_ms) * 1e-3)}
        if n_e < B and whole_ms < 40e3 and not args.no_small:
            # VERDICT r5 #1c: no extrapolation - the image of EVERY instance of the batch is written (two rotating buffers of
            # `chunk` instances; 2^21 instances of the 1 M-signal circuit = 68.7 TB through HBM, ~12 s)
            del bufs
            chunk = max(1, min(B, (4 << 30) // row_bytes))
            bufs = [torch.empty((chunk, circ.n_witness, 32), dtype=torch.uint8, device=dev) for _ in range(2)]
            seen_ = [0]

            def count_(first, n, ptr, st):
                seen_[0] += n
                return 0
            batch.stream_witnesses_device(0, min(B, 2 * chunk), chunk, bufs[0].data_ptr(), bufs[1].data_ptr(), noop)
            torch.cuda.synchronize()
            e0.record(streams[0])
            batch.stream_witnesses_device(0, B, chunk, bufs[0].data_ptr(), bufs[1].data_ptr(), count_)
            e1.record(streams[0])
            torch.cuda.synchronize()
            assert seen_[0] == B
            w_ms = e0.elapsed_time(e1)
            # the last instance of the last chunk against the per-instance egress (cw_get_witness)
            k_last = (B - 1) % chunk
            got_ = bufs[((B - 1) // chunk) % 2][k_last].cpu().numpy()
            full_ = np.frombuffer(batch_witness_bytes(batch, B - 1, circ.n_witness), dtype=np.uint8).reshape(circ.n_witness, 32)
            assert (got_ == full_).all(), "bulk egress differs from the per-instance egress"
            w_gbs = B * row_bytes / (w_ms * 1e-3) / 1e9
            egress.update({"sample": {"instances": n_e, "ms": e_ms, "GB/s": e_gbs}, "instances": B, "extrapolated_from": None, "chunk_instances": chunk,
                           "ms": w_ms, "GB/s": w_gbs, "frac_of_hbm_peak": w_gbs / HBM_PEAK_GBS, "whole_batch_ms": w_ms,
                           "bytes": float(B) * row_bytes, "witnesses_per_s_with_egress": B / ((step_ms + w_ms) * 1e-3)})
            n_e, e_ms = min(B, chunk * 8), None
        if e_ms is None:                                     # (the slice the overlapped run below uses, with the larger buffers)
            e0.record(streams[0])
            batch.stream_witnesses_device(0, n_e, chunk, bufs[0].data_ptr(), bufs[1].data_ptr(), noop)
            e1.record(streams[0])
            torch.cuda.synchronize()
            e_ms = e0.elapsed_time(e1)
        if n_fl > 1:
            # egress of this batch's slice on stream 0 while the other batch runs whole steps on stream 1
            k_steps = max(1, int(round(e_ms / max(step_ms, 1e-3))))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            batch.stream_witnesses_device(0, n_e, chunk, bufs[0].data_ptr(), bufs[1].data_ptr(), noop)
            for _ in range(k_steps):
                batches[1].run()
                batches[1].check_r1cs()
            torch.cuda.synchronize()
            both_ms = (time.perf_counter() - t1) * 1e3
            egress["overlapped"] = {"egress_instances": n_e, "steps_alongside": k_steps, "wall_ms": both_ms,
                                    "sum_of_parts_ms": e_ms + k_steps * step_ms,
                                    "egress_GB/s_while_evaluating": n_e * row_bytes / (both_ms * 1e-3) / 1e9}
        del bufs

    # packed boolean inputs (cw_set_inputs_bits_device): one bit per input and instance instead of 32 bytes
    packed = None
    if rank == 0 and batch.bitmode and args.workload.startswith("sha256_") and not args.packed_inputs:
        masks = (h_in if isinstance(h_in, BoolInputs) else BoolInputs(np.ascontiguousarray(h_in[:, :, 0]))).masks()
        d_masks = torch.from_numpy(masks.view(np.int64)).to(dev)
        for b_ in batches:
            b_.set_inputs_bits_device(d_masks.data_ptr())
        for i in range(n_fl + 1):
            step(i)
        torch.cuda.synchronize()
        iso2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        step(0, iso2)                                       # one step alone: evaluation without the 32-byte ingest
        torch.cuda.synchronize()
        isolated["eval_only_ms"] = iso2[0].elapsed_time(iso2[1])
        isolated["ingest_ms"] = max(isolated["eval_ms"] - isolated["eval_only_ms"], 1e-3)
        pe = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
        t1 = time.perf_counter()
        for s_ in range(args.steps):
            step(s_, pe[s_])
        torch.cuda.synchronize()
        p_el = time.perf_counter() - t1
        assert (batch.status() == 0).all()
        pub2 = torch.empty((B, circ.n_public, 32), dtype=torch.uint8, device=dev)
        batch.public_signals_device(pub2.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(pub2, pub_local), "packed inputs gave other public signals than the 32-byte inputs"
        packed = {"ms_per_step": p_el / args.steps * 1e3, "witnesses_per_s": B * args.steps / p_el,
                  "eval_ms": sum(e[0].elapsed_time(e[1]) for e in pe) / args.steps,
                  "input_bytes_per_step": int(masks.nbytes), "canonical_input_bytes_per_step": int(h_in.nbytes)}
        for b_ in batches:
            set_in(b_)
        del pub2

    # the stand-alone audit of an emitted-code batch (what a sceptical caller runs: CW_R1CS_AUDIT=1 re-checks every constraint
    # from the table instead of trusting the fused check): the emitted audit program, and the general kernels for comparison
    audit = None
    if rank == 0 and batch.bitmode and batch.jit and not args.no_small:
        audit = {}
        for key_, envs in (("emitted_audit_ms", {"CW_R1CS_AUDIT": "1"}), ("general_kernels_ms", {"CW_R1CS_AUDIT": "1", "CW_R1CS_AUDIT_GENERAL": "1"})):
            os.environ.update(envs)
            try:
                batch.check_r1cs(); batch.sync()
                batch.check_r1cs(); batch.sync()
                audit[key_] = batch.kernel_ms()["check"]
            finally:
                for k_ in envs:
                    del os.environ[k_]
        assert (batch.status() == 0).all(), "the audit flags instances the fused check passed"

    # Fp mul/s half of the metric: 2^24 lanes x 1024 dependent Montgomery products, bn128 and bls12381
    fp_mul = {}
    if rank == 0:
        from circom_amd.field import PRIMES
        rng = np.random.default_rng(5)
        n = args.fp_bench_lanes
        a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        b = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        a[:, 31] &= 0x1F
        b[:, 31] &= 0x1F
        for pname in ("bn128", "bls12381"):
            _, ms = rt.fp_mul_bench(PRIMES[pname], a, b, 1024, device=local_rank)
            fp_mul[pname] = n * 1024 / (ms * 1e-3)
    fp_mul_per_s = fp_mul.get("bn128")

    # the batch BASELINE.json's configs use for SHA-256 (4096), measured next to the throughput batch
    small = None
    if rank == 0 and world == 1 and args.workload == "sha256_2048" and B > 4096 and not args.batch and not args.no_small:
        b2 = circ.batch(4096, device=local_rank, stream=stream.cuda_stream)
        set_in(b2)
        ev2 = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(4)]
        for e in ev2:
            e[0].record(stream); b2.run(); e[1].record(stream); b2.check_r1cs(); e[2].record(stream)
        torch.cuda.synchronize()
        g2 = sum(e[0].elapsed_time(e[1]) for e in ev2[1:]) / 3
        c2 = sum(e[1].elapsed_time(e[2]) for e in ev2[1:]) / 3
        assert (b2.status() == 0).all()
        small = {"batch": 4096, "eval_ms": g2, "r1cs_check_ms": c2, "witnesses_per_s": 4096 / ((g2 + c2) * 1e-3),
                 "lanes_per_wave": b2.lanes}
        b2.close()

    if rank == 0:
        total_witnesses = n_total * args.steps
        value = total_witnesses / elapsed
        n_in, n_wit = circ.n_inputs, circ.n_witness
        alg_gen = 32.0 * (n_in + n_wit) * B            # B_gen of SURVEY §8d, per launch: the canonical 32-byte image
        alg_chk = 32.0 * n_wit * B                     # B_chk
        # rocprofv3 figures of this exact source (profiles/traffic.json, written by tools/summarize_prof.py: kernel
        # durations from the kernel trace, HBM bytes / instruction counts / clock from the PMC passes), else null
        prof = {}
        try:
            tj = json.load(open(ROOT / "profiles" / "traffic.json"))
            ent = tj.get("%s:%d" % (args.workload, B), {})
            ks = ent.get("kernel_sources") or {}
            if ent.get("source") == source_fingerprint():
                prof = dict(ent, counters_from="profiles/traffic.json: PMC passes of this exact source")
            elif ks.get("files") and ks.get("sha") == files_fingerprint(ks["files"]):
                # the counters were taken on an earlier commit, but every file that shapes THIS workload's kernels (listed in
                # the entry) is byte-identical to what it was then
                prof = dict(ent, counters_from="profiles/traffic.json: PMC passes of %s; the kernel sources of this workload (%s) "
                                               "are unchanged since" % (ks.get("profile", "an earlier commit"), ", ".join(ks["files"])))
        except Exception:
            pass
        bits = circ.bits_info() if batch.bitmode else {}
        ek = "cw_bits_eval_kernel" if batch.bitmode else "cw_eval_kernel"
        rk = "cw_bits_r1cs_{lut,int,wide}_kernel" if batch.bitmode else "cw_r1cs_stream_kernel"
        clk = prof.get("eval_clock_hz") or NOMINAL_CLOCK_HZ
        valu_peak = N_SIMD * clk / VALU_CLK_PER_WAVE_INST     # wave64 VALU instructions per second the chip can issue
        if batch.bitmode and batch.jit:
            # Emitted code (hip_elements/bitjit.py): one wave per 2 048 instances runs the circuit as straight-line v_bitop3_b32
            # instructions and writes every distinct signal value once - the bit table IS the witness (1 bit per value and
            # instance).  Roofs: HBM for the table's bytes (plus the rows the code re-reads: its register file cannot hold a
            # 1M-signal circuit), VALU issue for its instruction stream.  Durations are HIP-event intervals of THIS run (one step
            # alone; the evaluation's own time comes from the packed-input steps, whose ingest is a 10 us copy); instruction and
            # reload counts are the emitter's (it wrote every instruction: cp.jit_stats), counter figures from profiles/ if taken
            # on this source.
            js_static = getattr(cp, "jit_stats", {}) or {}
            # what a wave EXECUTES: a looped body (one per repeated template, bitjit.lower_jit) counts once per iteration;
            # js_static keeps the code's own size
            js = dict(js_static, **(js_static.get("executed") or {}))
            chunks = (B + 2047) // 2048
            ek = "cw_bits_jit (emitted per circuit)"
            kern_ms = isolated["kernels_ms"]["eval"] or isolated.get("eval_only_ms") or isolated["eval_ms"]
            tab_bytes = 8.0 * batch.bits_slots * batch.bits_groups
            reload_bytes = 256.0 * (js.get("prefetched", 0) + js.get("late_loads", 0)) * chunks
            insts = float(js.get("instructions", 0)) * chunks
            # VALU instructions only (what the 2-clock issue roof is a roof of): SQ_INSTS_VALU when profiles/ holds the counter
            # of this source, else the emitter's own count (every instruction that is not a memory instruction or a wait)
            non_valu = sum(js.get(k_, 0) for k_ in ("stores", "prefetched", "late_loads", "loads_for_check", "waits", "scratch_stores"))
            valu_insts = prof.get("eval_valu_insts") or float(js.get("instructions", 0) - non_valu) * chunks
            valu_src = "SQ_INSTS_VALU (profiles/)" if prof.get("eval_valu_insts") else "the emitter's count: instructions that are neither memory nor wait"
            roof_eval = {"bound": "hbm", "kernel": ek, "unit": "GB/s", "achieved": tab_bytes / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "frac": tab_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": kern_ms,
                         "kernel_ms_source": "HIP events around the emitted kernel on its stream (cw_batch_kernel_ms), one step alone; in_step_ms = "
                                             "the same interval inside the timed region, other batches in flight",
                         "in_step_ms": in_step["eval"],
                         "algorithmic_bytes_per_launch": tab_bytes,
                         "algorithmic_bytes_are": "the bit table: one 256-byte row per distinct signal value (%d rows) and chunk of 2 048 instances, written once" % batch.bits_slots,
                         "traffic": prof.get("eval"), "traffic_source": prof.get("counters_from"), "traffic_measured_in_run": False,
                         "traffic_estimate": {"table_rows_written": tab_bytes, "rows_re_read": reload_bytes,
                                              "GB/s_incl_re_reads": (tab_bytes + reload_bytes) / (kern_ms * 1e-3) / 1e9,
                                              "frac_incl_re_reads": (tab_bytes + reload_bytes) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "waves": chunks, "instructions_per_wave": js.get("instructions"), "gates_per_wave": js.get("gates"),
                         "code": {"instructions": js_static.get("instructions"), "loop": js_static.get("loop"),
                                  "code_object_bytes": js_static.get("code_bytes")},
                         "valu_issue": {"wave_insts_per_s": valu_insts / (kern_ms * 1e-3), "peak": valu_peak, "frac": valu_insts / (kern_ms * 1e-3) / valu_peak,
                                        "valu_insts_source": valu_src, "all_instructions_per_s": insts / (kern_ms * 1e-3),
                                        "lone_wave_frac": (js.get("instructions", 0) / (kern_ms * 1e-3)) / (clk / LONE_WAVE_CLK_PER_INST)},
                         "gate_evaluations_per_s": float(js.get("gates", 0)) * B / (kern_ms * 1e-3),
                         "fused_r1cs_check": {k[6:]: v for k, v in js.items() if k.startswith("check_")}}
            roof_eval = in_step_view(roof_eval, tab_bytes, HBM_PEAK_GBS, kern_ms, in_step["eval"], in_step_runs["eval"])
            te = roof_eval["traffic_estimate"]
            te["GB/s_incl_re_reads"] = (tab_bytes + reload_bytes) / (roof_eval["kernel_ms"] * 1e-3) / 1e9
            te["frac_incl_re_reads"] = te["GB/s_incl_re_reads"] / HBM_PEAK_GBS
            ms_use = roof_eval["kernel_ms"]
            roof_eval["valu_issue"].update({"wave_insts_per_s": valu_insts / (ms_use * 1e-3), "frac": valu_insts / (ms_use * 1e-3) / valu_peak,
                                            "all_instructions_per_s": insts / (ms_use * 1e-3),
                                            "lone_wave_frac": (js.get("instructions", 0) / (ms_use * 1e-3)) / (clk / LONE_WAVE_CLK_PER_INST),
                                            "isolated_frac": valu_insts / (kern_ms * 1e-3) / valu_peak})
            roof_eval["gate_evaluations_per_s"] = float(js.get("gates", 0)) * B / (ms_use * 1e-3)
            chk_kern_ms = isolated["r1cs_check_ms"]
            roof_r1cs = {"bound": "none (fused)", "kernel": "fused into cw_bits_jit; cw_bits_r1cs_* audit only the groups it flags", "kernel_ms": chk_kern_ms,
                         "frac": None, "traffic": prof.get("r1cs"), "stand_alone_audit": audit,
                         "note": "every non-trivial constraint is evaluated on the registers that hold its wires while the witness is generated "
                                 "(SURVEY 8d: B_chk -> 0); the stand-alone kernels remain as the audit (CW_R1CS_AUDIT=1, or after cw_device_bits)"}
            roof_valu = {"bound": "valu", "kernel": ek, "unit": "VALU wave-instructions/s", "achieved": valu_insts / (ms_use * 1e-3), "peak": valu_peak,
                         "frac": valu_insts / (ms_use * 1e-3) / valu_peak, "valu_insts_source": valu_src, "kernel_ms": ms_use, "clock_hz": clk,
                         "isolated": {"kernel_ms": kern_ms, "achieved": valu_insts / (kern_ms * 1e-3), "frac": valu_insts / (kern_ms * 1e-3) / valu_peak}}
            ing_ms = (isolated["kernels_ms"]["ingest"] if not args.packed_inputs else None) or isolated.get("ingest_ms")
            roof_ingest = None if not ing_ms else {"bound": "hbm", "kernel": "cw_bits_ingest_kernel", "unit": "GB/s", "achieved": 32.0 * n_in * B / (ing_ms * 1e-3) / 1e9,
                           "peak": HBM_PEAK_GBS, "frac": 32.0 * n_in * B / (ing_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": ing_ms,
                           "kernel_ms_source": "HIP events around table init + cw_bits_ingest_kernel on their stream (cw_batch_kernel_ms), one step "
                                               "alone; in_step_ms = the same interval inside the timed region, other batches in flight",
                           "in_step_ms": in_step["ingest"],
                           "algorithmic_bytes_per_launch": 32.0 * n_in * B,
                           "algorithmic_bytes_are": "the boundary's input image: 32 bytes per input signal and instance, read once",
                           "traffic": prof.get("ingest"), "traffic_source": prof.get("counters_from"), "traffic_measured_in_run": False}
            if roof_ingest:
                roof_ingest = in_step_view(roof_ingest, 32.0 * n_in * B, HBM_PEAK_GBS, ing_ms, in_step["ingest"], in_step_runs["ingest"])
        elif batch.bitmode:
            # The bit-plane engine holds ONE BIT per distinct signal value and instance: its kernels neither read nor write
            # the 32-byte image, so SURVEY 8d's byte roof does not bind them (round 2 divided the image's bytes by their
            # time and reported "fractions" of 53 and 96).  Their roofs: instruction issue for the evaluation, the scalar /
            # vector issue mix for the check; the image is priced where it is produced (`value_canonical`, `canonical_egress`).
            waves = ((B + 63) // 64) * (64 // batch.lanes)
            valu_insts = prof.get("eval_valu_insts") or float(BITS_VALU_PER_VROW) * bits["vrows"] * waves
            kern_ms = prof.get("eval_avg_us", 0.0) / 1e3 or isolated.get("eval_only_ms") or isolated["eval_ms"]
            roof_eval = {"bound": "valu", "kernel": ek, "unit": "wave-instructions/s", "achieved": valu_insts / (kern_ms * 1e-3),
                         "peak": valu_peak, "frac": valu_insts / (kern_ms * 1e-3) / valu_peak,
                         "traffic": prof.get("eval"), "kernel_ms": kern_ms,
                         "kernel_ms_source": "rocprofv3 kernel trace (profiles/)" if prof.get("eval_avg_us") else "HIP events, one step alone",
                         "valu_insts_source": "SQ_INSTS_VALU (profiles/)" if prof.get("eval_valu_insts") else
                         "%d VALU per vrow and wave (disassembly) x vrows x waves" % BITS_VALU_PER_VROW,
                         "clock_hz": clk, "waves": waves,
                         "all_instructions_per_vrow": BITS_INSTS_PER_VROW,
                         "lone_wave_issue_frac": BITS_INSTS_PER_VROW * bits["vrows"] / (kern_ms * 1e-3) / (clk / LONE_WAVE_CLK_PER_INST),
                         "gate_evaluations_per_s": bits["gate_lanes"] * B / (kern_ms * 1e-3)}
            # HBM view of the same launch: bytes the program must move (records once per XCD at best, the table's rows once)
            tab_bytes = 8.0 * bits["slots_per_group"] * ((B + 63) // 64)
            roof_eval["hbm"] = {"bit_table_bytes": tab_bytes, "algorithmic_bytes": tab_bytes,
                                "achieved_GB/s": tab_bytes / (kern_ms * 1e-3) / 1e9, "frac": tab_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "counter_bytes": prof.get("eval"),
                                "counter_frac": (prof["eval"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if prof.get("eval") else None}
            chk_kern_ms = prof.get("r1cs_avg_us", 0.0) / 1e3 or isolated["r1cs_check_ms"]
            chk_insts = prof.get("r1cs_valu_insts")
            roof_r1cs = {"bound": "valu", "kernel": rk, "unit": "wave-instructions/s",
                         "achieved": chk_insts / (chk_kern_ms * 1e-3) if chk_insts else None, "peak": valu_peak,
                         "frac": chk_insts / (chk_kern_ms * 1e-3) / valu_peak if chk_insts else None,
                         "traffic": prof.get("r1cs"), "kernel_ms": chk_kern_ms,
                         "hbm": {"algorithmic_bytes": tab_bytes, "achieved_GB/s": tab_bytes / (chk_kern_ms * 1e-3) / 1e9,
                                 "frac": tab_bytes / (chk_kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "counter_bytes": prof.get("r1cs")}}
            roof_valu = roof_eval
            ing_ms = prof.get("ingest_avg_us", 0.0) / 1e3 or isolated.get("ingest_ms")
            roof_ingest = None if not ing_ms else {"bound": "hbm", "kernel": "cw_bits_ingest_kernel", "unit": "GB/s", "achieved": 32.0 * n_in * B / (ing_ms * 1e-3) / 1e9,
                           "peak": HBM_PEAK_GBS, "frac": 32.0 * n_in * B / (ing_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": ing_ms,
                           "algorithmic_bytes_per_launch": 32.0 * n_in * B}
        else:
            if getattr(batch, "emitted", False):
                # the rows of the strand variant as emitted code (hip_elements/fpjit.py); with the fused check the kernel also
                # recomputes the R1CS rows it covers (about as much arithmetic again: Fp-mul/s below counts the evaluation's only)
                ek = "cw_fp_jit (emitted per circuit%s)" % (", R1CS check fused" if batch.fused_check else "")
                if batch.fused_check:
                    rk = "fused into cw_fp_jit; cw_r1cs_stream_kernel on the rows the code leaves + merge"
            gen_k = prof.get("eval_avg_us", 0.0) / 1e3 or isolated["kernels_ms"]["eval"] or isolated["eval_ms"]
            chk_k = prof.get("r1cs_avg_us", 0.0) / 1e3 or isolated["kernels_ms"]["check"] or isolated["r1cs_check_ms"]
            gen_gbs = alg_gen / (gen_k * 1e-3) / 1e9
            # (a fused check reads no witness image: its rows are recomputed on the evaluation's registers - no byte fraction)
            chk_gbs = None if getattr(batch, "fused_check", False) else alg_chk / (chk_k * 1e-3) / 1e9
            roof_eval = {"bound": "hbm", "kernel": ek, "achieved": gen_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gen_gbs / HBM_PEAK_GBS, "traffic": prof.get("eval"), "algorithmic_bytes_per_launch": alg_gen,
                         "kernel_ms": gen_k, "in_step_ms": in_step["eval"], "strands": batch.strands, "lanes_per_workgroup": batch.lanes,
                         "kernel_ms_source": "rocprofv3 kernel trace (profiles/)" if prof.get("eval_avg_us") else
                         "HIP events around the evaluation kernel on its stream (cw_batch_kernel_ms), one step alone",
                         "emitted_code": ([e for e in getattr(cp, "fpjit_stats", []) if e.get("n_strands") == batch.strands and
                                           bool(e.get("constraints_fused")) == bool(batch.fused_check)] or [None])[0]
                         if getattr(batch, "emitted", False) else None}
            roof_r1cs = {"bound": "hbm", "kernel": rk, "achieved": chk_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None if chk_gbs is None else chk_gbs / HBM_PEAK_GBS, "traffic": prof.get("r1cs"),
                         "algorithmic_bytes_per_launch": None if chk_gbs is None else alg_chk,
                         "kernel_ms": chk_k,
                         # the check of an arithmetic circuit is bound by instruction issue, not by bytes (Poseidon(2): 7.1e8 wave
                         # instructions per launch, most of them half-rate 32-bit multiplies): fraction of the 2-clock VALU peak
                         "valu_issue_frac": (prof["r1cs_valu_insts"] / (chk_k * 1e-3) / valu_peak) if prof.get("r1cs_valu_insts") else None}
            if not prof.get("eval_avg_us"):
                roof_eval = in_step_view(roof_eval, alg_gen, HBM_PEAK_GBS, gen_k, in_step["eval"], in_step_runs["eval"])
            if chk_gbs is not None and not prof.get("r1cs_avg_us"):
                roof_r1cs = in_step_view(roof_r1cs, alg_chk, HBM_PEAK_GBS, chk_k, in_step["check"], in_step_runs["check"])
            gen_iso, gen_k = gen_k, roof_eval["kernel_ms"]
            # Fp products per second: with several batches in flight the kernels of different steps run beside each other, so
            # the launch's own in-step duration is stretched by its neighbours - the rate the GPU SUSTAINS over the timed
            # region is products per step / ms_per_step (the driver's clock); `isolated` = the kernel running alone
            step_ms_fp = elapsed / args.steps * 1e3
            fpk = circ.n_mmul * B / (step_ms_fp * 1e-3)
            roof_valu = {"bound": "valu", "kernel": ek, "unit": "Fp-mul/s", "achieved": fpk, "peak": fp_mul_per_s,
                         "frac": fpk / fp_mul_per_s if fp_mul_per_s else None, "ms": step_ms_fp,
                         "is": "the evaluation's Fp products of one step / ms_per_step (whole timed region, %d batch(es) in flight)" % n_fl,
                         "in_step_kernel": {"kernel_ms": gen_k, "achieved": circ.n_mmul * B / (gen_k * 1e-3),
                                            "frac": circ.n_mmul * B / (gen_k * 1e-3) / fp_mul_per_s if fp_mul_per_s else None},
                         "isolated": {"kernel_ms": gen_iso, "achieved": circ.n_mmul * B / (gen_iso * 1e-3),
                                      "frac": circ.n_mmul * B / (gen_iso * 1e-3) / fp_mul_per_s if fp_mul_per_s else None},
                         "valu_wave_insts_per_s": (prof["eval_valu_insts"] / (gen_k * 1e-3)) if prof.get("eval_valu_insts") else None,
                         "valu_issue_frac": (prof["eval_valu_insts"] / (gen_k * 1e-3) / valu_peak) if prof.get("eval_valu_insts") else None}
            if batch.fused_check:
                # The emitted program ALSO performs the check's products (the R1CS check is recomputed behind the rows: every term
                # with a coefficient other than +-1 on a wire other than the constant, and one product per quadratic row - what the
                # stand-alone kernel multiplies too).  `frac` above counts the evaluation's products only; this is the
                # arithmetic the kernel really does per second against the same peak.
                q_ = circ.q
                n_chk = 0
                for a_, b_, c_ in cp.flat.constraints:
                    n_chk += sum(1 for part in (a_, b_, c_) for w_, co_ in part.items() if w_ != 0 and co_ % q_ not in (1, q_ - 1))
                    n_chk += 1 if (a_ and b_) else 0
                tot_ = (circ.n_mmul + n_chk) * B / (step_ms_fp * 1e-3)
                roof_valu["incl_check"] = {"check_products_per_witness": n_chk, "achieved": tot_,
                                           "frac": tot_ / fp_mul_per_s if fp_mul_per_s else None,
                                           "is": "evaluation + check products of one step (the check recomputed inside the emitted program; rows it "
                                                 "leaves are multiplied by the stand-alone kernel in the same step) / ms_per_step"}
            roof_ingest = None
        # What ONE STEP achieves (VERDICT r5 #1a): the bytes a step must move - the boundary's input image, and everything its
        # kernels move by the counters (or by the emitter's own count) - over ms_per_step of the timed region, against the HBM spec
        step_ms_ = elapsed / args.steps * 1e3
        in_bytes = 32.0 * n_in * B
        if batch.bitmode and batch.jit:
            all_bytes_est = (0.0 if args.packed_inputs else in_bytes) + tab_bytes + reload_bytes
            cnt = [prof.get(k_) for k_ in (("eval",) if args.packed_inputs else ("ingest", "eval"))]
            all_bytes_cnt = sum(cnt) if all(c_ is not None for c_ in cnt) else None
        elif batch.bitmode:
            all_bytes_est, all_bytes_cnt = in_bytes + 2 * tab_bytes, None
        else:
            all_bytes_est = 32.0 * (n_in + circ.n_signals) * B * (1.0 if getattr(batch, "fused_check", False) else 2.0)
            cnt = [prof.get(k_) for k_ in ("eval", "r1cs")]
            all_bytes_cnt = sum(cnt) if all(c_ is not None for c_ in cnt) else None
        iso_parts = [v_ for v_ in (isolated["kernels_ms"] or {}).values() if v_]
        step_obj = {"ms_per_step": step_ms_, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "input_bytes": in_bytes, "input_GB/s": in_bytes / (step_ms_ * 1e-3) / 1e9,
                    "input_frac": in_bytes / (step_ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "all_traffic_bytes": all_bytes_cnt or all_bytes_est,
                    "all_traffic_source": ("PMC counters (profiles/traffic.json), ingest + evaluation" if all_bytes_cnt else
                                           "estimate: input image + value / bit table written" + (" + the rows the emitted code re-reads (its own count)" if batch.bitmode and batch.jit else "")),
                    "all_traffic_GB/s": (all_bytes_cnt or all_bytes_est) / (step_ms_ * 1e-3) / 1e9,
                    "all_traffic_frac": (all_bytes_cnt or all_bytes_est) / (step_ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "sum_of_parts_alone_ms": sum(iso_parts) if iso_parts else None,
                    "in_step_kernel_ms": in_step, "in_step_runs": in_step_runs,
                    "is": "bytes per step / ms_per_step (the driver's clock) / 8 TB/s; sum_of_parts_alone_ms = the step's kernels each running "
                          "alone, added up: a step below it overlaps its parts"}
        # The Fp-mul half of the metric against ITS roof (VERDICT r5 #1b): a product is 266 VALU instructions, 162 of them the
        # half-rate v_mad_u64_u32; peak = 64 lanes / (162 / rate(v_mad_u64_u32) + 104 / rate(full-rate VALU)), both rates measured
        # chip-wide by tools/ubench_isa (profiles/r03_ubench_isa.json)
        fp_peak = 64.0 / (FPMUL_MADS / MAD_U64_WAVE_INSTS_PER_S + (FPMUL_INSTS - FPMUL_MADS) / SIMPLE_VALU_WAVE_INSTS_PER_S)
        roof_fpmul = None if not fp_mul_per_s else {
            "bound": "valu", "kernel": "cw_mulbench_kernel (2^%d lanes x 1024 dependent Montgomery products, bn128)" % (args.fp_bench_lanes.bit_length() - 1),
            "unit": "Fp-mul/s", "achieved": fp_mul_per_s, "peak": fp_peak, "frac": fp_mul_per_s / fp_peak,
            "instructions_per_product": FPMUL_INSTS, "v_mad_u64_u32_per_product": FPMUL_MADS,
            "rates_wave_insts_per_s": {"v_mad_u64_u32": MAD_U64_WAVE_INSTS_PER_S, "full_rate_valu": SIMPLE_VALU_WAVE_INSTS_PER_S,
                                       "source": "tools/ubench_isa, profiles/r03_ubench_isa.json (8 waves per SIMD)"},
            "by_prime": {k_: {"achieved": v_, "frac": v_ / fp_peak} for k_, v_ in fp_mul.items()}}
        # witnesses per second INCLUDING the 32-byte image (SURVEY 8d's definition of a witness's bytes): the step plus the
        # egress of the whole batch, sequentially; with the egress overlapped (another batch evaluating meanwhile) the step
        # hides behind it
        value_canonical = egress["witnesses_per_s_with_egress"] if egress else None
        out = {
            "metric": "witnesses/sec (batched inputs)",
            "value": value,
            "unit": "witnesses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": ("bit-plane boolean gates (v_bitop3_b32 on u32 lanes of 32 instances, emitted per circuit; u256 Montgomery fallback for "
                      "non-boolean instances)") if batch.bitmode and batch.jit else
            "bit-plane boolean gates on u64 instance masks (u256 Montgomery fallback for non-boolean instances)" if batch.bitmode
            else "u256 (9x29-bit limbs, Montgomery multiply)",
            "data": "synthetic",
            "config": {"workload": "%s %s --O0 (%d constraints), batch=%d per GPU" % (args.workload, cp.flat.prime, circ.n_constraints, B),
                       "value_is": ("witness generated + R1CS-verified per second, resident as bit planes (1 bit per signal value and "
                                    "instance); value_canonical includes writing the 32-byte-per-element image") if batch.bitmode else
                       ("witness generated + R1CS-verified per second, resident as 32-byte field elements in the value table "
                        "([signal][instance] order" + (", Montgomery form" if circ.montgomery else "") + "); value_canonical includes "
                        "writing the reference's image ([instance][witness element], canonical residues)"),
                       "n_signals": circ.n_signals, "n_witness": n_wit, "n_constraints": circ.n_constraints,
                       "at_reference_default_O1": _o1_size(cp.flat),
                       "engine": "bit-plane, emitted gfx950 code (one wave per 2 048 instances, 1 bit per signal value per instance)" if batch.bitmode and batch.jit else
                       "bit-plane (1 bit per signal value per instance)" if batch.bitmode else
                       (("256-bit schedule as emitted gfx950 code" + (" with the R1CS check fused in" if batch.fused_check else ""))
                        if getattr(batch, "emitted", False) else "256-bit schedule, interpreted") + (", signals in Montgomery form" if circ.montgomery else ""),
                       "bit_program": bits, "emitted_code": getattr(cp, "jit_stats", None) if batch.bitmode and batch.jit else None, "r1cs_check_classes": (circ.bits_r1cs_plan_stats() if batch.bitmode else None),
                       "schedule_rows": circ.n_rows, "fp_mul_per_witness": circ.n_mmul,
                       "parallelism": "instances sharded x%d, status + public-signal gather only" % world,
                       "inputs": "packed boolean masks (8 bytes per input and 64 instances)" if args.packed_inputs else
                       "canonical 32-byte field elements",
                       "in_flight": n_fl, "step_launch": ("one HIP graph per step (cw_run_check); in-step kernel durations from a plain pass behind the timed region"
                                                          if use_graph else "plain launches (cw_run + cw_check_r1cs)"),
                       "cu_partitions": args.cu_partitions or None, "lanes_per_wave": batch.lanes, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                       "compile_s": compile_s, "compile_cached": compile_cached, "compile_s_cold": getattr(cp, "compile_s_cold", None),
                       "shard_of": args.shard_of or None, "total_batch": args.total_batch or None},
            "value_canonical": value_canonical,
            "value_canonical_hbm_frac": (egress["frac_of_hbm_peak"] if egress else None),
            # the dominant kernel of THIS run (longest measured duration)
            "roofline": dominant_roofline(roof_eval, roof_r1cs, roof_ingest, gen_ms, chk_ms, bool(batch.bitmode and batch.jit), args.packed_inputs),
            "roofline_eval": roof_eval,
            "roofline_r1cs": roof_r1cs,
            "roofline_ingest": roof_ingest,
            "step": step_obj,
            "roofline_fpmul": roof_fpmul,
            # second bound of SURVEY §8d (integer work, no MFMA): VALU issue in bit-plane mode, Fp products per second
            # against the measured Fp-multiply peak (micro-benchmark, 2^24 x 1024) for the 256-bit schedule
            "roofline_valu": roof_valu,
            "valu_peak": {"wave_insts_per_s": valu_peak, "clk_per_wave_inst_per_simd": VALU_CLK_PER_WAVE_INST, "clock_hz": clk,
                          "calibration": "tools/ubench_isa (profiles/r03_ubench_isa.json): v_bitop3/v_bfi/v_and chains reach 1 "
                                         "instruction per 2.05 clocks per SIMD from 2 waves per SIMD; ONE wave issues one per 4.1-4.6"},
            "canonical_egress": egress,
            "packed_inputs": packed,
            "batch_4096": small,
            "fp_mul_per_s": fp_mul_per_s,
            "fp_mul_per_s_by_prime": fp_mul,
            # eval_ms / r1cs_check_ms: HIP-event intervals inside the timed region (with in_flight > 1 the regions of
            # consecutive steps overlap on the GPU, so each is stretched by the other batch's kernels and they add up to
            # more than ms_per_step); `isolated`: one step running alone; roofline*.kernel_ms: the kernel's own duration
            "eval_ms": gen_ms,
            "r1cs_check_ms": chk_ms,
            "isolated": isolated,
            "in_step_kernels_ms": in_step,             # ingest / eval / check: event pairs on each batch's stream, inside the timed region
            "failed_instances": n_bad,
            "parity": parity,
            "parity_checked": parity["parity_checked"] if parity else 0,
            "gathered": {"status_words": n_total, "public_signal_rows": n_pub_gathered,
                         "public_signals_per_instance": circ.n_public, "public_signals_sent_as": pub_form,
                         "bytes_per_instance": 4 + pub_bytes_per_instance},
            "cpu_baseline": None,
        }
        if args.shard_of:
            # BASELINE config 4 is ONE job of `total_batch` instances over `shard_of` GPUs: every GPU runs its shard once.  The
            # steady-state `value` above keeps several shards in flight on this GPU; the job itself costs one shard's latency
            # (a dependency chain, the same on every rank) plus the final gather of status words and public signals.
            gather_ms = xgmi_gather_ms(pub_bytes_per_instance, B)
            one = isolated["ms_per_step"]
            out["one_shot_job"] = {"shard_instances": B, "shard_ms_alone": one, "gather_ms_assumed": gather_ms,
                                   "predicted_job_ms": one + gather_ms, "ranks": args.shard_of,
                                   "predicted_witnesses_per_s": args.total_batch / ((one + gather_ms) * 1e-3),
                                   "same_job_on_one_gpu_note": "run --total-batch %d without --shard-of for the one-GPU time of the whole job" % args.total_batch}
        if world == 1:
            # What this line predicts for the N-GPU launches the driver runs (bench.py --gpus N: the same batch on every rank,
            # no exchange inside the timed steps; status words + public signals gathered once at the end): falsifiable numbers,
            # computed from this run's own measurements.  `steady_state`: N times this value (the ranks share nothing but the
            # host).  `one_shot`: one batch per rank run ONCE - its latency alone on the GPU plus the final gather (every peer
            # sends its rows to rank 0 over its own xGMI link: bytes / 153 GB/s + ~20 us).
            one = isolated["ms_per_step"]
            out["multi_gpu_prediction"] = {
                "assumptions": "xGMI link 153 GB/s per peer -> rank 0, 20 us per gather (MI355X_MICROARCH.md); unmeasured on this 1-GPU box",
                "by_ranks": {str(n_): {"steady_state_witnesses_per_s": n_ * value,
                                        "one_shot_job_ms": one + (xgmi_gather_ms(pub_bytes_per_instance, B) if n_ > 1 else 0.0),
                                        "one_shot_witnesses_per_s": n_ * B / ((one + (xgmi_gather_ms(pub_bytes_per_instance, B) if n_ > 1 else 0.0)) * 1e-3),
                                        "gather_bytes_per_peer": (4 + pub_bytes_per_instance) * B} for n_ in (1, 2, 4, 8)},
                "latency_bound": bool(not batch.bitmode and (B + batch.lanes - 1) // batch.lanes < 2 * 256),
                "latency_bound_note": "a batch whose workgroups do not cover the chip twice runs for the length of its dependency chain whatever "
                                      "its size: sharding such a job over more GPUs does not shorten it (DESIGN 7)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cp, args.workload)
            # the Fp-mul half on the same cores: the reference's own Fr_mul in a dependent chain per core (oracle/_ref, fr_shim.cpp)
            try:
                from oracle import ref_build
                fpm = {pn: ref_build.time_fp_mul(pn, 3.0) for pn in ("bn128", "bls12381")}
                if out["cpu_baseline"] is None:
                    out["cpu_baseline"] = {"value": None, "unit": "witnesses/s", "cores": fpm["bn128"]["cores"], "kind": "reference", "sample": "witness leg unavailable"}
                out["cpu_baseline"]["fp_mul_per_s"] = fpm["bn128"]["value"]
                out["cpu_baseline"]["fp_mul"] = fpm
                if fp_mul_per_s:
                    out["cpu_baseline"]["fp_mul_gpu_over_cpu"] = fp_mul_per_s / fpm["bn128"]["value"]
            except Exception as ex:                                   # noqa: BLE001  (a report, never a reason to fail the line)
                if out["cpu_baseline"] is not None:
                    out["cpu_baseline"]["fp_mul_error"] = repr(ex)[:200]
        if world == 1 and batch.bitmode and _o1_size.witness2signal is not None and not args.no_small and not args.batch:
            # What a prover takes (VERDICT r4 #4b): the witness of the system the reference builds by DEFAULT (--O1: constant and
            # renaming substitutions), as 32-byte field elements, for EVERY instance of the batch - no sample, no extrapolation.
            # The device still generates and checks the --O0 system; cw_set_witness_list makes every egress hand out the O1
            # wires (main.cpp:288-334 writes witness2signal's entries; constraint_list/src/lib.rs:187-193 builds that map).
            try:
                for b_ in batches:
                    b_.close()
                w2s = _o1_size.witness2signal
                circ.set_witness_list(w2s)
                b1 = circ.batch(B, device=local_rank, stream=stream.cuda_stream)
                masks = (h_in if isinstance(h_in, BoolInputs) else BoolInputs(np.ascontiguousarray(h_in[:, :, 0]))).masks()
                d_masks = torch.from_numpy(masks.view(np.int64)).to(dev)
                b1.set_inputs_bits_device(d_masks.data_ptr())
                b1.run(); b1.check_r1cs(); b1.sync()
                assert (b1.status() == 0).all()
                row_bytes = len(w2s) * 32
                # launches of >= 8 groups of 64 instances let the egress kernel walk the groups fastest (cw_bits.hip,
                # tools/ubench_egress.hip: 4.8 -> 5.2-5.8 TB/s); two buffers of up to 6 GiB, whole groups
                chunk = (6 << 30) // row_bytes
                chunk = max(1, min(B, chunk // 64 * 64 if chunk >= 64 else chunk))
                bufs = [torch.empty((chunk, len(w2s), 32), dtype=torch.uint8, device=dev) for _ in range(2)]
                seen = [0]

                def count(first, n, ptr, st):
                    seen[0] += n
                    return 0
                b1.stream_witnesses_device(0, min(B, 2 * chunk), chunk, bufs[0].data_ptr(), bufs[1].data_ptr(), lambda *a: 0)   # warm-up
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t1 = time.perf_counter()
                e0.record(stream)
                b1.stream_witnesses_device(0, B, chunk, bufs[0].data_ptr(), bufs[1].data_ptr(), count)
                e1.record(stream)
                torch.cuda.synchronize()
                wall_ms = (time.perf_counter() - t1) * 1e3
                e_ms = e0.elapsed_time(e1)
                assert seen[0] == B
                # one instance of the last chunk against the full witness of the same instance (the O1 wires of the O0 image)
                k_last = (B - 1) % chunk
                got = bufs[((B - 1) // chunk) % 2][k_last].cpu().numpy()
                full = np.frombuffer(batch_witness_bytes(b1, B - 1, len(w2s)), dtype=np.uint8).reshape(len(w2s), 32)
                assert (got == full).all(), "O1 egress differs from the per-instance egress"
                step_ms = elapsed / args.steps * 1e3
                out["value_canonical_O1"] = {
                    "witnesses_per_s": B / ((step_ms + e_ms) * 1e-3), "egress_only_witnesses_per_s": B / (e_ms * 1e-3),
                    "instances": B, "extrapolated": False, "wires": int(len(w2s)), "bytes_per_witness": row_bytes,
                    "egress_ms": e_ms, "egress_wall_ms": wall_ms, "GB/s": B * row_bytes / (e_ms * 1e-3) / 1e9,
                    "frac_of_hbm_peak": B * row_bytes / (e_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "chunk_instances": chunk,
                    "is": "the reference's default-level (--O1) witness as 32-byte elements for the WHOLE batch, written chunk by chunk "
                          "into two rotating device buffers of chunk_instances each (cw_stream_witnesses_device); witnesses_per_s = the step plus this egress"}
                b1.close()
            except Exception as ex:                                   # noqa: BLE001  (a report, never a reason to fail the line)
                out["value_canonical_O1"] = {"error": repr(ex)[:300]}
        line = json.dumps(out)
    for b_ in batches:
        b_.close()
    circ.close()
    # the ONE JSON line is the last thing on stdout: RCCL prints a version banner through C stdio, which a pipe only sees when that
    # buffer is flushed - without this it lands BEHIND the line (seen with CW_FORCE_DIST=1, profiles/r05t_*).  Every rank empties
    # its buffers, all meet, then rank 0 prints.
    def flush_all():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
    flush_all()
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        flush_all()
        print(line, flush=True)


if __name__ == "__main__":
    main()
