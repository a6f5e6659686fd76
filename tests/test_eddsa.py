"""BASELINE config 4 as a parity case: Semaphore-style circuit = Poseidon Merkle inclusion (depth 20) +
EdDSA-Poseidon signature verification over BabyJubjub (circom_amd/circuits/eddsa.py), 42 784 signals and
43 275 constraints at --O0, ~2 000 field divisions per witness.

Chain of evidence: plain-integer EdDSA/Merkle (eddsa_host.py) pins the outputs -> Python oracle ->
the reference's own C++ runtime compiled from /root/reference (byte-identical .wtns) -> HIP path."""
import random

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.frontend.flatten import FlatCircuit
from circom_amd.circuits import eddsa_host as H
from circom_amd.circuits.eddsa import CompConstant, EdDSAPoseidonVerifier, SemaphoreStyle, SUBGROUP_ORDER
from circom_amd.hip_elements.lower import lower
from circom_amd.hip_elements.writers import wtns_bytes
from oracle import ref_build
from oracle.field import PRIMES
from oracle.tape_eval import eval_flat, eval_tape, check_r1cs

Q = PRIMES["bn128"]
LEVELS = 20


def _inp(fc, row):
    return {fc.main_input_start + k: v for k, v in enumerate(row)}


def test_compconstant_matches_integer_compare():
    rng = random.Random(2)
    for ct in (SUBGROUP_ORDER - 1, Q - 1, 5):
        fc = FlatCircuit(Program(CompConstant(ct)))
        for x in (0, 1, ct - 1, ct, ct + 1, (1 << 254) - 1, rng.randrange(1 << 254), rng.randrange(1 << 254)):
            x = max(x, 0)
            bits = [(x >> i) & 1 for i in range(254)]
            sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, bits))
            assert failed is None and sig[1] == int(x > ct), (ct, x)
            assert check_r1cs(Q, fc.constraints, sig) is None


def test_host_signatures_verify_and_tampering_fails():
    rng = random.Random(4)
    s, A = H.keygen(Q, rng)
    R8, S = H.sign(Q, s, A, 99, rng)
    assert H.verify(Q, A, 99, R8, S)
    assert not H.verify(Q, A, 98, R8, S)
    assert not H.verify(Q, A, 99, R8, (S + 1) % SUBGROUP_ORDER)


@pytest.fixture(scope="module")
def sem(tmp_path_factory):
    d = tmp_path_factory.mktemp("sem")
    from conftest import emit_for_gpu
    return compile_program(Program(SemaphoreStyle(LEVELS)), str(d), "semaphore20", sym=False, fpjit=emit_for_gpu())


def test_semaphore_oracle_outputs_and_r1cs(sem):
    fc = sem.flat
    assert (fc.n_signals, len(fc.constraints)) == (42784, 43275)
    rng = random.Random(7)
    row, (root, nullifier) = H.semaphore_inputs(Q, LEVELS, rng)
    sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, row))
    assert failed is None and (sig[1], sig[2]) == (root, nullifier)
    assert check_r1cs(Q, fc.constraints, sig) is None
    # the lowered schedules (1 and 4 strands, race-checked replay) give the same signals
    for S in (1, 4):
        got, st = eval_tape(lower(fc, n_strands=S), _inp(fc, row))
        assert st == 0 and got == sig
    # tampered signature / S >= subgroup order / wrong message: an `===` fails
    for k, v in ((2, (row[2] + 1) % SUBGROUP_ORDER), (2, row[2] + SUBGROUP_ORDER), (5, row[5] ^ 1), (0, row[0] ^ 1)):
        bad = list(row)
        bad[k] = v % Q
        sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, bad))
        assert failed is not None


def test_semaphore_reference_runtime_wtns_equal_oracle(sem, tmp_path, ref_dir_bn128):
    try:
        ref_build.build_circuit(sem)
    except RuntimeError as e:
        pytest.skip(str(e))
    fc = sem.flat
    rng = random.Random(11)
    rows = [H.semaphore_inputs(Q, LEVELS, rng)[0] for _ in range(3)]
    raw = b"".join(v.to_bytes(32, "little") for r in rows for v in r)
    pre = str(tmp_path / "r_")
    ref_build.run_loop(sem, raw, len(rows), 1, wtns_prefix=pre)
    for i, r in enumerate(rows):
        want, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, r))
        assert failed is None
        assert open(pre + "%d.wtns" % i, "rb").read() == wtns_bytes(Q, want), i


@pytest.mark.gpu
def test_gpu_semaphore_matches_oracle_and_reference(sem, tmp_path):
    from circom_amd import runtime as rt
    fc = sem.flat
    rng = random.Random(13)
    B = 96
    rows, outs = zip(*(H.semaphore_inputs(Q, LEVELS, rng) for _ in range(B)))
    rows = [list(r) for r in rows]
    rows[5][2] = (rows[5][2] + 1) % SUBGROUP_ORDER          # instance 5: forged signature
    rows[6][6] = 2                                          # instance 6: path index is not a bit
    c = rt.Circuit(sem.tape_path, sem.dat_path, sem.r1cs_path)
    b = c.batch(B)
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    st = b.status()
    assert st[5] & rt.ST_ASSERT_FAILED and st[6] & rt.ST_ASSERT_FAILED
    assert (np.delete(st, [5, 6]) == 0).all()
    for i in (0, 1, 50, 95):
        assert (b.signal(i, 1), b.signal(i, 2)) == outs[i], i
    for i in (3, 64):
        want, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, rows[i]))
        assert failed is None and b.witness(i) == want
    _, loop = ref_build.binaries("bn128", "semaphore20")
    if loop.exists():                                       # prebuilt by __graft_entry__.build()
        good = [r for k, r in enumerate(rows) if k not in (5, 6)][:8]
        idx = [k for k in range(B) if k not in (5, 6)][:8]
        raw = b"".join(v.to_bytes(32, "little") for r in good for v in r)
        pre = str(tmp_path / "ref_")
        ref_build.run_loop(sem, raw, len(good), 1, wtns_prefix=pre)
        for j, i in enumerate(idx):
            g = tmp_path / ("gpu_%d.wtns" % i)
            b.write_wtns(i, g)
            assert g.read_bytes() == open(pre + "%d.wtns" % j, "rb").read(), i
    b.close(); c.close()


# ---- the same relation with the scalar multiplications' witnesses computed on a projective ladder ---------------------
@pytest.fixture(scope="module")
def semp(tmp_path_factory):
    d = tmp_path_factory.mktemp("semp")
    from conftest import emit_for_gpu
    return compile_program(Program(SemaphoreStyle(LEVELS, True)), str(d), "semaphore20p", sym=False, fpjit=emit_for_gpu())


def test_projective_ladder_same_outputs_fewer_chained_inversions(semp):
    fc = semp.flat
    rng = random.Random(7)
    row, (root, nullifier) = H.semaphore_inputs(Q, LEVELS, rng)
    sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, row))
    assert failed is None and (sig[1], sig[2]) == (root, nullifier)
    assert check_r1cs(Q, fc.constraints, sig) is None
    for S in (1, 16):
        tp = lower(fc, n_strands=S)
        got, st = eval_tape(tp, _inp(fc, row))
        assert st == 0 and got == sig
        # the ~1 000 inversions of the ladder do not depend on each other: Montgomery's trick leaves one per 64
        assert tp.stats["inv"] == tp.stats["inv_batches"] <= 24
    for k, v in ((2, (row[2] + 1) % SUBGROUP_ORDER), (5, row[5] ^ 1), (0, row[0] ^ 1)):
        bad = list(row)
        bad[k] = v % Q
        assert eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, bad))[1] is not None


def test_projective_ladder_reference_runtime_wtns_equal_oracle(semp, tmp_path, ref_dir_bn128):
    try:
        ref_build.build_circuit(semp)
    except RuntimeError as e:
        pytest.skip(str(e))
    fc = semp.flat
    rng = random.Random(12)
    rows = [H.semaphore_inputs(Q, LEVELS, rng)[0] for _ in range(2)]
    raw = b"".join(v.to_bytes(32, "little") for r in rows for v in r)
    pre = str(tmp_path / "r_")
    ref_build.run_loop(semp, raw, len(rows), 1, wtns_prefix=pre)
    for i, r in enumerate(rows):
        want, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, r))
        assert failed is None
        assert open(pre + "%d.wtns" % i, "rb").read() == wtns_bytes(Q, want), i


@pytest.mark.gpu
def test_gpu_projective_ladder_matches_oracle_and_reference(semp, tmp_path):
    from circom_amd import runtime as rt
    fc = semp.flat
    rng = random.Random(14)
    B = 70
    rows, outs = zip(*(H.semaphore_inputs(Q, LEVELS, rng) for _ in range(B)))
    rows = [list(r) for r in rows]
    rows[9][2] = (rows[9][2] + 1) % SUBGROUP_ORDER          # forged signature
    c = rt.Circuit(semp.tape_path, semp.dat_path, semp.r1cs_path)
    b = c.batch(B)
    b.set_inputs(rows)
    b.run(); b.check_r1cs(); b.sync()
    st = b.status()
    assert st[9] & rt.ST_ASSERT_FAILED and (np.delete(st, [9]) == 0).all()
    for i in (0, 1, 33, 69):
        assert (b.signal(i, 1), b.signal(i, 2)) == outs[i], i
    want, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, rows[40]))
    assert failed is None and b.witness(40) == want
    _, loop = ref_build.binaries("bn128", "semaphore20p")
    if loop.exists():
        idx = [0, 1, 2, 3]
        raw = b"".join(v.to_bytes(32, "little") for i in idx for v in rows[i])
        pre = str(tmp_path / "ref_")
        ref_build.run_loop(semp, raw, len(idx), 1, wtns_prefix=pre)
        for j, i in enumerate(idx):
            g = tmp_path / ("gpu_%d.wtns" % i)
            b.write_wtns(i, g)
            assert g.read_bytes() == open(pre + "%d.wtns" % j, "rb").read(), i
    b.close(); c.close()
