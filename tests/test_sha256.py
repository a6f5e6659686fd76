"""SHA-256 circuit (circomlib structure, re-authored): digest pinned to hashlib, R1CS satisfied, lowered
schedule == flat semantics, reference C++ runtime produces the identical .wtns; GPU batch bit-exact."""
import hashlib

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.frontend.flatten import flatten
from circom_amd.hip_elements.lower import lower
from circom_amd.circuits.sha256 import Sha256
from circom_amd.hip_elements.writers import wtns_bytes
from oracle import ref_build
from oracle.tape_eval import eval_flat, eval_rows, eval_tape, check_r1cs


def _bits(msg: bytes):
    return [(msg[i // 8] >> (7 - i % 8)) & 1 for i in range(8 * len(msg))]


def _digest_bits(msg: bytes):
    return _bits(hashlib.sha256(msg).digest())


@pytest.fixture(scope="module")
def sha64():
    return flatten(Program(Sha256(64)))


def test_sha256_one_block_digest_and_r1cs(sha64):
    fc = sha64
    assert fc.n_signals == 204329
    assert sum(1 for a, b, c in fc.constraints if a and b) == 30952     # ~30K non-linear constraints / block
    for msg in (b"abcdefgh", b"\x00" * 8, b"\xff" * 8):
        inp = {fc.main_input_start + i: b for i, b in enumerate(_bits(msg))}
        sig, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
        assert failed is None and sig[1:257] == _digest_bits(msg)
        assert check_r1cs(fc.fp.q, fc.constraints, sig) is None


def test_sha256_lowered_schedule_matches(sha64):
    fc = sha64
    t = lower(fc)
    inp = {fc.main_input_start + i: b for i, b in enumerate(_bits(b"GPU->wtn"))}
    a, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
    b, st = eval_tape(t, inp)
    assert failed is None and st == 0 and a == b
    # non-bit inputs: the circuit does not constrain its inputs; values must still agree until an assert trips
    inp[fc.main_input_start + 3] = 7
    a, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
    b, st = eval_tape(t, inp)
    assert (failed is None) == (st == 0)
    if failed is None:
        assert a == b


@pytest.mark.slow
def test_sha256_two_blocks_reference_runtime_parity(tmp_path, ref_dir_bn128):
    cp = compile_program(Program(Sha256(512)), str(tmp_path), "sha256_512", sym=False)
    fc = cp.flat
    assert fc.n_signals == 408529
    try:
        ref_build.build_circuit(cp)          # ~75 s the first time (10 MB of emitted C++)
    except RuntimeError as e:
        pytest.skip(str(e))
    rng = np.random.default_rng(2)
    bits = rng.integers(0, 2, size=(3, 512), dtype=np.uint8)
    raw = np.zeros((3, 512, 32), dtype=np.uint8)
    raw[:, :, 0] = bits
    pre = str(tmp_path / "r_")
    ref_build.run_loop(cp, raw.tobytes(), 3, 1, wtns_prefix=pre)
    for i in range(3):
        inp = {fc.main_input_start + k: int(b) for k, b in enumerate(bits[i])}
        want, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
        assert failed is None
        assert want[1:257] == _digest_bits(np.packbits(bits[i]).tobytes())
        assert open(pre + "%d.wtns" % i, "rb").read() == wtns_bytes(fc.fp.q, want)


@pytest.mark.gpu
def test_gpu_sha256_batch_bit_exact(tmp_path):
    from circom_amd import runtime as rt
    cp = compile_program(Program(Sha256(512)), str(tmp_path), "sha256_512", sym=False)
    fc = cp.flat
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    n = 192
    rng = np.random.default_rng(2)
    bits = rng.integers(0, 2, size=(n, 512), dtype=np.uint8)
    raw = np.zeros((n, 512, 32), dtype=np.uint8)
    raw[:, :, 0] = bits
    b = c.batch(n)
    b.set_inputs(raw)
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    for i in range(n):                                   # every instance: digest vs hashlib
        w = b.witness_bytes(i) if i % 64 == 0 else None
        dig = _digest_bits(np.packbits(bits[i]).tobytes())
        got = [b.signal(i, 1 + k) for k in (0, 1, 2, 100, 255)]
        assert got == [dig[k] for k in (0, 1, 2, 100, 255)], i
        if w is not None:                                # sampled: whole witness vs the oracle, byte for byte
            inp = {fc.main_input_start + k: int(x) for k, x in enumerate(bits[i])}
            want, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
            assert failed is None and w == b"".join(v.to_bytes(32, "little") for v in want), i
    # and against the reference runtime where its binary was prebuilt
    _, loop = ref_build.binaries("bn128", "sha256_512")
    if loop.exists():
        pre = str(tmp_path / "ref_")
        ref_build.run_loop(cp, raw[:2].tobytes(), 2, 1, wtns_prefix=pre)
        for i in range(2):
            g = tmp_path / ("g%d.wtns" % i)
            b.write_wtns(i, g)
            assert g.read_bytes() == open(pre + "%d.wtns" % i, "rb").read()
    b.close(); c.close()
