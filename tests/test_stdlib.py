"""circomlib-style building blocks (circuits/stdlib.py): values against plain Python, constraints satisfied, the
lowered schedules race-free, and the reference runtime's `.wtns` equal to the oracle's."""
import random

import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.frontend.flatten import flatten
from circom_amd.circuits.stdlib import SortPair, LessThan, BinSum, XOR, AND, OR, NOT, Mux1
from circom_amd.hip_elements.lower import lower
from circom_amd.hip_elements.writers import wtns_bytes
from oracle import ref_build
from oracle.field import PRIMES
from oracle.tape_eval import eval_flat, eval_tape, check_r1cs

Q = PRIMES["bn128"]


def _run(fc, row):
    inp = {fc.main_input_start + k: v for k, v in enumerate(row)}
    sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
    return inp, sig, failed


def test_gates_and_mux_truth_tables():
    for T, fn in ((XOR, lambda a, b: a ^ b), (AND, lambda a, b: a & b), (OR, lambda a, b: a | b)):
        fc = flatten(Program(T()))
        for a in (0, 1):
            for b in (0, 1):
                _, sig, failed = _run(fc, [a, b])
                assert failed is None and sig[1] == fn(a, b) and check_r1cs(Q, fc.constraints, sig) is None
    fc = flatten(Program(NOT()))
    assert [_run(fc, [a])[1][1] for a in (0, 1)] == [1, 0]
    fc = flatten(Program(Mux1()))
    assert [_run(fc, [7, 9, s])[1][1] for s in (0, 1)] == [7, 9]


def test_less_than_and_binsum_against_python():
    rng = random.Random(3)
    fc = flatten(Program(LessThan(32)))
    for a, b in [(0, 0), (0, 1), (1, 0), (2 ** 32 - 1, 2 ** 32 - 1), (2 ** 32 - 2, 2 ** 32 - 1)] + \
                [(rng.randrange(2 ** 32), rng.randrange(2 ** 32)) for _ in range(40)]:
        _, sig, failed = _run(fc, [a, b])
        assert failed is None and sig[1] == int(a < b) and check_r1cs(Q, fc.constraints, sig) is None
    fc = flatten(Program(BinSum(8, 3)))
    for _ in range(20):
        xs = [rng.randrange(256) for _ in range(3)]
        bits = [(x >> k) & 1 for x in xs for k in range(8)]
        _, sig, failed = _run(fc, bits)
        assert failed is None and sum(sig[1 + k] << k for k in range(10)) == sum(xs)
        assert check_r1cs(Q, fc.constraints, sig) is None


def test_sortpair_values_schedules_and_reference_runtime(tmp_path, ref_dir_bn128):
    n = 16
    cp = compile_program(Program(SortPair(n)), str(tmp_path), "sortpair16", sym=False, strands=(1,))
    fc = cp.flat
    rng = random.Random(8)
    rows = [[0, 0], [5, 5], [65535, 0], [0, 65535], [65535, 65535]] + [[rng.randrange(1 << n), rng.randrange(1 << n)] for _ in range(25)]
    tapes = [lower(fc, n_strands=S) for S in (1, 4, 16)]
    wants = []
    for a, b in rows:
        inp, sig, failed = _run(fc, [a, b])
        assert failed is None
        assert sig[1:5] == [min(a, b), max(a, b), int(a == b), a + b]
        assert check_r1cs(Q, fc.constraints, sig) is None
        for t in tapes:
            got, st = eval_tape(t, inp)
            assert st == 0 and got == sig
        wants.append(sig)
    # an input that does not fit n bits trips Num2Bits' `===` in both worlds
    _, _, failed = _run(fc, [1 << n, 3])
    assert failed is not None
    try:
        ref_build.build_circuit(cp)
    except RuntimeError as e:
        pytest.skip(str(e))
    raw = b"".join(v.to_bytes(32, "little") for r in rows for v in r)
    pre = str(tmp_path / "r_")
    ref_build.run_loop(cp, raw, len(rows), 1, wtns_prefix=pre)
    for i, want in enumerate(wants):
        assert open(pre + "%d.wtns" % i, "rb").read() == wtns_bytes(Q, want), i
    r = ref_build.run_cli(cp, '{"in": ["%d", "3"]}' % (1 << n), tmp_path / "x.wtns")
    assert r.returncode != 0
