"""The status word of a failing instance names the FIRST failing check of the reference's sequential program
(assert_bucket.rs:75-77, calcwit.cpp:104-114), whatever the schedule does with the rows: strands, batched inversions and
the pipelined variant all reorder them.  bits 8.. = index of the flat operation = what oracle.tape_eval.eval_flat reports."""
import os
import random

import numpy as np
import pytest

from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program, template
from circom_amd.frontend.flatten import flatten
from circom_amd.hip_elements.lower import lower
from oracle.field import PRIMES
from oracle.tape_eval import eval_flat, eval_tape

Q = PRIMES["bn128"]
N = 14


@template
def _Chain(c, n):
    """n checks `x[i]^2 + i === 3 x[i+1] + 7`, an integer division whose divisor is x[n] - 5, more checks behind it"""
    x = c.input("x", n + 1)
    out = c.output("out")
    s = c.signal("s", n)
    acc = c.const(0)
    for i in range(n):
        c.hint(s[i], x[i] * x[i] + i)
        c.enforce(s[i], x[i + 1] * 3 + 7)
        acc = acc + s[i] * (i + 1)
    d = c.signal("d")
    c.hint(d, (x[0] & 0xFFFF) // (x[n] - 5))
    t = c.signal("t")
    c.hint(t, d + x[1] / (x[2] + 1))
    c.enforce((t - d) * (x[2] + 1), x[1])
    c.set(out, acc + t)


def _row(rng, good: int, div_zero: bool):
    """inputs for which exactly the first `good` chain checks hold"""
    inv3 = pow(3, -1, Q)
    x = [rng.randrange(Q)]
    for i in range(N):
        nxt = (x[i] * x[i] + i - 7) * inv3 % Q
        x.append(nxt if i < good else (nxt + 1 + rng.randrange(5)) % Q)
    if div_zero:
        x[N] = 5
    return x


def _inp(fc, row):
    return {fc.main_input_start + k: v for k, v in enumerate(row)}


def test_replay_reports_the_index_eval_flat_reports():
    fc = flatten(Program(_Chain(N)))
    rng = random.Random(2)
    tapes = [lower(fc, **kw) for kw in (dict(n_strands=1), dict(n_strands=4), dict(n_strands=16), dict(pipe=(8, 8)),
                                        dict(n_strands=16, mont=True), dict(pipe=(4, 4), mont=True))]
    seen = set()
    for good in list(range(N + 1)) + [3, 7]:
        for dz in (False, True):
            if good == N and dz:            # x[N] is then forced by the chain: skip
                continue
            row = _row(rng, good, dz)
            want, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, row))
            if good == N and not dz:
                assert failed is None
            else:
                assert failed is not None
            for t in tapes:
                got, st = eval_tape(t, _inp(fc, row))
                if failed is None:
                    assert st == 0 and got == want
                else:
                    assert st >> 8 == failed and st & 3, (good, dz, t.stats.get("strands"))
            seen.add(failed)
    assert len(seen) >= N + 1              # every check was the first to fail for some input (and the division once)


@pytest.mark.gpu
@pytest.mark.parametrize("mont", [False, True])
def test_gpu_status_word_is_the_first_failing_operation_for_every_variant(tmp_path, mont):
    from circom_amd import runtime as rt
    cp = compile_program(Program(_Chain(N)), str(tmp_path), "chain", sym=False, mont=mont, pipe=(8, 8))
    fc = cp.flat
    rng = random.Random(5)
    rows, want = [], []
    for j in range(140):
        good = j % (N + 1)
        dz = (j // (N + 1)) % 2 == 1 and good != N
        r = _row(rng, good, dz)
        rows.append(r)
        want.append(eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, _inp(fc, r))[1])
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    for env in ({"CW_PIPE": "0", "CW_STRANDS": "1"}, {"CW_PIPE": "0", "CW_STRANDS": "4"}, {"CW_PIPE": "0", "CW_STRANDS": "16"},
                {"CW_PIPE": "1"}):
        os.environ.update(env)
        try:
            b = c.batch(len(rows))
        finally:
            for k in env:
                del os.environ[k]
        b.set_inputs(rows)
        b.run(); b.sync()
        st = b.status()
        for j, f in enumerate(want):
            if f is None:
                assert st[j] == 0, (env, j)
            else:
                assert int(st[j]) >> 8 == f and int(st[j]) & 3, (env, j, int(st[j]) >> 8, f)
        assert "operation %d" % want[1] in b.explain(1)
        b.close()
    c.close()
