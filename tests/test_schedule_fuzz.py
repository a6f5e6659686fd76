"""Random circuits through every lowering pass and strand count, replayed by the race-checking simulator of the
kernel's execution model (oracle/tape_eval.py: prefetch, register forwarding, LDS hand-offs, barrier elision raise
ScheduleHazard on any race) and compared with the straight evaluation of the flat witness code.  The shapes are
chosen to hit the passes: long small-coefficient sums (D_LINSUM and its splitting), field-sized coefficients
(D_DOTC), bit extraction (D_BIT, proved asserts), several divisions at one level (batched inversions), selects,
wide fan-out (extra destinations) and long dependent chains."""
import random

import pytest

from circom_amd.frontend.dsl import Program, template
from circom_amd.frontend.flatten import flatten
from circom_amd.hip_elements.lower import lower
from oracle.field import PRIMES
from oracle.tape_eval import eval_flat, eval_tape, check_r1cs

Q = PRIMES["bn128"]


@template
def _Sq(c):
    x = c.input("in")
    y = c.output("out")
    c.set(y, x * x)


@template
def _Leaf(c, k):
    # two inputs that the parent assigns at different times (the component fires when the LAST one arrives,
    # store_bucket.rs:660-735), an inner component, an intermediate signal
    a = c.input("a")
    b = c.input("b")
    out = c.output("out")
    mid = c.signal("mid")
    sq = c.component("sq", _Sq())
    c.set(sq["in"], a + k)
    c.set(mid, sq["out"] * b)
    c.set(out, mid + a - k)


def _random_template(seed, n_nodes, q=Q):
    rng = random.Random(seed)

    @template
    def Fuzz(c):
        ins = c.input("in", 4)
        vals = [ins[i] for i in range(4)]
        sig = c.signal("s", n_nodes)
        out = c.output("out", 3)
        pending = []                                         # sub-components still waiting for their second input
        n_leaf = 0
        for k in range(n_nodes):
            pick = lambda: vals[rng.randrange(len(vals))] if rng.random() < 0.5 else vals[-1 - rng.randrange(min(6, len(vals)))]
            if pending and rng.random() < 0.3:               # complete one: it fires now, its output becomes usable
                comp = pending.pop(rng.randrange(len(pending)))
                c.set(comp["b"], pick())
                vals.append(comp["out"])
            if rng.random() < 0.08:
                comp = c.component("leaf", _Leaf(rng.randrange(1, 4)), n_leaf)
                n_leaf += 1
                c.set(comp["a"], pick())
                pending.append(comp)
            r = rng.random()
            if r < 0.22:
                e = pick() * pick()
                c.set(sig[k], e)
            elif r < 0.34:
                e = pick() + pick() - pick() * rng.randrange(1, 9)
                c.set(sig[k], e)
            elif r < 0.44:                                   # long small-coefficient sum
                e = c.const(rng.randrange(5))
                for _ in range(rng.choice((3, 6, 30, 70))):
                    e = e + pick() * rng.randrange(-4, 9)
                c.set(sig[k], e)
            elif r < 0.52:                                   # field-sized coefficients
                e = pick() * rng.randrange(q) + pick() * rng.randrange(q) + pick() * rng.randrange(q) + rng.randrange(q)
                c.set(sig[k], e)
            elif r < 0.62:                                   # bits of a value, with the usual booleanity check
                x = pick()
                c.hint(sig[k], (x >> rng.randrange(0, 254)) & 1)
                c.enforce(sig[k] * (sig[k] - 1), 0)
            elif r < 0.74:                                   # divisions (several per level: batched)
                c.hint(sig[k], pick() / (pick() + rng.randrange(1, 5)))
            elif r < 0.80:
                c.hint(sig[k], c.select(pick().lt(pick()), pick(), pick() + 1))
            elif r < 0.88:                                   # a copy: becomes an extra destination
                c.set(sig[k], pick())
            else:
                c.hint(sig[k], (pick() & pick()) ^ (pick() >> 3))
            vals.append(sig[k])
        for comp in pending:
            c.set(comp["b"], pick())
            vals.append(comp["out"])
        for i in range(3):
            c.set(out[i], vals[-1 - i] + vals[rng.randrange(len(vals))] * (i + 2))

    return Fuzz()


@pytest.mark.parametrize("seed", range(40))
def test_random_circuits_lower_race_free_and_equivalent(seed):
    rng = random.Random(1000 + seed)
    fc = flatten(Program(_random_template(seed, 40 + 26 * (seed % 11))))
    tapes = [lower(fc, n_strands=S) for S in (1, 4, 16)]
    # the same circuits with signals in Montgomery form (integer operators get conversions) and as pipelined schedules
    tapes += [lower(fc, n_strands=S, mont=True) for S in (1, 16)]
    tapes += [lower(fc, pipe=(8, 8)), lower(fc, pipe=(4, 4), mont=True)]
    for trial in range(3):
        if trial == 0:
            row = [rng.randrange(Q) for _ in range(4)]
        elif trial == 1:
            row = [rng.randrange(4) for _ in range(4)]          # small values: run-time short paths, zero denominators
        else:
            row = [Q - 1 - rng.randrange(3), 0, rng.randrange(Q), 1]
        inp = {fc.main_input_start + k: v for k, v in enumerate(row)}
        sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
        assert failed is None                                    # the only asserts are provably true bit checks
        assert check_r1cs(Q, fc.constraints, sig) is None
        for t in tapes:
            got, st = eval_tape(t, inp)                          # raises ScheduleHazard on a race
            assert st == 0 and got == sig, (seed, trial, t.stats["strands"], t.kind, t.mont)


@pytest.mark.gpu
@pytest.mark.parametrize("mont", [False, True])
@pytest.mark.parametrize("seed", [3, 7, 18, 29])
def test_gpu_random_circuits_match_oracle(seed, mont, tmp_path):
    import numpy as np
    from circom_amd import runtime as rt
    from circom_amd.compiler import compile_program
    rng = random.Random(2000 + seed)
    cp = compile_program(Program(_random_template(seed, 40 + 26 * (seed % 11))), str(tmp_path), "fuzz%d" % seed, sym=False, mont=mont)
    assert cp.tape.mont == mont
    fc = cp.flat
    B = 130
    rows = [[rng.randrange(Q) for _ in range(4)] for _ in range(B - 10)] + [[rng.randrange(4) for _ in range(4)] for _ in range(10)]
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    for strands in ("1", "4", "16"):
        import os
        os.environ["CW_STRANDS"] = strands
        try:
            b = c.batch(B)
        finally:
            del os.environ["CW_STRANDS"]
        b.set_inputs(rows)
        b.run(); b.check_r1cs(); b.sync()
        assert (b.status() == 0).all(), (seed, strands)
        got = b.witnesses()
        for i in (0, 1, 64, 65, B - 11, B - 10, B - 1):
            inp = {fc.main_input_start + k: v for k, v in enumerate(rows[i])}
            sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
            assert failed is None
            assert got[i].tobytes() == b"".join(v.to_bytes(32, "little") for v in sig), (seed, strands, i)
        b.close()
    c.close()


@pytest.mark.parametrize("seed,prime", [(1, "bn128"), (5, "bn128"), (11, "bn128"), (23, "bn128"), (37, "bn128"),
                                        (2, "bls12381"), (9, "bls12381"), (4, "secq256r1"), (6, "bls12377")])
def test_random_circuits_match_the_reference_runtime(seed, prime, tmp_path):
    """The same random circuits, emitted as reference-style C++ and run by the reference's own runtime: its `.wtns`
    must equal the oracle's bytes (pins the flat evaluation order, the operator semantics and the writers on shapes
    no hand-written circuit has)."""
    from circom_amd.compiler import compile_program
    from circom_amd.hip_elements.writers import wtns_bytes
    from conftest import ensure_ref
    from oracle import ref_build
    ensure_ref(prime)
    Q = PRIMES[prime]
    cp = compile_program(Program(_random_template(seed, 40 + 26 * (seed % 11), Q), prime=prime), str(tmp_path),
                         "fuzz%d" % seed, sym=False, strands=(1,))
    try:
        ref_build.build_circuit(cp)
    except RuntimeError as e:
        pytest.skip(str(e))
    fc = cp.flat
    rng = random.Random(3000 + seed)
    rows = [[rng.randrange(Q) for _ in range(4)] for _ in range(4)] + [[rng.randrange(4) for _ in range(4)] for _ in range(3)]
    raw = b"".join(v.to_bytes(32, "little") for r in rows for v in r)
    pre = str(tmp_path / "r_")
    ref_build.run_loop(cp, raw, len(rows), 1, wtns_prefix=pre)
    for i, r in enumerate(rows):
        inp = {fc.main_input_start + k: v for k, v in enumerate(r)}
        sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
        assert failed is None
        assert open(pre + "%d.wtns" % i, "rb").read() == wtns_bytes(Q, sig), (seed, i)


def test_division_by_a_constant_power_of_two_is_a_shift():
    """`x \\ 2^k` / `x % 2^k` with a constant divisor become D_SHR / D_BAND rows (lower.py pass A: no Knuth-D division, no failing
    row); other constant divisors keep the division.  Same witness as the flat program on edge values."""
    from circom_amd.frontend.dsl import Program, template
    from circom_amd.frontend.flatten import flatten
    from circom_amd.hip_elements import lower as L
    from oracle.tape_eval import eval_flat, eval_tape

    @template
    def Limbs(c):
        x = c.input("x")
        out = c.output("out", 6)
        c.hint(out[0], x % (1 << 64))
        c.hint(out[1], x // (1 << 64))
        c.hint(out[2], x % 7)
        c.hint(out[3], x // 7)
        c.hint(out[4], (x + 5) // (1 << 200))
        c.hint(out[5], x % 1)
    for prime in ("bn128", "bls12381"):
        fc = flatten(Program(Limbs(), prime=prime))
        q = fc.fp.q
        t = L.lower(fc, n_strands=1)
        ops = [int(r[0]) & 0xFF for r in t.rows]
        assert ops.count(L.D_SHR) == 2 and ops.count(L.D_IDIV) == 1 and ops.count(L.D_MOD) == 1 and t.stats["pow2_divisions"] >= 4
        for x in (0, 1, (1 << 64) - 1, 1 << 64, (1 << 200) - 5, q - 1, q >> 1, 12345678901234567890123456789):
            inp = {fc.main_input_start: x % q}
            want, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
            got, st = eval_tape(t, inp)
            assert failed is None and st == 0 and got == want
            assert want[1:7] == [x % q % (1 << 64), x % q >> 64, x % q % 7, x % q // 7, ((x % q + 5) % q) >> 200, 0]
