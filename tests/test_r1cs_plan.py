"""Host-side plan of the staged R1CS check kernel (csrc/cw_r1cs_plan.h): built and replayed against the
LDS-DMA hazard rule on the CPU (cw_r1cs_plan_stats verifies every term reads the wire its row names)."""
import pytest

from circom_amd import runtime as rt
from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.circuits.basic import Multiplier2, Num2Bits
from circom_amd.circuits.poseidon import Poseidon
from circom_amd.circuits.sha256 import Sha256


def _circuit(tmp_path, prog, name):
    cp = compile_program(Program(prog), str(tmp_path), name, sym=False, strands=(1,))
    return cp, rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)


@pytest.mark.parametrize("entries", [6, 8, 10, 16, 64])
@pytest.mark.parametrize("chunks", [1, 3, 50])
def test_plan_is_hazard_free_and_complete(tmp_path, entries, chunks):
    for name, prog in (("m2", Multiplier2()), ("n2b", Num2Bits(64)), ("pos", Poseidon(2))):
        cp, c = _circuit(tmp_path, prog, name)
        st = c.r1cs_plan_stats(4096, chunks, entries)      # raises CwError on a hazard
        n_terms = sum(len(a) + len(b) + len(cc) for a, b, cc in cp.flat.constraints)
        assert st["terms"] == n_terms
        assert st["loads"] >= st["distinct_wires"] and st["entries"] == entries and st["depth"] == 4
        assert 1 <= st["chunks"] <= max(1, chunks)
        c.close()


def test_plan_caches_most_reuse_with_default_entries(tmp_path):
    cp, c = _circuit(tmp_path, Poseidon(2), "pos")
    st = c.r1cs_plan_stats(65536, 0, 0)
    # every wire is read ~2.7 times by the rows; the plan fetches it ~once per chunk
    assert st["terms"] > 2.5 * st["distinct_wires"]
    assert st["loads"] < 1.2 * st["distinct_wires"]
    assert st["filler_loads"] == 0
    c.close()


def test_small_lds_needs_filler_loads_but_stays_correct(tmp_path):
    cp, c = _circuit(tmp_path, Sha256(8), "sha8")
    few, many = c.r1cs_plan_stats(4096, 0, 6), c.r1cs_plan_stats(4096, 0, 24)
    assert few["filler_loads"] > 0 and few["loads"] > many["loads"]
    assert few["terms"] == many["terms"] and few["distinct_wires"] == many["distinct_wires"]
    c.close()


@pytest.mark.parametrize("seed", range(8))
def test_plans_of_random_circuits_are_hazard_free(tmp_path, seed):
    from test_schedule_fuzz import _random_template
    cp, c = _circuit(tmp_path, _random_template(seed, 60 + 30 * seed), "fuzz")
    n_terms = sum(len(a) + len(b) + len(cc) for a, b, cc in cp.flat.constraints)
    for chunks, entries in ((1, 6), (2, 7), (5, 10), (40, 16), (3, 64)):
        st = c.r1cs_plan_stats(1000, chunks, entries)       # raises CwError if a term could read a stale LDS entry
        assert st["terms"] == n_terms and st["loads"] >= st["distinct_wires"]
    c.close()


# ---- the term stream of the default kernel (cw_r1cs_stream_kernel), replayed on the CPU ------------------------------------
# What the kernel does with a term is restated here on Python integers (csrc/cw_kernels.hip r1_term); the verdict of the
# replay over a witness must be check_r1cs's (the oracle's plain A*B - C over the .r1cs rows) for every way of building the
# plan: boolean rows as one term or as three, riding inside the sum that reads the same bit or on their own, and - for a folded
# term - whichever of its two forms (select / product) the wave takes.
RBITS = 261
T_BOOL, COEF_CONST, COEF_BITSEL = 1 << 26, 1 << 31, 1 << 30


def _replay_stream(plan, q, w, mont, select=True):
    R = pow(2, RBITS, q)
    Rinv = pow(R, -1, q)
    V = [(x * R) % q for x in w] if mont else list(w)
    one = R if mont else 1
    ctab = [int.from_bytes(row.tobytes(), "little") for row in plan["ctab"]]
    mmul = lambda a, b: a * b * Rinv % q
    bad = None
    n_loads = 0
    for first, n, _, row0 in plan["chunk"]:
        A = B = cur = 0
        row = int(row0)
        allbool = False
        prev_slot = None
        for k in range(int(first), int(first) + int(n)):
            w0, ci = int(plan["terms"][k][0]), int(plan["terms"][k][1])
            slot, acc, endk = w0 & 0x3FFFFFF, (w0 >> 27) & 3, (w0 >> 29) & 3
            if slot != prev_slot:
                n_loads += 1
            prev_slot = slot
            wv = V[slot]
            ok = True
            if w0 & T_BOOL:
                ok = wv in (0, one)
                allbool = ok and select
            elif endk == 3:
                ok = cur == wv
            elif acc == 3:
                cur = wv
            else:
                x = wv
                if ci >> 31:
                    x = ctab[ci & 0x7FFFFFFF]
                elif (ci & COEF_BITSEL) and allbool:
                    x = ctab[(ci & 0x3FFFFFFF) + 1] if wv else 0
                    ci = 0
                elif ci >= 2:
                    x = mmul(wv, ctab[ci & 0x3FFFFFFF])
                cur = (cur - x) % q if ci == 1 else (cur + x) % q
                if (w0 >> 31) and acc != 2:
                    if acc == 0:
                        A = cur
                    else:
                        B = cur
                    cur = 0
                if endk == 2:
                    ok = cur == 0
                elif endk == 1:
                    ok = (mmul(A, B) == cur) if mont else ((A * B + (q - cur)) * Rinv % q == 0)
            if endk:
                if not ok:
                    oc = int(plan["row_orig"][row])
                    bad = oc if bad is None else min(bad, oc)
                row += 1
                if not (w0 & T_BOOL):
                    A = B = cur = 0
    return bad, n_loads


@pytest.mark.parametrize("mont", [False, True])
def test_stream_plan_folds_boolean_rows_into_the_sum_that_reads_the_bit(tmp_path, mont, monkeypatch):
    from oracle.tape_eval import eval_flat, check_r1cs
    monkeypatch.setenv("CW_MONT", "1" if mont else "0")
    n = 40
    cp, c = _circuit(tmp_path, Num2Bits(n), "n2b%d" % mont)
    assert c.montgomery == mont
    fc = cp.flat
    assert fc.n_main_inputs == 1
    w, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {fc.main_input_start: 0xA5A5A5A5A5 % (1 << n)})
    assert failed is None
    plans = {k: c.r1cs_stream_plan(16, **kw) for k, kw in
             (("fold", {}), ("nofold", {"no_fold": True}), ("nobool", {"no_bool": True}))}
    assert plans["fold"]["n_folded"] == n and plans["fold"]["n_bitsel"] == n - 1      # the bit of weight 1 needs no product
    assert plans["nofold"]["n_folded"] == 0 and plans["nobool"]["n_folded"] == 0
    assert plans["fold"]["n_terms"] == plans["nofold"]["n_terms"]
    for name, p in plans.items():
        assert sorted(p["row_orig"].tolist()) == list(range(len(fc.constraints))), name
    loads = {}
    for name, p in plans.items():
        for select in (True, False):
            got, loads[name] = _replay_stream(p, c.q, w, mont, select)
            assert got is None, (name, select)
    assert loads["fold"] <= loads["nofold"] - (n - 1)                   # every bit is read once, not twice
    # every wire set to 0, 1, 2, q - 1, its value + 1: the first violated row is the oracle's, whatever the plan
    for k in range(1, len(w)):
        for v in {0, 1, 2, c.q - 1, (w[k] + 1) % c.q}:
            w2 = list(w)
            w2[k] = v
            want = check_r1cs(c.q, fc.constraints, w2)
            for name, p in plans.items():
                for select in (True, False):
                    assert _replay_stream(p, c.q, w2, mont, select)[0] == want, (k, v, name, select)
    c.close()


@pytest.mark.parametrize("mont", [False, True])
@pytest.mark.parametrize("seed", range(6))
def test_stream_plans_of_random_circuits_give_the_oracles_verdict(tmp_path, seed, mont, monkeypatch):
    """random circuits with booleanity rows whose bits feed products, sums with small and field-sized coefficients and component
    wiring: folded or not, select or product, the replayed term stream names the row check_r1cs names"""
    import random
    from oracle.tape_eval import eval_flat, check_r1cs
    from test_schedule_fuzz import _random_template
    monkeypatch.setenv("CW_MONT", "1" if mont else "0")
    cp, c = _circuit(tmp_path, _random_template(seed, 80 + 40 * seed), "fuzz%d_%d" % (seed, mont))
    fc = cp.flat
    rng = random.Random(77 + seed)
    inp = {fc.main_input_start + k: rng.randrange(c.q) for k in range(4)}
    w, failed = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
    assert failed is None and check_r1cs(c.q, fc.constraints, w) is None
    plans = {k: c.r1cs_stream_plan(24, **kw) for k, kw in (("fold", {}), ("nofold", {"no_fold": True}), ("nobool", {"no_bool": True}))}
    assert plans["fold"]["n_folded"] > 0
    cases = [list(w)]
    for _ in range(60):
        w2 = list(w)
        for _ in range(rng.choice((1, 1, 1, 2, 3))):
            w2[rng.randrange(1, len(w))] = rng.choice((0, 1, 2, c.q - 1, rng.randrange(c.q)))
        cases.append(w2)
    n_bad = 0
    for w2 in cases:
        want = check_r1cs(c.q, fc.constraints, w2)
        n_bad += want is not None
        for name, p in plans.items():
            for select in (True, False):
                assert _replay_stream(p, c.q, w2, mont, select)[0] == want, (name, select)
    assert n_bad > 20
    c.close()
