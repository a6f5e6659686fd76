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
