"""cw_load must reject malformed artefacts with CW_EIO instead of crashing (the reference mmap()s its .dat blindly,
main.cpp:38-56; a library serving many circuits cannot)."""
import pytest

from circom_amd import runtime as rt
from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.circuits.basic import BasicMain


def test_truncated_and_corrupt_files_are_rejected(tmp_path):
    cp = compile_program(Program(BasicMain()), str(tmp_path), "basic", strands=(1, 4))
    tape = open(cp.tape_path, "rb").read()
    dat = open(cp.dat_path, "rb").read()
    r1cs = open(cp.r1cs_path, "rb").read()
    rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path).close()           # sanity: the originals load

    def attempt(t=None, d=None, r=None):
        tp, dp, rp = tmp_path / "x.cwt", tmp_path / "x.dat", tmp_path / "x.r1cs"
        tp.write_bytes(tape if t is None else t)
        dp.write_bytes(dat if d is None else d)
        rp.write_bytes(r1cs if r is None else r)
        return rt.Circuit(tp, dp, rp)

    for cut in (0, 3, 16, 60, 100, len(tape) // 2, len(tape) - 1):
        with pytest.raises(rt.CwError):
            attempt(t=tape[:cut])
    with pytest.raises(rt.CwError):
        attempt(t=b"XXXX" + tape[4:])                                       # bad magic
    with pytest.raises(rt.CwError):
        attempt(t=tape[:4] + (99).to_bytes(4, "little") + tape[8:])         # unknown version
    # (the runtime reads the hash map and the witness list of the .dat; the constant table after them is the
    #  reference runtime's business, so only truncation inside the first two parts is an error here)
    for cut in (0, 10, 256 * 24 + 5):
        with pytest.raises(rt.CwError):
            attempt(d=dat[:cut])
    for cut in (0, 11, 40, len(r1cs) // 2, len(r1cs) - 1):
        with pytest.raises(rt.CwError):
            attempt(r=r1cs[:cut])
    with pytest.raises(rt.CwError):
        attempt(r=b"r1cx" + r1cs[4:])
    # an .r1cs of another prime is refused
    other = compile_program(Program(BasicMain(), prime="bls12381"), str(tmp_path / "o"), "basic", strands=(1,))
    with pytest.raises(rt.CwError) as e:
        rt.Circuit(cp.tape_path, cp.dat_path, other.r1cs_path)
    assert "prime" in str(e.value)


def test_small_primes_are_rejected_not_miscomputed(tmp_path):
    """The device field code assumes circom's 253..256-bit primes (the short-path sums rely on q > 2^192); Goldilocks
    (64-bit, a separate runtime in the reference) must be refused by the lowering and by the library."""
    import numpy as np
    from circom_amd.frontend.dsl import Program
    from circom_amd.frontend.flatten import flatten
    from circom_amd.hip_elements.lower import lower
    from circom_amd.circuits.basic import Multiplier2
    with pytest.raises(ValueError, match="Goldilocks"):
        lower(flatten(Program(Multiplier2(), prime="goldilocks")))
    gold = 0xFFFFFFFF00000001
    a = np.zeros((4, 32), dtype=np.uint8)
    with pytest.raises(rt.CwError, match="unsupported prime"):
        rt.fp_mul_bench(gold, a, a, 4, device=0)
