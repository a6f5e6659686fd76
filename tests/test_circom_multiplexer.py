"""A circuit that is NOT a mirror of an eDSL circuit: circomlib-shaped multiplexer.circom (one-hot decoder hints pinned by
constraints, scalar products over two-dimensional signal arrays, a compile-time `while` in log2) checked against plain Python,
through the oracle, the simplifier and a failing selection."""
import os
import random

from circom_amd.frontend.circom_exec import program_from_text
from circom_amd.frontend.circom_simplify import simplify_o1
from circom_amd.frontend.flatten import flatten
from oracle.field import PRIMES
from oracle.tape_eval import check_r1cs, eval_flat

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "circom_amd", "circuits", "circomlib")
Q = PRIMES["bn128"]


def test_multiplexer(tmp_path):
    from circom_amd.frontend.circom_exec import build_program
    from circom_amd.frontend.circom_lang import parse_program
    f = tmp_path / "m.circom"
    f.write_text('include "multiplexer.circom";\n'
                 'template Main() { signal input inp[5][3]; signal input sel; signal output out[3]; signal output bits;\n'
                 '  component m = Multiplexer(3, 5); m.inp <== inp; m.sel <== sel; out <== m.out; bits <== log2(5) + log2(8) + log2(0); }\n'
                 'component main {public [sel]} = Main();\n')
    fc = flatten(build_program(parse_program(str(f), [LIB])))
    assert fc.n_pub_in == 1 and fc.inputs[0] == ("sel", 5, 1)          # the public input comes first
    rng = random.Random(6)
    words = [[rng.randrange(Q) for _ in range(3)] for _ in range(5)]
    flat = [v for w in words for v in w]
    sm = simplify_o1(fc)
    for sel in range(5):
        inp = {fc.main_input_start: sel}
        inp.update({fc.main_input_start + 1 + k: v for k, v in enumerate(flat)})
        sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
        assert failed is None and sig[1:4] == words[sel] and sig[4] == 4 + 4 + 0
        assert check_r1cs(Q, fc.constraints, sig) is None
        assert check_r1cs(Q, sm.constraints, [sig[s] for s in sm.witness2signal]) is None
    # an index outside the table: no decoder output fires, `dec.success === 1` fails at run time and in the R1CS
    inp[fc.main_input_start] = 7
    sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
    assert failed is not None
