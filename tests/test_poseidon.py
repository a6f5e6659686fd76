"""Poseidon circuit: regenerated constants and the circuit's output pinned to published values."""
import random

from circom_amd.frontend.dsl import Program
from circom_amd.frontend.flatten import flatten
from circom_amd.circuits.poseidon import Poseidon
from circom_amd.circuits.poseidon_constants import poseidon_params, poseidon_hash
from oracle.field import PRIMES
from oracle.tape_eval import eval_flat, check_r1cs

Q = PRIMES["bn128"]


def test_constants_match_circomlib_table_heads():
    C, M = poseidon_params(Q, 3)
    assert len(C) == 3 * (8 + 57)
    # circomlib poseidon_constants: POSEIDON_C(3)[0], POSEIDON_M(3)[0][0]
    assert C[0] == 0x0ee9a592ba9a9518d05986d656f40c2114c4993c11bb29938d21d47304cd8e6e
    assert M[0][0] == 0x109b7f411ba0e4c9b2b70caf5c36a7b194be7c11ad24378bfedb68592ba8118b


def test_published_test_vector():
    # circomlibjs test: poseidon([1,2])
    assert poseidon_hash(Q, [1, 2]) == 7853200120776062878684798364095072458815029376092732009249414926327459813530


def test_circuit_output_and_constraints():
    fc = flatten(Program(Poseidon(2)))
    assert fc.n_signals == 1108 and fc.inputs == [("inputs", 2, 2)]
    rng = random.Random(5)
    for ins in ([1, 2], [0, 0], [Q - 1, 5], [rng.randrange(Q), rng.randrange(Q)]):
        sig, failed = eval_flat(Q, fc.n_signals, fc.n_temps, fc.constants, fc.code, {2: ins[0], 3: ins[1]})
        assert failed is None and sig[1] == poseidon_hash(Q, ins)
        assert check_r1cs(Q, fc.constraints, sig) is None
    # 243 quadratic constraints (81 S-boxes x 3), the rest linear at --O0
    nq = sum(1 for a, b, c in fc.constraints if a and b)
    assert nq == 243
