"""Poseidon parameter generation (round constants + MDS matrix) for x^5 over a prime field.

circomlib is not in the reference tree nor in this container, so its `poseidon_constants.circom`
table is regenerated from its defining algorithm: the Grain-LFSR procedure of the Poseidon paper's
reference script (`generate_parameters_grain.sage 1 0 <n> <t> <R_F> <R_P> <p>`), which is how
circomlib's table was produced.  tests/test_poseidon.py pins the result against the well-known
constants C[0], M[0][0] for t=3 and the test vector poseidon([1,2]).
"""
from __future__ import annotations

from functools import lru_cache

# circomlib/circuits/poseidon.circom: N_ROUNDS_P for t = 2..17
N_ROUNDS_P = [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]
N_ROUNDS_F = 8


class _Grain:
    def __init__(self, field, sbox, n, t, r_f, r_p):
        bits = []
        for v, w in ((field, 2), (sbox, 4), (n, 12), (t, 12), (r_f, 10), (r_p, 10)):
            bits += [int(b) for b in bin(v)[2:].zfill(w)]
        bits += [1] * 30
        assert len(bits) == 80
        self.s = bits
        for _ in range(160):
            self._step()

    def _step(self):
        s = self.s
        nb = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(nb)
        return nb

    def bit(self):
        nb = self._step()
        while nb == 0:
            self._step()
            nb = self._step()
        return self._step()

    def bits(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v


@lru_cache(maxsize=None)
def poseidon_params(p: int, t: int, r_f: int = N_ROUNDS_F, r_p: int | None = None):
    """Returns (C, M): C flat list of (r_f+r_p)*t round constants, M the t x t MDS matrix."""
    if r_p is None:
        r_p = N_ROUNDS_P[t - 2]
    n = p.bit_length()
    g = _Grain(1, 0, n, t, r_f, r_p)
    C = []
    while len(C) < (r_f + r_p) * t:
        v = g.bits(n)
        while v >= p:
            v = g.bits(n)
        C.append(v)
    while True:
        rl = [g.bits(n) % p for _ in range(2 * t)]
        while len(set(rl)) != len(rl):
            rl = [g.bits(n) % p for _ in range(2 * t)]
        xs, ys = rl[:t], rl[t:]
        ok = all((xs[i] + ys[j]) % p != 0 for i in range(t) for j in range(t))
        if not ok:
            continue
        M = [[pow(xs[i] + ys[j], -1, p) for j in range(t)] for i in range(t)]
        return tuple(C), tuple(tuple(r) for r in M)


def poseidon_hash(p: int, inputs, transpose=False):
    """Plain-integer Poseidon permutation (reference for the circuit's expected output)."""
    t = len(inputs) + 1
    r_p = N_ROUNDS_P[t - 2]
    C, M = poseidon_params(p, t)
    st = [0] + [x % p for x in inputs]
    for r in range(N_ROUNDS_F + r_p):
        st = [(st[j] + C[r * t + j]) % p for j in range(t)]
        if r < N_ROUNDS_F // 2 or r >= N_ROUNDS_F // 2 + r_p:
            st = [pow(x, 5, p) for x in st]
        else:
            st[0] = pow(st[0], 5, p)
        if transpose:
            st = [sum(M[j][i] * st[j] for j in range(t)) % p for i in range(t)]
        else:
            st = [sum(M[i][j] * st[j] for j in range(t)) % p for i in range(t)]
    return st[0]
