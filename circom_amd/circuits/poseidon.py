"""Poseidon hash circuit in the structure of circomlib's `poseidon.circom` (Sigma / Ark / Mix
templates, x^5 S-box, 8 full rounds + N_ROUNDS_P[t-2] partial rounds, capacity element first).

circomlib is not available here, so the circuit is re-authored from the
algorithm; its constants come from `poseidon_constants.py` and its output is pinned to the
published test vector poseidon([1,2]) in tests/test_poseidon.py.
"""
from ..frontend.dsl import template
from .poseidon_constants import poseidon_params, N_ROUNDS_F, N_ROUNDS_P


@template
def Sigma(c):
    inp = c.input("in")
    out = c.output("out")
    in2 = c.signal("in2")
    in4 = c.signal("in4")
    c.set(in2, inp * inp)
    c.set(in4, in2 * in2)
    c.set(out, in4 * inp)


@template
def Ark(c, t, C, r):
    inp = c.input("in", t)
    out = c.output("out", t)
    for i in range(t):
        c.set(out[i], inp[i] + C[i + r])


@template
def Mix(c, t, M):
    inp = c.input("in", t)
    out = c.output("out", t)
    for i in range(t):
        lc = c.const(0)
        for j in range(t):
            lc = lc + M[i][j] * inp[j]
        c.set(out[i], lc)


@template
def Poseidon(c, nInputs):
    inputs = c.input("inputs", nInputs)
    out = c.output("out")
    t = nInputs + 1
    nRoundsF = N_ROUNDS_F
    nRoundsP = N_ROUNDS_P[t - 2]
    C, M = poseidon_params(c.fp.q, t)
    ark, mix = [], []
    for i in range(nRoundsF + nRoundsP):
        a = c.component("ark", Ark(t, C, t * i), i)
        ark.append(a)
        for j in range(t):
            if i == 0:
                if j > 0:
                    c.set(a["in"][j], inputs[j - 1])
                else:
                    c.set(a["in"][j], 0)
            else:
                c.set(a["in"][j], mix[i - 1]["out"][j])
        m = c.component("mix", Mix(t, M), i)
        mix.append(m)
        if i < nRoundsF // 2 or i >= nRoundsP + nRoundsF // 2:
            k = i if i < nRoundsF // 2 else i - nRoundsP
            for j in range(t):
                s = c.component("sigmaF", Sigma(), (k, j))
                c.set(s["in"], a["out"][j])
                c.set(m["in"][j], s["out"])
        else:
            k = i - nRoundsF // 2
            s = c.component("sigmaP", Sigma(), k)
            c.set(s["in"], a["out"][0])
            c.set(m["in"][0], s["out"])
            for j in range(1, t):
                c.set(m["in"][j], a["out"][j])
    c.set(out, mix[-1]["out"][0])
