"""Poseidon Merkle-tree inclusion proof (the core of Semaphore / Tornado-style circuits, BASELINE config 4's
"Poseidon Merkle depth 20" part) in the structure of the usual circomlib-based implementations:
MultiMux1 selects (left, right) by a path-index bit, Poseidon(2) hashes the pair, level by level.
Re-authored (circomlib absent); the root is pinned in tests against a plain-integer recomputation."""
from ..frontend.dsl import template
from .poseidon import Poseidon


@template
def MultiMux1(c, n):
    # circomlib mux1.circom: out[i] = (c[i][1] - c[i][0]) * s + c[i][0]
    cc = c.input("c", n, 2)
    s = c.input("s")
    out = c.output("out", n)
    for i in range(n):
        c.set(out[i], (cc[i][1] - cc[i][0]) * s + cc[i][0])


@template
def MerkleTreeInclusionProof(c, nLevels):
    leaf = c.input("leaf")
    pathIndices = c.input("pathIndices", nLevels)
    siblings = c.input("siblings", nLevels)
    root = c.output("root")
    hashes = c.signal("hashes", nLevels + 1)
    c.set(hashes[0], leaf)
    for i in range(nLevels):
        c.enforce(pathIndices[i] * (1 - pathIndices[i]), 0)
        mux = c.component("mux", MultiMux1(2), i)
        c.set(mux["c"][0][0], hashes[i])
        c.set(mux["c"][0][1], siblings[i])
        c.set(mux["c"][1][0], siblings[i])
        c.set(mux["c"][1][1], hashes[i])
        c.set(mux["s"], pathIndices[i])
        h = c.component("poseidons", Poseidon(2), i)
        c.set(h["inputs"][0], mux["out"][0])
        c.set(h["inputs"][1], mux["out"][1])
        c.set(hashes[i + 1], h["out"])
    c.set(root, hashes[nLevels])
