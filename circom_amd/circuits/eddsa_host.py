"""Plain-integer BabyJubjub / EdDSA-Poseidon / Merkle helpers used to SYNTHESISE valid inputs for the
Semaphore-style circuit (tests and bench need signatures that verify) and to pin the
circuit's outputs.  Host-side Python only; nothing here runs on the device."""
from __future__ import annotations

import random

from .babyjub import A, D, BASE8
from .eddsa import SUBGROUP_ORDER
from .poseidon_constants import poseidon_hash


def ed_add(p1, p2, q):
    x1, y1 = p1
    x2, y2 = p2
    t = D * x1 * x2 * y1 * y2 % q
    return ((x1 * y2 + y1 * x2) * pow(1 + t, -1, q) % q, (y1 * y2 - A * x1 * x2) * pow(1 - t, -1, q) % q)


def ed_mul(k, p, q):
    acc = (0, 1)
    for i in range(k.bit_length() - 1, -1, -1):
        acc = ed_add(acc, acc, q)
        if (k >> i) & 1:
            acc = ed_add(acc, p, q)
    return acc


def keygen(q, rng: random.Random):
    s = rng.randrange(1, SUBGROUP_ORDER)
    return s, ed_mul(s, BASE8, q)


def sign(q, s, A_pt, msg, rng: random.Random):
    """(R8, S) with S*B8 = R8 + (8*h)*A, h = Poseidon(R8x, R8y, Ax, Ay, msg); A = s*B8."""
    r = rng.randrange(1, SUBGROUP_ORDER)
    R8 = ed_mul(r, BASE8, q)
    h = poseidon_hash(q, [R8[0], R8[1], A_pt[0], A_pt[1], msg])
    S = (r + 8 * h * s) % SUBGROUP_ORDER
    return R8, S


def verify(q, A_pt, msg, R8, S):
    h = poseidon_hash(q, [R8[0], R8[1], A_pt[0], A_pt[1], msg])
    right = ed_add(R8, ed_mul(h, ed_mul(8, A_pt, q), q), q)
    return S < SUBGROUP_ORDER and ed_mul(S, BASE8, q) == right


def merkle_root(q, leaf, path_indices, siblings):
    h = leaf
    for bit, sib in zip(path_indices, siblings):
        h = poseidon_hash(q, [sib, h] if bit else [h, sib])
    return h


def semaphore_inputs(q, n_levels, rng: random.Random):
    """One valid input vector, in the circuit's declaration order, plus the expected (root, nullifierHash)."""
    s, A_pt = keygen(q, rng)
    msg = rng.randrange(q)
    R8, S = sign(q, s, A_pt, msg, rng)
    idx = [rng.randrange(2) for _ in range(n_levels)]
    sib = [rng.randrange(q) for _ in range(n_levels)]
    leaf = poseidon_hash(q, [A_pt[0], A_pt[1]])
    root = merkle_root(q, leaf, idx, sib)
    nullifier = poseidon_hash(q, [A_pt[0], A_pt[1], msg])
    row = [A_pt[0], A_pt[1], S, R8[0], R8[1], msg] + idx + sib
    return row, (root, nullifier)
