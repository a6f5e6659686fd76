#!/usr/bin/env python3
"""Generate tests/golden/reference_wtns.json by RUNNING THE REFERENCE's own C++ witness calculator
(common/main.cpp + calcwit.cpp + rendered generic/fr.cpp, compiled from /root/reference by oracle/Makefile) on fixed
inputs, in the container that has the reference tree.  The GPU box has no reference tree; the fixtures travel.

For every case the file records the inputs (decimal strings, main-component declaration order), the SHA-256 of the
reference's `.wtns` file, its length, and the first witness values (constant 1, outputs, inputs).  Small circuits
also keep the whole `.wtns` (hex).  tests/test_golden_wtns.py checks the Python oracle (CPU) and the HIP path (GPU)
against these bytes.

    python tests/golden/make_golden.py          # rewrites reference_wtns.json
"""
import hashlib
import json
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from circom_amd.compiler import compile_program           # noqa: E402
from circom_amd.frontend.dsl import Program                # noqa: E402
from oracle import ref_build                               # noqa: E402
from oracle.field import PRIMES                            # noqa: E402


def cases():
    """name -> (builder of the Program, prime, list of input rows)"""
    from circom_amd.circuits.basic import Multiplier2, Num2Bits, IsZero
    from circom_amd.circuits.poseidon import Poseidon
    from circom_amd.circuits.sha256 import Sha256
    from circom_amd.circuits.eddsa import SemaphoreStyle
    from circom_amd.circuits import eddsa_host as H
    q = PRIMES["bn128"]
    qb = PRIMES["bls12381"]
    rng = random.Random(20250923)
    out = {}
    out["multiplier2"] = (lambda: Program(Multiplier2()), "bn128",
                          [[3, 11], [0, 0], [q - 1, q - 1], [rng.randrange(q), rng.randrange(q)]])
    out["multiplier2_bls12381"] = (lambda: Program(Multiplier2(), prime="bls12381"), "bls12381",
                                   [[3, 11], [qb - 1, 2], [rng.randrange(qb), rng.randrange(qb)]])
    out["poseidon2"] = (lambda: Program(Poseidon(2)), "bn128",
                        [[1, 2], [0, 0], [q - 1, q - 1], [5, 2 ** 31 - 1], [rng.randrange(q), rng.randrange(q)]])
    out["poseidon2_bls12381"] = (lambda: Program(Poseidon(2), prime="bls12381"), "bls12381",
                                 [[1, 2], [rng.randrange(qb), rng.randrange(qb)]])
    out["num2bits16"] = (lambda: Program(Num2Bits(16)), "bn128", [[0], [1], [43690], [65535]])
    out["iszero"] = (lambda: Program(IsZero()), "bn128", [[0], [1], [q - 1], [rng.randrange(q)]])
    out["sha256_512"] = (lambda: Program(Sha256(512)), "bn128",
                         [[0] * 512, [1] * 512, [rng.randrange(2) for _ in range(512)]])
    # the BASELINE metric's workload: >= 1M constraints at --O0 (5 compression blocks)
    r3 = random.Random(2048)
    out["sha256_2048"] = (lambda: Program(Sha256(2048)), "bn128",
                          [[0] * 2048, [r3.randrange(2) for _ in range(2048)], [r3.randrange(2) for _ in range(2048)]])
    from circom_amd.circuits.opzoo import OperatorZoo
    half = q >> 1
    out["opzoo"] = (lambda: Program(OperatorZoo()), "bn128",
                    [[3, 11], [0, 0], [q - 1, q - 1], [half + 1, 255], [1 << 200, q - 3], [q - 2, 254], [12345678901234567890, 64],
                     [rng.randrange(q), rng.randrange(q)], [rng.randrange(q), rng.randrange(256)]])
    r2 = random.Random(77)
    out["semaphore20"] = (lambda: Program(SemaphoreStyle(20)), "bn128",
                          [H.semaphore_inputs(q, 20, r2)[0] for _ in range(2)])
    # the same relation with the scalar multiplications' hints computed on a projective ladder (circuits/babyjub.py)
    r4 = random.Random(78)
    out["semaphore20p"] = (lambda: Program(SemaphoreStyle(20, True)), "bn128",
                           [H.semaphore_inputs(q, 20, r4)[0] for _ in range(2)])
    # a Mixed component cluster (one template, different parameters per array element): the reference runtime executes it
    # through the io-map section of the .dat and _functionTable (store_bucket.rs:498-566)
    from circom_amd.circuits.basic import MixedArray
    r5 = random.Random(13)
    out["mixed_array"] = (lambda: Program(MixedArray(((2, 3), (1, 5), (3, 2), (2, 3)))), "bn128",
                          [list(range(1, 9)), [0] * 8, [q - 1] * 8, [r5.randrange(q) for _ in range(8)]])
    return out


def log_cases():
    """name -> (builder, prime, input rows as {name: value}): circuits whose log(...) output is pinned (stdout of the
    reference binary, main.cpp + the LogBucket code of log_bucket.rs:105-162)"""
    from circom_amd.circuits.basic import LogDemo
    q = PRIMES["bn128"]
    r = random.Random(5)
    return {"logdemo": (lambda: Program(LogDemo()), "bn128",
                        [{"a": 3, "b": 4}, {"a": 0, "b": 0}, {"a": q - 1, "b": 2}, {"a": 13, "b": 1},
                         {"a": r.randrange(q), "b": r.randrange(q)}])}


def make_logs():
    """tests/golden/reference_logs.json: stdout, exit status and .wtns digest of the reference CLI per input row.  For an
    instance that fails a run-time check only the lines BEFORE the reference's own failure trace are kept (`log`)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_logs.json")
    result = {"generator": "tests/golden/make_golden.py logs", "cases": {}}
    for name, (mk, prime, rows) in log_cases().items():
        d = tempfile.mkdtemp(prefix="goldenlog_")
        cp = compile_program(mk(), d, name, sym=False, strands=(1,))
        ref_build.build_circuit(cp)
        entries = []
        for row in rows:
            out = os.path.join(d, "o.wtns")
            if os.path.exists(out):
                os.unlink(out)
            r = ref_build.run_cli(cp, json.dumps({k: str(v) for k, v in row.items()}), out)
            ok = r.returncode == 0
            text = r.stdout if ok else r.stdout[:r.stdout.index("Failed assert")]
            entries.append({"inputs": {k: str(v) for k, v in row.items()}, "ok": ok, "log": text,
                            "wtns_sha256": hashlib.sha256(open(out, "rb").read()).hexdigest() if ok else None})
        result["cases"][name] = {"prime": prime, "vectors": entries}
        print(name, len(rows), "vectors")
    with open(path, "w") as f:
        json.dump(result, f, indent=1)


def goldilocks_cases():
    """name -> (builder, input rows): circuits on the 64-bit Goldilocks prime, run through the reference's OWN 64-bit
    runtime (goldilocks/fr.hpp + common64/{main,calcwit}.cpp).  The device path does not take this prime yet;
    the fixtures pin the ORACLE for it."""
    from circom_amd.circuits.basic import Multiplier2, IsZero, Num2Bits
    from circom_amd.circuits.opzoo import OperatorZoo
    from circom_amd.circuits.poseidon import Poseidon
    q = PRIMES["goldilocks"]
    r = random.Random(64)
    return {
        "multiplier2": (lambda: Program(Multiplier2(), prime="goldilocks"), [[3, 11], [0, 0], [q - 1, q - 1], [r.randrange(q), r.randrange(q)]]),
        "iszero": (lambda: Program(IsZero(), prime="goldilocks"), [[0], [1], [q - 1], [r.randrange(q)]]),
        "num2bits16": (lambda: Program(Num2Bits(16), prime="goldilocks"), [[0], [1], [43690], [65535]]),
        "opzoo": (lambda: Program(OperatorZoo(), prime="goldilocks"),
                  [[3, 11], [0, 0], [q - 1, q - 1], [(q >> 1) + 1, 63], [1 << 40, q - 3], [q - 2, 7], [12345678901234567890 % q, 64],
                   [r.randrange(q), r.randrange(q)], [r.randrange(q), r.randrange(64)]]),
        "poseidon2": (lambda: Program(Poseidon(2), prime="goldilocks"), [[1, 2], [0, 0], [q - 1, q - 1], [r.randrange(q), r.randrange(q)]]),
    }


def make_goldilocks():
    from circom_amd.frontend.flatten import flatten
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_wtns_goldilocks.json")
    result = {"generator": "tests/golden/make_golden.py goldilocks",
              "runtime": "reference goldilocks/fr.hpp + common64/{main,calcwit}.cpp (oracle/Makefile circuit64)", "cases": {}}
    for name, (mk, rows) in goldilocks_cases().items():
        fc = flatten(mk())
        cli = ref_build.build_circuit64(fc, name)
        d = tempfile.mkdtemp(prefix="golden64_")
        entries = []
        for row in rows:
            obj, pos = {}, 0
            for nm, _, size in fc.inputs:
                dims = fc.input_dims.get(nm) or []
                obj[nm] = str(row[pos]) if not dims else [str(v) for v in row[pos:pos + size]]
                pos += size
            out = os.path.join(d, "o.wtns")
            r = ref_build.run_cli64(cli, json.dumps(obj), out)
            assert r.returncode == 0, r.stderr
            b = open(out, "rb").read()
            entries.append({"inputs": [str(v) for v in row], "wtns_sha256": hashlib.sha256(b).hexdigest(), "wtns_len": len(b),
                            "wtns_hex": b.hex() if len(b) <= 4096 else None})
        result["cases"][name] = {"prime": "goldilocks", "vectors": entries}
        print(name, len(rows), "vectors", entries[0]["wtns_len"], "bytes each")
    with open(path, "w") as f:
        json.dump(result, f, indent=1)


def ecdsa_case():
    """BASELINE config 5: secp256k1 ECDSA verification over the BLS12-381 scalar field (circuits/secp256k1.py), n = 64, k = 4,
    stride 8; three valid signatures and one with another message hash (result = 0, every constraint still satisfied)"""
    from circom_amd.circuits import secp256k1 as S
    r = random.Random(5005)
    rows = [S.sign(S.SECP256K1, 64, 4, r) for _ in range(3)]
    bad = list(rows[0])
    bad[8] ^= 1
    rows.append(bad)
    return (lambda: Program(S.ECDSAVerifyNoPubkeyCheck(64, 4, S.SECP256K1, 8), prime="bls12381")), "bls12381", rows


def make_ecdsa():
    """`make_golden.py ecdsa` -> reference_wtns_ecdsa.json: the reference RUNTIME executes the emitted C++ of the verifier,
    bodies of the witness functions included (a Fermat inverse per curve operation: ~11 s per witness), and writes the 79 MB
    `.wtns` files whose digests are kept"""
    mk, prime, rows = ecdsa_case()
    d = tempfile.mkdtemp(prefix="golden_ecdsa_")
    cp = compile_program(mk(), d, "ecdsa_verify", sym=False, strands=(1,))
    ref_build.build_circuit(cp)
    raw = b"".join(int(v).to_bytes(32, "little") for r in rows for v in r)
    pre = os.path.join(d, "g_")
    ref_build.run_loop(cp, raw, len(rows), 1, wtns_prefix=pre)
    entries = []
    for i, r in enumerate(rows):
        b = open(pre + "%d.wtns" % i, "rb").read()
        nw = int.from_bytes(b[60:64], "little")
        head = [str(int.from_bytes(b[76 + 32 * k:108 + 32 * k], "little")) for k in range(8)]
        entries.append({"inputs": [str(v) for v in r], "wtns_sha256": hashlib.sha256(b).hexdigest(), "wtns_len": len(b),
                        "n_witness": nw, "witness_head": head})
        os.unlink(pre + "%d.wtns" % i)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_wtns_ecdsa.json")
    json.dump({"generator": "tests/golden/make_golden.py ecdsa", "runtime": "reference common/{main,calcwit}.cpp + generic/fr.cpp (GMP, no asm), bls12381",
               "n_signals": cp.flat.n_signals, "n_constraints": len(cp.flat.constraints),
               "cases": {"ecdsa_verify": {"prime": prime, "vectors": entries}}}, open(path, "w"), indent=1)
    print("wrote", path)


def main():
    """`make_golden.py` regenerates everything; `make_golden.py NAME...` only (re)generates the named cases and keeps
    the other entries of the JSON as they are."""
    if sys.argv[1:] == ["logs"]:
        return make_logs()
    if sys.argv[1:] == ["ecdsa"]:
        return make_ecdsa()
    if sys.argv[1:] == ["goldilocks"]:
        return make_goldilocks()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_wtns.json")
    only = set(sys.argv[1:])
    result = {"generator": "tests/golden/make_golden.py", "runtime": "reference common/{main,calcwit}.cpp + generic/fr.cpp (GMP, no asm)",
              "cases": {}}
    if only:
        result = json.load(open(path))
    for name, (mk, prime, rows) in cases().items():
        if only and name not in only:
            continue
        d = tempfile.mkdtemp(prefix="golden_")
        cp = compile_program(mk(), d, name.replace("_bls12381", ""), sym=False, strands=(1,))
        ref_build.build_circuit(cp)
        raw = b"".join(int(v).to_bytes(32, "little") for r in rows for v in r)
        pre = os.path.join(d, "g_")
        ref_build.run_loop(cp, raw, len(rows), 1, wtns_prefix=pre)
        entries = []
        for i, r in enumerate(rows):
            b = open(pre + "%d.wtns" % i, "rb").read()
            nw = int.from_bytes(b[60:64], "little")          # wtns: 12 B file header, 12 B section header, n8, q, nWitness
            head = [str(int.from_bytes(b[76 + 32 * k:108 + 32 * k], "little")) for k in range(min(nw, 8))]
            e = {"inputs": [str(v) for v in r], "wtns_sha256": hashlib.sha256(b).hexdigest(), "wtns_len": len(b),
                 "n_witness": nw, "witness_head": head}
            if len(b) <= 4096:
                e["wtns_hex"] = b.hex()
            entries.append(e)
        result["cases"][name] = {"prime": prime, "vectors": entries}
        print(name, len(rows), "vectors", entries[0]["wtns_len"], "bytes each")
    with open(path, "w") as f:
        json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
