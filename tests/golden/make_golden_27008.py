#!/usr/bin/env python3
"""Generate tests/golden/reference_wtns_sha256_27008.json by RUNNING THE REFERENCE's own C++ witness calculator (oracle/_ref,
compiled from /root/reference by oracle/Makefile) on the 53-block SHA-256 - the 1.07 M-constraint SHA at the reference's
default `--O1` (10.8 M constraints at `--O0`) - for two fixed messages.  Inputs are kept packed (hex of the message bytes, msb
first = input order); per vector the SHA-256 of the reference's 346 MB `.wtns`, its length and the digest bits.

    CW_CACHE=/tmp/cw_cache_27008 CW_ARTEFACT_FP=r06b python tests/golden/make_golden_27008.py
(the lowered artefacts of the circuit are reused when the cache has them; lowering takes ~30 minutes otherwise)"""
import hashlib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import bench                # noqa: E402
from oracle import ref_build  # noqa: E402

NAME, NBITS = "sha256_27008", 27008


def main():
    cache = os.environ.get("CW_CACHE") or os.path.join(tempfile.gettempdir(), "cw_cache_27008")
    cp, _, _ = bench.get_compiled(NAME, 1 << 19, cache, 0, None)
    ref_build.build_circuit(cp)
    rng = np.random.default_rng(27008)
    msgs = [bytes(NBITS // 8), rng.integers(0, 256, size=NBITS // 8, dtype=np.uint8).tobytes()]
    rows = np.zeros((len(msgs), NBITS, 32), dtype=np.uint8)
    for i, m in enumerate(msgs):
        rows[i, :, 0] = np.unpackbits(np.frombuffer(m, dtype=np.uint8))
    td = tempfile.mkdtemp(prefix="cw_gold27008_")
    pre = os.path.join(td, "r_")
    ref_build.run_loop(cp, rows.tobytes(), len(msgs), 1, wtns_prefix=pre)
    vecs = []
    for i, m in enumerate(msgs):
        h = hashlib.sha256()
        n = 0
        with open(pre + "%d.wtns" % i, "rb") as f:
            head = f.read(76 + 32 * 257)
            h.update(head)
            n += len(head)
            while True:
                blk = f.read(1 << 24)
                if not blk:
                    break
                h.update(blk)
                n += len(blk)
        # the digest signals (witness entries 1..256) as the circuit's own output must be SHA-256 of the message
        bits = [head[76 + 32 * (1 + k)] for k in range(256)]
        assert np.packbits(np.array(bits, dtype=np.uint8)).tobytes() == hashlib.sha256(m).digest()
        vecs.append({"message_hex": m.hex(), "wtns_sha256": h.hexdigest(), "wtns_len": n, "digest_hex": hashlib.sha256(m).hexdigest()})
        os.unlink(pre + "%d.wtns" % i)
    out = {"generator": "tests/golden/make_golden_27008.py", "circuit": "Sha256(27008), bn128, --O0: %d signals" % cp.flat.n_signals,
           "runtime": "reference C++ runtime (common/main.cpp + calcwit.cpp + generic/fr.cpp), oracle/_ref build", "vectors": vecs}
    with open(os.path.join(ROOT, "tests", "golden", "reference_wtns_sha256_27008.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out)[:400])


if __name__ == "__main__":
    main()
