"""The 64-bit Goldilocks prime (SURVEY row f4's tail): ORACLE side.

The reference has a second runtime for this prime - `goldilocks/fr.hpp` (field elements are plain u64 values) with
`common64/{main,calcwit}.cpp` - and a different shape of emitted code (values instead of pointers: compute_bucket.rs:353,
store_bucket.rs:575-657, value_bucket.rs:82-86).  The device path refuses the prime (its "q is large" shortcuts);
what is pinned here is the oracle: `oracle/emit_ref_cpp.py` emits the 64-bit code shape, `oracle/Makefile circuit64`
builds the reference's own 64-bit runtime around it, and the Python restatement reproduces its `.wtns` files byte for
byte - from the committed fixtures (tests/golden/reference_wtns_goldilocks.json) everywhere, live where the reference
tree is present."""
import hashlib
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_golden import goldilocks_cases                                            # noqa: E402

from circom_amd.frontend.flatten import flatten                                      # noqa: E402
from circom_amd.hip_elements.writers import wtns_bytes                              # noqa: E402
from oracle.tape_eval import eval_flat                                               # noqa: E402

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_wtns_goldilocks.json")))["cases"]
CASES = goldilocks_cases()


@pytest.mark.parametrize("name", sorted(GOLD))
def test_oracle_reproduces_the_64_bit_runtimes_wtns(name):
    mk, rows = CASES[name]
    fc = flatten(mk())
    q = fc.fp.q
    assert q == 18446744069414584321 and fc.fp.n64 == 1
    vecs = GOLD[name]["vectors"]
    assert [v["inputs"] for v in vecs] == [[str(x) for x in r] for r in rows]          # fixtures match the generator
    for vec in vecs:
        inp = {fc.main_input_start + k: int(v) for k, v in enumerate(vec["inputs"])}
        sig, failed = eval_flat(q, fc.n_signals, fc.n_temps, fc.constants, fc.code, inp)
        assert failed is None
        b = wtns_bytes(q, sig)                                                          # n8 = 8: one u64 per witness element
        assert len(b) == vec["wtns_len"] and hashlib.sha256(b).hexdigest() == vec["wtns_sha256"], name
        if vec["wtns_hex"]:
            assert b.hex() == vec["wtns_hex"]


def test_reference_64_bit_runtime_live(tmp_path):
    from oracle import ref_build
    if not ref_build.REF_ROOT.exists():
        pytest.skip("no reference tree")
    mk, rows = CASES["opzoo"]
    fc = flatten(mk())
    cli = ref_build.build_circuit64(fc, "opzoo")
    for row in rows[:4]:
        r = ref_build.run_cli64(cli, json.dumps({"a": str(row[0]), "b": str(row[1])}), tmp_path / "o.wtns")
        assert r.returncode == 0, r.stderr
        sig, _ = eval_flat(fc.fp.q, fc.n_signals, fc.n_temps, fc.constants, fc.code,
                           {fc.main_input_start: row[0], fc.main_input_start + 1: row[1]})
        assert (tmp_path / "o.wtns").read_bytes() == wtns_bytes(fc.fp.q, sig)


def test_the_device_lowering_still_refuses_the_prime():
    from circom_amd.hip_elements.lower import lower
    mk, _ = CASES["multiplier2"]
    with pytest.raises(ValueError, match="goldilocks"):
        lower(flatten(mk()))
