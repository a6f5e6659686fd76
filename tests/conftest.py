import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

REF_ROOT = Path(os.environ.get("CIRCOM_REF", "/root/reference"))


# The CPU suite does not assemble code objects that only a GPU can run: compile_program's default ("auto") emits the rows of
# every strand variant of every arithmetic circuit as gfx950 code (6 programs per circuit, ~150 compilations in this suite =
# a quarter of an hour of assembling).  Without a GPU device node the default is switched off here; tests/test_fpjit.py builds
# and replays the emitted code explicitly (hip_elements.fpjit.emit), the loader-robustness test re-enables it for its two
# circuits, and on a GPU box nothing is switched off: every arithmetic circuit of the GPU suite runs through the emitted code.
if not os.path.exists("/dev/kfd"):
    os.environ.setdefault("CW_FPJIT", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


def emit_for_gpu():
    """`fpjit=` argument for fixtures that compile LARGE arithmetic circuits shared by CPU and GPU tests: the emitted code of a
    40 000-signal circuit is six code objects of 12-18 MB (a minute of assembling) that only a GPU run ever executes, so the CPU
    suite compiles those circuits without it; small circuits always carry it (tests/test_fpjit.py covers both sides)."""
    import os
    return "auto" if os.path.exists("/dev/kfd") else False


def ensure_ref(prime: str) -> Path:
    """Make sure oracle/_ref/<prime>/ holds the compiled reference runtime; build it if the
    reference tree is present, otherwise skip (the GPU box only has the prebuilt files)."""
    out = ROOT / "oracle" / "_ref" / prime
    need = [out / n for n in ("libfr_shim.so", "main.o", "calcwit.o", "fr.o")]
    if not all(p.exists() for p in need):
        if not REF_ROOT.exists():
            pytest.skip("oracle/_ref not built and reference tree absent")
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "ref", f"PRIME={prime}", f"REF={REF_ROOT}"],
                       check=True, capture_output=True)
    return out


@pytest.fixture(scope="session")
def ref_dir_bn128():
    return ensure_ref("bn128")


@pytest.fixture(scope="session")
def ref_dir_bls12381():
    return ensure_ref("bls12381")
