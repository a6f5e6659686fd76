"""`python bench.py --gpus 2` started by hand must start TWO ranks itself (the flag used to be parsed
and ignored).  Rehearsed on CPU: --host-only swaps RCCL for gloo and device batches for host-only ones; everything else
- the re-exec under torch.distributed.run, rank 0 compiling once, the shards, the one gather - is the code the GPU run
uses."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(args, tmp_path, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args + ["--cache-dir", str(tmp_path / "cache")],
                          capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))


def test_bench_gpus_2_starts_two_ranks(tmp_path):
    r = _run(["--gpus", "2", "--host-only", "--workload", "poseidon2", "--total-batch", "101"], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 prints the one JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["host_only"] is True
    assert out["gathered"] == {"status_words": 101, "public_signal_rows": 101}
    assert out["config"]["compile_cached"] is False


def test_bench_refuses_a_world_that_is_not_what_was_asked_for(tmp_path):
    # a launcher environment of ONE rank with --gpus 2 must not print an `n_gpus: 1` line
    r = _run(["--gpus", "2", "--host-only", "--workload", "poseidon2"], tmp_path,
             {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_bench_gpus_8_weak_scaling_launch(tmp_path):
    """the launch the driver uses for the scaling curve: 8 ranks, weak scaling (every rank its own --batch instances)"""
    r = _run(["--gpus", "8", "--host-only", "--workload", "poseidon2", "--batch", "37"], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["host_only"] is True
    assert out["gathered"] == {"status_words": 8 * 37, "public_signal_rows": 8 * 37}


def test_config5_job_shape_over_8_ranks(tmp_path):
    """BASELINE config 5's launch - 1 024 instances of a tier-2 circuit on the BLS12-381 prime over 8 ranks - rehearsed with the
    verifier's building block (bigmultmodp: the same functions, the same launch / shard / gather code; the verifier's own
    artefacts take minutes to lower): every rank owns 128 instances, and the one exchange is a status word + the public signals
    (here the k = 3 output limbs) of every instance, from each peer to rank 0"""
    r = _run(["--gpus", "8", "--host-only", "--workload", "bigmultmodp", "--total-batch", "1024"], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong"
    assert out["shards"] == [128] * 8 and out["gathered"] == {"status_words": 1024, "public_signal_rows": 1024}
    assert out["gather_bytes_per_peer"] == (4 + 32 * 3) * 128 and 0.02 < out["predicted_gather_ms"] < 0.03
    sys.path.insert(0, str(ROOT))
    from circom_amd.sharding import shard_range
    assert [shard_range(1024, r_, 8) for r_ in (0, 7)] == [(0, 128), (896, 1024)]
    assert [b - a for a, b in (shard_range(1000, r_, 8) for r_ in range(8))] == [125] * 8


def test_batches_in_flight_follow_the_fill_of_the_chip():
    sys.path.insert(0, str(ROOT))
    import bench
    # bit-plane: waves = groups x slices; the 65 536-instance headline fills the SIMDs (two in flight), 4 096 a quarter (four)
    assert bench.auto_in_flight(True, 65536, 64) == 2 and bench.auto_in_flight(True, 4096, 16) == 4
    # 256-bit engine: the 1 024-instance shard of BASELINE config 4 is 64 workgroups (eight), 8 192 is the whole chip (two)
    assert bench.auto_in_flight(False, 1024, 16) == 8 and bench.auto_in_flight(False, 8192, 32) == 2
    assert bench.auto_in_flight(False, 65536, 64) == 2
