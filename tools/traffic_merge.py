"""Merge entries of gpurun_out/traffic.json (written on the GPU box by tools/summarize_prof.py) into profiles/traffic.json and
name the sources that shape that workload's kernels: bench.py quotes an entry only while those files are byte-identical.
usage: python tools/traffic_merge.py <key> <profile note> <file>..."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench

key, note, files = sys.argv[1], sys.argv[2], sys.argv[3:]
new = json.load(open(ROOT / "gpurun_out" / "traffic.json"))[key]
new["kernel_sources"] = {"files": files, "profile": note, "sha": bench.files_fingerprint(files)}
cur = json.load(open(ROOT / "profiles" / "traffic.json"))
cur[key] = new
json.dump(cur, open(ROOT / "profiles" / "traffic.json", "w"), indent=1, sort_keys=True)
print(key, {k: v for k, v in new.items() if k in ("eval", "r1cs", "eval_avg_us", "r1cs_avg_us", "eval_valu_frac", "r1cs_valu_frac", "eval_wait_frac")})
