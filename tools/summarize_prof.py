"""Summarise a tools/profile.sh output directory: per-kernel average duration (kernel trace) and PMC sums per
dispatch (FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; gfx950 note: FETCH_SIZE under-reports wide
coalesced reads by 2x — MI355X_MICROARCH.md §HBM)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(root, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


dur = defaultdict(list)
for r in rows("trace/**/*kernel_trace.csv"):
    try:
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    except (KeyError, ValueError):
        pass
print("kernel trace (us): name, calls, avg, min, max")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("  %-60s %5d %10.1f %10.1f %10.1f" % (k[:60], len(v), sum(v) / len(v), min(v), max(v)))
for tag in ("pmc_fetch", "pmc_write", "pmc_sq"):
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows(tag + "/**/*counter_collection.csv"):
        try:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        except (KeyError, ValueError):
            pass
    if acc:
        print("%s: per-dispatch average of each counter" % tag)
        for k, cs in acc.items():
            if "cw_" not in k:
                continue
            print("  %-50s %s" % (k[:50], "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in cs.items())))

# traffic.json for bench.py: HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE counts
# 64 B per 128-B request on wide coalesced reads; WRITE_SIZE as reported)
if len(sys.argv) > 3:
    import json
    key, out = sys.argv[2], sys.argv[3]
    fetch, write = {}, {}
    for tag, dst in (("pmc_fetch", fetch), ("pmc_write", write)):
        tmp = defaultdict(list)
        for r in rows(tag + "/**/*counter_collection.csv"):
            try:
                tmp[r["Kernel_Name"]].append(float(r["Counter_Value"]))
            except (KeyError, ValueError):
                pass
        for k, v in tmp.items():
            dst[k] = sum(v) / len(v)
    def short_name(k):
        if "ingest" in k:
            return "ingest"
        if "gather" in k:
            return "gather"
        if "assert_kernel" in k or "init_kernel" in k:
            return None
        if "eval_kernel" in k or "pipe_kernel" in k or "cw_bits_jit" in k or "cw_fp_jit" in k:
            return "eval"
        if "r1cs" in k:
            return "r1cs"
        return None

    res = {}
    for k in fetch:
        short = short_name(k)
        if short:                                       # the R1CS check may be several kernels: their traffic adds up
            res[short] = res.get(short, 0.0) + (2 * fetch[k] + write.get(k, 0.0)) * 1024.0
    # kernel durations from the kernel trace (the R1CS check is several kernels per step: their averages add up)
    for k, v in dur.items():
        short = short_name(k)
        if short:
            res[short + "_avg_us"] = res.get(short + "_avg_us", 0.0) + sum(v) / len(v)
            res[short + "_min_us"] = res.get(short + "_min_us", 0.0) + min(v)
    # instruction counts and the clock the kernel ran at, from the SQ pass.  GRBM_GUI_ACTIVE is reported SUMMED over the
    # 8 XCDs, so one XCD's clock = GRBM_GUI_ACTIVE / 8 / duration (of the SAME pass: the profiled clock differs from the
    # un-profiled one); VALU fraction = SQ_INSTS_VALU / (1024 SIMDs x GUI cycles / 2): a SIMD-32 issues a wave64 VALU
    # instruction over 2 clocks (MI355X_MICROARCH.md; tools/ubench_isa measures 2.05 with >= 2 waves per SIMD)
    sq = defaultdict(lambda: defaultdict(list))
    sq_dur = defaultdict(list)
    for r in rows("pmc_sq/**/*counter_collection.csv"):
        try:
            sq[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        except (KeyError, ValueError):
            pass
    for r in rows("pmc_sq/**/*kernel_trace.csv"):
        try:
            sq_dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        except (KeyError, ValueError):
            pass
    N_XCD = 8.0
    agg = defaultdict(lambda: defaultdict(float))
    for k, cs in sq.items():
        short = short_name(k)
        if short:
            for c, v in cs.items():
                agg[short][c] += sum(v) / len(v)
            if sq_dur.get(k):
                agg[short]["_dur_us"] += sum(sq_dur[k]) / len(sq_dur[k])
    for short, cs in agg.items():
        if cs.get("SQ_INSTS_VALU") and cs.get("GRBM_GUI_ACTIVE"):
            gui = cs["GRBM_GUI_ACTIVE"] / N_XCD
            res[short + "_valu_insts"] = cs["SQ_INSTS_VALU"]
            res[short + "_valu_frac"] = cs["SQ_INSTS_VALU"] / (1024.0 * gui / 2.0)
            res[short + "_gui_cycles"] = gui
            if cs.get("_dur_us"):
                res[short + "_clock_hz"] = gui / (cs["_dur_us"] * 1e-6)
            if cs.get("SQ_WAIT_ANY") and cs.get("SQ_WAVE_CYCLES"):
                res[short + "_wait_frac"] = cs["SQ_WAIT_ANY"] / max(cs["SQ_WAVE_CYCLES"], 1.0)
            for c in ("SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT",
                      "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU"):
                if cs.get(c):
                    res[short + "_" + c.lower()] = cs[c]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res["source"] = bench.source_fingerprint()          # bench.py only quotes these figures for the source they were measured on
    try:
        cur = json.load(open(out))
    except Exception:
        cur = {}
    cur[key] = res
    json.dump(cur, open(out, "w"), indent=1, sort_keys=True)
    print("traffic", key, res)
