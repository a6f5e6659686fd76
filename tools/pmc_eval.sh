R=$PWD; OUT=$R/gpurun_out/pmc_eval; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-small --in-flight 1 --fp-bench-lanes 65536 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_BUSY_max TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-small --in-flight 1 --fp-bench-lanes 65536 > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-small --in-flight 1 --fp-bench-lanes 65536 > $OUT/c.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for sub in "abc":
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("/root/repo/gpurun_out/pmc_eval/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "bits" in k:
            print(sub, k, {c: "%.3g" % (sum(v)/len(v)) for c, v in cs.items()})
PY
tail -3 $OUT/b.log
