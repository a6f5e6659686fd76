cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--workload semaphore20p --total-batch 8192 --shard-of 8 --steps 3 --warmup 1 --in-flight 1 --no-cpu-baseline --no-parity"
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_a -- python $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_IFETCH --output-format csv -d $R/gpurun_out/pmc_b -- python $R/bench.py $ARGS > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ('pmc_a','pmc_b'):
    fs=glob.glob('gpurun_out/%s/**/*counter_collection.csv'%d, recursive=True)
    if not fs: print(d,'no file'); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'cw_eval_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d, {c: '%.3g' % (sum(x)/len(x)) for c,x in acc.items()})
PY
