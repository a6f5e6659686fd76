set -x
R=$PWD
cd /tmp && export TMPDIR=/tmp
for S in 16; do
CW_STRANDS=$S rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_sha_S$S -- python $R/bench.py --workload sha256_512 --batch 4096 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
CW_STRANDS=$S rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_sha2_S$S -- python $R/bench.py --workload sha256_512 --batch 4096 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob,collections
for d in ("pmc_sha_S$S","pmc_sha2_S$S"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'cw_eval' in r['Kernel_Name'] or 'cw_r1cs' in r['Kernel_Name']:
                acc[r['Kernel_Name'][:20]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,cs in acc.items(): print("S=$S",k, {c: "%.4g"%(sum(v)/len(v)) for c,v in cs.items()})
PY
done
rm -rf $R/gpurun_out/pmc_sha*
