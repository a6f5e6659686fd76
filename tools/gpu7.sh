set -x
R=$PWD
python - <<'PY' 2>&1 | tail -8
import numpy as np, sys
sys.path.insert(0,'.')
from circom_amd import runtime as rt
from circom_amd.field import PRIMES
q=PRIMES['bn128']
rng=np.random.default_rng(5)
for n in (1<<16, 1<<17, 1<<18, 1<<19, 1<<22):
    a=rng.integers(0,256,size=(n,32),dtype=np.uint8); a[:,31]&=0x1f
    b=rng.integers(0,256,size=(n,32),dtype=np.uint8); b[:,31]&=0x1f
    out,ms=rt.fp_mul_bench(q,a,b,1024)
    print("fp mulbench n=%d (%.1f waves/SIMD) iters=1024: %.3f ms -> %.2f G mul/s, %.0f ns per wave-MMUL"%(n, n/65536, ms,n*1024/ms/1e6, ms*1e6/1024/max(1,n/65536)))
PY
cd /tmp && export TMPDIR=/tmp
for S in 1 4; do
CW_STRANDS=$S rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_v4_S$S -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_v4_S$S/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'cw_eval' in r['Kernel_Name'] or 'cw_r1cs' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:20]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,cs in acc.items(): print("S=$S",k, {c: "%.4g"%(sum(v)/len(v)) for c,v in cs.items()})
PY
done
rm -rf $R/gpurun_out/pmc_v4_S*
