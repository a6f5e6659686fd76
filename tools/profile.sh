# Collect the rocprofv3 evidence for profiles/: kernel trace + stats, then PMC passes (each in its own run).
# usage (on the GPU box): bash tools/profile.sh <tag> <workload:batch> [bench args...]
set -x
TAG=$1; KEY=$2; shift; shift
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-small --fp-bench-lanes 65536 "$@" > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-small --fp-bench-lanes 65536 "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-small --fp-bench-lanes 65536 "$@" > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-small --fp-bench-lanes 65536 "$@" > $OUT/pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -30
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
python $R/tools/summarize_prof.py $OUT "$KEY" $R/gpurun_out/traffic.json > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
# keep the merged-back payload small
find $OUT -name "*.csv" -size +4M -delete
