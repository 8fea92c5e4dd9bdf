#!/bin/bash
# Round evidence, run on the GPU box: bench line, rocprofv3 kernel stats of the same command,
# HBM request counters (separate --pmc passes) at the bench's 256 states per launch.
# Usage: bash tools/collect_profiles.sh r01
set -e
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.log 2> $OUT/bench.err
tail -n 1 $OUT/bench.log > $OUT/${TAG}_bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- \
  python3 $R/bench.py --steps 2 --warmup 1 --skip-aux --cpu-seconds 0.5 > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -n 1) $OUT/${TAG}_bench_kernel_stats.csv
export PMC_N=24 PMC_B=256   # bench.py's states per launch (32 GiB of state buffers)
rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/rd -o rd -- \
  python3 $R/tools/pmc_target.py > $OUT/rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/wr -o wr -- \
  python3 $R/tools/pmc_target.py > $OUT/wr.log 2>&1
python3 $R/tools/parse_pmc.py $OUT/${TAG}_pmc_k2_n24.json $(find $OUT/rd $OUT/wr -name "*counter_collection.csv")
cat $OUT/${TAG}_bench_n1.json
head -n 12 $OUT/${TAG}_bench_kernel_stats.csv
