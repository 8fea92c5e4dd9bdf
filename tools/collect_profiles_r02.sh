#!/bin/bash
# Round-2 evidence, run on the GPU box: bench line, rocprofv3 kernel stats of the same command,
# HBM request counters (separate --pmc passes) and SQ counters for the all-amplitudes-live K2
# plan (k_tile2), kernel stats + HBM counters for Meyer-Wallach at n = 28.
# Usage: bash tools/collect_profiles_r02.sh [tag]
set -e
TAG=${1:-r02}; export TAG
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[1] bench line"; python3 $R/bench.py > $OUT/bench.log 2> $OUT/bench.err
tail -n 1 $OUT/bench.log > $OUT/${TAG}_bench_n1.json
echo "[2] kernel stats of the headline command"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- \
  python3 $R/bench.py --steps 2 --warmup 1 --skip-aux > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -n 1) $OUT/${TAG}_bench_kernel_stats.csv
echo "[3] HBM counters, all-live K2 plan at the engine's 32 states per launch (4 GiB of states)"
export PMC_N=24 PMC_B=32 PMC_FLAGS=160   # QMLE_PLAN_NO_SPARSE | QMLE_PLAN_NO_ABSORB
rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/rd -o rd -- \
  python3 $R/tools/pmc_target.py > $OUT/rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/wr -o wr -- \
  python3 $R/tools/pmc_target.py > $OUT/wr.log 2>&1
python3 $R/tools/parse_pmc.py $OUT/${TAG}_pmc_k2_dense_n24.json $(find $OUT/rd $OUT/wr -name "*counter_collection.csv")
echo "[4] SQ counters of the three k_tile2 passes (32 states per launch)"
export PMC_B=32
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  --kernel-trace -d $OUT/sq1 -o sq1 --output-format csv -- python3 $R/tools/pmc_target.py > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES \
  --kernel-trace -d $OUT/sq2 -o sq2 --output-format csv -- python3 $R/tools/pmc_target.py > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_CYCLES \
  --kernel-trace -d $OUT/sq3 -o sq3 --output-format csv -- python3 $R/tools/pmc_target.py > $OUT/sq3.log 2>&1 || true
python3 - <<'PY' > $OUT/${TAG}_pmc_ktile2_sq_anatomy.txt
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/profiles_" + (os.environ.get("TAG") or "r02")
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/sq*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_tile2" in r["Kernel_Name"]:
            per[(int(r["Dispatch_Id"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    ids = sorted({k[0] for k in per})
    for i, d in enumerate(ids[-3:]):      # the last run's three passes
        for (dd, c), v in per.items():
            if dd == d:
                rows[i][c] = sum(v)
print("# k_tile2, all-live K2 plan (NO_SPARSE | NO_ABSORB), n = 24, 32 states per launch: SQ counters per launch")
for i in sorted(rows):
    print("pass", i + 1, {k: f"{v:.4g}" for k, v in sorted(rows[i].items())})
PY
echo "[5] Meyer-Wallach n = 28: kernel stats + HBM counters"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mwstats -o mw -- python3 $R/tools/mw_bench.py 28 > $OUT/mw.log 2>&1
cp $(find $OUT/mwstats -name "*kernel_stats.csv" | head -n 1) $OUT/${TAG}_mw_n28_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/mwrd -o rd -- \
  python3 $R/tools/mw_bench.py 28 > $OUT/mwrd.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/mwwr -o wr -- \
  python3 $R/tools/mw_bench.py 28 > $OUT/mwwr.log 2>&1
python3 $R/tools/parse_pmc.py $OUT/${TAG}_pmc_mw_n28.json $(find $OUT/mwrd $OUT/mwwr -name "*counter_collection.csv")
grep "n=28" $OUT/mw.log > $OUT/${TAG}_mw_n28.txt || true
echo "[6] K1 sweep kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k1stats -o k1 -- python3 $R/tools/k1_sweep.py > $OUT/k1.log 2>&1
cp $(find $OUT/k1stats -name "*kernel_stats.csv" | head -n 1) $OUT/${TAG}_k1_kernel_stats.csv
grep -v amdgpu.ids $OUT/k1.log > $OUT/${TAG}_k1_single_gate_n28.txt || true
cat $OUT/${TAG}_bench_n1.json | head -c 1500; echo
head -n 8 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-160
