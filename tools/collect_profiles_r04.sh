#!/bin/bash
# Round-4 evidence, run on the GPU box (bash tools/collect_profiles_r04.sh): bench line, rocprofv3 kernel
# stats of the same command, HBM request counters (separate --pmc passes) for the all-live K2 plan and for
# Meyer-Wallach at n = 28 (-> profiles/traffic.json, signed with the kernel-source hash), K1 kernel stats,
# the analysis loops' launch timeline, the whole-state regime and the fused Meyer-Wallach route.
set -e
TAG=${1:-r04}; export TAG
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
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
python3 $R/tools/parse_pmc.py $OUT/${TAG}_pmc_k2_dense_n24.json $(find $OUT/rd $OUT/wr -name "*counter_collection.csv") > $OUT/parse_k2.log
echo "[4] Meyer-Wallach n = 28 (resident state): kernel stats + HBM counters"
export MW_REPS=100
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mwstats -o mw -- python3 $R/tools/mw_bench.py 28 > $OUT/mw.log 2>&1
cp $(find $OUT/mwstats -name "*kernel_stats.csv" | head -n 1) $OUT/${TAG}_mw_n28_kernel_stats.csv
export MW_REPS=8
rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/mwrd -o rd -- \
  python3 $R/tools/mw_bench.py 28 > $OUT/mwrd.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/mwwr -o wr -- \
  python3 $R/tools/mw_bench.py 28 > $OUT/mwwr.log 2>&1
python3 $R/tools/parse_pmc.py $OUT/${TAG}_pmc_mw_n28.json $(find $OUT/mwrd $OUT/mwwr -name "*counter_collection.csv") > $OUT/parse_mw.log
export MW_REPS=100
python3 $R/tools/mw_bench.py 28 24 2>/dev/null | grep "^n=" > $OUT/${TAG}_mw_n28.txt || true
echo "[4b] Meyer-Wallach out of the producing pass (n = 28 tiled, opt-in; 12-qubit loop, default): kernel stats"
QMLE_MW_FUSE_TILED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mwf -o mwf -- python3 $R/tools/mw_fused_profile.py > $OUT/mwf.log 2>&1
cp $(find $OUT/mwf -name "*kernel_stats.csv" | head -n 1) $OUT/${TAG}_mw_fused_kernel_stats.csv
echo "[5] K1 sweep kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k1stats -o k1 -- python3 $R/tools/k1_sweep.py > $OUT/k1.log 2>&1
cp $(find $OUT/k1stats -name "*kernel_stats.csv" | head -n 1) $OUT/${TAG}_k1_kernel_stats.csv
grep -v -E "amdgpu.ids|rocprofv3|^[WEI][0-9]" $OUT/k1.log > $OUT/${TAG}_k1_single_gate_n28.txt || true
echo "[5b] K1 controlled gates: block order (control positions 7 / 8) and the 4-row burst form, per target wire"
cd $R
python3 tools/k1_block_order.py 16 17 18 19 20 21 2>/dev/null > $OUT/${TAG}_k1_block_order.txt || true
(for d in 1 -1 5 -7; do for b in 0 9; do echo "control = target wire + ($d), QMLE_K1_CTRL_BURST=$b"; K1_CTRL_DELTA=$d QMLE_K1_CTRL_BURST=$b K1_MULS=0 python3 tools/k1_block_order.py 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 2>/dev/null; done; done) > $OUT/${TAG}_k1_ctrl_burst_raw.txt || true
cd /tmp
echo "[6] traffic.json (signed with the source hash)"
cd $R
python3 tools/update_traffic.py $OUT/${TAG}_pmc_k2_dense_n24.json 24 32 dense \
  "profiles/${TAG}_pmc_k2_dense_n24.json: rocprofv3 --pmc (separate read / write passes) on tools/pmc_target.py with PMC_FLAGS=160 (NO_SPARSE|NO_ABSORB), 32 states per launch, average over the k_tile2 launches of the plan (read+write pass and measuring pass)" > $OUT/traffic1.log
python3 tools/update_traffic.py $OUT/${TAG}_pmc_mw_n28.json 28 0 mw \
  "profiles/${TAG}_pmc_mw_n28.json: every k_mw_* launch of one qmle_meyer_wallach call (rocprofv3 --pmc, separate read / write passes, on tools/mw_bench.py 28 with MW_REPS=8), averaged over the calls of the run)" > $OUT/traffic2.log
cp profiles/traffic.json $OUT/traffic.json
echo "[7] analysis loops: wall / GPU time per call, then the launch timeline of the same loops"
python3 tools/loops_anatomy.py c3 c4 c4_api mw 2>/dev/null | grep -E "^(c3|c4|c4_api|mw):" > $OUT/${TAG}_loops_anatomy.txt || true
cd /tmp
rocprofv3 --kernel-trace -d $OUT/loops -o loops -- python3 $R/tools/loops_anatomy.py c3 c4 mw > $OUT/loops.log 2>&1
cd $R
python3 tools/rocpd_timeline.py $(find $OUT/loops -name "*_results.db" | head -n 1) 24 > $OUT/${TAG}_loops_timeline.txt || true
echo "[8] whole-state regime, deep circuits, configs"
python3 tools/whole_state_bench.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_ws_bench_after.txt || true
python3 tools/accum_probe.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_accum_probe.txt || true
python3 tools/configs_bench.py 2>/dev/null | grep "^|" > $OUT/${TAG}_configs.md || true
echo "[9] bench line (after the traffic records: its roofline.traffic is the PMC figure of THIS tree)"
cd /tmp
python3 $R/bench.py > $OUT/bench.log 2> $OUT/bench.err
tail -n 1 $OUT/bench.log > $OUT/${TAG}_bench_n1.json
head -c 600 $OUT/${TAG}_bench_n1.json; echo
head -n 8 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-160
