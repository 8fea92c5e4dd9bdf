#!/bin/bash
# Round-5 evidence, run on the GPU box (bash tools/collect_profiles_r05.sh [part ...]); parts: stats pmc mw sq legs
# k1 misc rows ab bench (default: all).  Everything lands in gpurun_out/profiles_r05/, the files to keep are named r05_*.
set -e
TAG=r05; export TAG
PARTS=${*:-stats pmc mw sq legs k1 misc rows ab bench}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
stats_of() {  # stats_of <name> <program ...>: rocprofv3 kernel stats of one target -> r05_<name>_kernel_stats.csv
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$name -o $name -- "$@" > $OUT/st_$name.log 2>&1
  cp $(find $OUT/st_$name -name "*kernel_stats.csv" | head -n 1) $OUT/${TAG}_${name}_kernel_stats.csv
}
if has stats; then
  echo "[stats] kernel stats of the headline command"
  stats_of bench python3 $R/bench.py --steps 2 --warmup 1 --skip-aux
fi
if has pmc; then
  echo "[pmc] HBM counters, all-live K2 plan at the engine's 32 states per launch (4 GiB of states)"
  export PMC_N=24 PMC_B=32 PMC_FLAGS=160   # QMLE_PLAN_NO_SPARSE | QMLE_PLAN_NO_ABSORB
  rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/rd -o rd -- \
    python3 $R/tools/pmc_target.py > $OUT/rd.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/wr -o wr -- \
    python3 $R/tools/pmc_target.py > $OUT/wr.log 2>&1
  python3 $R/tools/parse_pmc.py $OUT/${TAG}_pmc_k2_dense_n24.json $(find $OUT/rd $OUT/wr -name "*counter_collection.csv") > $OUT/parse_k2.log
  (cd $R && python3 tools/update_traffic.py $OUT/${TAG}_pmc_k2_dense_n24.json 24 32 dense \
    "profiles/${TAG}_pmc_k2_dense_n24.json: rocprofv3 --pmc (separate read / write passes) on tools/pmc_target.py with PMC_FLAGS=160 (NO_SPARSE|NO_ABSORB), 32 states per launch, average over the k_tile2 launches of the plan (read+write pass and measuring pass)" > $OUT/traffic1.log)
fi
if has mw; then
  echo "[mw] Meyer-Wallach n = 28: kernel stats + HBM counters, resident and fused routes"
  export MW_REPS=100
  stats_of mw_n28 python3 $R/tools/mw_bench.py 28
  export MW_REPS=8
  rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/mwrd -o rd -- \
    python3 $R/tools/mw_bench.py 28 > $OUT/mwrd.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/mwwr -o wr -- \
    python3 $R/tools/mw_bench.py 28 > $OUT/mwwr.log 2>&1
  python3 $R/tools/parse_pmc.py $OUT/${TAG}_pmc_mw_n28.json $(find $OUT/mwrd $OUT/mwwr -name "*counter_collection.csv") > $OUT/parse_mw.log
  (cd $R && python3 tools/update_traffic.py $OUT/${TAG}_pmc_mw_n28.json 28 0 mw \
    "profiles/${TAG}_pmc_mw_n28.json: every k_mw_* launch of one qmle_meyer_wallach call (rocprofv3 --pmc, separate read / write passes, on tools/mw_bench.py 28 with MW_REPS=8), averaged over the calls of the run)" > $OUT/traffic2.log)
  rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/mwfrd -o rd -- \
    python3 $R/tools/mw_fused_target.py > $OUT/mwfrd.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/mwfwr -o wr -- \
    python3 $R/tools/mw_fused_target.py > $OUT/mwfwr.log 2>&1
  python3 $R/tools/parse_pmc.py $OUT/${TAG}_pmc_mw_fused_n28.json $(find $OUT/mwfrd $OUT/mwfwr -name "*counter_collection.csv") > $OUT/parse_mwf.log
  (cd $R && python3 tools/update_traffic.py $OUT/${TAG}_pmc_mw_fused_n28.json 28 8 mwfused \
    "profiles/${TAG}_pmc_mw_fused_n28.json: the k_mw_read_later launches of one fused call (the default route, rocprofv3 --pmc, separate read / write passes, tools/mw_fused_target.py with MW_REPS=8): bytes fetched after the circuit" > $OUT/traffic3.log)
  export MW_REPS=30
  stats_of mw_fused python3 $R/tools/mw_fused_profile.py
  export MW_REPS=100
  python3 $R/tools/mw_bench.py 28 24 2>/dev/null | grep "^n=" > $OUT/${TAG}_mw_n28.txt || true
  cp $R/profiles/traffic.json $OUT/traffic.json
fi
if has sq; then
  echo "[sq] SQ counters: Meyer-Wallach reads (resident + fused last pass), the deep default engine, C2, whole-state"
  export MW_REPS=4
  bash $R/tools/sq_counters.sh mw_resident "k_mw_read" -- python3 $R/tools/mw_bench.py 28
  cp $R/gpurun_out/sq_mw_resident.txt $OUT/${TAG}_mw_sq_resident.txt
  bash $R/tools/sq_counters.sh mw_fused "k_mw_read|k_tile2" -- python3 $R/tools/mw_fused_target.py
  cp $R/gpurun_out/sq_mw_fused.txt $OUT/${TAG}_mw_sq_fused.txt
  export DT_N=24 DT_LAYERS=4 DT_B=32 DT_FLAGS=0 DT_REPS=2
  bash $R/tools/sq_counters.sh deep_default "k_tile" -- python3 $R/tools/deep_target.py
  cp $R/gpurun_out/sq_deep_default.txt $OUT/${TAG}_deep_default_sq.txt
  export DT_FLAGS=160
  bash $R/tools/sq_counters.sh deep_all_live "k_tile" -- python3 $R/tools/deep_target.py
  cp $R/gpurun_out/sq_deep_all_live.txt $OUT/${TAG}_deep_all_live_sq.txt
  export DT_N=20 DT_B=256 DT_FLAGS=0
  bash $R/tools/sq_counters.sh c2 "k_tile" -- python3 $R/tools/deep_target.py
  cp $R/gpurun_out/sq_c2.txt $OUT/${TAG}_c2_sq.txt
  bash $R/tools/ws_sq.sh r05_n10 10:6:65535 > /dev/null; cp $R/gpurun_out/ws_sq_r05_n10.txt $OUT/${TAG}_ws_sq_n10.txt
  bash $R/tools/ws_sq.sh r05_n12 12:3:32768 > /dev/null; cp $R/gpurun_out/ws_sq_r05_n12.txt $OUT/${TAG}_ws_sq_n12.txt
  cd /tmp
fi
if has legs; then
  echo "[legs] kernel stats of the bench legs added in round 5 (one target each)"
  export DT_REPS=3
  export DT_N=20 DT_LAYERS=4 DT_B=1024 DT_FLAGS=0 DT_DRU=1; stats_of c2_b1024 python3 $R/tools/deep_target.py
  export DT_B=1; export DT_REPS=20; stats_of c2_single python3 $R/tools/deep_target.py
  export DT_REPS=3
  export DT_B=1024 DT_FLAGS=160; stats_of c2_b1024_all_live python3 $R/tools/deep_target.py
  export DT_N=24 DT_LAYERS=4 DT_B=64 DT_FLAGS=0; stats_of deep_default python3 $R/tools/deep_target.py
  export DT_FLAGS=160; stats_of deep_all_live python3 $R/tools/deep_target.py
  export DT_LAYERS=1 DT_DRU=0 DT_CIRCUIT=Circuit_19; stats_of k2_circuit19 python3 $R/tools/deep_target.py
  unset DT_CIRCUIT DT_DRU DT_LAYERS DT_N DT_B DT_FLAGS DT_REPS
fi
if has k1; then
  echo "[k1] K1 sweep kernel stats (RX RZ CX CRX CRZ CZ CPhase)"
  stats_of k1 python3 $R/tools/k1_sweep.py
  grep -v -E "amdgpu.ids|rocprofv3|^[WEI][0-9]" $OUT/st_k1.log > $OUT/${TAG}_k1_single_gate_n28.txt || true
fi
if has misc; then
  echo "[misc] analysis loops, whole-state regime, configs"
  cd $R
  python3 tools/loops_anatomy.py c3 c4 c4_api mw 2>/dev/null | grep -E "^(c3|c4|c4_api|mw):" > $OUT/${TAG}_loops_anatomy.txt || true
  python3 tools/whole_state_bench.py 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_ws_bench.txt || true
  python3 tools/configs_bench.py 2>/dev/null | grep "^|" > $OUT/${TAG}_configs.md || true
  cd /tmp
fi
if has rows; then
  echo "[rows] widened rows (SURVEY 8-f): wall-clock table, the noisy plan under the round-5 switches"
  cd $R
  python3 tools/next_rows_bench.py > $OUT/next_rows_now.md 2> $OUT/next_rows.err || true
  {
    echo "== one LDS sweep per superoperator, gates unmerged (QMLE_NO_REG2Q=1 QMLE_NO_MERGE_2Q=1)"
    QMLE_NO_REG2Q=1 QMLE_NO_MERGE_2Q=1 python3 tools/noise_plan_profile.py 2>&1 | grep -v amdgpu.ids
    echo "== 4x4 operators in register-tile groups (QMLE_NO_MERGE_2Q=1)"
    QMLE_NO_MERGE_2Q=1 python3 tools/noise_plan_profile.py 2>&1 | grep -v amdgpu.ids
    echo "== default: gates and channels merged into the 4x4 operators"
    python3 tools/noise_plan_profile.py 2>&1 | grep -v amdgpu.ids
    echo "== host profile of one noisy Model(10, 2) call, 64 parameter sets (compiled call)"
    NP_N=10 NP_L=2 NP_B=64 NP_TOP=12 python3 tools/noise_profile.py 2>&1 | grep -v amdgpu.ids
  } > $OUT/${TAG}_noise_plan.txt
  cd /tmp
fi
if has ab; then
  echo "[ab] schedule / engine switches against the defaults, per-pass times"
  cd $R
  bash tools/top_first_ab.sh > $OUT/${TAG}_top_first_ab.txt 2>&1 || true
  bash tools/chunk_overlap_ab.sh > $OUT/${TAG}_chunk_overlap_ab.txt 2>&1 || true
  bash tools/pad_high_ab.sh > $OUT/${TAG}_pad_high_ab.txt 2>&1 || true
  cd /tmp
fi
if has bench; then
  echo "[bench] bench line (after the traffic records: its roofline.traffic is the PMC figure of THIS tree)"
  python3 $R/bench.py > $OUT/bench.log 2> $OUT/bench.err
  tail -n 1 $OUT/bench.log > $OUT/${TAG}_bench_n1.json
  python3 -c "import json,sys; d=json.load(open('$OUT/${TAG}_bench_n1.json')); print(json.dumps(d['summary'])[:3000])"
fi
echo "[done] $PARTS"
