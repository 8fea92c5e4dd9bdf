set -x
R=$GRAFT_REPO_ROOT
export WS_SHAPES=10:6:65536,12:3:32768,10:6:4096,12:3:2048
python tools/whole_state_bench.py 2>/dev/null | grep "^n=" > gpurun_out/r05_ws_skip_after.txt
QMLE_BUILD_ALL_MATRICES=1 python tools/whole_state_bench.py 2>/dev/null | grep "^n=" > gpurun_out/r05_ws_skip_before.txt
cat gpurun_out/r05_ws_skip_before.txt gpurun_out/r05_ws_skip_after.txt
export PMC_N=24 PMC_B=32 PMC_FLAGS=160
bash tools/sq_counters.sh k2_headline "k_tile" -- python3 $R/tools/pmc_target.py > /dev/null 2>&1
grep -v "raw:" gpurun_out/sq_k2_headline.txt
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests_2.log 2>&1; tail -3 gpurun_out/r05_gputests_2.log
