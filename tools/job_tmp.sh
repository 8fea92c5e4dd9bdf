R=$GRAFT_REPO_ROOT
cd $R
for m in 0 1 2; do echo "pairing $m"; QMLE_MW_PAIRING=$m python tools/mw_lean_ab.py 2>/dev/null | tail -1; done
echo "last-pass NT off:"; QMLE_LAST_PASS_NT=0 python tools/mw_lean_ab.py 2>/dev/null | tail -1
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests_7.log 2>&1; tail -3 gpurun_out/r05_gputests_7.log
