R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export QMLE_MW_FUSE_TILED=1 MW_REPS=40 QMLE_MW_PAIRING=1
for m in 1 0; do
export QMLE_MW_NT=$m
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mwnt_$m -o p -- python3 $R/tools/mw_fused_target.py > $R/gpurun_out/mwnt_$m.log 2>&1
echo "QMLE_MW_NT=$m"; python3 - $(find $R/gpurun_out/mwnt_$m -name "*kernel_stats.csv" | head -n 1) <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].replace('(anonymous namespace)::','')[:50]
    if 'k_mw_read' in n or 'k_tile2<' in n and 'true, false, false, true>' in n: print(f"   {n:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
done
cd $R; unset QMLE_MW_NT
python tools/mw_lean_ab.py 2>/dev/null | tail -1
