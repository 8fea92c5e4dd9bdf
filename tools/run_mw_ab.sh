# A/B of Meyer-Wallach variants under rocprofv3 (kernel trace): usage bash tools/run_mw_ab.sh "ENV=1 ..." ...
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/mw2
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/mw2/prof_$i
  rm -rf $out
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/mw_bench.py $MW_ARGS 28 > $GRAFT_REPO_ROOT/gpurun_out/mw2/bench_$i.txt 2>&1
  echo "== $v: $(grep '^n=' $GRAFT_REPO_ROOT/gpurun_out/mw2/bench_$i.txt)"
  python3 - "$out" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_mw' in r['Name']:
        print('   %-40s calls %3s avg %8.1f us min %8.1f max %8.1f' % (r['Name'].replace('(anonymous namespace)::','')[:40], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
