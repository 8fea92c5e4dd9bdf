R=$GRAFT_REPO_ROOT
cd $R
for v in 0 1 0 1; do echo "QMLE_FILL_NT=$v"; QMLE_FILL_NT=$v python bench.py --steps 10 --warmup 3 --skip-aux 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], [p['avg_launch_ms'] for p in d['roofline']['per_pass']], d['roofline']['frac'])"; done
for fl in 0 32; do echo "deep n=24 flags=$fl"; DEEP_DEFAULT=$fl python tools/deep_anatomy.py 2>/dev/null | tail -1; done
for fl in 0 32; do echo "c2 n=20 flags=$fl"; DEEP_N=20 DEEP_B=256 DEEP_DEFAULT=$fl python tools/deep_anatomy.py 2>/dev/null | tail -1; done
