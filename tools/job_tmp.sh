R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -2
export WS_SHAPES=10:6:65536,12:3:32768,10:6:4096,12:3:2048
python tools/whole_state_bench.py 2>/dev/null | grep "^n="
python bench.py --steps 10 --warmup 3 --skip-aux 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], [p['avg_launch_ms'] for p in d['roofline']['per_pass']], d['roofline']['frac'])"
DEEP_DEFAULT=0 python tools/deep_anatomy.py 2>/dev/null | tail -1
DEEP_DEFAULT=160 python tools/deep_anatomy.py 2>/dev/null | tail -1
