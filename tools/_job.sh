cd $GRAFT_REPO_ROOT
for e in 0 1; do echo "QMLE_SMALL_LAST_TILE=$e"; export QMLE_SMALL_LAST_TILE=$e
DEEP_DEFAULT=0 python tools/deep_anatomy.py 2>/dev/null | tail -1
DEEP_N=20 DEEP_B=256 DEEP_DEFAULT=0 python tools/deep_anatomy.py 2>/dev/null | tail -1
DEEP_N=22 DEEP_B=128 DEEP_LAYERS=2 DEEP_DEFAULT=0 python tools/deep_anatomy.py 2>/dev/null | tail -1
DEEP_N=24 DEEP_B=64 DEEP_LAYERS=3 DEEP_DEFAULT=0 python tools/deep_anatomy.py 2>/dev/null | tail -1
python tools/mw_lean_ab.py 2>/dev/null | tail -1 | cut -c1-200
done
