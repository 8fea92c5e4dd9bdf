cd $GRAFT_REPO_ROOT
for e in 0 1; do echo "QMLE_TILE_NO_NT=$e"; if [ $e == 1 ]; then export QMLE_TILE_NO_NT=1; fi
DEEP_DEFAULT=160 python tools/deep_anatomy.py 2>/dev/null | tail -1 | cut -c1-300
DEEP_N=20 DEEP_B=1024 DEEP_DEFAULT=160 python tools/deep_anatomy.py 2>/dev/null | tail -1 | cut -c1-300
DEEP_N=20 DEEP_B=16 DEEP_DEFAULT=160 python tools/deep_anatomy.py 2>/dev/null | tail -1 | cut -c1-300
done
