R=$GRAFT_REPO_ROOT
cd $R
for fl in 0 265216 265472; do echo "deep n=24 flags=$fl"; DEEP_DEFAULT=$fl python tools/deep_anatomy.py 2>/dev/null | tail -2; done
for fl in 0 265216; do echo "c2 n=20 flags=$fl"; DEEP_N=20 DEEP_B=256 DEEP_DEFAULT=$fl python tools/deep_anatomy.py 2>/dev/null | tail -2; done
