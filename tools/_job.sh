cd $GRAFT_REPO_ROOT
for fl in 0 199680 134144 199936; do echo "flags=$fl"
DEEP_DEFAULT=$fl python tools/deep_anatomy.py 2>/dev/null | tail -1 | cut -c1-300
DEEP_N=20 DEEP_B=256 DEEP_DEFAULT=$fl python tools/deep_anatomy.py 2>/dev/null | tail -1 | cut -c1-300
done
