R=$GRAFT_REPO_ROOT
cd $R
export WS_SHAPES=10:6:65536,12:3:32768
for pad in 0 1024 2560 5120 8192; do echo "pad $pad"; QMLE_T2_LDS_PAD=$pad python tools/whole_state_bench.py 2>/dev/null | grep "^n=" | cut -c1-80; done
