cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputests_9.log 2>&1; tail -2 gpurun_out/r05_gputests_9.log
for f in fuzz_sparse fuzz_sparse_kernels fuzz_multi_zin fuzz_engines fuzz_gradients fuzz_map_path fuzz_analysis fuzz_autotune; do
  echo "== $f"; FUZZ_SEED=909 timeout -k 10 170 python tools/$f.py 2>/dev/null | tail -1 | cut -c1-200
done
