#!/bin/bash
# Every differential fuzzer once (GPU box): one "== name" header and the tool's last line each.
cd "$(dirname "$0")/.."
for f in fuzz_sparse fuzz_sparse_kernels fuzz_multi_zin fuzz_engines fuzz_gradients fuzz_map_path fuzz_analysis fuzz_autotune fuzz_coefficients fuzz_autograd fuzz_chunks; do
  echo "== $f"
  timeout -k 10 400 python3 tools/$f.py 2>&1 | grep -v amdgpu.ids | tail -n 1
done
