#!/bin/bash
# Expressibility(12 q, 1024 pairs) wall-clock with the library's Philox sampler (thread counts) vs numpy's loop
mkdir -p gpurun_out/c3
for t in 1 2 4; do QMLE_RNG_THREADS=$t python tools/c3_anatomy.py 2>&1 | grep -v amdgpu.ids | head -45 > gpurun_out/c3/lib_sampler_t$t.txt; done
QMLE_NUMPY_SAMPLER=1 python tools/c3_anatomy.py 2>&1 | grep -v amdgpu.ids | head -45 > gpurun_out/c3/numpy_sampler.txt
grep "per call" gpurun_out/c3/lib_sampler_t*.txt gpurun_out/c3/numpy_sampler.txt
grep -E "philox_uniform|utils.py:100" gpurun_out/c3/lib_sampler_t*.txt gpurun_out/c3/numpy_sampler.txt
