#!/bin/bash
# A/B of the last stage's padding: all-live HE circuits, per-pass HIP-event times
set -e
mkdir -p gpurun_out/pad
for cfg in "20 1" "22 1" "26 1" "28 1" "24 2" "24 4" "22 3"; do
  set -- $cfg
  for v in 0 1; do
    if [ "$v" = 0 ]; then K2_N=$1 K2_LAYERS=$2 python tools/k2_passes.py; else K2_N=$1 K2_LAYERS=$2 QMLE_PAD_HIGH=$v python tools/k2_passes.py; fi
  done
done > gpurun_out/pad/k2_pad_sizes.txt 2>&1
grep "us/state" -A12 gpurun_out/pad/k2_pad_sizes.txt | grep -v amdgpu.ids
