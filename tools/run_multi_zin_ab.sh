#!/bin/bash
# A/B of multi-tile walks on stages with known zeros inside the tile (QMLE_NO_MULTI_ZIN=1 = round-2 behaviour)
mkdir -p gpurun_out/mz
for v in 0 1; do
  if [ "$v" = 1 ]; then export QMLE_NO_MULTI_ZIN=1; else unset QMLE_NO_MULTI_ZIN; fi
  echo "== QMLE_NO_MULTI_ZIN=${QMLE_NO_MULTI_ZIN:-unset}"
  DEEP_DEFAULT=0 python tools/deep_anatomy.py 2>/dev/null | grep "^DBG"
  for nn in 24 22; do SWEEP_N=$nn python tools/layers_sweep.py 2>/dev/null | grep layers | cut -c1-70; done
done > gpurun_out/mz/ab.txt 2>&1
cat gpurun_out/mz/ab.txt
