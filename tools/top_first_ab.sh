#!/bin/bash
# A/B of the round-5 top-first schedules (first tile of a run from |0..0> on the TOP 14 positions; all-live plans
# only): per-pass HIP-event times of tools/deep_anatomy.py with and without QMLE_NO_TOP_FIRST=1.
cd "$(dirname "$0")/.."
run() {
  local name=$1; shift
  for off in 1 0; do
    if [ $off = 1 ]; then echo "== $name: round-4 candidates (QMLE_NO_TOP_FIRST=1)"; env "$@" QMLE_NO_TOP_FIRST=1 python3 tools/deep_anatomy.py 2>&1 | grep -v amdgpu.ids
    else echo "== $name: default"; env "$@" python3 tools/deep_anatomy.py 2>&1 | grep -v amdgpu.ids; fi
  done
}
run "n24 1 layer (re-uploading) all-live"  DEEP_LAYERS=1
run "n24 4 layers all-live"      DEEP_LAYERS=4
run "n20 4 layers all-live b1024" DEEP_N=20 DEEP_LAYERS=4 DEEP_B=1024
run "n22 3 layers all-live b256" DEEP_N=22 DEEP_LAYERS=3 DEEP_B=256
run "n26 1 layer all-live b16"   DEEP_N=26 DEEP_LAYERS=1 DEEP_B=16
run "n28 1 layer all-live b4"    DEEP_N=28 DEEP_LAYERS=1 DEEP_B=4
run "n24 circuit19 all-live"     DEEP_LAYERS=1 DEEP_CIRCUIT=Circuit_19
run "n18 2 layers all-live b1024" DEEP_N=18 DEEP_LAYERS=2 DEEP_B=1024
run "n16 2 layers all-live b1024" DEEP_N=16 DEEP_LAYERS=2 DEEP_B=1024
