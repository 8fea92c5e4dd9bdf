#!/bin/bash
# A/B of QMLE_PAD_HIGH (the LAST tile stage padded from the top of the register instead of from position 0)
# over the bench's circuits: per-pass HIP-event times of tools/deep_anatomy.py under both settings.
cd "$(dirname "$0")/.."
run() {  # name, env...
  local name=$1; shift
  for ph in 0 1; do
    echo "== $name QMLE_PAD_HIGH=$ph"
    env "$@" QMLE_PAD_HIGH=$ph python3 tools/deep_anatomy.py 2>&1 | grep -v amdgpu.ids
  done
}
run "n24 1 layer all-live"       DEEP_LAYERS=1
run "n24 4 layers all-live"      DEEP_LAYERS=4
run "n24 4 layers default"       DEEP_LAYERS=4 DEEP_DEFAULT=0
run "n20 4 layers default b1024" DEEP_N=20 DEEP_LAYERS=4 DEEP_DEFAULT=0 DEEP_B=1024
run "n20 4 layers all-live b1024" DEEP_N=20 DEEP_LAYERS=4 DEEP_B=1024
run "n26 1 layer all-live"       DEEP_N=26 DEEP_LAYERS=1 DEEP_B=16
run "n24 circuit19 all-live"     DEEP_LAYERS=1 DEEP_CIRCUIT=Circuit_19
run "n22 3 layers default"       DEEP_N=22 DEEP_LAYERS=3 DEEP_DEFAULT=0 DEEP_B=256
