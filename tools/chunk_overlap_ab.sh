#!/bin/bash
# A/B of round 5's chunk pipeline (QMLE_NO_CHUNK_OVERLAP=1: the one-stream loop): µs per state of batches that need
# several chunks, tools/deep_anatomy.py circuits under the default engine and all-live.
cd "$(dirname "$0")/.."
run() {
  local name=$1; shift
  for off in 1 0; do
    if [ $off = 1 ]; then echo "== $name: one stream (QMLE_NO_CHUNK_OVERLAP=1)"; env "$@" QMLE_NO_CHUNK_OVERLAP=1 python3 tools/deep_anatomy.py 2>&1 | grep -v amdgpu.ids | grep -v "stages:"
    else echo "== $name: two streams, one stage apart"; env "$@" python3 tools/deep_anatomy.py 2>&1 | grep -v amdgpu.ids | grep -v "stages:"; fi
  done
}
run "n24 1 layer all-live b256"        DEEP_LAYERS=1 DEEP_B=256
run "n24 4 layers all-live b256"       DEEP_LAYERS=4 DEEP_B=256
run "n24 4 layers default b1024"       DEEP_LAYERS=4 DEEP_DEFAULT=0 DEEP_B=1024
run "n24 1 layer default b1024"        DEEP_LAYERS=1 DEEP_DEFAULT=0 DEEP_B=1024
run "n20 4 layers all-live b2048"      DEEP_N=20 DEEP_LAYERS=4 DEEP_B=2048
run "n20 4 layers default b8192"       DEEP_N=20 DEEP_LAYERS=4 DEEP_DEFAULT=0 DEEP_B=8192
run "n26 1 layer all-live b64"         DEEP_N=26 DEEP_LAYERS=1 DEEP_B=64
run "n28 1 layer all-live b16"         DEEP_N=28 DEEP_LAYERS=1 DEEP_B=16
run "n24 circuit19 all-live b256"      DEEP_LAYERS=1 DEEP_CIRCUIT=Circuit_19 DEEP_B=256
run "n22 3 layers default b2048"       DEEP_N=22 DEEP_LAYERS=3 DEEP_DEFAULT=0 DEEP_B=2048
