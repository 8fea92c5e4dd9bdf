#!/bin/bash
# Per-instance L2 / memory-side counters of the K1 CX launches (control = target + 1) on a few target
# wires at n = 28: are the 128 L2 channel instances evenly loaded, and where does the time go for
# the wires whose control sits on byte-address bit 10 / 11 (target wires 19 / 18)?  DESIGN section 5.
out=$GRAFT_REPO_ROOT/gpurun_out/k1pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
wires="10 14 18 19 25"
i=0
for set in "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_BUSY" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL" "TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_LEVEL TCC_TAG_STALL"; do
  i=$((i+1))
  rm -rf $out/p$i
  rocprofv3 --pmc $set --output-format json -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/k1_cx_target.py $wires > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*results.json" | head -1)
  echo "## counters: $set   (target wires $wires, 4 launches each, in this order)"
  python3 $GRAFT_REPO_ROOT/tools/parse_pmc_channels.py $f k_direct_1q
done
