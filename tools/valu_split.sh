#!/bin/bash
# VALU / SALU / LDS instructions per wave of the all-live K2 measuring pass with the gate groups
# and / or the epilogue switched off (QMLE_DBG_T2 = 0..3): where the pass's instructions are.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/valu_split
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PMC_N=24 PMC_B=32 PMC_FLAGS=160
for d in 0 1 2 3; do
  export QMLE_DBG_T2=$d
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM_RD \
    --kernel-trace -d $OUT/d$d -o d$d --output-format csv -- python3 $R/tools/pmc_target.py > $OUT/d$d.log 2>&1
done
python3 - <<'PY' | tee $OUT/r02_valu_split.txt
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/valu_split"
print("# k_tile2 passes of the all-live K2 plan, n = 24, 32 states: instructions per wave (dbg: 1 = no groups, 2 = no epilogue)")
for d in range(4):
    f = glob.glob(out + f"/d{d}/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_tile2" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(per)[-3:]
    for i, k in enumerate(ids):
        c = per[k]; w = c["SQ_WAVES"] or 1
        print(f"dbg={d} pass {i+1}: waves {w:.0f} VALU {c['SQ_INSTS_VALU']/w:.0f} SALU {c['SQ_INSTS_SALU']/w:.0f} "
              f"LDS {c['SQ_INSTS_LDS']/w:.0f} SMEM {c['SQ_INSTS_SMEM']/w:.0f} VMEM_RD {c['SQ_INSTS_VMEM_RD']/w:.1f}")
PY
