#!/bin/bash
# SQ-level anatomy of k_tile (run on the GPU box): two --pmc passes, per-dispatch CSVs.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ktile
mkdir -p $OUT
export PMC_N=${PMC_N:-24} PMC_B=${PMC_B:-32}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  --kernel-trace -d $OUT/p1 -o p1 --output-format csv -- python3 $R/tools/pmc_target.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES \
  --kernel-trace -d $OUT/p2 -o p2 --output-format csv -- python3 $R/tools/pmc_target.py > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE SQ_CYCLES \
  --kernel-trace -d $OUT/p3 -o p3 --output-format csv -- python3 $R/tools/pmc_target.py > $OUT/p3.log 2>&1 || true
find $OUT -name "*counter_collection.csv" | head
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/pmc_ktile"
rows=collections.defaultdict(dict)
for f in sorted(glob.glob(out+"/**/*counter_collection.csv", recursive=True)):
    tag=f.split("/pmc_ktile/")[1].split("/")[0]
    per=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if ("k_tile" in r["Kernel_Name"] or "k_reg_measure" in r["Kernel_Name"]) and "k_fold" not in r["Kernel_Name"]:
            per[(int(r["Dispatch_Id"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    ids=sorted({k[0] for k in per})
    # last run's three passes
    for i,d in enumerate(ids[-3:]):
        for (dd,c),v in per.items():
            if dd==d: rows[i][c]=sum(v)
for i in sorted(rows):
    print("pass",i+1,{k:f"{v:.4g}" for k,v in sorted(rows[i].items())})
PY
