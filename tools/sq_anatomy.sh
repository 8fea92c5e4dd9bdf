#!/bin/bash
# SQ counters of the k_tile2 passes of the all-live K2 plan (32 states per launch): three
# rocprofv3 --pmc runs, summary printed and saved.  Usage: bash tools/sq_anatomy.sh [tag]
set -e
TAG=${1:-sq}; export TAG
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PMC_N=24 PMC_B=32 PMC_FLAGS=${PMC_FLAGS:-160}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  --kernel-trace -d $OUT/sq1 -o sq1 --output-format csv -- python3 $R/tools/pmc_target.py > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES \
  --kernel-trace -d $OUT/sq2 -o sq2 --output-format csv -- python3 $R/tools/pmc_target.py > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM \
  --kernel-trace -d $OUT/sq3 -o sq3 --output-format csv -- python3 $R/tools/pmc_target.py > $OUT/sq3.log 2>&1 || true
python3 - <<'PY' > $OUT/${TAG}_pmc_ktile2_sq_anatomy.txt
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/sq_" + os.environ["TAG"]
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/sq*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_tile2" in r["Kernel_Name"]:
            per[(int(r["Dispatch_Id"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    ids = sorted({k[0] for k in per})
    for i, d in enumerate(ids[-3:]):      # the last run's three passes
        for (dd, c), v in per.items():
            if dd == d:
                rows[i][c] = sum(v)
print("# k_tile2, K2 plan flags", os.environ.get("PMC_FLAGS"), "n = 24, 32 states per launch: SQ counters per launch")
for i in sorted(rows):
    print("pass", i + 1, {k: f"{v:.4g}" for k, v in sorted(rows[i].items())})
    r = rows[i]
    if r.get("SQ_WAVES"):
        w = r["SQ_WAVES"]
        print("   per wave: VALU %.0f SALU %.0f SMEM %.0f LDS %.0f VMEM_RD %.1f VMEM_WR %.1f; wave cycles/4 %.0f; VALU active share %.2f" % (
            r.get("SQ_INSTS_VALU", 0) / w, r.get("SQ_INSTS_SALU", 0) / w, r.get("SQ_INSTS_SMEM", 0) / w, r.get("SQ_INSTS_LDS", 0) / w,
            r.get("SQ_INSTS_VMEM_RD", 0) / w, r.get("SQ_INSTS_VMEM_WR", 0) / w, r.get("SQ_WAVE_CYCLES", 0) / w,
            r.get("SQ_ACTIVE_INST_VALU", 0) / max(r.get("SQ_WAVE_CYCLES", 1), 1)))
PY
cat $OUT/${TAG}_pmc_ktile2_sq_anatomy.txt
