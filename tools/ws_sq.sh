#!/bin/bash
# SQ counters of the whole-state (LDS-resident) circuit kernel, one shape per run:
#   bash tools/ws_sq.sh <tag> [n:layers:B]      (default 10:6:65536)
# three rocprofv3 --pmc passes over tools/whole_state_bench.py, summary per wave printed and saved
# to gpurun_out/ws_sq_<tag>.txt
set -e
TAG=${1:-ws}; SHAPE=${2:-10:6:65536}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ws_sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WS_SHAPES=$SHAPE WS_REPS=3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  --kernel-trace -d $OUT/sq1 -o sq1 --output-format csv -- python3 $R/tools/whole_state_bench.py > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES \
  --kernel-trace -d $OUT/sq2 -o sq2 --output-format csv -- python3 $R/tools/whole_state_bench.py > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY \
  --kernel-trace -d $OUT/sq3 -o sq3 --output-format csv -- python3 $R/tools/whole_state_bench.py > $OUT/sq3.log 2>&1 || true
TAG=$TAG SHAPE=$SHAPE python3 - <<'PY' > $R/gpurun_out/ws_sq_$TAG.txt
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/ws_sq_" + os.environ["TAG"]
tot = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/sq*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        key = "circuit" if ("k_tile2" in name or "k_ws" in name) else "k_build_matrices" if "k_build_matrices" in name else None
        if key:
            per[(key, int(r["Dispatch_Id"]))][r["Counter_Name"]] += float(r["Counter_Value"])
    for key in ("circuit", "k_build_matrices"):
        ids = sorted(d for k, d in per if k == key)
        if ids:
            tot[key].update(per[(key, ids[-1])])   # the last launch
n, layers, B = (int(v) for v in os.environ["SHAPE"].split(":"))
print(f"# whole-state regime, HE n={n} layers={layers} B={B}: SQ counters of the LAST launch (sums over the chip)")
for key, r in tot.items():
    print(key, {k: f"{v:.4g}" for k, v in sorted(r.items())})
    w = r.get("SQ_WAVES")
    if w:
        per_state = w / B
        print("   per wave: VALU %.0f SALU %.0f SMEM %.0f LDS %.0f VMEM_RD %.1f VMEM_WR %.1f | waves per state %.2f" % (
            r.get("SQ_INSTS_VALU", 0) / w, r.get("SQ_INSTS_SALU", 0) / w, r.get("SQ_INSTS_SMEM", 0) / w, r.get("SQ_INSTS_LDS", 0) / w,
            r.get("SQ_INSTS_VMEM_RD", 0) / w, r.get("SQ_INSTS_VMEM_WR", 0) / w, per_state))
    wc = r.get("SQ_WAVE_CYCLES")
    if wc:
        print("   of wave cycles: VALU issue %.3f, LDS issue %.3f, scalar %.3f, waiting (s_waitcnt / barrier) %.3f, issue stalls %.3f (LDS part %.3f); waves per busy SIMD cycle %.2f" % (
            r.get("SQ_ACTIVE_INST_VALU", 0) / wc, r.get("SQ_ACTIVE_INST_LDS", 0) / wc, r.get("SQ_ACTIVE_INST_SCA", 0) / wc, r.get("SQ_WAIT_ANY", 0) / wc,
            r.get("SQ_WAIT_INST_ANY", 0) / wc, r.get("SQ_WAIT_INST_LDS", 0) / wc, wc / max(r.get("SQ_BUSY_CYCLES", 1), 1)))
    if r.get("SQ_LDS_IDX_ACTIVE"):
        print("   LDS array cycles %.4g, of them bank conflicts %.4g (%.1f %%)" % (
            r["SQ_LDS_IDX_ACTIVE"], r.get("SQ_LDS_BANK_CONFLICT", 0), 100 * r.get("SQ_LDS_BANK_CONFLICT", 0) / r["SQ_LDS_IDX_ACTIVE"]))
PY
grep -h "M states" $OUT/sq1.log | tail -2 >> $R/gpurun_out/ws_sq_$TAG.txt || true
cat $R/gpurun_out/ws_sq_$TAG.txt
