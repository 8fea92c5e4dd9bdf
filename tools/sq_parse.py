#!/usr/bin/env python3
"""Per-kernel summary of the three counter passes of tools/sq_counters.sh.
    sq_parse.py <family-regex> <out-dir> [<command line, for the header>]
Kernels whose name matches the regex are grouped by their (shortened) name; counters are averaged over the
launches of a kernel.  Derived figures (MI355X: 256 CUs, 1024 SIMDs, 8 XCDs; GRBM_GUI_ACTIVE is summed over the
8 XCDs, so the launch lasts GRBM_GUI_ACTIVE / 8 cycles):
  VALU busy   = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / launch cycles   (every wave64 VALU instruction issues for 4)
  LDS busy    = SQ_LDS_IDX_ACTIVE / 256 CUs / launch cycles             (one LDS pipe per CU)
  scalar busy = (SQ_INSTS_SALU + SQ_INSTS_SMEM) / 256 CUs / launch cycles (one scalar issue per CU and cycle)
  of a wave's cycles: issuing VALU / LDS / scalar, waiting (s_waitcnt, barrier), stalled at issue."""
import collections
import csv
import glob
import re
import sys

fam_re, out = re.compile(sys.argv[1]), sys.argv[2]
print(f"# SQ counters per kernel ({sys.argv[1]}); target: {sys.argv[3] if len(sys.argv) > 3 else ''}")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name.replace("void ", ""))
    return name[:90]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/sq*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if fam_re.search(r["Kernel_Name"]):
            per[(short(r["Kernel_Name"]), int(r["Dispatch_Id"]))][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _d), cs in per.items():
        for c, v in cs.items():
            acc[k][c].append(v)
for k, cs in acc.items():
    r = {c: sum(v) / len(v) for c, v in cs.items()}
    n = max(len(v) for v in cs.values())
    print(f"{k}  [{n} launches averaged]")
    print("   raw:", {c: f"{v:.4g}" for c, v in sorted(r.items())})
    cyc = r.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc:
        print("   launch %.4g cycles; VALU busy %.3f, LDS busy %.3f (bank conflicts %.1f %% of LDS cycles), scalar busy %.3f" % (
            cyc, r.get("SQ_INSTS_VALU", 0) * 4 / 1024 / cyc, r.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc,
            100 * r.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, r.get("SQ_LDS_IDX_ACTIVE", 0)),
            (r.get("SQ_INSTS_SALU", 0) + r.get("SQ_INSTS_SMEM", 0)) / 256 / cyc))
    w = r.get("SQ_WAVES")
    if w:
        print("   per wave: VALU %.0f SALU %.0f SMEM %.0f LDS %.0f VMEM_RD %.1f VMEM_WR %.1f (%.0f waves)" % (
            r.get("SQ_INSTS_VALU", 0) / w, r.get("SQ_INSTS_SALU", 0) / w, r.get("SQ_INSTS_SMEM", 0) / w,
            r.get("SQ_INSTS_LDS", 0) / w, r.get("SQ_INSTS_VMEM_RD", 0) / w, r.get("SQ_INSTS_VMEM_WR", 0) / w, w))
    wc = r.get("SQ_WAVE_CYCLES")
    if wc:
        print("   of wave cycles: VALU issue %.3f, LDS issue %.3f, scalar %.3f, VMEM issue %.3f, waiting %.3f, issue stalls %.3f (LDS part %.3f)" % (
            r.get("SQ_ACTIVE_INST_VALU", 0) / wc, r.get("SQ_ACTIVE_INST_LDS", 0) / wc, r.get("SQ_ACTIVE_INST_SCA", 0) / wc,
            r.get("SQ_ACTIVE_INST_VMEM", 0) / wc, r.get("SQ_WAIT_ANY", 0) / wc, r.get("SQ_WAIT_INST_ANY", 0) / wc,
            r.get("SQ_WAIT_INST_LDS", 0) / wc))
