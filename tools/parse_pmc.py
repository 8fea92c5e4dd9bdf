#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel-family average of
each counter per launch.  HBM traffic per launch follows MI355X_MICROARCH.md (HBM section):
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB... rocprofv3 reports them in units
of 1 KiB? -> we derive bytes from the raw request counters instead:
  read bytes  = TCC_EA0_RDREQ_sum * 64 B, DOUBLED for wide coalesced streams (gfx950
                tallies 128-B requests at 64 B -- guide line 298),
  write bytes = TCC_EA0_WRREQ_64B_sum * 64 + (TCC_EA0_WRREQ_sum - TCC_EA0_WRREQ_64B_sum) * 32.
Usage: parse_pmc.py out.json csv [csv ...]"""
import collections
import csv
import json
import sys


def family(name):
    for k in ("k_tile2", "k_mw_read", "k_mw_purity", "k_reg_measure_mono", "k_reg_measure", "k_product_stream", "k_tile_product", "k_fold_columns", "k_mono_coef", "k_tile", "k_direct_1q", "k_expval_partial", "k_expval_final", "k_build_matrices",
              "k_probs", "k_overlap", "k_cross", "k_init_zero", "k_fill_zero"):
        if k in name:
            return k
    return None


def main():
    out_path, files = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    for f in files:
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            if not fam:
                continue
            agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[fam][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
    res = {}
    for fam, cs in agg.items():
        res[fam] = {c: v / max(1, len(launches[fam][c])) for c, v in cs.items()}
        res[fam]["launches"] = max(len(v) for v in launches[fam].values())
        d = res[fam]
        if "TCC_EA0_RDREQ_sum" in d:
            d["hbm_read_bytes_per_launch_x2_corrected"] = d["TCC_EA0_RDREQ_sum"] * 64 * 2
        if "TCC_EA0_WRREQ_sum" in d:
            w64 = d.get("TCC_EA0_WRREQ_64B_sum", d["TCC_EA0_WRREQ_sum"])
            d["hbm_write_bytes_per_launch"] = w64 * 64 + (d["TCC_EA0_WRREQ_sum"] - w64) * 32
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    for fam, d in res.items():
        print(fam, {k: f"{v:.4g}" for k, v in d.items()})


if __name__ == "__main__":
    main()
