#!/usr/bin/env python3
"""profiles/traffic.json entries (HBM bytes per launch, read by bench.py for roofline.traffic) from
the per-kernel PMC averages that tools/parse_pmc.py wrote.  Usage: update_traffic.py pmc.json n states"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import source_sha16  # the records are signed with the kernel sources they were collected for
SHA = source_sha16()
pm = json.load(open(sys.argv[1])); n = int(sys.argv[2]); states = int(sys.argv[3])
tag = sys.argv[4] if len(sys.argv) > 4 else "fused"   # "dense": the all-amplitudes-live plan
source = sys.argv[5] if len(sys.argv) > 5 else None
# the all-live initialising pass = k_fill_zero (the zeros) + a k_tile2 launch for tile 0 of every state:
# its fill traffic belongs to k_tile2's per-launch average (one k_tile2 launch per pass either way)
# (round 3: the initialising pass of the from-|0..0> variant is k_fill_zero + the GENERIC kernel on a
# 2^14-amplitude tile -- then the fill belongs to k_tile and k_tile2's average covers passes 2 and 3)
if tag == "dense" and "k_fill_zero" in pm and "k_tile" in pm and \
        pm["k_tile"].get("launches") == pm["k_fill_zero"].get("launches"):
    for key in ("hbm_read_bytes_per_launch_x2_corrected", "hbm_write_bytes_per_launch"):
        pm["k_tile"][key] = pm["k_tile"].get(key, 0.0) + pm["k_fill_zero"].get(key, 0.0)
elif tag == "dense" and "k_fill_zero" in pm and "k_tile2" in pm:
    f, k = pm["k_fill_zero"], pm["k_tile2"]
    share = f.get("launches", 0) / max(1, k.get("launches", 1))
    for key in ("hbm_read_bytes_per_launch_x2_corrected", "hbm_write_bytes_per_launch"):
        k[key] = k.get(key, 0.0) + f.get(key, 0.0) * share
path = os.path.join(root, "profiles", "traffic.json")
t = json.load(open(path))
for fam, d in pm.items():
    if "hbm_read_bytes_per_launch_x2_corrected" not in d or "hbm_write_bytes_per_launch" not in d:
        continue
    if not fam.startswith(("k_tile", "k_product", "k_reg_measure", "k_direct", "k_mw")):
        continue
    rd, wr = round(d["hbm_read_bytes_per_launch_x2_corrected"]), round(d["hbm_write_bytes_per_launch"])
    t[f"{fam}:n{n}:{tag}"] = {"read": rd, "write": wr, "hbm_bytes_per_launch": rd + wr,
                              "states_per_launch": states, "launches_averaged": d.get("launches"),
                              "source_sha16": SHA}
    if source:
        t[f"{fam}:n{n}:{tag}"]["source"] = source
# Meyer-Wallach: bytes per qmle_meyer_wallach CALL = every k_mw_* launch of one call (tag "mw": the
# counters come from tools/mw_bench.py, `states` = calls profiled)
if tag == "mw":
    states = pm.get("k_mw_purity", {}).get("launches", states)   # one purity launch per call
    rd = sum(d.get("hbm_read_bytes_per_launch_x2_corrected", 0.0) * d.get("launches", 0)
             for f, d in pm.items() if f.startswith("k_mw")) / max(1, states)
    wr = sum(d.get("hbm_write_bytes_per_launch", 0.0) * d.get("launches", 0)
             for f, d in pm.items() if f.startswith("k_mw")) / max(1, states)
    t[f"meyer_wallach:n{n}"] = {"read": round(rd), "write": round(wr), "hbm_bytes_per_launch": round(rd + wr),
                                "states_per_launch": 1, "calls_averaged": states, "source": source,
                                "source_sha16": SHA}
# fused route (tag "mwfused", `states` = calls profiled by tools/mw_fused_target.py): what is fetched AFTER the circuit
# = the k_mw_read_later launches of one call (the circuit's own last pass reports the first read's sums)
if tag == "mwfused":
    d = pm.get("k_mw_read", {})
    calls = max(1, states)
    rd = d.get("hbm_read_bytes_per_launch_x2_corrected", 0.0) * d.get("launches", 0) / calls
    wr = d.get("hbm_write_bytes_per_launch", 0.0) * d.get("launches", 0) / calls
    t[f"meyer_wallach_fused:n{n}"] = {"read": round(rd), "write": round(wr), "hbm_bytes_per_launch": round(rd + wr),
                                      "states_per_launch": 1, "calls_averaged": calls,
                                      "later_read_launches_per_call": d.get("launches", 0) / calls,
                                      "source": source, "source_sha16": SHA}
json.dump(t, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in t.items() if k != "_how"}, indent=1))
