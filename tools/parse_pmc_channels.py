#!/usr/bin/env python3
"""Per-INSTANCE view of rocprofv3 --pmc counters (JSON output): for every dispatch of the kernels
matching a pattern, per counter: number of instances (XCD x L2 channel), sum, min, max and how many
instances saw less than a quarter of the mean -- the evidence for "this access pattern uses half
the channels".  Usage: parse_pmc_channels.py results.json [kernel substring]"""
import json, sys, collections
doc = json.load(open(sys.argv[1]))["rocprofiler-sdk-tool"][0]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_direct_1q"
names = {}
for c in doc.get("counters", []):
    names[c["id"]["handle"] if isinstance(c.get("id"), dict) else c.get("id")] = c.get("name")
ksym = {k["kernel_id"]: k.get("formatted_kernel_name") or k.get("kernel_name") for k in doc.get("kernel_symbols", [])}
recs = doc["callback_records"].get("counter_collection") or doc["buffer_records"].get("counter_collection")
out = []
for r in recs:
    info = r["dispatch_data"]["dispatch_info"]
    kname = ksym.get(info["kernel_id"], "?")
    if pat not in kname:
        continue
    per = collections.defaultdict(list)
    for x in r["records"]:
        per[names.get(x["counter_id"]["handle"], str(x["counter_id"]["handle"]))].append(x["value"])
    dur = (r["dispatch_data"]["end_timestamp"] - r["dispatch_data"]["start_timestamp"]) / 1e3
    row = {"dispatch": info["dispatch_id"], "kernel": kname.split("(")[0][-40:], "us": round(dur, 1)}
    for cname, vals in sorted(per.items()):
        mean = sum(vals) / len(vals)
        row[cname] = {"instances": len(vals), "sum": sum(vals), "min": min(vals), "max": max(vals),
                      "idle_instances": sum(1 for v in vals if v < 0.25 * mean)}
    out.append(row)
for row in out:
    print(json.dumps(row))
