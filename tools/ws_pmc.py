"""Largest value of every collected counter per kernel from `rocprofv3 --pmc ... --output-format csv`
directories (tools/whole_state_bench.py under SQ_* counters): python tools/ws_pmc.py <dir> [<dir> ...]."""
import csv, glob, collections, sys
for d in sys.argv[1:]:
    f = glob.glob(d + "/*counter_collection.csv")
    if not f: print("no csv in", d); continue
    rows = list(csv.DictReader(open(f[0])))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(k, {n: (round(max(v), 0), len(v)) for n, v in c.items()})
