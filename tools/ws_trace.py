"""Median per-launch time of each kernel and shape from a `rocprofv3 --kernel-trace --output-format csv`
run of tools/whole_state_bench.py (five shapes x 23 launches): python tools/ws_trace.py, run from the
repo root after `rocprofv3 ... -d gpurun_out/ws/prof -o ws -- python3 tools/whole_state_bench.py`."""
import csv,glob,collections
f=glob.glob("gpurun_out/ws/prof/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"][:40]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items():
    # 5 shapes x 23 launches each
    print(k, len(v), [round(sorted(v[i*23+3:(i+1)*23])[10],1) for i in range(5) if len(v)>=(i+1)*23])
