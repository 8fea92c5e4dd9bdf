#!/usr/bin/env python3
"""Per-pass HIP-event times of the all-live 4-layer data-re-uploading circuit at n = 24 (k2_deep of
bench.py) -- run under QMLE_DBG_T2 = 0 / 4 / 8 / 1 to see what the gate loop's scalar side costs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from qml_essentials_amd import simulation
from qml_essentials_amd.model import Model

n, B = int(os.environ.get("DEEP_N", "24")), int(os.environ.get("DEEP_B", "64"))
flags = N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB if os.environ.get("DEEP_DEFAULT") is None else int(os.environ["DEEP_DEFAULT"]) & ~1
simulation.PLAN_FLAGS = flags
model = Model(n, int(os.environ.get("DEEP_LAYERS", "4")), os.environ.get("DEEP_CIRCUIT", "Hardware_Efficient"), data_reupload=True)
rng = np.random.default_rng(1000)
params = rng.uniform(0, 2 * np.pi, (B, *model.params.shape[1:])).astype(np.float32)
x = np.full((1, 1), 0.5, dtype=np.float32)
tape, _ = model.record_tape(params=params[:2], inputs=x)
low = simulation.LoweredTape(tape, n)
top = simulation.get_plan(low)
plan = top.executed("expval")
d = plan.describe()
pd, xd = torch.from_numpy(params).cuda(), torch.from_numpy(x).cuda()
for _ in range(2):
    out = model(params=pd, inputs=xd)
torch.cuda.synchronize()
reps = 3
plan.profile_begin(len(d["stages"]) * reps * 8 + 16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out = model(params=pd, inputs=xd)
e1.record(); torch.cuda.synchronize()
ms, cnt, _ = plan.profile_end()
per = [m / reps / B * 1e3 for m in ms]
print("   stages:", [(s["T"], s["L"], s["bits"][s["L"] if s["L"] < s["T"] else 0:], s.get("expval_kernel"), bin(s["zero_in"]).count("1")) for s in d["stages"]], "absorbed", top.describe().get("absorbed_ops"))
print(f"DBG={os.environ.get('QMLE_DBG_T2', '0')}: {e0.elapsed_time(e1) / reps / B * 1e3:.1f} us/state; per pass (groups: us): ",
      [(len(s.get('fast_groups') or s.get('groups') or []), round(t, 1)) for s, t in zip(d["stages"], per)], flush=True)
