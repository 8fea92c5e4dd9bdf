import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter("ignore")
from qml_essentials_amd import _native as N, simulation
from qml_essentials_amd.model import Model
n, L, B = 20, 4, 256
m = Model(n, L, "Hardware_Efficient")
rng = np.random.default_rng(0)
P = rng.uniform(0, 6.28, (B, *m.params.shape[1:])).astype(np.float32)
tape, _ = m.record_tape(params=P[:2], inputs=np.array([0.5], dtype=np.float32))
low = simulation.LoweredTape(tape, n)
for label, flags in (("auto", 0), ("dense", N.PLAN_NO_SPARSE), ("tape order", N.PLAN_TAPE_ORDER)):
    top = N.Plan(low.ops, n, low.n_slots, low.consts, flags)
    plan = top.executed("expval")
    d = plan.describe()
    ang = torch.from_numpy(np.ascontiguousarray(low.angle_table(2)[:1].repeat(B, 0))).cuda()
    ang = torch.from_numpy(rng.uniform(0, 6.28, (B, low.n_slots)).astype(np.float32)).cuda()
    ws = torch.empty(top.workspace_bytes(B, "expval", n), dtype=torch.uint8, device="cuda")
    obs = list(range(n))
    top.run(ang, "expval", obs, workspace=ws); torch.cuda.synchronize()
    reps = 5
    plan.profile_begin(len(d["stages"]) * reps * 4 + 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): top.run(ang, "expval", obs, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    ms, cnt, _ = plan.profile_end()
    print(label, "total %.2f ms per %d states" % (e0.elapsed_time(e1) / reps, B),
          [(s["T"], s["L"], s["n_lowered"], s["lds_round_trips"], bin(s["zero_in"]).count("1"), round(t_ / max(c, 1) * 1e3)) for s, t_, c in zip(d["stages"], ms, cnt)], flush=True)
