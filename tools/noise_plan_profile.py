#!/usr/bin/env python3
"""The engine plan of a noisy Model call (vec(rho) on the doubled register): stages, groups and HIP-event time
per stage.  NP_N / NP_L / NP_B as in tools/noise_profile.py."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N, simulation
from qml_essentials_amd.model import Model
from qml_essentials_amd.tape import recording
from qml_essentials_amd.utils import PRNGKey
n, layers, B = int(os.environ.get("NP_N", "10")), int(os.environ.get("NP_L", "2")), int(os.environ.get("NP_B", "64"))
NOISE = {"BitFlip": 0.01, "PhaseFlip": 0.02, "Depolarizing": 0.03, "AmplitudeDamping": 0.05, "PhaseDamping": 0.06}
rng = np.random.default_rng(1000)
m = Model(n, layers, "Hardware_Efficient")
m.noise_params = dict(NOISE)
P = rng.uniform(0, 6.28, (B, *m.params.shape[1:])).astype(np.float32)
x = np.array([0.5], dtype=np.float32)
with recording() as tape:
    m._variational(P.T if False else P[0], x, random_key=PRNGKey(0), noise_params=m.noise_params)
seg = simulation.doubled_tape(tape, n)
low = simulation.LoweredTape(seg, 2 * n)
plan = simulation.get_plan(low)
d = plan.describe()
print("ops on the doubled register:", len(seg), " lowered:", d["n_lowered"], " stages:", len(d["stages"]), " candidate", d.get("candidate"))
ang = torch.from_numpy(np.repeat(low.angle_table(1), B, 0)).cuda()
for _ in range(2): out = plan.run(ang, "state")
torch.cuda.synchronize()
reps = 3
plan.profile_begin(len(d["stages"]) * reps * 8 + 16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): out = plan.run(ang, "state")
e1.record(); torch.cuda.synchronize()
ms, cnt, _ = plan.profile_end()
print(f"{e0.elapsed_time(e1) / reps:.3f} ms per {B}-state run")
for s, t in zip(d["stages"], ms):
    kinds = {}
    for g in s.get("groups", []): kinds[g["kind"]] = kinds.get(g["kind"], 0) + 1
    print(f"  {s['kind']:6s} T={s.get('T')} L={s.get('L')} ops={s['n_lowered']:3d} groups by kind={kinds} fast={s.get('fast')} bits={s.get('bits')}  {t / reps:.3f} ms")
