#!/usr/bin/env python3
"""Randomised check of the multi-tile walks over stages with known zeros inside the tile (round 3):
random ansatz / layer count / encoding at n = 21..23, 32 states -- the state must be identical bit for
bit with QMLE_NO_MULTI_ZIN=1 (one tile per workgroup) and equal to the all-live plan's at float32 level."""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from qml_essentials_amd import simulation
from qml_essentials_amd.ansaetze import Ansaetze
from qml_essentials_amd.model import Model

warnings.simplefilter("ignore")
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "3")))
names = [a.__name__ for a in Ansaetze.get_available()]
bad = walks = 0
for trial in range(int(os.environ.get("FUZZ_N", "16"))):
    n = int(rng.integers(21, 24))
    B = 32 if n < 23 else 16
    L = int(rng.integers(1, 4))
    name = str(rng.choice(names))
    model = Model(n, L, name, data_reupload=bool(rng.integers(2)))
    params = rng.uniform(0, 2 * np.pi, (B, *model.params.shape[1:])).astype(np.float32)
    x = np.array([[float(rng.uniform(-1, 1))]], dtype=np.float32)
    tape, _ = model.record_tape(params=params, inputs=x)
    low = simulation.LoweredTape(tape, n)
    ang = torch.from_numpy(low.angle_table(B)).cuda()
    plan = N.Plan(low.ops, n, low.n_slots, low.consts, 0)
    stages = plan.describe()["stages"]
    zin = [s for s in stages[1:] if s["kind"] == "tile" and s["zero_in"]
           and all((s["zero_in"] >> b) & 1 == 0 for b in range(n) if b not in s["bits"])]
    os.environ.pop("QMLE_NO_MULTI_ZIN", None)
    a = plan.run(ang, "state")
    za = plan.run(ang, "expval", list(range(n)))
    os.environ["QMLE_NO_MULTI_ZIN"] = "1"
    b = plan.run(ang, "state")
    zb = plan.run(ang, "expval", list(range(n)))
    os.environ.pop("QMLE_NO_MULTI_ZIN", None)
    dense = N.Plan(low.ops, n, low.n_slots, low.consts, N.PLAN_NO_SPARSE).run(ang[:4], "state")
    ok = torch.equal(a, b) and float((za - zb).abs().max()) < 1e-6 and float((a[:4] - dense).abs().max()) < 3e-6
    walks += bool(zin)
    bad += not ok
    print(trial, name, n, L, "stages with in-tile zeros only:", len(zin), "ok" if ok else "MISMATCH",
          float((a[:4] - dense).abs().max()), flush=True)
    del a, b, dense
    torch.cuda.empty_cache()
print(f"mismatches: {bad} (trials with a candidate stage: {walks})")
