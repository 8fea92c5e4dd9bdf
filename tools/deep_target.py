#!/usr/bin/env python3
"""Counter target: a few calls of a model's expval batch through the product API (no timing).
    DT_N (24) DT_LAYERS (4) DT_B (64) DT_FLAGS (0 = the default engine; 160 = all amplitudes live) DT_CIRCUIT DT_DRU (1)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import simulation
from qml_essentials_amd.model import Model

n, layers, B = int(os.environ.get("DT_N", "24")), int(os.environ.get("DT_LAYERS", "4")), int(os.environ.get("DT_B", "64"))
simulation.PLAN_FLAGS = int(os.environ.get("DT_FLAGS", "0"))
dru = os.environ.get("DT_DRU", "1") != "0"
model = Model(n, layers, os.environ.get("DT_CIRCUIT", "Hardware_Efficient"), data_reupload=dru)
params = np.random.default_rng(1000).uniform(0, 2 * np.pi, (B, *model.params.shape[1:])).astype(np.float32)
pd = torch.from_numpy(params).cuda()
xd = torch.full((1, 1), 0.5, device="cuda") if dru else None
for _ in range(int(os.environ.get("DT_REPS", "3"))):
    out = model(params=pd, inputs=xd) if dru else model(params=pd)
torch.cuda.synchronize()
print("ok", tuple(out.shape))
