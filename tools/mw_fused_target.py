#!/usr/bin/env python3
"""Counter target: MW_REPS calls of the circuit + Meyer-Wallach through QMLE_MEAS_MEYER_WALLACH at n = 28 (the
fused tiled route -- the default since round 5; QMLE_MW_FUSE_TILED=0 selects the stand-alone reads) -- nothing else on the GPU."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from qml_essentials_amd import _native as N
n = int(os.environ.get("MW_N", "28"))
ops, slots = bench._he_layer_ops(n)
ang = torch.from_numpy(np.random.default_rng(6).uniform(0, 6.28, (1, slots)).astype(np.float32)).cuda()
plan = N.Plan(ops, n, slots)
ws_m = torch.empty(plan.workspace_bytes(1, "mw"), dtype=torch.uint8, device="cuda")
for _ in range(int(os.environ.get("MW_REPS", "8"))):
    q = plan.run(ang, "mw", workspace=ws_m)
torch.cuda.synchronize()
print("Q", float(q[0, 0]))
