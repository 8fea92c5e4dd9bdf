#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 --pmc passes: K2 plan on a few states."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from tests.test_abi_cpu import he_layer_ops

n = int(os.environ.get("PMC_N", "24")); B = int(os.environ.get("PMC_B", "8"))
flags = int(os.environ.get("PMC_FLAGS", "0"))
ops, slots = he_layer_ops(n)
ang = torch.from_numpy(np.random.default_rng(1000).uniform(0, 6.28, (B, slots)).astype(np.float32)).cuda()
plan = N.Plan(ops, n, slots, flags=flags)
ws = torch.empty(plan.workspace_bytes(B, "expval", n), dtype=torch.uint8, device="cuda")
for _ in range(2):
    plan.run(ang, "expval", list(range(n)), workspace=ws)
torch.cuda.synchronize()
