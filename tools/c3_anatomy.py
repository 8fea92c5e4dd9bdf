#!/usr/bin/env python3
"""Where the wall-clock of BASELINE config 3 (Expressibility, 12 qubits, 1024 pairs) goes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from qml_essentials_amd.expressibility import Expressibility
from qml_essentials_amd.model import Model
import cProfile, pstats

m = Model(12, 3, "Hardware_Efficient", data_reupload=False)
for _ in range(3):
    Expressibility.kl_divergence_to_haar(m, n_samples=1024, n_bins=75, random_key=1000)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    Expressibility.kl_divergence_to_haar(m, n_samples=1024, n_bins=75, random_key=1000)
torch.cuda.synchronize()
print("per call ms", (time.perf_counter() - t0) / 10 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    Expressibility.kl_divergence_to_haar(m, n_samples=1024, n_bins=75, random_key=1000)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
