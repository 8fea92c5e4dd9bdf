#!/usr/bin/env python3
"""Randomised Model-level check of known-zero tracking: every ansatz / encoding / layer count at
n = 15..19 gives the same expval / probs / state with and without QMLE_PLAN_NO_SPARSE."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qml_essentials_amd import _native as N
from qml_essentials_amd import simulation
from qml_essentials_amd.ansaetze import Ansaetze, Encoding
from qml_essentials_amd.model import Model

warnings.simplefilter("ignore")
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "7")))
names = [a.__name__ for a in Ansaetze.get_available()]
bad = 0
for trial in range(int(os.environ.get("FUZZ_N", "60"))):
    n = int(rng.integers(15, 20))
    kw = dict(circuit_type=str(rng.choice(names)), data_reupload=bool(rng.integers(2)))
    if rng.random() < 0.3:
        kw["encoding"] = Encoding("hamming", ["RX"])
    if rng.random() < 0.3:
        kw["output_qubit"] = [0, int(n - 1)] if rng.random() < 0.5 else [[0, 1], [int(n - 2), int(n - 1)]]
    L = int(rng.integers(1, 3))
    et = str(rng.choice(["expval", "expval", "probs", "state"]))
    res = {}
    for mode, fl in (("sparse", 0), ("dense", N.PLAN_NO_SPARSE)):
        simulation.PLAN_FLAGS = fl
        simulation.clear_plan_cache()
        try:
            m = Model(n, L, **kw)
            r2 = np.random.default_rng(trial)
            P = r2.uniform(0, 2 * np.pi, (2, *m.params.shape[1:])).astype(np.float32)
            X = r2.uniform(0, 1, (2, m.n_input_feat)).astype(np.float32)
            res[mode] = np.asarray(m(params=P, inputs=X, execution_type=et))
        except Exception as e:  # same error on both sides is fine
            res[mode] = repr(e)
    a, b = res["sparse"], res["dense"]
    if isinstance(a, str) or isinstance(b, str):
        ok = a == b
        err = a if not ok else "both raise"
    else:
        err = float(np.abs(a - b).max())
        ok = err < (3e-6 if et == "probs" else 2e-6)  # partial probs: LDS bins + one float atomic per bin and workgroup
    if not ok and not isinstance(a, str) and et == "probs":
        # partial probabilities are summed with float atomics (k_marginal): measure the run-to-run
        # noise of ONE mode before calling the difference between the two a mismatch
        again = np.asarray(m(params=P, inputs=X, execution_type=et))
        noise = float(np.abs(again - b).max())
        print("   run-to-run difference of the dense mode alone:", noise, flush=True)
        ok = err < 3e-6 + 4 * noise
    if not ok:
        bad += 1
    print(trial, n, L, et, kw["circuit_type"], kw.get("encoding") is not None, "output" in str(kw.keys()), err, "" if ok else "<<< MISMATCH", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
