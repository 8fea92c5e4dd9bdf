#!/usr/bin/env python3
"""Differential fuzz of the three gradient routes through Model.gradient: parameter shift on the
complex64 engine, the fused adjoint sweep, and parameter shift on the complex128 engine (x64 mode), for
random ansatz / size / layers / encoding / wrt -- all three must agree at float32 level."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd.ansaetze import Ansaetze
from qml_essentials_amd.model import Model
from qml_essentials_amd.utils import x64_scope

warnings.simplefilter("ignore")
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
names = [a.__name__ for a in Ansaetze.get_available()]
bad = 0
for trial in range(int(os.environ.get("FUZZ_N", "60"))):
    n = int(rng.integers(1, 8)) if rng.random() < 0.8 else int(rng.integers(14, 17))
    kw = dict(n_qubits=n, n_layers=int(rng.integers(1, 3)), circuit_type=str(rng.choice(names)),
              data_reupload=bool(rng.integers(2)))
    if rng.random() < 0.3:
        kw["encoding"] = [str(rng.choice(["RX", "RY", "RZ"]))]
    wrt = str(rng.choice(["params", "params", "inputs"]))
    nb = int(rng.choice([1, 3]))
    x = rng.uniform(-1, 1, nb)
    tag = f"{trial} n={n} {kw['circuit_type']} L={kw['n_layers']} dru={kw['data_reupload']} enc={kw.get('encoding')} wrt={wrt} nb={nb}"
    try:
        m = Model(**kw)
        p = rng.uniform(0, 2 * np.pi, m.params.shape)
        g_ps = np.asarray(m.gradient(params=p, inputs=x, wrt=wrt, force_mean=True, method="parameter-shift"))
        g_ad = np.asarray(m.gradient(params=p, inputs=x, wrt=wrt, force_mean=True, method="adjoint"))
        with x64_scope(True):
            g_64 = np.asarray(m.gradient(params=p, inputs=x, wrt=wrt, force_mean=True, method="parameter-shift"))
    except (NotImplementedError, ValueError) as e:
        print(tag, "->", type(e).__name__, str(e)[:90], flush=True)
        continue
    except Exception as e:
        bad += 1
        print(tag, "-> ERROR", type(e).__name__, str(e)[:160], flush=True)
        continue
    ok = g_ps.shape == g_ad.shape == g_64.shape
    e1 = float(np.abs(g_ps - g_64).max()) if ok and g_ps.size else 0.0
    e2 = float(np.abs(g_ad - g_64).max()) if ok and g_ps.size else 0.0
    ok = ok and e1 < 2e-5 and e2 < 2e-5
    bad += not ok
    if not ok or trial % 15 == 0:
        print(tag, g_ps.shape, g_ad.shape, g_64.shape, "shift-vs-x64", e1, "adjoint-vs-x64", e2, "ok" if ok else "MISMATCH", flush=True)
print(f"mismatches / errors: {bad}")
