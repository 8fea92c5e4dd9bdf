#!/usr/bin/env python3
"""complex128 gradients of Model(16, 4, Hardware_Efficient) (240 parameters), mean of the 16 <Z> outputs: the one-sweep
adjoint (qmle_adjoint_gradient_f64, round 5) against the parameter-shift route (2 x 240 shifted circuits on the
complex128 engine, what x64 mode contracted in rounds 3-4), wall-clock per gradient and max |difference|."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd.model import Model

n, layers = int(os.environ.get("GB_N", "16")), int(os.environ.get("GB_LAYERS", "4"))
m = Model(n, layers, "Hardware_Efficient", x64=True)
x = np.array([0.5])


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


ta, ga = timed(lambda: np.asarray(m.gradient(inputs=x, method="adjoint", force_mean=True)), 5)
ts, gs = timed(lambda: np.asarray(m.gradient(inputs=x, method="parameter-shift", force_mean=True)), 2)
m32 = Model(n, layers, "Hardware_Efficient")
t32, g32 = timed(lambda: np.asarray(m32.gradient(inputs=x, method="adjoint", force_mean=True)), 5)
print(f"n={n} layers={layers} params={ga.size}: complex128 adjoint sweep {ta * 1e3:.2f} ms, complex128 parameter shift "
      f"{ts * 1e3:.2f} ms ({ts / ta:.1f} x), complex64 adjoint sweep {t32 * 1e3:.2f} ms; "
      f"max |adjoint - shift| = {np.abs(ga - gs).max():.2e}, max |complex128 - complex64| = {np.abs(ga - g32).max():.2e}")
