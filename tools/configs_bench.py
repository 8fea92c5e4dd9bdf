#!/usr/bin/env python3
"""Wall-clock of the BASELINE.json configurations C1..C5 through the drop-in API, with the
oracle's CPU time beside it (bounded samples).  Prints a markdown table."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
entry.build()
from oracle import c_port, circuits as OC, einsum_sim as OE, analysis as OA
from qml_essentials_amd import _native as N
from qml_essentials_amd.model import Model
from qml_essentials_amd.expressibility import Expressibility
from qml_essentials_amd.entanglement import Entanglement
from qml_essentials_amd.coefficients import Coefficients

rng = np.random.default_rng(1000)
threads = c_port.lib().svc_max_threads()
rows = []


def gpu_time(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


# C1
m = Model(4, 2, "Hardware_Efficient"); p = rng.uniform(0, 6.28, m.params.shape[1:]).astype(np.float32)
g = gpu_time(lambda: m(params=p, inputs=np.array([0.5], dtype=np.float32)))
spec = OC.ModelSpec(4, 2, "Hardware_Efficient")
t0 = time.perf_counter(); OE.simulate_and_measure(OC.model_tape(spec, p, [0.5]), 4, "expval", [("PauliZ", [q]) for q in range(4)]); c = time.perf_counter() - t0
rows.append(("C1 Model(4,2,HE) expval, 1 sample", g, c, "numpy einsum oracle, 1 thread"))
# C2
m = Model(20, 4, "Hardware_Efficient"); p = rng.uniform(0, 6.28, m.params.shape[1:]).astype(np.float32)
g = gpu_time(lambda: m(params=p, inputs=np.array([0.5], dtype=np.float32)))
spec = OC.ModelSpec(20, 4, "Hardware_Efficient"); tape = OC.model_tape(spec, p, [0.5])
c_port.lib().svc_set_threads(threads)
t0 = time.perf_counter(); psi = c_port.simulate(tape, 20, threads=threads); c_port.expval_z(psi, 20, list(range(20))); c = time.perf_counter() - t0
rows.append(("C2 Model(20,4,HE) expval, 1 sample (480 gates)", g, c, f"C/OpenMP port, {threads} threads"))
P = rng.uniform(0, 6.28, (256, *m.params.shape[1:])).astype(np.float32)
g = gpu_time(lambda: m(params=P, inputs=np.array([0.5], dtype=np.float32)), reps=3)
rows.append(("C2 batch of 256 parameter sets", g, c * 256, "port, extrapolated x256"))
# C3
m = Model(12, 3, "Hardware_Efficient", data_reupload=False)
g = gpu_time(lambda: Expressibility.kl_divergence_to_haar(m, n_samples=1024, n_bins=75, random_key=1000))
spec = OC.ModelSpec(12, 3, "Hardware_Efficient", data_reupload=False)
PP = np.asarray(m.params)
t0 = time.perf_counter()
st = np.array([c_port.simulate(OC.model_tape(spec, PP[i], [0.0]), 12, threads=1) for i in range(64)])
OA.fidelities_pure(st, 32); c = (time.perf_counter() - t0) * (2048 / 64)
rows.append(("C3 Expressibility 12q, 1024 pairs, 75 bins (pure-state form)", g, c, "port 1 thread (fastest at 32 KiB states), 64 of 2048 states x32"))
# C4
m = Model(10, 6, "Hardware_Efficient")
x = (2 * np.pi * np.arange(4096) / 4096).astype(np.float32).reshape(-1, 1)
g = gpu_time(lambda: m(inputs=x, force_mean=True), reps=3)
spec = OC.ModelSpec(10, 6, "Hardware_Efficient"); p = np.asarray(m.params[0])
t0 = time.perf_counter()
for i in range(64):
    psi = c_port.simulate(OC.model_tape(spec, p, x[i], zero_inputs_batch1=False), 10, threads=1); c_port.expval_z(psi, 10, list(range(10)))
c = (time.perf_counter() - t0) * (4096 / 64)
rows.append(("C4 Fourier sweep Model(10,6,HE), 4096-point grid (340 gates each)", g, c, "port 1 thread, 64 of 4096 x64"))
g2 = gpu_time(lambda: Coefficients.get_spectrum(m), reps=3)
rows.append(("C4' Coefficients.get_spectrum (reference grid, 121 points)", g2, float("nan"), "-"))
# C5
n = 28
ops = [(gname, [q], [i * n + q], -1) for i, gname in enumerate(("RY", "RZ", "RY")) for q in range(n)]
from oracle.circuits import bricks
ops += [("CX", [a, b], [], -1) for a, b in bricks(n, mirror=False) + bricks(n, offset=-1, modulo=True, wrap=True, mirror=False)]
plan = N.Plan(ops, n, 3 * n)
ang = torch.from_numpy(rng.uniform(0, 6.28, (1, 3 * n)).astype(np.float32)).cuda()
state = plan.run(ang, "state")
g_sim = gpu_time(lambda: plan.run(ang, "state", out=state), reps=3)
g_mw = gpu_time(lambda: N.meyer_wallach(state), reps=5)
q = float(N.meyer_wallach(state)[0])
rows.append((f"C5 28q HE layer (112 gates) statevector, fused passes", g_sim, float("nan"), "-"))
rows.append((f"C5 Meyer-Wallach on the 2 GiB state (Q = {q:.4f}), 3 reads", g_mw, float("nan"), "-"))
print("| config | MI355X wall | CPU oracle | CPU note |\n|---|---|---|---|")
for name, g, c, note in rows:
    print(f"| {name} | {g * 1e3:.3f} ms | {'' if c != c else f'{c * 1e3:.1f} ms'} | {note} |")
