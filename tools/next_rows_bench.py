#!/usr/bin/env python3
"""Wall-clock of the widened rows (multi-register entanglement circuits, the density-matrix / noise
path, shots) through the drop-in API on the GPU, with the oracle's CPU time for one sample beside it where the
oracle states the same computation.  Prints a markdown table (profiles/rNN_next_rows.md)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib
import __graft_entry__ as entry
with contextlib.redirect_stdout(sys.stderr):
    entry.build()
from oracle import noise as ON, einsum_sim as OE
from qml_essentials_amd.model import Model
from qml_essentials_amd.entanglement import Entanglement
from qml_essentials_amd.tape import recording
from qml_essentials_amd.utils import PRNGKey
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from helpers import frontend_to_oracle

rng = np.random.default_rng(1000)
rows = []
NOISE = {"BitFlip": 0.01, "PhaseFlip": 0.02, "Depolarizing": 0.03, "AmplitudeDamping": 0.05, "PhaseDamping": 0.06}


def gpu_time(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def row(name, g, c=None, note="-"):
    rows.append((name, g, c, note))
    print(f"{name}: {g * 1e3:.3f} ms" + (f" (oracle {c * 1e3:.1f} ms, {note})" if c else ""), file=sys.stderr, flush=True)


# ---- rank 3: density-matrix / noise path -------------------------------------------------------------
for n, layers, B in ((6, 3, 256), (8, 3, 256), (10, 2, 64), (12, 1, 16)):
    m = Model(n, layers, "Hardware_Efficient")
    P = rng.uniform(0, 6.28, (B, *m.params.shape[1:])).astype(np.float32)
    x = np.array([0.5], dtype=np.float32)
    for et in ("expval", "density"):
        if et == "density" and n > 10:
            continue
        g = gpu_time(lambda: m(params=P, inputs=x, noise_params=dict(NOISE), execution_type=et), reps=3)
        c = None
        if et == "expval" and n <= 8:
            with recording() as tape:
                m._variational(P[0], x, random_key=PRNGKey(0), noise_params=m.noise_params)
            ot = frontend_to_oracle(tape)
            t0 = time.perf_counter(); rho = ON.simulate_mixed(ot, n); c = (time.perf_counter() - t0) * B
        row(f"noisy Model({n},{layers},HE), 5 channels per gate layer, {et}, batch {B} ({2 * n}-wire register)", g, c,
            f"oracle/noise.py simulate_mixed (numpy einsum, complex128), 1 sample x{B}" if c else "-")

# ---- rank 1: multi-register entanglement circuits --------------------------------------------------
for n, S in ((6, 512), (10, 256), (12, 64)):
    m = Model(n, 2, "Hardware_Efficient")
    g = gpu_time(lambda: Entanglement.bell_measurements(m, n_samples=S, random_key=PRNGKey(1000)), reps=3)
    row(f"Entanglement.bell_measurements Model({n},2,HE), {S} samples ({2 * n}-qubit circuits, Z-parities out of the last pass)", g)
for n, S in ((4, 512), (8, 64)):
    m = Model(n, 2, "Hardware_Efficient")
    g = gpu_time(lambda: Entanglement.concentratable_entanglement(m, n_samples=S, random_key=PRNGKey(1000)), reps=3)
    row(f"Entanglement.concentratable_entanglement Model({n},2,HE), {S} samples ({3 * n}-qubit swap tests)", g)
m = Model(4, 2, "Hardware_Efficient")
g = gpu_time(lambda: Entanglement.entanglement_of_formation(m, n_samples=64, random_key=PRNGKey(1000)), reps=3)
row("Entanglement.entanglement_of_formation Model(4,2,HE), 64 samples (densities from the engine, eigh on the host)", g)

# ---- rank 4: shots -----------------------------------------------------------------------------------
for n, B, shots in ((10, 256, 1024), (16, 64, 4096), (20, 8, 8192)):
    m = Model(n, 2, "Hardware_Efficient")
    P = rng.uniform(0, 6.28, (B, *m.params.shape[1:])).astype(np.float32)
    x = np.array([0.5], dtype=np.float32)
    g0 = gpu_time(lambda: m(params=P, inputs=x, execution_type="expval"), reps=3)
    ms = Model(n, 2, "Hardware_Efficient", shots=shots)
    g = gpu_time(lambda: ms(params=P, inputs=x, execution_type="expval"), reps=3)
    row(f"Model({n},2,HE) expval from {shots} shots, batch {B} (exact expval: {g0 * 1e3:.3f} ms)", g)
    g = gpu_time(lambda: ms(params=P, inputs=x, execution_type="probs"), reps=3)
    row(f"Model({n},2,HE) probs from {shots} shots, batch {B}", g)

print("| widened row | MI355X wall | CPU oracle | CPU note |")
print("|---|---|---|---|")
for name, g, c, note in rows:
    print(f"| {name} | {g * 1e3:.3f} ms | {'%.1f ms' % (c * 1e3) if c else ''} | {note} |")
