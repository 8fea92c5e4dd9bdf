#!/usr/bin/env python3
"""BASELINE config 4 call, host side: time until the launches are enqueued, until the values are on the host,
until the FFT is done (medians over 200 calls); then the same with the GPU work already finished by the time
the host asks (a sync before the copy) to separate host time from waiting."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd.model import Model

m = Model(10, 6, "Hardware_Efficient")
x = torch.from_numpy((2 * np.pi * np.arange(4096) / 4096).astype(np.float32).reshape(-1, 1)).cuda()
for _ in range(20):
    np.fft.fft(m(inputs=x, force_mean=True).cpu().numpy().astype(np.float64))
A, B, Cc, D = [], [], [], []
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = m(inputs=x, force_mean=True)
    t1 = time.perf_counter()
    h = y.cpu()
    t2 = time.perf_counter()
    h = h.numpy().astype(np.float64)
    c = np.fft.fft(h) / 4096
    t3 = time.perf_counter()
    A.append(t1 - t0); B.append(t2 - t1); Cc.append(t3 - t2); D.append(t3 - t0)
med = lambda v: sorted(v)[len(v) // 2] * 1e6
print(f"enqueue (model call returns) {med(A):.1f} us | device->host copy returns +{med(B):.1f} us | astype + FFT +{med(Cc):.1f} us | total {med(D):.1f} us")
import scipy.fft as sf
hh = h.copy()
def mirror(r, n):
    o = np.empty(n, dtype=np.complex128); o[:n // 2 + 1] = r; o[n // 2 + 1:] = np.conj(r[1:n // 2][::-1]); return o
for name, f in (("np.fft.fft", lambda: np.fft.fft(hh) / 4096), ("np.fft.rfft + mirror", lambda: mirror(np.fft.rfft(hh, norm="forward"), 4096)),
                ("scipy.fft.fft", lambda: sf.fft(hh) / 4096), ("scipy.fft.rfft + mirror", lambda: mirror(sf.rfft(hh, norm="forward"), 4096))):
    for _ in range(100): f()
    t = time.perf_counter()
    for _ in range(1000): f()
    print(f"  {name:26s} {(time.perf_counter() - t) * 1e3:.1f} us; max |d| vs np.fft.fft {np.abs(f() - np.fft.fft(hh) / 4096).max():.1e}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    y = m(inputs=x, force_mean=True)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); print(f"{st.total_calls / 200:.0f} Python calls per model call"); st.sort_stats("tottime").print_stats(18)
