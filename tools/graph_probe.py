#!/usr/bin/env python3
"""Does a torch.cuda.CUDAGraph capture the library's launches?  The C3 chain (sampler with a device
key -> angle table -> matrices -> circuit -> pair fidelities -> histogram) captured once, replayed
with a new key, compared with the eager chain and timed."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from qml_essentials_amd.expressibility import Expressibility
from qml_essentials_amd.model import Model
from qml_essentials_amd import utils

S, BINS = 1024, 75
m = Model(12, 3, "Hardware_Efficient", data_reupload=False)
kl_ref = Expressibility.kl_divergence_to_haar(m, n_samples=S, n_bins=BINS, random_key=1000)
print("eager kl", kl_ref)

# the compiled call behind the model (recorded by the eager call above)
params = m.device_params()
got = m._forward_device(params, None, None, "state", False, _want_call=True)
cc, leaves, divs, mods, B = got
print("B", B, "n_slots", cc.n_slots)
shape = tuple(params.shape)
n_vals = int(np.prod(shape))
key_host = torch.empty(2, dtype=torch.int64).pin_memory()
key_dev = torch.empty(2, dtype=torch.int64, device="cuda")
lib = N.lib()


def set_key(seed):
    k = utils.as_key(seed)
    _next, sub = utils.safe_random_split(k)
    w = utils._philox_words(sub)
    key_host.numpy().view(np.uint64)[:] = w
    key_dev.copy_(key_host, non_blocking=True)


def chain():
    p = torch.empty(shape, dtype=torch.float32, device="cuda")
    N.check(lib.qmle_philox_uniform_f32_device_key(C.c_void_p(key_dev.data_ptr()), n_vals, 0.0, 2 * np.pi,
                                                  C.c_void_p(p.data_ptr()), N._stream_ptr()))
    states = cc.run([p], divs, mods, B, 0)
    fid = N.pair_fidelity(states)
    return N.histogram(fid, BINS, 0.0, 1.0)


set_key(1000)
for _ in range(3):
    counts_eager = chain()
torch.cuda.synchronize()
print("eager chain counts sum", int(counts_eager.sum()))
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        chain()
torch.cuda.current_stream().wait_stream(side)
with torch.cuda.graph(g):
    counts_static = chain()
set_key(1000)
g.replay()
torch.cuda.synchronize()
print("graph == eager:", bool(torch.equal(counts_static, counts_eager)))
_, haar = Expressibility.haar_integral(12, BINS)
z = counts_static.cpu().numpy() / S
print("graph kl", Expressibility.kullback_leibler_divergence(z, haar))
set_key(7)
g.replay()
torch.cuda.synchronize()
c7 = counts_static.clone()
set_key(7)
c7e = chain()
print("re-seeded graph == eager:", bool(torch.equal(c7, c7e)))

for name, fn in (("eager", lambda: chain().cpu()), ("graph", lambda: (set_key(1000), g.replay(), counts_static.cpu())[-1])):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name}: median {ts[25] * 1e6:.1f} us, min {ts[0] * 1e6:.1f} us")
