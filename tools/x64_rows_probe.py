import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from qml_essentials_amd.model import Model
rng = np.random.default_rng(0)
def gpu_time(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        t0=time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    return sorted(ts)[len(ts)//2]
for n,L,B in ((6,3,256),(10,2,256),(16,2,16),(20,2,4)):
    for x64 in (False, True):
        m = Model(n, L, "Hardware_Efficient", x64=x64 or None)
        P = rng.uniform(0,6.28,(B,*m.params.shape[1:]))
        x = np.array([0.5])
        g = gpu_time(lambda: m(params=P, inputs=x, execution_type="expval"))
        print(f"Model({n},{L}) expval batch {B} x64={x64}: {g*1e3:.3f} ms", flush=True)
    m = Model(n, L, "Hardware_Efficient")
    P = rng.uniform(0,6.28,(min(B,16),*m.params.shape[1:]))
    for meth in ("adjoint", "parameter-shift"):
        try:
            g = gpu_time(lambda: m.gradient(params=P, inputs=np.array([0.5]), method=meth), reps=3)
            print(f"Model({n},{L}).gradient(method={meth}) batch {len(P)}: {g*1e3:.3f} ms", flush=True)
        except Exception as e:
            print(meth, "failed", repr(e)[:200])
