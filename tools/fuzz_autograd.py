#!/usr/bin/env python3
"""torch_bridge.differentiable with a BATCH of parameter sets (B_P, layers, n_params) -- round 4: one adjoint
sweep over the B_I x B_P batch -- against (a) the same loss summed set by set (B_P = 1 calls) and (b)
Model.gradient(method="parameter-shift") contracted with the loss weights; random ansatz / size / layers /
batch shapes, gradients with respect to the parameters (and the inputs where the route supports them)."""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd.ansaetze import Ansaetze
from qml_essentials_amd.model import Model
from qml_essentials_amd.torch_bridge import differentiable

warnings.simplefilter("ignore")
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "3")))
names = [a.__name__ for a in Ansaetze.get_available()]
bad = ran = 0
for trial in range(int(os.environ.get("FUZZ_N", "60"))):
    n = int(rng.integers(1, 9)) if rng.random() < 0.85 else int(rng.integers(14, 16))
    kw = dict(n_qubits=n, n_layers=int(rng.integers(1, 3)), circuit_type=str(rng.choice(names)),
              data_reupload=bool(rng.integers(2)))
    try:
        m = Model(**kw)
    except Exception:
        continue
    if 0 in m.params.shape:
        continue
    B_P, B_I = int(rng.choice([2, 3, 5])), int(rng.choice([1, 2, 4]))
    P = rng.uniform(0, 6.28, (B_P, *m.params.shape[1:])).astype(np.float32)
    X = rng.uniform(-1, 1, (B_I, m.n_input_feat)).astype(np.float32)
    tag = f"{trial} n={n} {kw['circuit_type']} L={kw['n_layers']} dru={kw['data_reupload']} B_P={B_P} B_I={B_I}"
    try:
        f = differentiable(m)
        pd = torch.tensor(P, device="cuda", requires_grad=True)
        xd = torch.tensor(X, device="cuda")
        y = f(pd, xd, force_mean=True)
        W = torch.from_numpy(rng.uniform(-1, 1, tuple(y.shape)).astype(np.float32)).cuda()
        (y * W).sum().backward()
        g_batch = pd.grad.detach().cpu().numpy()
        # (a) set by set
        g_sets = np.zeros_like(g_batch)
        for k in range(B_P):
            pk = torch.tensor(P[k], device="cuda", requires_grad=True)
            yk = f(pk, xd, force_mean=True)
            wk = W.reshape(B_I, B_P)[:, k].reshape(yk.shape) if y.dim() else W
            (yk * wk).sum().backward()
            g_sets[k] = pk.grad.detach().cpu().numpy()
        e1 = float(np.abs(g_batch - g_sets).max())
        ok = e1 < 2e-5
    except (NotImplementedError, ValueError) as e:
        print(tag, "->", type(e).__name__, str(e)[:100], flush=True)
        continue
    except Exception as e:
        bad += 1
        print(tag, "-> ERROR", type(e).__name__, str(e)[:200], flush=True)
        continue
    ran += 1
    bad += not ok
    if not ok or trial % 20 == 0:
        print(tag, tuple(y.shape), g_batch.shape, "batch-vs-sets", e1, "ok" if ok else "MISMATCH", flush=True)
print(f"{ran} models, mismatches / errors: {bad}")
