import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from qml_essentials_amd import _native as N
from dense_profile import he_ops
def t(n, B, layers, flags):
    ops, slots = he_ops(n, layers)
    ang = torch.from_numpy(np.random.default_rng(1).uniform(0, 6.28, (B, slots)).astype(np.float32)).cuda()
    top = N.Plan(ops, n, slots, flags=flags)
    ws = torch.empty(top.workspace_bytes(B, "expval", n), dtype=torch.uint8, device="cuda")
    obs = list(range(n))
    out = top.run(ang, "expval", obs, workspace=ws); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): out = top.run(ang, "expval", obs, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3 / B * 1e3, out
for n, layers in ((24, 1), (24, 2), (24, 3), (24, 4), (20, 4), (22, 3), (16, 2), (18, 3)):
    os.environ.pop("QMLE_X", None)
    a, oa = t(n, 64, layers, 0)
    b, ob = t(n, 64, layers, N.PLAN_NO_ABSORB)
    d = N.Plan(*((lambda o: (o[0], n, o[1]))(he_ops(n, layers)))).describe()
    print(f"n={n} layers={layers}: engine choice ({'folded' if 'expval_plan' in d else 'applied'}) {a:.1f} us/state, NO_ABSORB {b:.1f} us/state, max diff {float((oa-ob).abs().max()):.2e}", flush=True)
