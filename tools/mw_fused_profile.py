import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from qml_essentials_amd import _native as N
n = 28
ops, slots = bench._he_layer_ops(n)
ang = torch.from_numpy(np.random.default_rng(6).uniform(0, 6.28, (1, slots)).astype(np.float32)).cuda()
plan = N.Plan(ops, n, slots)
st = plan.run(ang, "state")
ws_m = torch.empty(plan.workspace_bytes(1, "mw"), dtype=torch.uint8, device="cuda")
for _ in range(30):
    plan.run(ang, "state", out=st)
for _ in range(30):
    plan.run(ang, "mw", workspace=ws_m)
for _ in range(30):
    N.meyer_wallach(st)
torch.cuda.synchronize()
# 12-qubit loop
from qml_essentials_amd.entanglement import Entanglement
from qml_essentials_amd.model import Model
m = Model(12, 3, "Hardware_Efficient", data_reupload=False)
for _ in range(10):
    Entanglement.meyer_wallach(m, n_samples=2048, random_key=1000)
torch.cuda.synchronize()
