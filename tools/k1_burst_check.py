import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from qml_essentials_amd import _native as N
n = 20
for gate, slots in (("CX", 0), ("CRX", 1)):
    for c, t in ((1, 0), (0, 1), (5, 2), (2, 9), (10, 3), (3, 10), (9, 8)):
        st0 = torch.randn((2, 1 << n, 2), device="cuda", dtype=torch.float32)
        st0 = torch.view_as_complex(st0).contiguous()
        ang = torch.full((2, max(1, slots)), 0.7, device="cuda")
        plan = N.Plan([(gate, [c, t], [0] if slots else [], -1)], n, slots, flags=N.PLAN_NO_FUSION)
        outs = []
        for burst in ("0", "9"):
            os.environ["QMLE_K1_CTRL_BURST"] = burst
            s = st0.clone()
            N.apply_inplace(plan, ang, s)
            outs.append(s)
        assert torch.equal(outs[0], outs[1]), (gate, c, t)
print("mode 8 == mode 2 bit for bit")
