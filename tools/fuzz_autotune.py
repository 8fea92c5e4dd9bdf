#!/usr/bin/env python3
"""qmle_plan_autotune on random tapes: whatever schedule the tuner adopts (stages, tables and device image are
swapped inside a live plan handle), the states, probabilities and <Z> values must stay those of the default
schedule (float32 rounding of another gate order) and of the oracle.  Random 1- / 2- / 3-wire gate tapes of
40-120 gates on 15-22 qubits, default and all-live flags, every measurement kind, a second tuning call on the
same handle (cache hit) and a fresh handle of the same tape (per-process cache)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_port
from qml_essentials_amd import _native as N
from tests.helpers import random_tape, tape_to_native

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "5")))
bad = ran = adopted = 0
for trial in range(int(os.environ.get("FUZZ_N", "40"))):
    n = int(rng.integers(15, 23))
    tape = random_tape(n, int(rng.integers(40, 120)), rng, three_q=False)  # (the C port has no 3-wire gates)
    ops, angles, consts = tape_to_native(tape, n)
    flags = int(rng.choice([0, N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB, N.PLAN_NO_SPARSE]))
    B = int(rng.choice([1, 3, 8]))
    ang = torch.from_numpy(np.ascontiguousarray(np.repeat(angles[None, :], B, 0), dtype=np.float32)).cuda()
    obs = list(range(n))
    plan = N.Plan(ops, n, len(angles), consts, flags)
    st0 = plan.run(ang, "state").clone()
    ez0 = plan.run(ang, "expval", obs).clone()
    meas = str(rng.choice(["state", "expval"]))
    rep = plan.autotune(meas, len(obs) if meas == "expval" else 0, batch=B, top_k=int(rng.integers(2, 7)), reps=2)
    adopted += rep["ms_after"] < rep["ms_before"]
    st1 = plan.run(ang, "state")
    ez1 = plan.run(ang, "expval", obs)
    pr1 = plan.run(ang, "probs") if n <= 20 else None
    again = plan.autotune(meas, len(obs) if meas == "expval" else 0, batch=B, top_k=3, reps=2)
    st2 = plan.run(ang, "state")
    fresh = N.Plan(ops, n, len(angles), consts, flags)
    fresh.autotune(meas, len(obs) if meas == "expval" else 0, batch=B, top_k=3, reps=2)
    st3 = fresh.run(ang, "state")
    psi = c_port.simulate(tape, n)
    d = lambda a, b: float((torch.view_as_real(a) - torch.view_as_real(b)).abs().max())
    errs = dict(tuned_vs_default=d(st1, st0), again=d(st2, st0), fresh=d(st3, st0),
                ez=float((ez1 - ez0).abs().max()), oracle=float(np.abs(st1[0].cpu().numpy() - psi).max()))
    if pr1 is not None:
        errs["probs"] = float((pr1 - (st1.real ** 2 + st1.imag ** 2)).abs().max())
    ran += 1
    if max(errs.values()) > 2e-6 or not np.isfinite(list(errs.values())).all():
        bad += 1
        print("MISMATCH", n, len(tape), flags, B, meas, rep, errs)
print(f"{ran} tapes, schedule changed on {adopted}, mismatches: {bad}")
