#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): gate-applies/s and statevectors/s of the
data-reuploading Model hot path at n_qubits=24, batch=1024 per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A *step* = one call of ``Model(24, 1, "Hardware_Efficient", data_reupload=False)`` on a
batch of 1024 parameter sets per GPU through the drop-in API (``Model.__call__`` ->
``Script.execute`` -> ``libqmle_sv``): 96 reference gates per statevector (72 one-qubit +
24 CX, SURVEY.md 8-d "K2"), PauliZ expectation on all 24 wires.  Parameters are
synthetic U[0, 2 pi) float32 from ``numpy.random.default_rng(1000)`` and are resident in
HBM (a CUDA tensor) before the timed region; the per-sample angle table is built on the GPU
(``qmle_build_angles``), the statevectors are produced and consumed on the GPU and the result
is a CUDA tensor -- no host<->device traffic inside a step.  Weak scaling: every rank simulates its own 1024
states and one RCCL all-gather returns the (1024 N, 24) expectation values.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
``roofline`` (dominant kernel, timed live with HIP events on the launch stream) and
``cpu_baseline`` (the oracle's C/OpenMP port on a bounded sample, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md:36


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-qubits", type=int, default=24)
    ap.add_argument("--batch", type=int, default=1024, help="statevectors per GPU per step")
    ap.add_argument("--no-fusion", action="store_true", help="one HBM pass per reference gate")
    ap.add_argument("--skip-aux", action="store_true", help="skip K1 / expressibility / CPU legs")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def k1_single_gate(n=28, reps=10):
    """K1 of SURVEY.md 8-d: one gate per launch on a 2^n state, HIP-event timed."""
    from qml_essentials_amd import _native as N

    D = 1 << n
    st = torch.randn((1, D, 2), device="cuda", dtype=torch.float32)
    st = torch.view_as_complex(st / st.norm()).contiguous()
    ang = torch.full((1, 1), 1.234, device="cuda")
    out = {}
    for gate, wires_list, bytes_per_amp in (("RX", (0, 13, 27), 16), ("RZ", (13,), 16),
                                            ("CX", (0, 13), 8), ("CRX", (13,), 8)):
        for w in wires_list:
            wires = [w] if gate in ("RX", "RZ") else [w, (w + 1) % n]
            slots = [0] if gate != "CX" else []
            plan = N.Plan([(gate, wires, slots, -1)], n, 1, flags=N.PLAN_NO_FUSION)
            ws = torch.empty(plan.workspace_bytes(1, "state"), dtype=torch.uint8, device="cuda")
            for _ in range(2):
                N.apply_inplace(plan, ang, st, ws)
            plan.profile_begin(reps + 1)
            for _ in range(reps):
                N.apply_inplace(plan, ang, st, ws)
            ms, cnt, _ = plan.profile_end()
            avg = ms[0] / max(1, cnt[0])
            gbps = bytes_per_amp * D / avg / 1e6
            out[f"{gate}_wire{w}"] = {"ms": round(avg, 4), "GBps": round(gbps, 1),
                                      "frac_of_8TBps": round(gbps / HBM_PEAK_GBPS, 3)}
    del st
    torch.cuda.empty_cache()
    return out


def expressibility_wallclock(n=12, samples=1024):
    """BASELINE config 3: KL-to-Haar, 12 qubits, 1024 pairs, HE 3 layers, no DRU."""
    from qml_essentials_amd.expressibility import Expressibility
    from qml_essentials_amd.model import Model

    m = Model(n, 3, "Hardware_Efficient", data_reupload=False)
    Expressibility.kl_divergence_to_haar(m, n_samples=64, n_bins=75, random_key=1)  # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kl = Expressibility.kl_divergence_to_haar(m, n_samples=samples, n_bins=75, random_key=1000)
    torch.cuda.synchronize()
    return {"seconds": round(time.perf_counter() - t0, 5), "kl": float(np.mean(kl)),
            "n_qubits": n, "pairs": samples}


def adjoint_gradient_wallclock(n=20, layers=4):
    """Extra: gradient of mean_q <Z_q> w.r.t. all parameters of BASELINE config 2's model
    (20 qubits, 4 layers, 300 parameters) by the fused adjoint sweep, CUDA tensors in and out."""
    from qml_essentials_amd.model import Model

    m = Model(n, layers, "Hardware_Efficient")
    p = torch.tensor(np.asarray(m.params[0]), dtype=torch.float32, device="cuda")
    x = torch.tensor([[0.5]], dtype=torch.float32, device="cuda")
    cot = torch.ones((1,), dtype=torch.float32, device="cuda")
    m.vjp_device(p, x, cot, force_mean=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g, _ = m.vjp_device(p, x, cot, force_mean=True)
    torch.cuda.synchronize()
    return {"ms": round((time.perf_counter() - t0) / 10 * 1e3, 3), "n_qubits": n,
            "n_params": int(p.numel()), "finite": bool(torch.isfinite(g).all())}


def cpu_baseline(n, params_row, budget_s):
    """Oracle C/OpenMP port on a bounded sample of the same workload (rank 0, N=1)."""
    from oracle import c_port, circuits as OC

    spec = OC.ModelSpec(n, 1, "Hardware_Efficient", data_reupload=False)
    threads = c_port.lib().svc_max_threads()
    done, t0 = 0, time.perf_counter()
    while True:
        tape = OC.model_tape(spec, params_row[done % len(params_row)], [0.0])
        psi = c_port.simulate(tape, n)
        c_port.expval_z(psi, n, list(range(n)))
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 64:
            break
    gates = sum(1 for g in tape if g[0] != "Barrier")
    return {"value": round(done * gates / el, 2), "unit": "gate-applies/s", "cores": threads,
            "kind": "port", "statevectors_per_s": round(done / el, 4),
            "sample": f"{done} of the statevectors of one step (same tape: {gates} gates + <Z> on "
                      f"{n} wires, n={n}), oracle/sv_cpu.c with {threads} OpenMP threads, "
                      f"{el:.1f} s"}


def dense_state_run(n, B, params_dev, n_gates, folded):
    """The same step with known-zero tracking switched off (QMLE_PLAN_NO_SPARSE): every pass
    reads / computes / stores all 2^n amplitudes, as a circuit without exploitable zeros would.
    Reported beside `value` so that the gain from skipping known zeros is visible as such."""
    import torch
    from qml_essentials_amd import _native as N
    from qml_essentials_amd import simulation
    from qml_essentials_amd.model import Model

    saved = simulation.PLAN_FLAGS
    simulation.PLAN_FLAGS = saved | N.PLAN_NO_SPARSE
    try:
        model = Model(n, 1, "Hardware_Efficient", data_reupload=False)
        tape, _ = model.record_tape(params=params_dev[:2].cpu().numpy())
        plan = simulation.get_plan(simulation.LoweredTape(tape, n))
        plan = plan.expval_child() or plan
        st = plan.describe()["stages"]
        moved = sum(s_["read_bytes_from_zero"] + (s_["write_bytes_from_zero"] if i + 1 < len(st) else 0.0)
                    for i, s_ in enumerate(st))
        model(params=params_dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model(params=params_dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        simulation.PLAN_FLAGS = saved
    return {"ms_per_step": round(dt * 1e3, 3), "gate_applies_per_s": round(n_gates * B / dt, 1),
            "statevectors_per_s": round(B / dt, 2), "hbm_passes_per_state": len(st),
            "hbm_bytes_moved_per_state": moved,
            "moved_GBps": round(moved * B / dt / 1e9, 1),
            "moved_frac_of_8TBps": round(moved * B / dt / 1e9 / HBM_PEAK_GBPS, 4),
            "note": "QMLE_PLAN_NO_SPARSE: all 2^n amplitudes read / computed / stored in every pass"}


def main():
    a = parse_args()
    from qml_essentials_amd import distributed
    import __graft_entry__ as entry

    rank, size = distributed.init_from_env()
    if rank == 0:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):  # stdout carries the ONE JSON line only
            entry.build()
    distributed.barrier()
    from qml_essentials_amd import _native as N
    from qml_essentials_amd import simulation
    from qml_essentials_amd.model import Model

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    if size == 1:
        torch.cuda.set_device(0)
    if a.no_fusion:
        simulation.PLAN_FLAGS = N.PLAN_NO_FUSION

    n, B = a.n_qubits, a.batch
    model = Model(n, 1, "Hardware_Efficient", data_reupload=False)
    rng = np.random.default_rng(1000)
    params = rng.uniform(0, 2 * np.pi, (B * size, *model.params.shape[1:])).astype(np.float32)

    tape, _ = model.record_tape(params=params[:2])
    low = simulation.LoweredTape(tape, n)
    plan = simulation.get_plan(low)
    # <Z> runs the plan without the trailing CX layers (folded into parity observables); the
    # folded gates' algorithmic bytes are credited to its last pass (describe())
    folded = plan.describe().get("absorbed_ops", 0)
    plan = plan.expval_child() or plan
    desc = plan.describe()
    n_gates = len(low.ops)

    # inputs resident in HBM before the timed region: the (B*size, 1, 72) parameter tensor
    params_dev = torch.from_numpy(params).cuda()

    def step():
        return model(params=params_dev)  # CUDA tensor (B*size, n) on every rank (one all-gather)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    distributed.barrier()
    n_stages = len(desc["stages"])
    if rank == 0:
        plan.profile_begin(n_stages * B * a.steps + 16)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    distributed.barrier()
    elapsed = time.perf_counter() - t0
    if size > 1:
        dev = "cuda" if torch.distributed.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert tuple(out.shape) == (B * size, n) and bool(torch.isfinite(out).all())

    expr = None
    if not a.skip_aux:  # sharded over all ranks -> every rank takes part
        expr = expressibility_wallclock()
    if rank != 0:
        distributed.barrier()
        return
    stage_ms, stage_cnt, overflow = plan.profile_end()
    total_states = B * size * a.steps
    value = n_gates * total_states / elapsed
    D = float(1 << n)

    # dominant kernel = the kernel family with the largest summed device time
    fam = {}
    for i, st in enumerate(desc["stages"]):
        k = {"tile": "k_tile", "direct": "k_direct_1q", "diag_all": "k_diag_all"}[st["kind"]]
        if st["kind"] == "tile" and st.get("product") and 0 < i < len(desc["stages"]) - 1:
            # (+ k_fold_columns, inside the same timed scope); the streaming layout takes over
            # when bit 0 is live and >= 512 live amplitudes x states are in flight (launch_tile)
            live = bin(~st["zero_in"] & ((1 << n) - 1)).count("1")
            k = "k_product_stream" if live >= 9 and not st["zero_in"] & 1 else "k_tile_product"
        if i == len(desc["stages"]) - 1 and st["kind"] == "tile":
            k = st["expval_kernel"].replace("_fold", "")  # <Z> out of the last pass: which kernel runs it
        f = fam.setdefault(k, {"ms": 0.0, "launches": 0, "algo": 0.0, "moved": 0.0})
        f["ms"] += stage_ms[i]
        f["launches"] += stage_cnt[i]
        per_launch_states = (B * a.steps) / max(1, stage_cnt[i])
        f["algo"] += st["algo_bytes_per_state"] * per_launch_states * stage_cnt[i]
        # HBM bytes the plan compiler expects this stage to move in a run from |0..0>
        # (known-zero amplitudes are never read, all-zero tiles never stored)
        moved = st["read_bytes_from_zero"] + st["write_bytes_from_zero"]
        if i == len(desc["stages"]) - 1 and st["kind"] == "tile":
            moved = st["read_bytes_from_zero"]  # <Z> straight out of the last pass: nothing stored
        f["moved"] += moved * per_launch_states * stage_cnt[i]
    dom_name = max(fam, key=lambda k: fam[k]["ms"])
    dom = fam[dom_name]
    achieved = dom["algo"] / (dom["ms"] * 1e-3) / 1e9 if dom["ms"] > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath)).get(f"{dom_name}:n{n}:{'nofusion' if a.no_fusion else 'fused'}")
            traffic = rec["hbm_bytes_per_launch"] if rec else None
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm", "kernel": dom_name,
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
        "avg_launch_ms": round(dom["ms"] / max(1, dom["launches"]), 5),
        "launches": dom["launches"],
        "algorithmic_bytes_per_launch": round(dom["algo"] / max(1, dom["launches"])),
        "bytes_moved_per_launch": round(dom["moved"] / max(1, dom["launches"])),
        "moved_GBps": round(dom["moved"] / (dom["ms"] * 1e-3) / 1e9, 1) if dom["ms"] > 0 else 0.0,
        "moved_frac": round(dom["moved"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        if dom["ms"] > 0 else 0.0,
        "note": "algorithmic bytes = sum of the per-gate bytes (SURVEY 8-d) of the reference "
                "gates one launch applies; a fused pass applies many gates per HBM round trip and "
                "a run from |0..0> never reads or stores amplitudes that are still exactly zero, "
                "so achieved exceeds the HBM peak; bytes_moved_per_launch / moved_GBps / moved_frac "
                "are the bytes the dominant kernel really streams (they match the PMC traffic); "
                "dense_state is the same step with every amplitude read / computed / stored",
        "event_pool_overflow": overflow,
    }
    result = {
        "metric": "gate_applies_per_s", "value": round(value, 1), "unit": "gate-applies/s",
        "n_gpus": size, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "complex64", "data": "synthetic",
        "config": {"workload": f"K2: Model({n}, 1, Hardware_Efficient, data_reupload=False) expval "
                               f"on all wires, {n_gates} gates/state (72 1q + 24 CX at n=24), "
                               f"batch {B} statevectors per GPU per step",
                   "n_qubits": n, "batch_per_gpu": B, "gates_per_state": n_gates,
                   "hbm_passes_per_state": len(desc["stages"]), "fusion": not a.no_fusion,
                   "gates_folded_into_observables": folded,
                   "parallelism": f"batch-sharded x{size}"},
        "statevectors_per_s": round(total_states / elapsed, 2),
        # transparency: `value` counts every gate of the reference tape; this one leaves out the
        # trailing CX layer that <Z> folds into its observables instead of applying to the state
        "gate_applies_per_s_state_applied_only": round((n_gates - folded) * total_states / elapsed, 1),
        "roofline": roofline,
    }
    if not a.skip_aux and size == 1 and not a.no_fusion:
        result["dense_state"] = dense_state_run(n, B, params_dev, n_gates, folded)
    if not a.skip_aux and size == 1:
        result["cpu_baseline"] = cpu_baseline(n, params[:64], a.cpu_seconds)
        try:
            result["k1_single_gate_28q"] = k1_single_gate()
        except Exception as e:  # pragma: no cover - e.g. not enough free HBM
            result["k1_single_gate_28q"] = {"error": str(e)}
    if expr is not None:
        result["expressibility_12q_1024pairs"] = expr
        if size == 1:
            try:
                result["adjoint_gradient_20q"] = adjoint_gradient_wallclock()
            except Exception as e:  # extras never break the headline line
                result["adjoint_gradient_20q"] = {"error": str(e)}
    distributed.barrier()
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
