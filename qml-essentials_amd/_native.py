"""ctypes binding of ``libqmle_sv.so`` (C ABI declared in ``include/qmle_sv.h``).

PyTorch is used only for device memory and streams (``tensor.data_ptr()``,
``torch.cuda.current_stream().cuda_stream``); no torch type crosses the ABI.
There is NO CPU fallback: if the library is missing, or a compute entry point is
called without a GPU, this module raises.
"""
from __future__ import annotations

import ctypes as C
import functools
import json
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libqmle_sv.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)

MAX_QUBITS = 32

# qmle_status -> (exception type, message) -- messages follow the reference where it
# has one (operations.py:140-146, simulation.py:271)
OK = 0
_STATUS_EXC = {
    -1: ValueError,
    -2: ValueError,
    -3: ValueError,
    -4: ValueError,
    -5: ValueError,
    -6: ValueError,
    -7: RuntimeError,
    -8: RuntimeError,
    -9: RuntimeError,
    -10: NotImplementedError,
    -11: ValueError,
}

OPCODES = {
    "Id": 0, "PauliX": 1, "PauliY": 2, "PauliZ": 3, "H": 4, "S": 5,
    "RX": 6, "RY": 7, "RZ": 8, "Rot": 9,
    "CX": 10, "CY": 11, "CZ": 12, "CRX": 13, "CRY": 14, "CRZ": 15,
    "CPhase": 16, "ControlledPhaseShift": 16, "SWAP": 17,
    "RXX": 18, "RYY": 19, "RZZ": 20, "RZX": 21, "CCX": 22, "CSWAP": 23,
    "MAT1": 24, "MAT2": 25, "DIAG_ALL": 26, "MAT4": 27,
}
MEAS = {"state": 0, "probs": 1, "expval": 2, "density": 3, "mw": 4}

PLAN_DEFAULT = 0
PLAN_NO_FUSION = 1
PLAN_FORCE_GLOBAL = 2
PLAN_FORCE_TILE = 4
PLAN_NO_REGTILE = 8
PLAN_PREFETCH = 16
PLAN_NO_ABSORB = 32
PLAN_NO_MERGE = 64
PLAN_NO_SPARSE = 128
PLAN_TAPE_ORDER = 1 << 24


def plan_flags(no_fusion=False, force_global=False, force_tile=False, tile_bits=0, low_bits=0,
               prefetch=False, no_absorb=False, no_sparse=False, tape_order=False):
    f = (PLAN_PREFETCH if prefetch else 0) | (PLAN_NO_ABSORB if no_absorb else 0)
    f |= PLAN_NO_SPARSE if no_sparse else 0
    f |= PLAN_TAPE_ORDER if tape_order else 0
    if no_fusion:
        f |= PLAN_NO_FUSION
    if force_global:
        f |= PLAN_FORCE_GLOBAL
    if force_tile:
        f |= PLAN_FORCE_TILE
    return f | ((tile_bits & 0xFF) << 8) | ((low_bits & 0xFF) << 16)


class QmleOp(C.Structure):
    _fields_ = [
        ("opcode", C.c_uint16),
        ("wire", C.c_int16 * 4),
        ("slot", C.c_int32 * 3),
        ("mat_off", C.c_int32),
    ]


class QmleAngleMap(C.Structure):
    """qmle_angle_map (include/qmle_sv.h): qmle_build_angles' arguments for qmle_run_batch_map."""
    _fields_ = [
        ("d_leaves", C.POINTER(C.c_void_p)),
        ("leaf_strides", C.POINTER(C.c_int64)),
        ("leaf_div", C.POINTER(C.c_int32)),
        ("leaf_mod", C.POINTER(C.c_int32)),
        ("n_leaves", C.c_int32),
        ("d_ptr", C.c_void_p), ("d_arg", C.c_void_p), ("d_idx", C.c_void_p),
        ("d_coef", C.c_void_p), ("d_const", C.c_void_p), ("d_period", C.c_void_p),
        ("batch_offset", C.c_int64),
    ]


class AngleMapArgs:
    """A reusable qmle_angle_map: the map's device arrays are fixed, the leaves change per call."""

    def __init__(self, n_leaves, d_ptr, d_arg, d_idx, d_coef, d_const, d_period):
        k = max(1, n_leaves)
        self.lp, self.ls = (C.c_void_p * k)(), (C.c_int64 * k)()
        self.ld, self.lm = (C.c_int32 * k)(), (C.c_int32 * k)()
        self._keep = (d_ptr, d_arg, d_idx, d_coef, d_const, d_period)
        self.c = QmleAngleMap(C.cast(self.lp, C.POINTER(C.c_void_p)), C.cast(self.ls, C.POINTER(C.c_int64)),
                              C.cast(self.ld, C.POINTER(C.c_int32)), C.cast(self.lm, C.POINTER(C.c_int32)),
                              n_leaves, d_ptr.data_ptr(), d_arg.data_ptr(), d_idx.data_ptr(), d_coef.data_ptr(),
                              d_const.data_ptr(), d_period.data_ptr() if d_period is not None else None, 0)
        self.ref = C.byref(self.c)

    def set(self, leaves, strides, divs, mods, batch_offset=0):
        for k, t in enumerate(leaves):
            self.lp[k] = t.data_ptr()
            self.ls[k] = int(strides[k])
            self.ld[k] = int(divs[k])
            self.lm[k] = int(mods[k])
        self.c.batch_offset = int(batch_offset)
        return self


_lib = None

# Opt-in plan autotuner (QMLE_AUTOTUNE=1 or set_autotune(True)): a plan's first "state" / "expval" run on
# the GPU times the cost model's best schedules for that batch size and keeps the fastest
# (Plan.autotune / qmle_plan_autotune).  Off by default: the default schedule is deterministic.
_AUTOTUNE = os.environ.get("QMLE_AUTOTUNE", "0") not in ("", "0")


def set_autotune(on: bool = True) -> None:
    global _AUTOTUNE
    _AUTOTUNE = bool(on)


# every symbol include/qmle_sv.h declares: (name, restype, argtypes)
_VP, _I, _SZ, _F = C.c_void_p, C.c_int, C.c_size_t, C.c_float
SYMBOLS = [
    ("qmle_sv_version", _I, []),
    ("qmle_status_string", C.c_char_p, [_I]),
    ("qmle_device_count", _I, []),
    ("qmle_plan_create", _I, [C.POINTER(QmleOp), _I, _I, _I, C.POINTER(_F), _I, C.c_uint,
                               C.POINTER(_VP)]),
    ("qmle_plan_destroy", _I, [_VP]),
    ("qmle_plan_expval_child", _VP, [_VP]),
    ("qmle_plan_executed", _VP, [_VP, _I]),
    ("qmle_plan_describe", _I, [_VP, C.c_char_p, _SZ]),
    ("qmle_plan_stats", _I, [_VP, C.POINTER(C.c_int64)]),
    ("qmle_plan_autotune", _I, [_VP, _I, _I, _I, _I, _I, _VP, C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                C.POINTER(C.c_double)]),
    ("qmle_workspace_bytes", _SZ, [_VP, _I, _I, _I, _I]),
    ("qmle_run_batch", _I, [_VP, _VP, _I, _I, C.POINTER(C.c_int32), _I, _VP, _VP, _SZ, _VP]),
    ("qmle_run_batch_parity", _I, [_VP, _VP, _I, C.POINTER(C.c_uint32), _I, _VP, _VP, _SZ, _VP]),
    ("qmle_build_angles", _I, [C.POINTER(_VP), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                               C.POINTER(C.c_int32), _I, _VP, _VP, _VP, _VP, _VP, _VP, _I,
                               C.c_int64, C.c_int64, _VP, _VP]),
    ("qmle_run_batch_map", _I, [_VP, _VP, _VP, _I, _I, C.POINTER(C.c_int32), _I, _VP, _VP, _SZ, _VP]),
    ("qmle_apply_inplace", _I, [_VP, _VP, _I, _VP, _VP, _SZ, _VP]),
    ("qmle_profile_begin", _I, [_VP, _I]),
    ("qmle_profile_end", _I, [_VP, C.POINTER(C.c_double), C.POINTER(C.c_int64), _I]),
    ("qmle_expval_z", _I, [_VP, _I, _I, C.POINTER(C.c_int32), _I, _VP, _VP, _SZ, _VP]),
    ("qmle_expval_workspace_bytes", _SZ, [_I, _I]),
    ("qmle_probs", _I, [_VP, _I, _I, _VP, _VP]),
    ("qmle_density", _I, [_VP, _I, _I, _VP, _VP]),
    ("qmle_marginal_probs", _I, [_VP, _I, _I, C.POINTER(C.c_int32), _I, _VP, _VP]),
    ("qmle_pair_fidelity", _I, [_VP, _I, _I, _VP, _VP, _SZ, _VP]),
    ("qmle_pair_fidelity_workspace_bytes", _SZ, [_I, _I]),
    ("qmle_density_probs", _I, [_VP, _I, _I, _VP, _VP]),
    ("qmle_density_expval_z", _I, [_VP, _I, _I, C.POINTER(C.c_int32), _I, _VP, _VP]),
    ("qmle_overlap", _I, [_VP, _VP, _I, _I, _VP, _VP, _SZ, _VP]),
    ("qmle_overlap_workspace_bytes", _SZ, [_I, _I]),
    ("qmle_expval_parity", _I, [_VP, _I, _I, C.POINTER(C.c_uint32), _I, _VP, _VP, _SZ, _VP]),
    ("qmle_expval_parity_workspace_bytes", _SZ, [_I, _I]),
    ("qmle_meyer_wallach", _I, [_VP, _I, _I, _VP, _VP, _VP, _SZ, _VP]),
    ("qmle_meyer_wallach_workspace_bytes", _SZ, [_I, _I]),
    ("qmle_meyer_wallach_reads", _I, [_I]),
    ("qmle_philox_uniform_f32", _I, [_VP, C.c_uint64, C.c_double, C.c_double, _VP]),
    ("qmle_philox_uniform_f32_device", _I, [_VP, C.c_uint64, C.c_double, C.c_double, _VP, _VP]),
    ("qmle_philox_uniform_f32_device_key", _I, [_VP, C.c_uint64, C.c_double, C.c_double, _VP, _VP]),
    ("qmle_run_batch_f64", _I, [_VP, _VP, _I, _I, C.POINTER(C.c_uint32), _I, _VP, _VP, _SZ, _VP]),
    ("qmle_workspace_bytes_f64", _SZ, [_VP, _I, _I]),
    ("qmle_plan_set_consts_f64", _I, [_VP, C.POINTER(C.c_double), _I]),
    ("qmle_apply_inplace_f64", _I, [_VP, _VP, _I, _VP, _VP, _SZ, _VP]),
    ("qmle_apply_inplace_f64_workspace_bytes", _SZ, [_VP, _I]),
    ("qmle_histogram", _I, [_VP, C.c_int64, _I, _F, _F, _VP, _VP]),
    ("qmle_adjoint_gradient", _I, [_VP, _VP, _VP, _VP, _I, _VP, C.POINTER(C.c_uint32), _I, _VP, _I,
                                   _VP, _I, _VP, _SZ, _VP]),
    ("qmle_adjoint_workspace_bytes", _SZ, [_VP, _VP, _I]),
    ("qmle_adjoint_gradient_f64", _I, [_VP, _VP, _VP, _VP, _I, _VP, C.POINTER(C.c_uint32), _I, _VP, _I,
                                       _VP, _I, _VP, _SZ, _VP]),
    ("qmle_adjoint_workspace_bytes_f64", _SZ, [_VP, _VP, _I]),
    ("qmle_sample_counts", _I, [_VP, _I, _I, _I, C.c_uint64, C.c_uint64, _VP, _VP, _VP, _SZ,
                                _VP]),
    ("qmle_sample_workspace_bytes", _SZ, [_I, _I]),
    ("qmle_probs_diag_expval", _I, [_VP, _I, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), _VP, _I, _VP, _VP, _SZ, _VP]),
    ("qmle_probs_diag_expval_workspace_bytes", _SZ, [_I]),
]


def lib() -> C.CDLL:
    """Load libqmle_sv.so (built by ``__graft_entry__.build()``); fail loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        # PyTorch-ROCm bundles its own libamdhip64 (soname libamdhip64.so.7).  It must be
        # in the process BEFORE libqmle_sv.so resolves that soname, otherwise a second
        # HIP runtime (/opt/rocm) is loaded and torch's device pointers are foreign to it.
        import torch  # noqa: F401

        handle = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(handle, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


class Unsupported(NotImplementedError):
    """QMLE_ERR_UNSUPPORTED: the engine has no kernel for this request (callers may fall back
    to another plan shape)."""


def check(status: int, what: str = "") -> None:
    if status == OK:
        return
    if status == -10:
        raise Unsupported(f"{what}: {lib().qmle_status_string(status).decode()} (qmle status -10; "
                          "this is also what a plan answers when it is run on another GPU than the "
                          "one of its first run -- build one Plan per device)")
    msg = lib().qmle_status_string(status).decode()
    raise _STATUS_EXC.get(status, RuntimeError)(f"{what}: {msg} (qmle status {status})")


_torch_gpu = None  # torch, once a GPU has been seen (the answer does not change within a process)


def require_gpu():
    global _torch_gpu
    if _torch_gpu is not None:
        return _torch_gpu
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError(
            "qml-essentials_amd needs an AMD GPU (gfx950); no CPU fallback exists. "
            "Run on the MI355X box (gpurun)."
        )
    _torch_gpu = torch
    return torch


PINNED_RESULT_BYTES = 1 << 20  # results at least this large leave the GPU through a page-locked buffer


def to_host(t) -> np.ndarray:
    """Device tensor -> numpy array.  Statevectors, density matrices and probability tables are tens to
    hundreds of MiB per call: a copy into pageable memory moves 6.5 GiB/s on the MI355X host, the same copy into
    a page-locked buffer 53 GiB/s (``tools/d2h_probe.py``).  Large results are therefore written into a pinned
    tensor (torch's caching host allocator: the block is reused once the array is dropped) and returned as the
    numpy view of it -- no second copy."""
    torch = require_gpu()
    if not t.is_cuda:
        return t.numpy()
    t = t.detach()
    if t.numel() * t.element_size() < PINNED_RESULT_BYTES:
        return t.cpu().numpy()
    if not t.is_contiguous():
        t = t.contiguous()
    try:
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    except RuntimeError:  # (no page-locked memory to be had -- a memlock limit, a fragmented host: pageable copy)
        return t.cpu().numpy()
    h.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return h.numpy()


def current_device():
    """``torch.device`` of the current GPU (cached objects: this sits on every call's path)."""
    torch = require_gpu()
    i = torch.cuda.current_device()
    d = _devices.get(i)
    if d is None:
        d = _devices[i] = torch.device("cuda", i)
    return d


_devices: dict = {}


def _stream_ptr():
    # the raw hipStream_t of torch's current stream (what `current_stream().cuda_stream` returns,
    # without building a Stream object per launch)
    torch = require_gpu()
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _i32(values: Sequence[int]):
    arr = (C.c_int32 * max(1, len(values)))(*[int(v) for v in values])
    return arr


class Plan:
    """A compiled tape (``qmle_plan``).  Host-only until the first ``run``.

    ops: list of ``(name, wires, slots, mat_off)``; ``slots`` index columns of the
    per-sample angle table; ``consts`` is the float32 blob for MAT1/MAT2/DIAG_ALL.
    """

    def __init__(self, ops: List[Tuple[str, Sequence[int], Sequence[int], int]], n_qubits: int,
                 n_slots: int, consts: Optional[np.ndarray] = None, flags: int = 0):
        L = lib()
        arr = (QmleOp * max(1, len(ops)))()
        for i, (name, wires, slots, mat_off) in enumerate(ops):
            code = OPCODES.get(name) if isinstance(name, str) else int(name)
            if code is None:
                raise ValueError(f"Unknown gate {name!r}")
            arr[i].opcode = code
            wires = list(wires)
            if len(wires) > 4:
                raise ValueError(f"{name} expects at most 4 wires, got {len(wires)}: {wires}")
            for k in range(4):
                arr[i].wire[k] = int(wires[k]) if k < len(wires) else -1
            for k in range(3):
                arr[i].slot[k] = int(slots[k]) if k < len(slots) else -1
            arr[i].mat_off = int(mat_off)
        if consts is None:
            consts = np.zeros(0, dtype=np.float32)
        self._consts = np.ascontiguousarray(consts, dtype=np.float32)
        handle = C.c_void_p()
        rc = L.qmle_plan_create(
            arr, len(ops), int(n_qubits), int(n_slots),
            self._consts.ctypes.data_as(C.POINTER(C.c_float)), int(self._consts.size),
            C.c_uint(flags), C.byref(handle),
        )
        check(rc, "qmle_plan_create")
        self._h = handle
        self.n_qubits = int(n_qubits)
        self.n_slots = int(n_slots)
        self.n_ops = len(ops)
        self.flags = flags

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and _lib is not None and getattr(self, "_owner", None) is None:
            _lib.qmle_plan_destroy(h)
            self._h = None

    def expval_child(self) -> Optional["Plan"]:
        """The child plan that runs when trailing CX / SWAP / diagonal gates were folded into the Z
        observables (non-owning view; None if nothing was folded)."""
        return self._view(lib().qmle_plan_expval_child(self._h))

    def executed(self, meas: str = "expval") -> "Plan":
        """The plan object ``run(..., meas)`` really executes (``qmle_plan_executed``): the folded
        child for "expval", and of that / of this plan the schedule compiled for runs from |0..0>
        when there is one.  Describe and profile THIS view; it is ``self`` when nothing differs."""
        h = lib().qmle_plan_executed(self._h, MEAS[meas])
        if not h or h == self._h.value:
            return self
        return self._view(h)

    def _view(self, h) -> Optional["Plan"]:
        if not h:
            return None
        child = Plan.__new__(Plan)
        child._h = C.c_void_p(h)
        child._owner = self        # keeps the parent (and with it the handle) alive
        child._consts = self._consts
        child.n_qubits, child.n_slots, child.flags = self.n_qubits, self.n_slots, self.flags
        child.n_ops = child.stats()["n_ops"]
        return child

    def describe(self) -> dict:
        L = lib()
        need = L.qmle_plan_describe(self._h, None, 0)
        buf = C.create_string_buffer(need + 1)
        L.qmle_plan_describe(self._h, buf, need + 1)
        return json.loads(buf.value.decode())

    def stats(self) -> dict:
        arr = (C.c_int64 * 8)()
        check(lib().qmle_plan_stats(self._h, arr), "qmle_plan_stats")
        keys = ["n_ops", "n_passes", "whole_state_lds", "tile_bits", "mat_floats",
                "direct_passes", "n_lowered", "algo_bytes_per_state"]
        return dict(zip(keys, [int(v) for v in arr]))

    def autotune(self, meas: str, n_obs: int = 0, batch: int = 32, top_k: int = 4, reps: int = 3) -> dict:
        """Opt-in (``qmle_plan_autotune``): time the cost model's best ``top_k`` schedules for this
        measurement ("state" / "expval") and batch size on the current GPU and keep the fastest.
        -> ``{"candidate", "padding", "ms_before", "ms_after"}`` (candidate -1: nothing to tune; padding -2:
        not tuned, the scratch allocation failed).  Choices are remembered per executed plan: where "state"
        and "expval" run the same plan the first measurement tuned decides for both."""
        require_gpu()
        chosen = (C.c_int32 * 2)(-1, -1)
        before, after = C.c_double(0.0), C.c_double(0.0)
        check(lib().qmle_plan_autotune(self._h, MEAS[meas], int(n_obs), int(batch), int(top_k), int(reps),
                                       _stream_ptr(), chosen, C.byref(before), C.byref(after)),
              "qmle_plan_autotune")
        self.__dict__.pop("_wsb", None)  # the schedule may have changed: workspace sizes are asked for again
        return {"candidate": int(chosen[0]), "padding": int(chosen[1]), "ms_before": before.value,
                "ms_after": after.value}

    def profile_begin(self, capacity: int) -> None:
        check(lib().qmle_profile_begin(self._h, int(capacity)), "qmle_profile_begin")

    def profile_end(self):
        """-> (ms per pass, launches per pass, pool_overflowed)."""
        # (the engine times the plan it executes -- executed() -- whose pass count may differ from this one's)
        n = max(1, self.stats()["n_passes"], self.executed("state").stats()["n_passes"])
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        rc = lib().qmle_profile_end(self._h, ms, cnt, n)
        if rc < 0:
            check(rc, "qmle_profile_end")
        return list(ms), [int(c) for c in cnt], bool(rc)

    def workspace_bytes(self, batch: int, meas: str, n_obs: int = 0, states_in_flight: int = 0):
        key = (batch, meas, n_obs, states_in_flight)  # (a pure function of the plan: memoised)
        cache = self.__dict__.setdefault("_wsb", {})
        v = cache.get(key)
        if v is None:
            if len(cache) > 256:
                cache.clear()
            v = cache[key] = int(lib().qmle_workspace_bytes(self._h, batch, MEAS[meas], n_obs, states_in_flight))
        return v

    def _workspace(self, B: int, meas: str, n_obs: int, states_in_flight: int, workspace, dev):
        """Workspace tensor of the size the engine asks for.  The default asks for state buffers
        for up to 32 GiB worth of states per launch; when the device cannot spare that, fall
        back to fewer states in flight (the engine adapts to whatever it is handed)."""
        torch = require_gpu()
        need = self.workspace_bytes(B, meas, n_obs, states_in_flight)
        if workspace is not None and workspace.numel() >= need:
            return workspace
        try:
            return torch.empty(need, dtype=torch.uint8, device=dev)
        except torch.OutOfMemoryError:
            if states_in_flight == 1:
                raise
        s = states_in_flight if states_in_flight > 0 else B
        while s > 1:
            s = max(1, s // 4)
            try:
                return torch.empty(self.workspace_bytes(B, meas, n_obs, s), dtype=torch.uint8,
                                   device=dev)
            except torch.OutOfMemoryError:
                if s == 1:
                    raise
        raise torch.OutOfMemoryError("qmle workspace")

    def _out(self, B: int, meas: str, n_obs: int, dev):
        torch = require_gpu()
        D = 1 << self.n_qubits
        if meas == "state":
            return torch.empty((B, D), dtype=torch.complex64, device=dev)
        if meas == "probs":
            return torch.empty((B, D), dtype=torch.float32, device=dev)
        if meas == "expval":
            return torch.empty((B, n_obs), dtype=torch.float32, device=dev)
        if meas == "mw":  # (Q, purity of wire 0 .. n-1) per state: QMLE_MEAS_MEYER_WALLACH
            return torch.empty((B, self.n_qubits + 1), dtype=torch.float32, device=dev)
        return torch.empty((B, D, D), dtype=torch.complex64, device=dev)

    def _tune_once(self, meas: str, n_obs: int, B: int) -> bool:
        """Opt-in autotuner: the plan's first run of this measurement times the candidates."""
        if not (_AUTOTUNE and meas in ("state", "expval")):
            return False
        tuned = self.__dict__.setdefault("_tuned", set())
        if meas in tuned:
            return False
        got = self.autotune(meas, n_obs, batch=B)  # (raises on failure: not remembered, tried again)
        if got["padding"] != -2:  # -2: no scratch memory next to the caller's buffers -- retry next run
            tuned.add(meas)
        return True  # (a workspace sized for the old schedule is void)

    def run(self, angles, meas: str, obs_wires: Sequence[int] = (), out=None, workspace=None,
            states_in_flight: int = 0):
        """simulate_and_measure for a batch.  ``angles``: float32 cuda tensor [B, n_slots]."""
        torch = require_gpu()
        if meas not in MEAS:
            raise ValueError(f"Unknown measurement type: {meas!r}")  # simulation.py:271
        dev = current_device()
        if angles is None:
            angles = torch.zeros((1, max(1, self.n_slots)), dtype=torch.float32, device=dev)
        if angles.dtype != torch.float32 or not angles.is_cuda or not angles.is_contiguous():
            angles = angles.to(device=dev, dtype=torch.float32).contiguous()
        if angles.dim() != 2 or (self.n_slots and angles.shape[1] != self.n_slots):
            raise ValueError(f"angles must be [B, {self.n_slots}], got {tuple(angles.shape)}")
        B = int(angles.shape[0])
        n_obs = len(obs_wires)
        if out is None:
            out = self._out(B, meas, n_obs, dev)
        if self._tune_once(meas, n_obs, B):
            workspace = None
        workspace = self._workspace(B, meas, n_obs, states_in_flight, workspace, dev)
        rc = lib().qmle_run_batch(
            self._h, C.c_void_p(angles.data_ptr()), B, MEAS[meas], _i32(obs_wires), n_obs,
            C.c_void_p(out.data_ptr()), C.c_void_p(workspace.data_ptr()),
            C.c_size_t(workspace.numel()), _stream_ptr(),
        )
        check(rc, "qmle_run_batch")
        return out

    def run_map(self, amap: "AngleMapArgs", B: int, meas: str, obs_wires: Sequence[int] = (), out=None,
                workspace=None):
        """:meth:`run` with the angle table built inside the same native call (qmle_run_batch_map):
        ``amap`` holds the affine map and this call's leaves; the table itself is scratch."""
        torch = require_gpu()
        if meas not in MEAS:
            raise ValueError(f"Unknown measurement type: {meas!r}")  # simulation.py:271
        dev = current_device()
        n_obs = len(obs_wires)
        angles = torch.empty((B, max(1, self.n_slots)), dtype=torch.float32, device=dev)
        if out is None:
            out = self._out(B, meas, n_obs, dev)
        if self._tune_once(meas, n_obs, B):
            workspace = None
        workspace = self._workspace(B, meas, n_obs, 0, workspace, dev)
        rc = lib().qmle_run_batch_map(
            self._h, amap.ref, C.c_void_p(angles.data_ptr()), B, MEAS[meas], _i32(obs_wires), n_obs,
            C.c_void_p(out.data_ptr()), C.c_void_p(workspace.data_ptr()),
            C.c_size_t(workspace.numel()), _stream_ptr(),
        )
        check(rc, "qmle_run_batch_map")
        return out

    def run64(self, angles, meas: str, wire_groups: Sequence[Sequence[int]] = ()):
        """complex128 execution of the plan (the reference's ``jax_enable_x64`` mode):
        ``angles`` float64 [B, n_slots]; returns complex128 states / density matrices, float64
        probabilities, or float64 ``<Z..Z>`` of every wire group [B, len(wire_groups)]."""
        torch = require_gpu()
        if meas not in MEAS:
            raise ValueError(f"Unknown measurement type: {meas!r}")  # simulation.py:271
        dev = current_device()
        if angles is None:
            angles = torch.zeros((1, max(1, self.n_slots)), dtype=torch.float64, device=dev)
        angles = angles.to(device=dev, dtype=torch.float64).contiguous()
        if angles.dim() != 2 or (self.n_slots and angles.shape[1] != self.n_slots):
            raise ValueError(f"angles must be [B, {self.n_slots}], got {tuple(angles.shape)}")
        B, D, n_obs = int(angles.shape[0]), 1 << self.n_qubits, len(wire_groups)
        masks = (C.c_uint32 * max(1, n_obs))()
        for k, g in enumerate(wire_groups):
            m = 0
            for w in g:
                m |= 1 << int(w)
            masks[k] = m
        out = {"state": lambda: torch.empty((B, D), dtype=torch.complex128, device=dev),
               "probs": lambda: torch.empty((B, D), dtype=torch.float64, device=dev),
               "expval": lambda: torch.empty((B, n_obs), dtype=torch.float64, device=dev),
               "density": lambda: torch.empty((B, D, D), dtype=torch.complex128, device=dev)}[meas]()
        wsb = int(lib().qmle_workspace_bytes_f64(self._h, B, MEAS[meas]))
        ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
        check(lib().qmle_run_batch_f64(self._h, C.c_void_p(angles.data_ptr()), B, MEAS[meas], masks, n_obs,
                                       C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                                       C.c_size_t(ws.numel()), _stream_ptr()), "qmle_run_batch_f64")
        return out

    def set_consts64(self, consts) -> None:
        """The batch-constant blob at full precision (complex128 callers with explicit matrices)."""
        a = np.ascontiguousarray(consts, dtype=np.float64).reshape(-1)
        check(lib().qmle_plan_set_consts_f64(self._h, a.ctypes.data_as(C.POINTER(C.c_double)), int(a.size)),
              "qmle_plan_set_consts_f64")

    def run_parity(self, angles, wire_groups: Sequence[Sequence[int]], workspace=None,
                   states_in_flight: int = 0):
        """<Z..Z> over every wire group, measured out of the last pass (no stored state).
        Returns float32 [B, len(wire_groups)]."""
        torch = require_gpu()
        dev = current_device()
        if angles is None:
            angles = torch.zeros((1, max(1, self.n_slots)), dtype=torch.float32, device=dev)
        angles = angles.to(device=dev, dtype=torch.float32).contiguous()
        if angles.dim() != 2 or (self.n_slots and angles.shape[1] != self.n_slots):
            raise ValueError(f"angles must be [B, {self.n_slots}], got {tuple(angles.shape)}")
        B, n_obs = int(angles.shape[0]), len(wire_groups)
        masks = (C.c_uint32 * max(1, n_obs))()
        for k, grp in enumerate(wire_groups):
            m = 0
            for w in grp:
                if not 0 <= int(w) < self.n_qubits:
                    raise ValueError(f"wire {w} out of range for {self.n_qubits} qubits")
                m |= 1 << int(w)
            masks[k] = m
        out = torch.empty((B, n_obs), dtype=torch.float32, device=dev)
        workspace = self._workspace(B, "expval", n_obs, states_in_flight, workspace, dev)
        check(lib().qmle_run_batch_parity(
            self._h, C.c_void_p(angles.data_ptr()), B, masks, n_obs, C.c_void_p(out.data_ptr()),
            C.c_void_p(workspace.data_ptr()), C.c_size_t(workspace.numel()), _stream_ptr()),
            "qmle_run_batch_parity")
        return out


def build_angles(leaves, strides, divs, mods, d_ptr, d_arg, d_idx, d_coef, d_const, n_slots,
                 batch, batch_offset=0, out=None, d_period=None):
    """Angle table [batch, n_slots] on device from device-resident leaf tensors;
    ``d_period`` [n_slots] float64: slots reduced into (-period/2, period/2]."""
    torch = require_gpu()
    k = len(leaves)
    dev = d_const.device
    if out is None:
        out = torch.empty((batch, max(n_slots, 0)), dtype=torch.float32, device=dev)
    if n_slots == 0:
        return out
    lp = (_VP * max(1, k))(*[t.data_ptr() for t in leaves])
    ls = (C.c_int64 * max(1, k))(*[int(v) for v in strides])
    ld = (C.c_int32 * max(1, k))(*[int(v) for v in divs])
    lm = (C.c_int32 * max(1, k))(*[int(v) for v in mods])
    check(lib().qmle_build_angles(lp, ls, ld, lm, k, C.c_void_p(d_ptr.data_ptr()),
                                  C.c_void_p(d_arg.data_ptr()), C.c_void_p(d_idx.data_ptr()),
                                  C.c_void_p(d_coef.data_ptr()), C.c_void_p(d_const.data_ptr()),
                                  C.c_void_p(d_period.data_ptr() if d_period is not None else None),
                                  int(n_slots), int(batch), int(batch_offset),
                                  C.c_void_p(out.data_ptr()), _stream_ptr()), "qmle_build_angles")
    return out


def apply_inplace(plan: Plan, angles, states, workspace=None):
    """Apply ``plan``'s gates in place to resident ``states`` [B, 2^n] (no init)."""
    torch = require_gpu()
    B = int(states.shape[0])
    if angles is None:
        angles = torch.zeros((B, max(1, plan.n_slots)), dtype=torch.float32, device=states.device)
    need = plan.workspace_bytes(B, "state")
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=states.device)
    check(lib().qmle_apply_inplace(plan._h, C.c_void_p(angles.data_ptr()), B,
                                   C.c_void_p(states.data_ptr()), C.c_void_p(workspace.data_ptr()),
                                   C.c_size_t(workspace.numel()), _stream_ptr()),
          "qmle_apply_inplace")
    return states


def apply_inplace64(plan: Plan, angles, states):
    """complex128 counterpart of :func:`apply_inplace` (``qmle_apply_inplace_f64``): ``plan``'s
    operators on resident complex128 ``states`` [B <= 65535, 2^n], float64 ``angles``."""
    torch = require_gpu()
    B = int(states.shape[0])
    if states.dtype != torch.complex128 or not states.is_cuda or not states.is_contiguous():
        raise ValueError("states must be a contiguous complex128 CUDA tensor [B, 2^n]")
    if angles is None:
        angles = torch.zeros((B, max(1, plan.n_slots)), dtype=torch.float64, device=states.device)
    angles = angles.to(dtype=torch.float64).contiguous()
    need = int(lib().qmle_apply_inplace_f64_workspace_bytes(plan._h, B))
    ws = torch.empty(need, dtype=torch.uint8, device=states.device)
    check(lib().qmle_apply_inplace_f64(plan._h, C.c_void_p(angles.data_ptr()), B,
                                       C.c_void_p(states.data_ptr()), C.c_void_p(ws.data_ptr()),
                                       C.c_size_t(need), _stream_ptr()),
          "qmle_apply_inplace_f64")
    return states


# ---- stand-alone measurement / analysis kernels -------------------------------------
MAX_ROWS = 65535  # one launch covers at most this many samples (grid.y); the wrappers below cut longer batches


def _row_chunked(*tensor_args: int, max_rows: int = MAX_ROWS):
    """Run the wrapped call on slices of at most ``max_rows`` rows of the positional tensor
    arguments ``tensor_args`` and concatenate the results (a tensor or a tuple of tensors / None)."""
    def deco(fn):
        @functools.wraps(fn)
        def run(*args, **kwargs):
            rows = int(args[tensor_args[0]].shape[0]) if args[tensor_args[0]].dim() > 1 else 1
            if rows <= max_rows:
                return fn(*args, **kwargs)
            torch = require_gpu()
            parts = []
            for lo in range(0, rows, max_rows):
                a = list(args)
                for i in tensor_args:
                    a[i] = args[i][lo:lo + max_rows].contiguous()
                parts.append(fn(*a, **kwargs))
            if isinstance(parts[0], tuple):
                return tuple(None if p0 is None else torch.cat([p[k] for p in parts], dim=0)
                             for k, p0 in enumerate(parts[0]))
            return torch.cat(parts, dim=0)
        return run
    return deco


def _states_info(states):
    torch = require_gpu()
    if states.dtype == torch.complex128 and states.is_cuda:
        # complex128 states (utils.enable_x64): the analysis kernels (fidelities, Meyer-Wallach,
        # marginals, ...) are complex64 -- their results carry float32 accuracy either way
        states = states.to(torch.complex64)
    if states.dtype != torch.complex64 or not states.is_cuda:
        raise ValueError("states must be a complex64 CUDA tensor [B, 2^n]")
    states = states.contiguous()
    if states.dim() == 1:
        states = states.unsqueeze(0)
    B, D = int(states.shape[0]), int(states.shape[1])
    n = D.bit_length() - 1
    if 1 << n != D:
        raise ValueError(f"state length {D} is not a power of two")
    return torch, states, B, n


@_row_chunked(0)
def expval_z(states, obs_wires: Sequence[int]):
    torch, states, B, n = _states_info(states)
    out = torch.empty((B, len(obs_wires)), dtype=torch.float32, device=states.device)
    wsb = int(lib().qmle_expval_workspace_bytes(n, B))
    ws = torch.empty(wsb, dtype=torch.uint8, device=states.device)
    check(lib().qmle_expval_z(C.c_void_p(states.data_ptr()), n, B, _i32(obs_wires), len(obs_wires),
                              C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), wsb,
                              _stream_ptr()), "qmle_expval_z")
    return out


def probs(states):
    torch, states, B, n = _states_info(states)
    out = torch.empty((B, 1 << n), dtype=torch.float32, device=states.device)
    check(lib().qmle_probs(C.c_void_p(states.data_ptr()), n, B, C.c_void_p(out.data_ptr()),
                           _stream_ptr()), "qmle_probs")
    return out


@_row_chunked(0)
def density(states):
    torch, states, B, n = _states_info(states)
    out = torch.empty((B, 1 << n, 1 << n), dtype=torch.complex64, device=states.device)
    check(lib().qmle_density(C.c_void_p(states.data_ptr()), n, B, C.c_void_p(out.data_ptr()),
                             _stream_ptr()), "qmle_density")
    return out


@_row_chunked(0)
def marginal_probs(states, keep: Sequence[int]):
    torch, states, B, n = _states_info(states)
    out = torch.empty((B, 1 << len(keep)), dtype=torch.float32, device=states.device)
    check(lib().qmle_marginal_probs(C.c_void_p(states.data_ptr()), n, B, _i32(keep), len(keep),
                                    C.c_void_p(out.data_ptr()), _stream_ptr()),
          "qmle_marginal_probs")
    return out


def pair_fidelity(states):
    """|<psi_i|psi_{i+S}>|^2 for states [2S, 2^n] -> [S]."""
    torch, states, B, n = _states_info(states)
    if B % 2:
        raise ValueError("pair_fidelity needs an even number of states")
    S = B // 2
    if S > MAX_ROWS:  # pairs (i, i + S): cut the pair index range, both halves per slice
        return torch.cat([pair_fidelity(torch.cat([states[lo:min(S, lo + MAX_ROWS)],
                                                   states[S + lo:S + min(S, lo + MAX_ROWS)]]))
                          for lo in range(0, S, MAX_ROWS)])
    out = torch.empty((S,), dtype=torch.float32, device=states.device)
    wsb = int(lib().qmle_pair_fidelity_workspace_bytes(n, S))
    ws = torch.empty(wsb, dtype=torch.uint8, device=states.device)
    check(lib().qmle_pair_fidelity(C.c_void_p(states.data_ptr()), n, S, C.c_void_p(out.data_ptr()),
                                   C.c_void_p(ws.data_ptr()), wsb, _stream_ptr()),
          "qmle_pair_fidelity")
    return out


@_row_chunked(0)
def density_probs(rho_vec, n_qubits: int):
    """diag(rho) of vectorised density matrices [B, 4^n] -> float32 [B, 2^n]."""
    torch = require_gpu()
    rho_vec = rho_vec.contiguous()
    B = int(rho_vec.shape[0])
    out = torch.empty((B, 1 << n_qubits), dtype=torch.float32, device=rho_vec.device)
    check(lib().qmle_density_probs(C.c_void_p(rho_vec.data_ptr()), n_qubits, B,
                                   C.c_void_p(out.data_ptr()), _stream_ptr()), "qmle_density_probs")
    return out


@_row_chunked(0)
def density_expval_z(rho_vec, n_qubits: int, obs_wires: Sequence[int]):
    torch = require_gpu()
    rho_vec = rho_vec.contiguous()
    B = int(rho_vec.shape[0])
    out = torch.empty((B, len(obs_wires)), dtype=torch.float32, device=rho_vec.device)
    check(lib().qmle_density_expval_z(C.c_void_p(rho_vec.data_ptr()), n_qubits, B,
                                      _i32(obs_wires), len(obs_wires),
                                      C.c_void_p(out.data_ptr()), _stream_ptr()),
          "qmle_density_expval_z")
    return out


@_row_chunked(0, 1)
def overlap(a, b):
    """<a_i|b_i> for two [B, 2^n] complex64 tensors -> complex64 [B]."""
    torch, a, B, n = _states_info(a)
    _, b, Bb, nb = _states_info(b)
    if (B, n) != (Bb, nb):
        raise ValueError("overlap: shape mismatch")
    out = torch.empty((B,), dtype=torch.complex64, device=a.device)
    wsb = int(lib().qmle_overlap_workspace_bytes(n, B))
    ws = torch.empty(wsb, dtype=torch.uint8, device=a.device)
    check(lib().qmle_overlap(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n, B,
                             C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), wsb,
                             _stream_ptr()), "qmle_overlap")
    return out


@_row_chunked(0)
def expval_parity(states, wire_groups):
    """<Z..Z> on each group of wires -> float32 [B, len(groups)]."""
    torch, states, B, n = _states_info(states)
    masks = []
    for g in wire_groups:
        m = 0
        for w in g:
            if not 0 <= int(w) < n:
                raise ValueError(f"wire {w} out of range for {n} qubits")
            m |= 1 << int(w)
        masks.append(m)
    arr = (C.c_uint32 * max(1, len(masks)))(*masks)
    out = torch.empty((B, len(masks)), dtype=torch.float32, device=states.device)
    wsb = int(lib().qmle_expval_parity_workspace_bytes(n, B))
    ws = torch.empty(wsb, dtype=torch.uint8, device=states.device)
    check(lib().qmle_expval_parity(C.c_void_p(states.data_ptr()), n, B, arr, len(masks),
                                   C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), wsb,
                                   _stream_ptr()), "qmle_expval_parity")
    return out


@_row_chunked(0)
def meyer_wallach(states, return_purities: bool = False):
    torch, states, B, n = _states_info(states)
    out = torch.empty((B,), dtype=torch.float32, device=states.device)
    pur = torch.empty((B, n), dtype=torch.float32, device=states.device) if return_purities else None
    wsb = int(lib().qmle_meyer_wallach_workspace_bytes(n, B))
    ws = torch.empty(wsb, dtype=torch.uint8, device=states.device)
    check(lib().qmle_meyer_wallach(C.c_void_p(states.data_ptr()), n, B, C.c_void_p(out.data_ptr()),
                                   C.c_void_p(pur.data_ptr()) if pur is not None else None,
                                   C.c_void_p(ws.data_ptr()), wsb, _stream_ptr()),
          "qmle_meyer_wallach")
    return (out, pur) if return_purities else out


def mw_reads(n_qubits: int) -> int:
    """HBM reads of the state per ``meyer_wallach`` call at this size (host only)."""
    return int(lib().qmle_meyer_wallach_reads(int(n_qubits)))


def philox_uniform(key, count: int, low: float, high: float) -> np.ndarray:
    """``numpy.random.Generator(numpy.random.Philox(key=key)).uniform(low, high, count)
    .astype(float32)``, bit for bit, from the library's host-side generator (no GPU involved)."""
    key = np.ascontiguousarray(key, dtype=np.uint64)
    if key.shape != (2,):
        raise ValueError("Philox4x64 takes a key of two 64-bit words")
    out = np.empty(int(count), dtype=np.float32)
    check(lib().qmle_philox_uniform_f32(key.ctypes.data, int(count), float(low), float(high),
                                          out.ctypes.data))
    return out


def philox_uniform_device(key, count: int, low: float, high: float):
    """:func:`philox_uniform` written by the GPU: float32 CUDA tensor [count], the same bits."""
    torch = require_gpu()
    key = np.ascontiguousarray(key, dtype=np.uint64)
    if key.shape != (2,):
        raise ValueError("Philox4x64 takes a key of two 64-bit words")
    out = torch.empty((int(count),), dtype=torch.float32, device=current_device())
    check(lib().qmle_philox_uniform_f32_device(key.ctypes.data, int(count), float(low), float(high),
                                               C.c_void_p(out.data_ptr()), _stream_ptr()),
          "qmle_philox_uniform_f32_device")
    return out


def histogram(values, n_bins: int, lo: float = 0.0, hi: float = 1.0):
    torch = require_gpu()
    values = values.to(dtype=torch.float32).contiguous().reshape(-1)
    counts = torch.empty((n_bins,), dtype=torch.int32, device=values.device)
    check(lib().qmle_histogram(C.c_void_p(values.data_ptr()), values.numel(), n_bins,
                               C.c_float(lo), C.c_float(hi), C.c_void_p(counts.data_ptr()),
                               _stream_ptr()), "qmle_histogram")
    return counts


def sample_counts(probs, shots: int, seed: int, row_offset: int = 0, want_probs: bool = True):
    """``shots`` inverse-CDF draws per row of ``probs`` [B, 2^n] (float32, CUDA) ->
    ``(counts int32 [B, 2^n], estimated probs float32 [B, 2^n] or None)``."""
    torch = require_gpu()
    if probs.dtype != torch.float32 or not probs.is_cuda or probs.dim() != 2:
        raise ValueError("probs must be a float32 CUDA tensor [B, 2^n]")
    probs = probs.contiguous()
    B, D = int(probs.shape[0]), int(probs.shape[1])
    n = D.bit_length() - 1
    if (1 << n) != D:
        raise ValueError(f"row length {D} is not a power of two")
    counts = torch.empty((B, D), dtype=torch.int32, device=probs.device)
    est = torch.empty((B, D), dtype=torch.float32, device=probs.device) if want_probs else None
    for b0 in range(0, B, 65535):
        bc = min(65535, B - b0)
        ws = torch.empty(lib().qmle_sample_workspace_bytes(n, bc), dtype=torch.uint8,
                         device=probs.device)
        check(lib().qmle_sample_counts(
            C.c_void_p(probs[b0:].data_ptr()), n, bc, int(shots),
            C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_uint64(int(row_offset) + b0),
            C.c_void_p(counts[b0:].data_ptr()),
            C.c_void_p(est[b0:].data_ptr()) if est is not None else None,
            C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), _stream_ptr()),
            "qmle_sample_counts")
    return counts, est


def probs_diag_expval(probs, obs: Sequence[Tuple[Sequence[int], Optional[Sequence[float]]]]):
    """``sum_i p[b, i] * diag(O_k)[i]`` for observables given as ``(wires, diagonal)``;
    ``diagonal=None`` means the Z-parity over ``wires``.  Returns float32 [B, n_obs]."""
    torch = require_gpu()
    probs = probs.contiguous()
    B, D = int(probs.shape[0]), int(probs.shape[1])
    n = D.bit_length() - 1
    wires, counts, offs, table = [], [], [], []
    for w, d in obs:
        wires += list(w)
        counts.append(len(w))
        if d is None:
            offs.append(-1)
        else:
            d = np.asarray(d, dtype=np.float32).reshape(-1)
            if d.size != 2 ** len(w):
                raise ValueError(f"diagonal has {d.size} entries for {len(w)} wire(s)")
            offs.append(sum(t.size for t in table))
            table.append(d)
    d_table = (torch.from_numpy(np.concatenate(table)).to(probs.device) if table else None)
    out = torch.empty((B, len(obs)), dtype=torch.float32, device=probs.device)
    ws = torch.empty(max(1, lib().qmle_probs_diag_expval_workspace_bytes(len(obs))),
                     dtype=torch.uint8, device=probs.device)
    for b0 in range(0, B, 65535):
        bc = min(65535, B - b0)
        check(lib().qmle_probs_diag_expval(
            C.c_void_p(probs[b0:].data_ptr()), n, bc, _i32(wires), _i32(counts), _i32(offs),
            C.c_void_p(d_table.data_ptr()) if d_table is not None else None, len(obs),
            C.c_void_p(out[b0:].data_ptr()), C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()),
            _stream_ptr()), "qmle_probs_diag_expval")
    return out


class AdjointTerm(C.Structure):
    _fields_ = [("out_slot", C.c_int32), ("x_wires", C.c_uint32), ("z_wires", C.c_uint32),
                ("proj_wires", C.c_uint32), ("n_y", C.c_int32), ("coef", C.c_float),
                ("marks_off", C.c_int32)]


@_row_chunked(2, 3, 4, max_rows=32767)  # (psi and lambda of a sample share a launch: 2 B <= 65535)
def adjoint_gradient(fwd: Plan, rev: Plan, angles_fwd, angles_rev, weights,
                     wire_groups: Sequence[Sequence[int]], terms, n_grad_slots: int):
    """One backward sweep: d/d(angle) of sum_k weights[b, k] <Z..Z>_k -> float32 [B, n_grad_slots]
    (float64 tensors in: the complex128 sweep, ``qmle_adjoint_gradient_f64``, float64 out).
    ``terms``: one ``(out_slot, x_wires, z_wires, proj_wires, n_y, coef, marks_off)`` per op of
    ``rev`` (the reversed, daggered NO_FUSION plan)."""
    torch = require_gpu()
    f64 = weights.dtype == torch.float64
    if f64 and (angles_fwd.dtype != torch.float64 or angles_rev.dtype != torch.float64):
        raise ValueError("the complex128 sweep takes float64 angle tables")
    B, n_obs = int(weights.shape[0]), len(wire_groups)
    if weights.dim() != 2 or weights.shape[1] != n_obs:
        raise ValueError(f"weights must be [B, {n_obs}], got {tuple(weights.shape)}")
    masks = (C.c_uint32 * max(1, n_obs))()
    for k, grp in enumerate(wire_groups):
        m = 0
        for wq in grp:
            if not 0 <= int(wq) < fwd.n_qubits:
                raise ValueError(f"wire {wq} out of range for {fwd.n_qubits} qubits")
            m |= 1 << int(wq)
        masks[k] = m
    arr = (AdjointTerm * max(1, len(terms)))()
    for i, t in enumerate(terms):
        (arr[i].out_slot, arr[i].x_wires, arr[i].z_wires, arr[i].proj_wires, arr[i].n_y,
         arr[i].coef, arr[i].marks_off) = t
    dev = weights.device
    if f64:
        out = torch.empty((B, n_grad_slots), dtype=torch.float64, device=dev)
        ws = torch.empty(lib().qmle_adjoint_workspace_bytes_f64(fwd._h, rev._h, B), dtype=torch.uint8, device=dev)
        check(lib().qmle_adjoint_gradient_f64(
            fwd._h, rev._h, C.c_void_p(angles_fwd.data_ptr()), C.c_void_p(angles_rev.data_ptr()), B,
            C.c_void_p(weights.data_ptr()), masks, n_obs, arr, len(terms),
            C.c_void_p(out.data_ptr()), int(n_grad_slots), C.c_void_p(ws.data_ptr()),
            C.c_size_t(ws.numel()), _stream_ptr()), "qmle_adjoint_gradient_f64")
        return out
    out = torch.empty((B, n_grad_slots), dtype=torch.float32, device=dev)
    ws = torch.empty(lib().qmle_adjoint_workspace_bytes(fwd._h, rev._h, B), dtype=torch.uint8,
                     device=dev)
    check(lib().qmle_adjoint_gradient(
        fwd._h, rev._h, C.c_void_p(angles_fwd.data_ptr()), C.c_void_p(angles_rev.data_ptr()), B,
        C.c_void_p(weights.data_ptr()), masks, n_obs, arr, len(terms),
        C.c_void_p(out.data_ptr()), int(n_grad_slots), C.c_void_p(ws.data_ptr()),
        C.c_size_t(ws.numel()), _stream_ptr()), "qmle_adjoint_gradient")
    return out
