"""ctypes front of ``oracle/sv_cpu.c`` (ORACLE / CPU baseline -- test infra only).

Runs an oracle tape ``[(name, wires, params)]`` with the C/OpenMP pair-update
kernels; gate matrices come from ``oracle/gates.py``.  Used by the ``cpu_baseline``
leg of ``bench.py`` and validated against ``einsum_sim`` in the CPU tests.
"""
import ctypes as C
import os

import numpy as np

from . import gates as G

_LIB = None
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsv_cpu.so")


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run __graft_entry__.build()")
        h = C.CDLL(LIB_PATH)
        h.svc_max_threads.restype = C.c_int
        _LIB = h
    return _LIB


def _fm(mat):
    m = np.asarray(mat, dtype=np.complex64)
    return np.ascontiguousarray(np.stack([m.real, m.imag], axis=-1).reshape(-1), dtype=np.float32)


def simulate(tape, n_qubits, threads=None):
    L = lib()
    if threads:
        L.svc_set_threads(int(threads))
    psi = np.empty(2**n_qubits, dtype=np.complex64)
    pp = psi.ctypes.data_as(C.c_void_p)
    L.svc_init_zero(pp, n_qubits)
    for name, wires, params in tape:
        if name == "Barrier":
            continue
        k = len(wires)
        if k == 1:
            fm = _fm(G.matrix(name, params))
            L.svc_apply_1q(pp, n_qubits, int(wires[0]), fm.ctypes.data_as(C.c_void_p))
        elif k == 2 and name in ("CX", "CY", "CZ", "CRX", "CRY", "CRZ", "CPhase"):
            fm = _fm(G.matrix(name, params)[2:, 2:])
            L.svc_apply_c1q(pp, n_qubits, int(wires[0]), int(wires[1]), fm.ctypes.data_as(C.c_void_p))
        elif k == 2:
            fm = _fm(G.matrix(name, params))
            L.svc_apply_2q(pp, n_qubits, int(wires[0]), int(wires[1]), fm.ctypes.data_as(C.c_void_p))
        else:
            raise NotImplementedError(f"C port: {k}-qubit gate {name}")
    return psi


def expval_z(psi, n_qubits, wires):
    w = np.asarray(wires, dtype=np.int32)
    out = np.empty(len(w), dtype=np.float32)
    lib().svc_expval_z(psi.ctypes.data_as(C.c_void_p), n_qubits, w.ctypes.data_as(C.c_void_p),
                       len(w), out.ctypes.data_as(C.c_void_p))
    return out
