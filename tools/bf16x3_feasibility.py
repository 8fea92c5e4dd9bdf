#!/usr/bin/env python3
"""Numerics check for the matrix-core idea: apply chains of random 16x16 unitaries
to a complex64 vector with the operator and the amplitudes split into bf16 pieces and the
products accumulated in fp32 (what v_mfma_f32_32x32x16_bf16 does), for 3 / 6 / 9 product terms,
against float64.  Host NumPy only."""
import numpy as np

def bf16(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).astype(np.uint32).view(np.float32)

def split(x, k):
    parts, rest = [], np.asarray(x, dtype=np.float32)
    for _ in range(k):
        p = bf16(rest); parts.append(p); rest = (rest - p).astype(np.float32)
    return parts

def matmul_terms(A, X, pairs):
    a, x = split(A, 3), split(X, 3)
    acc = np.zeros((A.shape[0], X.shape[1]), dtype=np.float32)
    for i, j in pairs:
        acc = (acc + (a[i].astype(np.float32) @ x[j].astype(np.float32)).astype(np.float32)).astype(np.float32)
    return acc

TERMS = {3: [(0, 0), (0, 1), (1, 0)],
         6: [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)],
         9: [(i, j) for i in range(3) for j in range(3)]}
rng = np.random.default_rng(0)
N, chain = 4096, 60
psi = rng.normal(size=(16, N)) + 1j * rng.normal(size=(16, N)); psi /= np.linalg.norm(psi)
ref = psi.astype(np.complex128)
cur = {k: psi.astype(np.complex64) for k in TERMS}
f32 = psi.astype(np.complex64)
for step in range(chain):
    q, _ = np.linalg.qr(rng.normal(size=(16, 16)) + 1j * rng.normal(size=(16, 16)))
    ref = q @ ref
    f32 = (q.astype(np.complex64) @ f32).astype(np.complex64)
    Ar = np.block([[q.real, -q.imag], [q.imag, q.real]]).astype(np.float32)     # real 32x32
    for k, pairs in TERMS.items():
        X = np.concatenate([cur[k].real, cur[k].imag]).astype(np.float32)       # 32 x N
        Y = matmul_terms(Ar, X, pairs)
        cur[k] = (Y[:16] + 1j * Y[16:]).astype(np.complex64)
    if step + 1 in (1, 10, 30, 60):
        scale = np.abs(ref).max()
        print(f"after {step + 1:2d} operators: fp32 vector {np.abs(f32 - ref).max() / scale:.2e}",
              " ".join(f"bf16x{k} {np.abs(cur[k] - ref).max() / scale:.2e}" for k in TERMS))
