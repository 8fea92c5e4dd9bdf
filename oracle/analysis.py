"""Analysis loops of the reference, restated in NumPy/SciPy (oracle; test infra only).

Two flavours per quantity:
  * ``*_reference_form``: the formula exactly as the reference computes it (full
    density matrices, ``scipy.linalg.sqrtm``, partial traces) -- feasible only at
    small n;
  * ``*_pure``: the mathematically equal pure-state formula the HIP path
    implements.
Tests assert both flavours agree at small n, which is what licenses the pure-state
form at the BASELINE sizes.
"""
import math

import numpy as np
from scipy import integrate
from scipy.linalg import sqrtm
from scipy.special import rel_entr


# --- Expressibility (expressibility.py) -------------------------------------
def fidelities_reference_form(rhos, n_samples):
    """expressibility.py:48-64: F_i = |Tr sqrtm(sqrt(rho_i) sigma_i sqrt(rho_i))|^2,
    sample i paired with sample i + n_samples."""
    sqrt1 = np.array([sqrtm(m) for m in rhos[:n_samples]])
    inner = sqrt1 @ rhos[n_samples:] @ sqrt1
    f = np.trace(np.array([sqrtm(m) for m in inner]), axis1=1, axis2=2) ** 2
    return np.abs(f)


def fidelities_pure(states, n_samples):
    """|<psi_i|psi_{i+S}>|^2 (math.py:60-86 without the renormalisation)."""
    a, b = states[:n_samples], states[n_samples:]
    return np.abs(np.einsum("ab,ab->a", np.conj(a), b)) ** 2


def fidelity_histogram(fidelities, n_bins, n_samples):
    """expressibility.py:104-110."""
    y = np.linspace(0, 1, n_bins + 1)
    z, _ = np.histogram(fidelities, bins=y)
    return y, z / n_samples


def haar_probability(fid, n_qubits):
    """expressibility.py:115-131."""
    N = 2**n_qubits
    return (N - 1) * (1 - fid) ** (N - 2)


def haar_integral(n_qubits, n_bins):
    """expressibility.py:133-152 (numeric quad per bin)."""
    dist = np.zeros(n_bins)
    for idx in range(n_bins):
        dist[idx], _ = integrate.quad(
            haar_probability, idx / n_bins, (idx + 1) / n_bins, args=(n_qubits,)
        )
    return dist


def kl_divergence(p, haar):
    """expressibility.py:205-235."""
    p = np.asarray(p)
    if p.ndim == 1:
        p = p.reshape(1, -1)
    return np.array([np.sum(rel_entr(row, haar)) for row in p])


# --- Entanglement (entanglement.py / jaqsi.py) -------------------------------
def partial_trace(rho, n_qubits, keep):
    """jaqsi.py:60-76: trace out every qubit not in ``keep``; kept wires stay in
    ascending order."""
    rho_t = rho.reshape((2,) * (2 * n_qubits))
    trace_out = sorted(set(range(n_qubits)) - set(keep))
    for q in reversed(trace_out):
        n_rem = rho_t.ndim // 2
        rho_t = np.trace(rho_t, axis1=q, axis2=q + n_rem)
    d = 2 ** len(keep)
    return rho_t.reshape(d, d)


def meyer_wallach_reference_form(rho, n_qubits):
    """entanglement.py:86-101 for one density matrix."""
    qb = list(range(n_qubits))
    entropy = 0.0
    for j in range(n_qubits):
        keep = qb[:j] + qb[j + 1:]
        d = partial_trace(rho, n_qubits, keep)
        entropy += np.trace((d @ d).real)
    return 2 * (1 - entropy / n_qubits)


def qubit_purities_pure(state, n_qubits):
    """Tr rho_j^2 = a^2 + d^2 + 2|c|^2 for every wire j."""
    psi = state.reshape((2,) * n_qubits)
    out = np.zeros(n_qubits)
    for j in range(n_qubits):
        m = np.moveaxis(psi, j, 0).reshape(2, -1)
        a = np.sum(np.abs(m[0]) ** 2)
        d = np.sum(np.abs(m[1]) ** 2)
        c = np.sum(m[0] * np.conj(m[1]))
        out[j] = a * a + d * d + 2 * np.abs(c) ** 2
    return out


def meyer_wallach_pure(state, n_qubits):
    return 2 * (1 - np.sum(qubit_purities_pure(state, n_qubits)) / n_qubits)


def marginalize_probs(probs, n_qubits, keep):
    """jaqsi.py:106-146 (batched or not); kept wires in ascending order."""
    p = np.asarray(probs).reshape(-1, 2**n_qubits)
    trace_out = tuple(q for q in range(n_qubits - 1, -1, -1) if q not in keep)
    out = []
    for row in p:
        t = row.reshape((2,) * n_qubits)
        for q in trace_out:
            t = t.sum(axis=q)
        out.append(t.ravel())
    return np.array(out)


# --- Coefficients (coefficients.py:109-150) ----------------------------------
def fourier_grid(degree, mfs=1, mts=1):
    """coefficients.py:114-127: per-feature sample points and the flattened grid."""
    n_freqs = [mfs * d for d in degree]
    axes = [np.arange(0, 2 * mts * np.pi, 2 * np.pi / nf) for nf in n_freqs]
    grid = np.array(np.meshgrid(*axes)).T.reshape(-1, len(degree))
    return axes, grid, n_freqs


def fourier_transform(outputs, axes, n_freqs, mts=1):
    """coefficients.py:130-150: fftn over the feature axes / prod(N_i); fftfreq."""
    F = len(axes)
    out = np.asarray(outputs).reshape(*[a.shape[0] for a in axes], -1).squeeze()
    coeffs = np.fft.fftn(out, axes=list(range(F)))
    freqs = [np.fft.fftfreq(int(mts * n_freqs[i]), 1 / n_freqs[i]) for i in range(F)]
    return coeffs / math.prod(out.shape[0:F]), freqs
