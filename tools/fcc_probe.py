import sys, numpy as np
sys.path.insert(0, '/root/repo')
from qml_essentials_amd.coefficients import FCC, Coefficients
from qml_essentials_amd.model import Model
from tests.test_gpu_fcc import _oracle_spectrum
for ct in ("Hardware_Efficient", "Circuit_17"):
    for x64 in (False, True):
        model = Model(n_qubits=6, n_layers=1, circuit_type=ct, output_qubit=-1, encoding=["RY"], x64=x64)
        fcc = FCC.get_fcc(model=model, n_samples=500, scale=True)
        print(ct, 'x64' if x64 else 'c64', 'engine fcc', fcc, flush=True)
    P = np.asarray(model.params, dtype=np.float64)
    coeffs, freqs = _oracle_spectrum(ct, 6, P)
    keep = FCC._calculate_mask(freqs)
    want = FCC._correlate(coeffs[keep].T)
    low = np.tril(np.ones(want.shape, dtype=bool), k=-1)
    print(ct, 'oracle fcc on the same', P.shape[0], 'params:', np.abs(want[low]).mean(), 'max |coeff| per freq', np.abs(coeffs).max(axis=1))
    # engine spectrum x64 on same params
    model.params = P
    c2, f2 = Coefficients.get_spectrum(model, shift=True, force_mean=True, execution_type="expval")
    print('   engine x64 spectrum vs oracle: max abs diff', np.abs(c2 - coeffs).max(), 'per freq', np.abs(c2 - coeffs).max(axis=1))
