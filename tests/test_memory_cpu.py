"""HBM model + batch chunking (``memory.py``): the properties the reference asserts for its host
RAM model in ``tests/test_jaqsi.py:1711-1816`` that carry over to the engine's HBM model."""
import numpy as np
import pytest

from qml_essentials_amd import memory


@pytest.fixture(autouse=True)
def _free_hbm(monkeypatch):
    monkeypatch.setattr(memory, "available_memory_bytes", lambda: 200 * 1024**3)


def test_estimate_peak_bytes_basic_and_scaling():
    e1 = memory.estimate_peak_bytes(5, 1, "state", False)
    e100 = memory.estimate_peak_bytes(5, 100, "state", False)
    assert 0 < e1 < e100 <= e1 * 200
    assert memory.estimate_peak_bytes(5, 100, "density", True) > e100
    assert memory.estimate_peak_bytes(19, 10, "state", False) > \
        memory.estimate_peak_bytes(14, 10, "state", False) * 10
    # a noisy tape materialises vec(rho) per sample even when only probabilities come back
    assert memory.estimate_peak_bytes(8, 5, "probs", True) > memory.estimate_peak_bytes(8, 5, "probs", False)


def test_compute_chunk_size():
    assert memory.compute_chunk_size(2, 10, "state", False) == 10
    c = memory.compute_chunk_size(n_qubits=30, batch_size=1000000, type="density",
                                  use_density=True, n_obs=0)
    assert 1 <= c < 1000000
    assert memory.compute_chunk_size(n_qubits=30, batch_size=100, type="density", use_density=True,
                                     n_obs=0, memory_fraction=0.00001) >= 1
    # expval at <= 14 qubits never leaves LDS: a huge batch still fits
    assert memory.compute_chunk_size(6, 209498, "expval", False, 6, n_ops=30) == 209498
    # ... while statevector outputs of the same batch at 24 qubits do not (128 MiB each)
    assert memory.compute_chunk_size(24, 4096, "state", False) < 4096


def test_execute_chunked_concatenates_uneven_chunks():
    calls = []

    def run(s, e):
        calls.append((s, e))
        return np.arange(s, e, dtype=np.float32)[:, None] * np.ones((1, 3), dtype=np.float32)

    out = memory.execute_chunked(run, 7, 3)          # chunks of 3, 3, 1
    assert calls == [(0, 3), (3, 6), (6, 7)] and out.shape == (7, 3)
    assert np.array_equal(out[:, 0], np.arange(7))


def test_complex128_mode_doubles_the_model(monkeypatch):
    """in x64 mode the engine holds 16-byte amplitudes, float64 matrix rows and (above 13
    qubits) 4 GiB of states in flight; observables it does not measure itself keep every state."""
    for meas in ("state", "probs", "density"):
        a = memory.estimate_peak_bytes(20, 8, meas, False, 4, n_ops=50)
        b = memory.estimate_peak_bytes(20, 8, meas, False, 4, n_ops=50, x64=True)
        # ("state": complex64 computes in place in the output, complex128 in its own buffers on top)
        lo, hi = (3.9, 4.1) if meas == "state" else (1.9, 2.1)
        assert lo < b / a < hi, (meas, a, b)
    # expval at 14 qubits: LDS-resident in complex64, streaming (states in HBM) in complex128
    assert memory.estimate_peak_bytes(14, 1000, "expval", False, 14, x64=True) > \
        100 * memory.estimate_peak_bytes(14, 1000, "expval", False, 14)
    g = memory.estimate_peak_bytes(12, 64, "expval", False, 2, x64=True, general_obs=True)
    assert g > 64 * (1 << 12) * 16 * 2
    monkeypatch.setattr(memory, "available_memory_bytes", lambda: 1 << 30)
    c32 = memory.compute_chunk_size(20, 512, "state", False)
    c64 = memory.compute_chunk_size(20, 512, "state", False, x64=True)
    assert 1 <= c64 <= c32 // 2 + 4 and c32 < 512  # (the constant megabyte of slack is not doubled)


def test_complex128_states_in_flight_for_every_measurement_type():
    """above 13 qubits the complex128 engine (qmle_workspace_bytes_f64) works in its own
    round of state buffers for every measurement type -- "state" included -- and at <= 13 qubits a
    density batch holds psi of every sample next to the output."""
    s20 = 16 << 20
    out = 8 * s20
    e = memory.estimate_peak_bytes(20, 8, "state", False, x64=True)
    assert e >= out + 8 * s20  # output + the round's state buffers
    # a large batch: 4 GiB of states in flight, not the whole batch
    big = memory.estimate_peak_bytes(20, 4096, "state", False, x64=True)
    assert 4096 * s20 + (4 << 30) <= big <= 1.1 * (4096 * s20 + (4 << 30) + 4096 * 64) + (2 << 20)
    s10 = 16 << 10
    d = memory.estimate_peak_bytes(10, 32, "density", False, x64=True)
    assert d >= 32 * s10 * 1024 + 32 * s10


def test_batches_of_several_chunks_count_two_sets_of_state_buffers():
    """Round 5: a batch larger than one chunk of state buffers keeps two chunks in flight (two internal streams,
    qmle_engine.hip), so the engine's workspace holds two sets of buffers -- the model must too, or a batch sized
    to just fit would be sent whole and fail inside the call."""
    from qml_essentials_amd import memory as M

    n = 24
    state = (1 << n) * 8
    per_chunk = M.IN_FLIGHT_TARGET_BYTES // state
    one = M.estimate_peak_bytes(n, per_chunk, "expval", n_obs=n)
    two = M.estimate_peak_bytes(n, per_chunk + 1, "expval", n_obs=n)
    assert two - one > 0.9 * per_chunk * state          # the second slot
    assert M.estimate_peak_bytes(n, 4 * per_chunk, "expval", n_obs=n) - two < 0.1 * per_chunk * state  # and no third
