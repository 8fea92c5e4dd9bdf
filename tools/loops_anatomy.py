#!/usr/bin/env python3
"""Where the wall-clock of the three sampling loops goes (BASELINE configs 3 / 4 and a sampled
Meyer-Wallach): wall per call, GPU time per call (HIP events on the launch stream), Python calls
per invocation and the functions that own the host time.

    python tools/loops_anatomy.py [c3] [c4] [mw] [--profile]
"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd.coefficients import Coefficients  # noqa: E402
from qml_essentials_amd.entanglement import Entanglement  # noqa: E402
from qml_essentials_amd.expressibility import Expressibility  # noqa: E402
from qml_essentials_amd.model import Model  # noqa: E402


def legs():
    m3 = Model(12, 3, "Hardware_Efficient", data_reupload=False)
    m4 = Model(10, 6, "Hardware_Efficient")
    x4 = torch.from_numpy((2 * np.pi * np.arange(4096) / 4096).astype(np.float32).reshape(-1, 1)).cuda()
    m5 = Model(12, 3, "Hardware_Efficient", data_reupload=False)

    def c4():
        y = m4(inputs=x4, force_mean=True)
        return Coefficients._fft_real(y.cpu().numpy().astype(np.float64))

    return {
        "c3": lambda: Expressibility.kl_divergence_to_haar(m3, n_samples=1024, n_bins=75, random_key=1000),
        "c4": c4,
        "c4_api": lambda: Coefficients.get_spectrum(m4, mfs=34, mts=1),  # 34 * 121 = 4114 grid points
        "mw": lambda: Entanglement.meyer_wallach(m5, n_samples=2048, random_key=1000),
    }


def main():
    want = [a for a in sys.argv[1:] if not a.startswith("-")] or ["c3", "c4", "mw"]
    prof = "--profile" in sys.argv
    all_legs = legs()
    for name in want:
        fn = all_legs[name]
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        reps = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        walls, gpus = [], []
        for _ in range(reps):  # wall: the call alone (it returns host values); GPU share: calls of their own
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            walls.append(time.perf_counter() - t0)
        for _ in range(reps):
            torch.cuda.synchronize()
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            gpus.append(e0.elapsed_time(e1))
        walls.sort(), gpus.sort()
        print(f"{name}: wall median {walls[reps // 2] * 1e3:.3f} ms (min {walls[0] * 1e3:.3f}), "
              f"first-to-last GPU event {gpus[reps // 2]:.3f} ms (min {gpus[0]:.3f})", flush=True)
        if prof:
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            pr.disable()
            st = pstats.Stats(pr)
            print(f"{name}: {st.total_calls / 20:.0f} Python calls per invocation")
            st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
