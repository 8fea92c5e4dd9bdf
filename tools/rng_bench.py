#!/usr/bin/env python3
"""Host-side Philox sampler (csrc/qmle_rng.cpp) vs numpy's loop: ms per draw of N values."""
import os, sys, time, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    from qml_essentials_amd import _native as N
    n = int(sys.argv[1])
    key = np.random.SeedSequence(1000).generate_state(2, np.uint64)
    def t(f, reps=200):
        f(); best = 1e9
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(reps): f()
            best = min(best, (time.perf_counter() - t0) / reps * 1e3)
        return best
    print(f"threads={os.environ.get('QMLE_RNG_THREADS', 'auto'):>4} n={n}: library {t(lambda: N.philox_uniform(key, n, 0.0, 6.28)):.4f} ms, "
          f"numpy {t(lambda: np.random.Generator(np.random.Philox(key=key)).uniform(0, 6.28, n).astype(np.float32), 50):.4f} ms", flush=True)
else:
    for n in (73728, 1 << 20):
        for th in ("1", "2", "4", "8", None):
            env = dict(os.environ)
            if th: env["QMLE_RNG_THREADS"] = th
            else: env.pop("QMLE_RNG_THREADS", None)
            subprocess.run([sys.executable, __file__, str(n)], env=env)
