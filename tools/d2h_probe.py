#!/usr/bin/env python3
"""Device -> host copy of a large result: pageable `.cpu()` against a pinned staging buffer (torch's caching host
allocator), MiB sizes from argv."""
import sys, time, torch
for mib in [int(a) for a in sys.argv[1:]] or [16, 128, 512]:
    t = torch.randn(mib * 1024 * 1024 // 8, 2, device="cuda")
    torch.cuda.synchronize()
    def timed(f, reps=4):
        f(); ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]
    a = timed(lambda: t.cpu())
    def pinned():
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True); torch.cuda.synchronize()
        return h.numpy()
    b = timed(pinned)
    def pinned_then_copy():
        return pinned().copy()
    c = timed(pinned_then_copy)
    print(f"{mib} MiB: .cpu() {a * 1e3:.2f} ms ({mib / 1024 / a:.1f} GiB/s); pinned staging {b * 1e3:.2f} ms ({mib / 1024 / b:.1f} GiB/s); pinned + copy into a pageable array {c * 1e3:.2f} ms")
