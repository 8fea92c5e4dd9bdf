"""CPU-only: `bench.py --gpus N` starts N ranks by itself, the traffic guard
refuses stale PMC records (item 3), and the JSON-line bookkeeping is what the driver parses."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True,
                          env=env, timeout=timeout, cwd=ROOT)


def test_gpus_2_spawns_two_ranks_through_the_launcher():
    """The parent never touches the GPU; it starts torch.distributed.run as a child and relays rank
    0's ONE JSON line.  gloo here (no GPU in this container): on a GPU box the same path runs nccl."""
    p = _run(["--gpus", "2", "--rendezvous-only"], {"QMLE_DIST_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == [0, 1]
    assert rec["collective_backend"] == "gloo" and rec["launched_by"] == "bench.py"
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node=2" in p.stderr
    # the one collective of every sharded leg timed by itself, on the payload the leg gathers --
    # the strong-scaling legs (C3: 1024 pairs, C4: 4096 x 10 values) and their saturated weak-scaling companions
    legs = rec["scaling_legs"]
    assert set(legs) == {"c3_expressibility_12q_1024pairs", "c4_fourier_10q_6l_4096grid",
                         "c3_saturated_weak", "c4_saturated_weak"}
    assert legs["c3_expressibility_12q_1024pairs"]["rows_per_rank"] == 512
    assert legs["c4_fourier_10q_6l_4096grid"]["payload_bytes_per_rank"] == 2048 * 10 * 4
    assert legs["c3_saturated_weak"]["rows_per_rank"] == 16384 and legs["c4_saturated_weak"]["rows_per_rank"] == 32768
    for leg in legs.values():
        assert leg["collective_ms"] is not None and 0.0 < leg["collective_ms"] < 5000.0


def test_world_size_must_match_gpus():
    """A run whose process group is not --gpus ranks wide exits non-zero instead of printing a
    line with a wrong n_gpus (round 2: `--gpus 8` ran one rank and said n_gpus 1)."""
    p = _run(["--gpus", "3", "--rendezvous-only"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0
    assert "--gpus 3" in p.stderr and not p.stdout.strip()


def test_non_nccl_backend_needs_an_explicit_rehearsal_switch():
    p = _run(["--gpus", "2", "--rendezvous-only"])  # no GPU here -> gloo, but nobody asked for it
    assert p.returncode != 0
    assert "needs nccl" in p.stderr


def test_single_rank_bookkeeping():
    p = _run(["--gpus", "1", "--rendezvous-only"])
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads(p.stdout.strip())
    assert rec["n_gpus"] == 1 and rec["ranks_seen"] == [0] and rec["launched_by"] == "python"


def test_stale_traffic_records_are_refused(tmp_path):
    sys.path.insert(0, ROOT)
    import bench

    sha = bench.source_sha16()
    path = tmp_path / "traffic.json"
    path.write_text(json.dumps({
        "fresh": {"hbm_bytes_per_launch": 123, "source": "x", "states_per_launch": 32, "source_sha16": sha},
        "stale": {"hbm_bytes_per_launch": 456, "source": "x", "source_sha16": "0" * 16},
        "unsigned": {"hbm_bytes_per_launch": 789}}))
    assert bench.load_traffic("fresh", str(path)) == (123, "x", 32, None)
    for key in ("stale", "unsigned", "absent"):
        got = bench.load_traffic(key, str(path))
        assert got[0] is None and got[3], (key, got)
    assert "STALE" in bench.load_traffic("stale", str(path))[3]


def test_committed_traffic_file_is_signed():
    """Every record bench.py reads carries the hash of the kernel sources it was collected for
    (the hash may lag behind the tree while kernels are being edited: then bench.py reports
    traffic null + the reason; it never reports stale bytes)."""
    doc = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for key in ("k_tile2:n24:dense", "meyer_wallach:n28"):
        assert key in doc, key
        assert len(doc[key].get("source_sha16", "")) == 16, key
