"""CPU-only: the C-ABI library loads, exports every declared symbol, and the
host-side plan compiler validates / schedules tapes (no compute without a GPU)."""
import os
import re

import numpy as np
import pytest

from qml_essentials_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_header_symbols():
    lib = N.lib()
    assert lib.qmle_sv_version() == 149
    header = open(os.path.join(ROOT, "include", "qmle_sv.h")).read()
    declared = set(re.findall(r"\b(qmle_[a-z_0-9]+)\s*\(", header))
    declared -= {"qmle_op", "qmle_plan"}
    bound = {name for name, _, _ in N.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name)


def test_header_opcodes_match_binding():
    header = open(os.path.join(ROOT, "include", "qmle_sv.h")).read()
    codes = dict(re.findall(r"QMLE_OP_([A-Z0-9_]+) = (\d+)", header))
    pairs = {"RX": "RX", "RY": "RY", "RZ": "RZ", "ROT": "Rot", "CX": "CX", "CRX": "CRX",
             "CPHASE": "CPhase", "CCX": "CCX", "CSWAP": "CSWAP", "DIAG_ALL": "DIAG_ALL",
             "H": "H", "X": "PauliX", "Z": "PauliZ", "SWAP": "SWAP", "RZX": "RZX"}
    for h, p in pairs.items():
        assert int(codes[h]) == N.OPCODES[p]


def test_plan_validation_errors_map_to_valueerror():
    # operations.py:140-146: wrong wire count / duplicate wires -> ValueError
    with pytest.raises(ValueError, match="wires"):
        N.Plan([("CX", [0], [], -1)], 2, 0)
    with pytest.raises(ValueError, match="duplicate"):
        N.Plan([("CX", [1, 1], [], -1)], 2, 0)
    with pytest.raises(ValueError, match="range"):
        N.Plan([("RX", [3], [0], -1)], 2, 1)
    with pytest.raises(ValueError, match="slot"):
        N.Plan([("RX", [0], [2], -1)], 2, 1)
    with pytest.raises(ValueError):
        N.Plan([("Nope", [0], [], -1)], 2, 0)


def he_layer_ops(n):
    ops, slot = [], 0
    for g in ("RY", "RZ", "RY"):
        for q in range(n):
            ops.append((g, [q], [slot], -1))
            slot += 1
    from oracle.circuits import bricks
    for c, t in bricks(n, mirror=False) + bricks(n, offset=-1, modulo=True, wrap=True, mirror=False):
        ops.append(("CX", [c, t], [], -1))
    return ops, slot


def test_small_circuit_runs_whole_state_in_lds():
    ops, slots = he_layer_ops(12)
    st = N.Plan(ops, 12, slots).stats()
    assert st["whole_state_lds"] == 1 and st["n_passes"] == 1
    # RY.RZ.RY on each wire merge into one 2x2 (commutation-aware)
    assert st["n_lowered"] == 12 + 12


def test_no_fusion_is_one_pass_per_gate():
    ops, slots = he_layer_ops(24)
    p = N.Plan(ops, 24, slots, flags=N.plan_flags(no_fusion=True))
    st = p.stats()
    # every gate but one streams through the direct kernel -- since round 2 also the CX with the
    # control on bits 2 / 3 and the target on bits 1 / 2 (wires 21 -> 22, 20 -> 21: lane-exchange
    # mode); CX 22 -> 23 (control on bit 1, target on bit 0) keeps its single-gate tile pass
    assert st["n_passes"] == 96 and st["direct_passes"] == 95
    # a control on bits 1..3 with a HIGH target still takes a single-gate tile pass
    hi = N.Plan([("CX", [22, 3], [], -1)], 24, 0, flags=N.plan_flags(no_fusion=True)).stats()
    assert hi["n_passes"] == 1 and hi["direct_passes"] == 0
    # (72*256 + 24*128) MiB per state
    assert st["algo_bytes_per_state"] == (72 * 256 + 24 * 128) * 2**20


def test_fused_schedule_covers_every_gate_once_and_respects_order():
    ops, slots = he_layer_ops(24)
    p = N.Plan(ops, 24, slots)
    d = p.describe()
    seen = [s for st in d["stages"] for s in st["src_ops"]]
    assert sorted(seen) == list(range(len(ops)))
    assert d["n_qubits"] == 24 and len(d["stages"]) <= 6
    # dependency order: for gates sharing a wire, stage index must be non-decreasing
    stage_of = {}
    for si, st in enumerate(d["stages"]):
        for s in st["src_ops"]:
            stage_of[s] = si
    last = {}
    for i, (_, wires, _, _) in enumerate(ops):
        for w in wires:
            if w in last:
                assert stage_of[last[w]] <= stage_of[i]
            last[w] = i
    for st in d["stages"]:
        if st["kind"] == "tile":
            assert st["T"] in (12, 13) and st["bits"][: st["L"]] == list(range(st["L"]))
            assert st["T"] == d["tile_bits"] and 1 <= st["lds_round_trips"] <= st["n_lowered"]


def test_compute_without_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = N.Plan([("H", [0], [], -1)], 1, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        p.run(None, "state")


def test_hot_kernels_have_no_scratch_and_keep_their_occupancy():
    """hipcc's resource-usage remarks, saved by ``__graft_entry__.build()``: no kernel may use
    scratch or spill VGPRs (a 16-amplitude register tile that falls into scratch costs 3.5x --
    seen once while restructuring the gate loop), and the tile kernels must stay at
    >= 4 waves per SIMD, the occupancy their latency hiding was tuned for."""
    import json
    import os

    import __graft_entry__ as G

    if not os.path.exists(G.RESOURCES):
        G.build(force=True)
    res = json.load(open(G.RESOURCES))
    assert len(res) >= 40
    for name, r in res.items():
        if "k_build_matrices" in name:  # fp64 products of <= 4x4 matrices, indexed at run time:
            continue                    # 20 us per 1024 x 47 matrices, not worth unrolling
        if "k_tile_pf" in name:         # the opt-in LDS-DMA experiment (measured slower, kept
            continue                    # for A/B only): its 16x16-operator variant spills 36 B at 512 threads
        assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0, (name, r)
    tiles = {k: v for k, v in res.items() if "6k_tileILb" in k}
    assert len(tiles) == 4  # <DENSE4> x <MW>
    for name, r in tiles.items():
        assert r["Occupancy"] >= 4 and r["VGPRs"] <= 128, (name, r)
    direct = [v for k, v in res.items() if "k_direct_1q" in k]
    assert direct and all(r["Occupancy"] >= 7 for r in direct)
    # the fast tile kernel: far below the 96 VGPRs that 5 workgroups of 32 KiB per CU allow (the 16
    # amplitudes live in 32 registers, no copies around the gate dispatch); the multi-tile variants
    # (next tile's 8 float4 per lane in flight / tile loop around the epilogue) stay within 96 too
    fast = {k: v for k, v in res.items() if "k_tile2" in k}
    # <NT, MEASURE, MULTI> x 8 + the two whole-state instantiations (<.., WS>) + the three that report
    # Meyer-Wallach sums behind the store (<.., MW>: 40 more sums per work item, still 5 workgroups per CU)
    # + (round 5) the two that walk several tiles with Z-parity observables (<.., MASKS>: 24 accumulators per work
    # item, 4 workgroups of 32 KiB per CU)
    assert len(fast) == 15
    import re

    for name, r in fast.items():
        nt, me, mu, ws, mw, mk = (c == "1" for c in
                                  re.search(r"k_tile2ILb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)EEEv", name).groups())
        if mk:
            assert r["Occupancy"] >= 4 and r["VGPRs"] <= 128 and me and mu and not (ws or mw), (name, r)
            continue
        assert r["Occupancy"] >= 5 and r["VGPRs"] <= (96 if mu or mw else 64), (name, r)
        assert not (ws and (nt or mu)) and not (mw and (mu or not me)), name


def test_xor_addressed_tile_kernels_have_no_static_lds():
    """The tile kernels address their LDS tile as ``base ^ offset`` (swizzle and gather offsets in
    one XOR), which needs the dynamic-LDS window to start at offset 0, i.e. NO static __shared__
    in those kernels (ADVICE r2 / VERDICT r2 item 7).  The library checks it on the host at the
    first launch (``lds_base_is_zero`` -> QMLE_ERR_INTERNAL, no device-side trap any more); this
    is the same check on the build's resource table, where it fails without a GPU."""
    import json
    import os

    import __graft_entry__ as G

    if not os.path.exists(G.RESOURCES):
        G.build(force=True)
    res = json.load(open(G.RESOURCES))
    xor_kernels = {k: v for k, v in res.items()
                   if any(t in k for t in ("6k_tileILb", "k_tile2", "k_tile_pf", "k_mw_tile2", "k_mw_read"))}
    assert len(xor_kernels) >= 14
    for name, r in xor_kernels.items():
        assert r["LDS Size"] == 0, (name, r)
    for unit in G.UNITS:
        assert "__builtin_trap" not in open(os.path.join(G.CSRC, unit)).read(), unit
    assert N.lib().qmle_status_string(-12).decode().startswith("internal invariant")


def test_from_zero_variant_of_the_all_live_k2_plan(monkeypatch):
    """Plan compiler, host only.  qmle_run_batch always starts from |0..0>, where the first stage computes
    ONE tile per state whatever the tile: the plan it executes (the from-zero variant, reported by executed())
    stages 2^14 amplitudes first.  Round 5: on the TOP 14 positions {10..23} (Stage::shift), which leaves the
    low positions, the deferred CX(10 -> 9) and the wrap-around CX(0 -> 23) to ONE measuring pass on {0..10, 23}
    -- two passes.  Round 3's schedule (first tile on {0..13}, position 6 carried so that the read+write pass's
    tile is {0-3, 6, 13-19}, three passes) is candidate 26.  The plan's own stages, which qmle_apply_inplace and
    the adjoint sweep apply to LIVE states, keep the round-2 schedule."""
    ops, slots = he_layer_ops(24)
    top = N.Plan(ops, 24, slots, flags=N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB)
    own = top.describe()
    assert own["zero_run"] is False and [s["T"] for s in own["stages"]] == [12, 12, 12]
    assert all(s["shift"] == 0 for s in own["stages"])
    assert [sum(g["n_ops"] for g in s["fast_groups"]) for s in own["stages"]] == [12, 8, 4]
    assert top.expval_child() is None   # NO_ABSORB: nothing folded; executed() names the variant
    var = top.executed("expval")
    assert var is not top
    d = var.describe()
    assert d["zero_run"] is True and d["model_cost"] < own["model_cost"] and d["candidate"] >= 48
    st = d["stages"]
    assert len(st) == 2 and [s["T"] for s in st] == [14, 12] and [s["shift"] for s in st] == [10, 0]
    assert st[0]["bits"] == list(range(10, 24)) and not st[0]["fast"]   # one tile per state, generic kernel
    assert st[1]["bits"] == list(range(11)) + [23] and st[1]["fast"] and len(st[1]["fast_groups"]) == 3
    assert sorted(o for s in st for o in s["src_ops"]) == list(range(96))
    # the round-3 schedule, still a candidate (what QMLE_NO_TOP_FIRST=1 falls back to)
    monkeypatch.setenv("QMLE_FORCE_CAND", "26")
    d = N.Plan(ops, 24, slots, flags=N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB).executed("expval").describe()
    monkeypatch.delenv("QMLE_FORCE_CAND")
    assert d["zero_run"] is True and d["model_cost"] < own["model_cost"]
    st = d["stages"]
    assert len(st) == 3 and [s["T"] for s in st] == [14, 12, 12]
    assert st[0]["bits"] == list(range(14)) and not st[0]["fast"]      # 2^14 amplitudes: generic kernel, one tile per state
    assert st[1]["bits"] == [0, 1, 2, 3, 6] + list(range(13, 20)) and st[1]["fast"]
    assert [len(s["fast_groups"]) for s in st[1:]] == [2, 1]           # 6 + 4 dense gates: the measuring pass is ONE group
    assert sum(s["n_lowered"] for s in st) == 48
    # every gate exactly once, in an order that respects wires (the generic checker of the fused schedule)
    seen = sorted(o for s in st for o in s["src_ops"])
    assert seen == list(range(96))
    # known-zero runs keep the round-2 geometry (their first passes are launch-bound special kernels)
    dflt = N.Plan(ops, 24, slots).expval_child().describe()
    assert [s["T"] for s in dflt["stages"]] == [12, 12, 12] and dflt["stages"][1]["product"]
    # forced geometries, one-pass-per-gate plans and whole-state plans have no variant
    unf = N.Plan(ops, 24, slots, flags=N.PLAN_NO_ABSORB | N.PLAN_NO_FUSION)
    assert unf.expval_child() is None and unf.executed("expval") is unf and unf.executed("state") is unf
    ops12, slots12 = he_layer_ops(12)
    small = N.Plan(ops12, 12, slots12, flags=N.PLAN_NO_ABSORB)
    assert small.expval_child() is None and small.executed("expval") is small


def test_meyer_wallach_read_count():
    """Host only: reads of the state per Meyer-Wallach call (bench.py's bytes-moved accounting):
    one cache-resident sweep per wire below the tile size, then 1 + ceil((n - 12) / 8)."""
    lib = N.lib()
    assert [lib.qmle_meyer_wallach_reads(n) for n in (1, 4, 11)] == [1, 4, 11]
    assert [lib.qmle_meyer_wallach_reads(n) for n in (12, 13, 16, 20, 21, 24, 28, 29, 32)] == \
        [1, 2, 2, 2, 3, 3, 3, 4, 4]
    assert lib.qmle_meyer_wallach_reads(0) == 0 and lib.qmle_meyer_wallach_reads(33) == 0
    for n in range(12, 33):   # rows of every read + the purities fit the advertised workspace
        assert lib.qmle_meyer_wallach_workspace_bytes(n, 1) >= (1 << (n - 12)) // 16 * 48 * 4


def test_known_zero_tracking_and_kernel_choice_of_the_k2_plan():
    """Plan compiler, host only: the K2 plan (one HE layer at n = 24, <Z> on every wire) after
    observable folding -- ready gates are scheduled low bits first, every stage knows which bit
    positions are still exactly zero when it starts, the middle pass qualifies for the product
    kernels and the last one for the register-resident measuring kernel."""
    ops, slots = he_layer_ops(24)
    top = N.Plan(ops, 24, slots)
    d = top.describe()
    assert d["absorbed_ops"] == 24
    st = d["expval_plan"]["stages"]
    assert [s["kind"] for s in st] == ["tile"] * 3
    assert st[0]["bits"] == list(range(12)) and st[0]["zero_in"] == (1 << 24) - 1
    assert st[1]["zero_in"] == ((1 << 24) - 1) & ~((1 << 12) - 1)        # bits 0..11 rotated so far
    assert st[2]["zero_in"] == 0xF << 20 and st[2]["bits"][-4:] == [20, 21, 22, 23]
    assert st[0]["next_tile"] and st[1]["next_tile"] and not st[2]["next_tile"]
    assert st[1]["product"] and st[1]["lds_round_trips"] == 2
    assert st[2]["expval_kernel"] == "k_reg_measure_mono"
    # HBM bytes per state: one 32 KiB tile, 2^12 -> 2^20 amplitudes, 2^20 amplitudes read
    assert st[0]["read_bytes_from_zero"] == 0 and st[0]["write_bytes_from_zero"] == 8 * 2**12
    assert st[1]["read_bytes_from_zero"] == 8 * 2**12 and st[1]["write_bytes_from_zero"] == 8 * 2**20
    assert st[2]["read_bytes_from_zero"] == 8 * 2**20
    # switched off: dense passes.  The folded form would end in k_reg_measure on live input; the
    # engine now applies the trailing CX layer instead (free in the fast tile path, single-bit
    # epilogue) unless told to fold regardless
    top_dense = N.Plan(ops, 24, slots, flags=N.plan_flags(no_sparse=True)).describe()
    assert "expval_plan" not in top_dense and all(s["fast"] for s in top_dense["stages"])
    assert top_dense["stages"][1]["read_bytes_from_zero"] == 8 * 2**24
    # deeper circuits: folding stays where the model says it pays (4 layers), not where the
    # folded plan's general-mask epilogue costs more than the CX layer (2 layers)
    def layers(k):
        o, s = [], 0
        for _ in range(k):
            oo, ss = he_layer_ops(24)
            o += [(g, w, [x + s for x in sl], m) for g, w, sl, m in oo]
            s += ss
        return o, s
    assert "expval_plan" not in N.Plan(*((lambda t: (t[0], 24, t[1]))(layers(2)))).describe()
    assert N.Plan(*((lambda t: (t[0], 24, t[1]))(layers(4)))).describe()["absorbed_ops"] == 24
    # tape order keeps the old schedule (high wires first)
    tape = N.Plan(ops, 24, slots, flags=N.plan_flags(tape_order=True)).describe()["expval_plan"]["stages"]
    assert tape[0]["bits"] == [0, 1, 2, 3] + list(range(16, 24))


def test_fast_tile_groups_fold_cx_into_the_layout():
    """Plan compiler, host only (round 2): the all-amplitudes-live K2 plan.  Every stage takes the
    fast tile path; X / CX never cost a register-tile group (they change the LDS layout map the
    address tables are built from), so the passes after the initialisation need 2 groups each
    for their 12 gates instead of 3, and a stage whose layout is not the identity at its end
    re-lays the tile out in its last group."""
    ops, slots = he_layer_ops(24)
    d = N.Plan(ops, 24, slots, flags=N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB).describe()
    assert "absorbed_ops" not in d or d["absorbed_ops"] == 0
    st = d["stages"]
    assert len(st) == 3 and all(s["fast"] for s in st)
    assert sum(s["n_lowered"] for s in st) == 48                 # 24 merged 2x2 + 24 CX
    # CX gates wait for a pass that holds both wires anyway ("lazy" schedule): dense gates per
    # pass 12 / 8 / 4, so the measuring pass is ONE group
    assert [sum(g["n_ops"] for g in s["fast_groups"]) for s in st] == [12, 8, 4]
    assert [len(s["fast_groups"]) for s in st] == [3, 2, 1]
    for s in st[1:]:
        groups = s["fast_groups"]
        assert len(s["groups"]) >= 3                             # generic grouping needs >= 3
        assert groups[-1]["relayout"] == 1 and all(g["relayout"] == 0 for g in groups[:-1])
        assert s["read_bytes_from_zero"] == 8 * 2**24            # all amplitudes live
    # a stage with a gate the fast path does not cover (CCX, 4x4) keeps the generic kernel
    mixed = N.Plan([("H", [0], [], -1), ("CCX", [0, 1, 2], [], -1), ("RXX", [3, 4], [0], -1)], 16, 1,
                   flags=N.PLAN_NO_SPARSE).describe()["stages"]
    assert not any(s["fast"] for s in mixed)
    # forced geometries outside 10..13 tile bits fall back as well
    small = N.Plan(ops, 24, slots, flags=N.PLAN_NO_SPARSE | N.plan_flags(tile_bits=8, low_bits=4))
    assert not any(s["fast"] for s in small.describe()["stages"])


def test_philox_sampler_is_numpys_stream_bit_for_bit():
    """qmle_philox_uniform_f32 (host-side, csrc/qmle_rng.cpp) == numpy's Philox4x64-10 stream
    through Generator.uniform + float32 cast: every length around the block / batch / thread
    boundaries, spawned keys, other ranges -- and `utils.uniform`, the sampler behind
    `Model.initialize_params` (reference: `model.py:687-693`, threefry stream: SURVEY 8-c), draws
    the same parameters with either loop."""
    import os

    from qml_essentials_amd import utils

    for seed in (0, 1000, 2**40 + 3):
        for n in (0, 1, 2, 3, 4, 5, 11, 12, 13, 31, 32, 33, 16383, 16384, 32767, 32769, 73728, 131072, 262147,
                  524288 + 5, (1 << 20) + 3, 3 * (1 << 20) + 1):  # (the last three: 2, 4 and 8 threads)
            seq = np.random.SeedSequence(seed)
            ref = np.random.Generator(np.random.Philox(np.random.SeedSequence(seed))).uniform(0, 2 * np.pi, n).astype(np.float32)
            got = N.philox_uniform(seq.generate_state(2, np.uint64), n, 0.0, 2 * np.pi)
            assert np.array_equal(ref, got), (seed, n)
    for child in np.random.SeedSequence(5).spawn(3):
        ref = np.random.Generator(np.random.Philox(child)).uniform(-1.5, 3.25, 1001).astype(np.float32)
        again = np.random.SeedSequence(entropy=child.entropy, spawn_key=child.spawn_key)
        assert np.array_equal(ref, N.philox_uniform(again.generate_state(2, np.uint64), 1001, -1.5, 3.25))
    k = utils.key(7).split(3)[1]
    mine = utils.uniform(k, (64, 3, 11), 0.0, 2 * np.pi)
    os.environ["QMLE_NUMPY_SAMPLER"] = "1"
    try:
        theirs = utils.uniform(k, (64, 3, 11), 0.0, 2 * np.pi)
    finally:
        del os.environ["QMLE_NUMPY_SAMPLER"]
    assert mine.dtype == np.float32 and mine.shape == theirs.shape and np.array_equal(mine, theirs)
    with pytest.raises(ValueError):
        N.philox_uniform(np.zeros(3, np.uint64), 4, 0.0, 1.0)


def test_every_schedule_candidate_places_every_gate_exactly_once(monkeypatch):
    """Plan-compiler invariant, all 60 candidates (tile geometry x lazy CX x wide first tile / carried
    position 6 / first tile on the top positions) and the last-stage padding switch: whatever the schedule, the stages' `src_ops` are a
    permutation of the tape -- no gate dropped, none applied twice -- and the tile of every stage
    holds the positions its gates act on (wire w <-> position n - 1 - w)."""
    n = 20
    ops, slots = [], 0
    for _ in range(3):
        o, s_ = he_layer_ops(n)
        ops += [(g, w, [x + slots for x in sl], m) for g, w, sl, m in o]
        slots += s_
    seen = set()
    for pad in (None, "1"):
        if pad:
            monkeypatch.setenv("QMLE_PAD_HIGH", pad)
        for k in range(60):  # (48..59: the first tile on the top 14 positions -- tuner candidates, round 5)
            monkeypatch.setenv("QMLE_FORCE_CAND", str(k))
            top = N.Plan(ops, n, slots, flags=N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB)
            d = top.executed("expval").describe()
            if k >= 48:
                assert d["candidate"] == k and d["stages"][0]["shift"] == n - 14
            else:
                assert all(st.get("shift", 0) == 0 for st in d["stages"])
            placed = sorted(s for st in d["stages"] for s in st["src_ops"])
            assert placed == list(range(len(ops))), (k, pad)
            for st in d["stages"]:
                if st["kind"] != "tile":
                    continue
                tile = set(st["bits"])
                for s in st["src_ops"]:
                    assert all(n - 1 - w in tile for w in ops[s][1]), (k, pad, s)
            seen.add(tuple(tuple(st["bits"]) for st in d["stages"] if st["kind"] == "tile"))
    assert len(seen) >= 8  # (the candidates really are different schedules)


def test_round4_entry_points_refuse_bad_arguments_before_touching_a_device():
    """qmle_run_batch_map / qmle_plan_autotune / qmle_plan_executed (include/qmle_sv.h, round 4): argument
    errors are status codes, decided on the host -- no HIP call is needed to get them."""
    import ctypes as C

    lib = N.lib()
    ops, slots = he_layer_ops(4)
    plan = N.Plan(ops, 4, slots)
    null = C.c_void_p(None)
    assert lib.qmle_run_batch_map(null, null, null, 1, 2, None, 0, null, null, 0, null) == -1       # no plan
    assert lib.qmle_run_batch_map(plan._h, null, null, 1, 2, None, 0, null, null, 0, null) == -1    # no map
    amap = N.QmleAngleMap()
    assert lib.qmle_run_batch_map(plan._h, C.byref(amap), null, 0, 2, None, 0, null, null, 0, null) == -1  # batch 0
    assert lib.qmle_run_batch_map(plan._h, C.byref(amap), null, 1, 2, None, 0, null, null, 0, null) == -1  # slots, no table
    chosen = (C.c_int32 * 2)()
    a, b = C.c_double(), C.c_double()
    assert lib.qmle_plan_autotune(null, 0, 0, 1, 2, 2, null, chosen, C.byref(a), C.byref(b)) == -1
    assert lib.qmle_plan_executed(null, 0) is None
    assert plan.executed("state").stats()["n_ops"] == plan.stats()["n_ops"]


def test_matrices_of_folded_cx_gates_are_not_built():
    """Round 5: an X / CX inside a register-tile group is a swap of amplitudes or a change of the LDS layout
    map -- no tile kernel reads its 2x2 matrix, so the forward engine's matrix builder skips those build
    groups (`build_groups_needed` of `qmle_plan_describe`; the adjoint sweep and the complex128 engine build all).
    Direct (one gate per launch) stages apply a CX through its matrix: there every group stays needed."""
    ops, slots = he_layer_ops(10)
    plan = N.Plan(ops, 10, slots)
    d = plan.describe()
    n_cx = sum(1 for o in ops if o[0] == "CX")
    assert d["build_groups"] - d["build_groups_needed"] == n_cx > 0
    ops24, slots24 = he_layer_ops(16)
    unfused = N.Plan(ops24, 16, slots24, flags=N.PLAN_NO_FUSION | N.PLAN_FORCE_GLOBAL).describe()
    direct_cx = sum(1 for st in unfused["stages"] if st["kind"] == "direct")
    assert direct_cx > 0 and unfused["build_groups"] - unfused["build_groups_needed"] < n_cx + 100
    assert unfused["build_groups_needed"] >= direct_cx
