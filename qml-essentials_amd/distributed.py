"""Batch-axis sharding over the GPUs of one node (one process per GPU, RCCL/xGMI).

The reference has no multi-device code; its authors mark the ``jax.vmap`` in
``Script._execute_batched`` as "the exact boundary to replace with
``jax.shard_map``" (``qml_essentials/script.py:443-453``: batched args ``P(0)``,
broadcast args ``P()``, outputs ``P(0)``).  This module is that replacement:
contiguous blocks of the flattened batch per rank, the plan replicated, and exactly
ONE collective per call -- an all-gather of KiB-sized results (fidelities, expvals,
Meyer-Wallach values).  A single statevector is never split across GPUs
("replicas only").

``torch.distributed`` backend ``nccl`` is RCCL on ROCm; ``gloo`` is used by the
CPU tests.  Rendezvous must use 127.0.0.1 (container hostnames may not resolve).

**Contract.**  Sharding is opt-in: it is switched on by ``init_from_env()`` / ``enable()``
(or ``QMLE_SHARD=1``), never by the mere existence of a ``torch.distributed`` process
group -- a DDP-style program whose ranks each hold their OWN minibatch must not have its
rows mixed with other ranks'.  Once on, every rank must make the same calls with
IDENTICAL arguments (same params / inputs / keys); each rank computes rows
``shard_bounds(B)`` of that common batch and the all-gather returns all ``B`` rows on every
rank.  ``QMLE_SHARD_CHECK=1`` verifies the contract with one extra all-reduce per call (a
hash of the arguments).  One process drives one GPU (libqmle_sv keeps per-device state
keyed by the current device, but a plan's device blob lives on the device of its first run).
"""
from __future__ import annotations

import contextlib
import os
import threading
from typing import List, Sequence, Tuple

import numpy as np

_state = threading.local()
_opt_in = os.environ.get("QMLE_SHARD", "0") not in ("", "0")


def _dist():
    import torch.distributed as dist

    return dist


def is_initialized() -> bool:
    try:
        dist = _dist()
        return dist.is_available() and dist.is_initialized()
    except Exception:  # pragma: no cover - torch without distributed
        return False


def world() -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when no process group exists."""
    if not is_initialized():
        return 0, 1
    dist = _dist()
    return dist.get_rank(), dist.get_world_size()


def enable(on: bool = True) -> None:
    """Opt in to (or out of) batch sharding for this process; see the module contract."""
    global _opt_in
    _opt_in = bool(on)


def enabled() -> bool:
    """True when calls should shard their batch across ranks."""
    return _opt_in and world()[1] > 1 and not getattr(_state, "local_only", False)


def check_same_arguments(*arrays) -> None:
    """Debug aid (``QMLE_SHARD_CHECK=1``): raise if the ranks were handed different arguments.
    One all-reduce (MIN and MAX of a 63-bit hash) -- off by default."""
    if os.environ.get("QMLE_SHARD_CHECK", "0") in ("", "0") or not enabled():
        return
    import hashlib

    import torch

    h = hashlib.blake2b(digest_size=8)
    for a in arrays:
        if a is None:
            h.update(b"none")
            continue
        if hasattr(a, "detach"):
            a = a.detach().cpu().numpy()
        try:
            a = np.ascontiguousarray(a)
            if a.dtype == object:
                raise TypeError
        except (TypeError, ValueError):
            h.update(repr(a).encode())
            continue
        h.update(str(a.shape).encode() + str(a.dtype).encode() + a.tobytes())
    v = int.from_bytes(h.digest(), "little") >> 1
    dist = _dist()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([v, -v], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t[0].item()) != -int(t[1].item()):
        raise RuntimeError("qml_essentials_amd.distributed: ranks called with different arguments; "
                           "batch sharding needs identical arguments on every rank "
                           "(distributed.enable(False) for per-rank minibatches)")


@contextlib.contextmanager
def local_only():
    """Suppress sharding inside the block (the caller shards by itself)."""
    prev = getattr(_state, "local_only", False)
    _state.local_only = True
    try:
        yield
    finally:
        _state.local_only = prev


def init_from_env(backend: str = None) -> Tuple[int, int]:
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*
    (set by ``python -m torch.distributed.run``) and opt in to batch sharding; no-op for a
    single process."""
    import torch

    size = int(os.environ.get("WORLD_SIZE", "1"))
    if size > 1:
        enable(True)
    if size <= 1 or is_initialized():
        return world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist = _dist()
    # QMLE_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsals on a 1-GPU box)
    backend = backend or os.environ.get("QMLE_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
    dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=size)
    return world()


def shard_bounds(n: int, rank: int = None, size: int = None) -> Tuple[int, int]:
    """Contiguous block ``[lo, hi)`` for ``rank``: the balanced split ``rank*n//size ..
    (rank+1)*n//size`` -- block sizes differ by at most one and no rank is left empty
    when ``n >= size`` (an empty shard would skip the engine and hang the others in the
    all-gather)."""
    if rank is None or size is None:
        rank, size = world()
    return rank * n // size, (rank + 1) * n // size


def my_block(n: int, *arguments) -> Tuple[int, int, bool]:
    """``(lo, hi, sharded)`` of this rank for a batch of ``n`` rows: the whole range when
    sharding is off or there are fewer rows than ranks (then every rank computes all rows and
    no collective runs).  The one place the four call sites (Script, Model's compiled device
    path, Expressibility pairs, Meyer-Wallach samples) take their block from; they hand over
    the call's ``arguments`` (params / inputs / ...), which ``QMLE_SHARD_CHECK=1`` verifies to be
    identical on every rank before any row is split (:func:`check_same_arguments`)."""
    if enabled() and n >= world()[1]:
        check_same_arguments(n, *arguments)
        lo, hi = shard_bounds(n)
        return lo, hi, True
    return 0, n, False


def all_shard_bounds(n: int, size: int) -> List[Tuple[int, int]]:
    return [shard_bounds(n, r, size) for r in range(size)]


def all_gather_rows(local, n_total: int):
    """Concatenate every rank's rows (rank r holds ``shard_bounds(n_total, r)``) along
    axis 0 and return the full array on every rank -- one ``all_gather`` of blocks padded to
    the largest shard (shards differ by at most one row; when they are equal the gather
    lands in place and nothing is re-sliced).  ``local``: torch tensor (any device) or
    ndarray; an empty shard (``n_total < size``) contributes a zero-row block."""
    import torch

    rank, size = world()
    if not is_initialized():
        return local
    # (a one-rank process group still goes through the collective: the product never shards at
    # world size 1 -- `enabled()` -- but tests/test_gpu_distributed.py exercises RCCL this way)
    is_np = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local.contiguous()
    dist = _dist()
    if dist.get_backend() == "nccl" and not t.is_cuda:
        t = t.cuda()
    elif dist.get_backend() == "gloo" and t.is_cuda:
        t = t.cpu()
    bounds = all_shard_bounds(n_total, size)
    lo, hi = bounds[rank]
    if t.shape[0] != hi - lo:
        raise ValueError(f"all_gather_rows: rank {rank} holds {t.shape[0]} rows, its shard of "
                         f"{n_total} has {hi - lo}")
    per = max(h - l for l, h in bounds)
    even = all(h - l == per for l, h in bounds)
    if even:
        pad = t
    else:
        pad = torch.zeros((per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
    out = torch.empty((size * per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if t.is_complex():  # gather as real pairs: safest across backends
        dist.all_gather_into_tensor(torch.view_as_real(out), torch.view_as_real(pad))
    else:
        dist.all_gather_into_tensor(out, pad)
    if even:
        full = out
    else:
        full = torch.cat([out[r * per: r * per + (h - l)] for r, (l, h) in enumerate(bounds)], dim=0)
    if is_np:
        return full.cpu().numpy()
    if local.is_cuda and not full.is_cuda:
        return full.to(local.device)
    return full if local.is_cuda or not full.is_cuda else full.cpu()


def barrier() -> None:
    if is_initialized():
        _dist().barrier()
