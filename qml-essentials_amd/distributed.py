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
"""
from __future__ import annotations

import contextlib
import os
import threading
from typing import List, Sequence, Tuple

import numpy as np

_state = threading.local()


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


def enabled() -> bool:
    """True when calls should shard their batch across ranks."""
    return world()[1] > 1 and not getattr(_state, "local_only", False)


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
    (set by ``python -m torch.distributed.run``); no-op for a single process."""
    import torch

    size = int(os.environ.get("WORLD_SIZE", "1"))
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
    """Contiguous block ``[lo, hi)`` of ``ceil(n / size)`` items for ``rank``."""
    if rank is None or size is None:
        rank, size = world()
    per = -(-n // size)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def all_shard_bounds(n: int, size: int) -> List[Tuple[int, int]]:
    return [shard_bounds(n, r, size) for r in range(size)]


def all_gather_rows(local, n_total: int):
    """Concatenate every rank's rows (rank r holds ``shard_bounds(n_total, r)``) along
    axis 0 and return the full array on every rank -- one ``all_gather`` of padded,
    equal-sized blocks.  ``local``: torch tensor (any device) or ndarray."""
    import torch

    rank, size = world()
    if size == 1:
        return local
    is_np = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local.contiguous()
    dist = _dist()
    if dist.get_backend() == "nccl" and not t.is_cuda:
        t = t.cuda()
    elif dist.get_backend() == "gloo" and t.is_cuda:
        t = t.cpu()
    per = -(-n_total // size)
    pad = torch.zeros((per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    out = torch.empty((size * per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if t.is_complex():  # gather as real pairs: safest across backends
        dist.all_gather_into_tensor(torch.view_as_real(out), torch.view_as_real(pad))
    else:
        dist.all_gather_into_tensor(out, pad)
    pieces = [out[r * per: r * per + (hi - lo)]
              for r, (lo, hi) in enumerate(all_shard_bounds(n_total, size))]
    full = torch.cat(pieces, dim=0)
    if is_np:
        return full.cpu().numpy()
    if local.is_cuda and not full.is_cuda:
        return full.to(local.device)
    return full if local.is_cuda or not full.is_cuda else full.cpu()


def barrier() -> None:
    if is_initialized():
        _dist().barrier()
