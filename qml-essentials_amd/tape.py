"""Per-thread recording tapes.

Behavioural mirror of ``qml_essentials/tape.py:10-55,92-138``: instantiating an
``Operation`` while a ``recording()`` block is active appends it to the innermost
tape of the current thread.  Pulse-event tapes (``tape.py:58-89``) are out of
scope (pulse-level simulation is not on the hot path).
"""
from __future__ import annotations

import copy
import threading
from contextlib import contextmanager
from typing import Callable, Iterator, List, Optional

_tls = threading.local()


def _stack() -> list:
    st = getattr(_tls, "tapes", None)
    if st is None:
        st = _tls.tapes = []
    return st


def active_tape() -> Optional[list]:
    """Innermost tape being recorded on this thread, or ``None``."""
    st = _stack()
    return st[-1] if st else None


@contextmanager
def recording() -> Iterator[list]:
    """Open a fresh tape; nested blocks get independent tapes."""
    st = _stack()
    tape: list = []
    st.append(tape)
    try:
        yield tape
    finally:
        # pop *our* tape even if an inner block leaked one
        while st and st[-1] is not tape:
            st.pop()
        if st:
            st.pop()


def shift_and_append(tape_ops: List, offset: int) -> None:
    """Append wire-shifted shallow copies of ``tape_ops`` to the active tape
    (multi-register circuits, ``tape.py:92-114``)."""
    target = active_tape()
    if target is None:
        return
    for op in tape_ops:
        clone = copy.copy(op)
        clone._wires = [w + offset for w in op.wires]
        target.append(clone)


def copy_to_tape(fn: Callable[[], None], offset: int) -> None:
    """Record ``fn`` on a side tape and replay it shifted by ``offset`` wires
    (``tape.py:117-138``)."""
    with recording() as side:
        fn()
    shift_and_append(side, offset)


@contextmanager
def batch_context(batch: int) -> Iterator[None]:
    """Batch size of the recording in progress (per thread).  Gates that draw per-sample
    random numbers while being recorded (``UnitaryGates.GateError``) read it, because a
    batch-constant angle carries no batch axis of its own."""
    prev = getattr(_tls, "batch", 1)
    _tls.batch = int(batch)
    try:
        yield
    finally:
        _tls.batch = prev


def current_batch() -> int:
    return getattr(_tls, "batch", 1)
