"""The GPU box compiles libqmle_sv from a clean slate: the `.so` that travels with
the snapshot is accepted by content hash, so without this test hipcc would never run there."""
import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.gpu
def test_clean_build_on_this_box(tmp_path):
    import torch  # noqa: F401  (its HIP runtime first, like _native.lib())

    import __graft_entry__ as G
    from qml_essentials_amd import _native as N

    out = tmp_path / "libqmle_sv_clean.so"
    G.build_library(str(out), force=True, obj_dir=str(tmp_path / "obj"))  # every unit through hipcc, then the link
    clean = ctypes.CDLL(str(out))
    assert clean.qmle_sv_version() == N.lib().qmle_sv_version()
    for name, _res, _args in N.SYMBOLS:
        assert hasattr(clean, name), name
    # and the library in the tree is the one its sources describe
    assert open(G.LIB + ".sha").read().strip() == G.source_hash()
