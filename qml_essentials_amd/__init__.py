"""Import shim: ``import qml_essentials_amd`` -> sources in ``qml-essentials_amd/``.

The product package directory is named ``qml-essentials_amd`` (not a valid Python
identifier), so this two-line package extends its ``__path__`` to that directory.
Everything (modules, ``csrc/``, the built ``libqmle_sv.so``) lives there.
"""
import os as _os

_REAL = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "qml-essentials_amd"
)
__path__.insert(0, _REAL)
PACKAGE_DIR = _REAL

from ._version import __version__  # noqa: E402,F401
