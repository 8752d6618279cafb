"""Import shim: the package directory is named `rust-doom_amd` (not a valid identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module('rust-doom_amd')
sys.modules[__name__] = _pkg
