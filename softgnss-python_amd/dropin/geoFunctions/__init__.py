"""Drop-in package name of the reference's geoFunctions (the functions the position solution uses)."""
import importlib as _il
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
_mod = _il.import_module("softgnss-python_amd.geoFunctions")
globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("__")})
