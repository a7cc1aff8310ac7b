"""MI355X-native GPS L1 C/A acquisition + tracking engine with SoftGNSS-python's API.

Drop-in surface (same names, arguments and error behaviour as the reference modules):
    initialize.Settings / Result / TruePosition     reference initialize.py
    acquisition.AcquisitionResult                   reference acquisition.py
    tracking.TrackingResult                         reference tracking.py
All signal processing runs in hand-written gfx950 HIP kernels reached through the ctypes
C-ABI of libsgx.so (include/sgx.h).  There is no CPU implementation in this package.

The directory name contains a hyphen; import it with
    importlib.import_module("softgnss-python_amd")
or put softgnss-python_amd/dropin on sys.path and `import initialize, acquisition, tracking`
exactly as the reference's scripts do.
"""
from . import _native, engine, synth   # noqa: F401
from .acquisition import AcquisitionResult   # noqa: F401
from .initialize import Result, Settings, TruePosition   # noqa: F401
from .record import DeviceFile, DeviceSignal   # noqa: F401
from .postNavigation import NavigationResult   # noqa: F401
from .tracking import TrackingResult   # noqa: F401

__all__ = ["Settings", "Result", "TruePosition", "AcquisitionResult", "TrackingResult", "NavigationResult", "DeviceFile",
           "DeviceSignal", "synth", "engine"]
