"""Device-context cache: one libsgx context per (settings, device)."""
import os

from . import _native

_contexts = {}


def default_device():
    """GPU index of this process: SGX_DEVICE, else LOCAL_RANK (one process per GPU), else 0."""
    for key in ("SGX_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(key)
        if v not in (None, ""):
            return int(v)
    return 0


def get_context(settings, device=None):
    """Return the cached device context for these settings; raises when there is no GPU/library."""
    dev = default_device() if device is None else int(device)
    key = (bytes(_native.settings_struct(settings)), dev)
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = _native.Context(settings, dev)
        _contexts[key] = ctx
    return ctx


def close_all():
    for c in list(_contexts.values()):
        c.close()
    _contexts.clear()
