"""Device-context cache: one libsgx context per (settings, device).

A thread that wants a context of its own - several independent receivers multiplexed on one GPU, each with its
stream, scratch and record - wraps its calls in `with engine.private_context(settings, device):`."""
import contextlib
import os
import threading

from . import _native

_contexts = {}
_local = threading.local()


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
    own = getattr(_local, "ctx", None)
    if own is not None and own[0] == key:
        return own[1]
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = _native.Context(settings, dev)
        _contexts[key] = ctx
    return ctx


@contextlib.contextmanager
def private_context(settings, device=None, priority=0):
    """A context used only by the calling thread for the duration of the block (closed afterwards).
    priority: stream priority class (-1, 0, +1) - give contexts that run concurrently different classes."""
    dev = default_device() if device is None else int(device)
    key = (bytes(_native.settings_struct(settings)), dev)
    ctx = _native.Context(settings, dev, priority)
    prev = getattr(_local, "ctx", None)
    _local.ctx = (key, ctx)
    try:
        yield ctx
    finally:
        _local.ctx = prev
        ctx.close()


def close_all():
    for c in list(_contexts.values()):
        c.close()
    _contexts.clear()
