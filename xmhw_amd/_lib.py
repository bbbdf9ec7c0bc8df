"""Loader of the HIP extension.  There is NO CPU fallback: if the gfx950
library or its pybind11 module is missing the product path fails loudly."""
import importlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_mod = None
_err = None


class HipExtensionMissing(ImportError):
    pass


def hip():
    """Return the pybind11 module ``xmhw_amd._xmhw_hip`` or raise."""
    global _mod, _err
    if _mod is not None:
        return _mod
    if not os.path.exists(os.path.join(_HERE, "libxmhw_amd.so")):
        raise HipExtensionMissing(
            "xmhw_amd/libxmhw_amd.so not found: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C xmhw_amd/csrc`")
    try:
        _mod = _RetryOnNoMem(importlib.import_module("xmhw_amd._xmhw_hip"))
    except ImportError as e:  # pragma: no cover - build problem
        raise HipExtensionMissing(f"cannot import xmhw_amd._xmhw_hip: {e}") from e
    return _mod


class _RetryOnNoMem:
    """The pybind11 module with one behaviour added: a call that runs out of device memory
    (XMHW_ERR_NOMEM -> MemoryError) is repeated once after the cached large device buffers of
    xmhw_amd.device have been returned to the driver -- allocations made inside the C ABI (plan
    tables, scratch rows, the float32 threshold copy) cannot see that cache otherwise."""

    def __init__(self, mod):
        self._mod = mod
        self._wrapped = {}

    def __getattr__(self, name):
        w = self._wrapped.get(name)
        if w is not None:
            return w
        attr = getattr(self._mod, name)
        if not callable(attr) or isinstance(attr, type):
            return attr

        def call(*args, **kw):
            try:
                return attr(*args, **kw)
            except MemoryError:
                from .device import release_device_cache
                if not release_device_cache():
                    raise
                return attr(*args, **kw)

        call.__name__ = name
        self._wrapped[name] = call
        return call


def require_gpu():
    """Raise unless a HIP device is usable."""
    h = hip()
    n = h.device_count()          # raises HipError without a device/driver
    if n < 1:
        raise h.HipError("no HIP device visible")
    return n
