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
        _mod = importlib.import_module("xmhw_amd._xmhw_hip")
    except ImportError as e:  # pragma: no cover - build problem
        raise HipExtensionMissing(f"cannot import xmhw_amd._xmhw_hip: {e}") from e
    return _mod


def require_gpu():
    """Raise unless a HIP device is usable."""
    h = hip()
    n = h.device_count()          # raises HipError without a device/driver
    if n < 1:
        raise h.HipError("no HIP device visible")
    return n
