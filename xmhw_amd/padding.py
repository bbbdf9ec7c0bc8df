"""maxPadLength: the host half of ``ts.interpolate_na(dim=tdim, max_gap=maxPadLength)``
(xmhw/xmhw.py:157-160, :407-410).

The reference hands the whole job to xarray (``DataArray.interpolate_na``, linear, ``use_coordinate=True``,
``xarray/core/missing.py``: ``interp_na``, ``get_clean_interp_index``, ``_get_nan_block_lengths``).  What
that does, and what is mirrored here and in the ``pad_gaps`` kernel (csrc/kernels_ingest.hip):

* the interpolation abscissa is the time COORDINATE: datetime axes become float64 nanoseconds since
  1970-01-01, numeric axes are used as they are; it must be strictly increasing (ValueError otherwise);
* ``max_gap`` is compared with the coordinate distance between the valid samples either side of a
  run of NaN (n missing daily steps are a gap of n + 1 days).  On a datetime axis it must be a
  timedelta (``numpy.timedelta64``, ``datetime.timedelta``, ``pandas.Timedelta`` or a string pandas
  can parse) and a bare number raises TypeError -- also in the reference, whose documentation
  nevertheless shows an integer; on a numeric axis it must be a number;
* values: ``numpy.interp`` in float64, stored in the series' dtype; leading / trailing runs stay NaN.
"""
import datetime as _dt
import numbers

import numpy as np

_EPOCH = np.datetime64("1970-01-01T00:00:00", "ns")


def interp_index(time):
    """float64 abscissa of xarray's get_clean_interp_index(arr, dim, use_coordinate=True)"""
    t = np.asarray(time)
    if t.dtype.kind == "M":
        x = (t.astype("datetime64[ns]") - _EPOCH) / np.timedelta64(1, "ns")
    elif t.dtype.kind in "fiu":
        x = t.astype(np.float64)
    else:
        raise TypeError(f"Index {t.dtype!r} must be castable to float64 to support interpolation or curve fitting")
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.shape[0] > 1:
        d = np.diff(x)
        if not (d >= 0).all():
            raise ValueError("Index 'time' must be monotonically increasing")
        if (d == 0).any():
            raise ValueError("Index 'time' has duplicate values")
    return x


def max_gap_value(max_gap, time):
    """float64 ``max_gap`` in the units of interp_index(time), with xarray's type rules"""
    if isinstance(max_gap, (np.ndarray, list, tuple, dict, set)):
        raise ValueError("max_gap must be a scalar.")
    if np.asarray(time).dtype.kind == "M":
        if isinstance(max_gap, np.timedelta64):
            return float(max_gap.astype("timedelta64[ns]") / np.timedelta64(1, "ns"))
        if isinstance(max_gap, _dt.timedelta):
            return float(np.timedelta64(max_gap).astype("timedelta64[ns]") / np.timedelta64(1, "ns"))
        if isinstance(max_gap, str):
            try:
                import pandas as pd
                return float(pd.to_timedelta(max_gap).to_timedelta64().astype("timedelta64[ns]") / np.timedelta64(1, "ns"))
            except Exception as e:
                raise ValueError(f"Could not convert {max_gap!r} to timedelta64 using pandas.to_timedelta") from e
        if hasattr(max_gap, "to_timedelta64"):                       # pandas.Timedelta
            return float(max_gap.to_timedelta64().astype("timedelta64[ns]") / np.timedelta64(1, "ns"))
        raise TypeError("Expected value of type str, pandas.Timedelta, datetime.timedelta or numpy.timedelta64, "
                        f"but received {type(max_gap).__name__}")
    if not isinstance(max_gap, (numbers.Number, np.number)) or isinstance(max_gap, bool):
        raise TypeError(f"Expected integer or floating point max_gap on a numeric axis. Received {type(max_gap).__name__}.")
    return float(max_gap)


class PadSpec:
    """interp_index + max_gap of one call, with the device copy of the abscissa made on first use."""

    def __init__(self, time, max_gap):
        self.x = interp_index(time)
        self.max_gap = max_gap_value(max_gap, time)
        self._dev = None

    def apply(self, d_ts_ptr, itemsize, T, C, ld=None, stream=0):
        """interpolate the device series (T, C) in place"""
        from ._lib import hip
        from .device import DeviceBuffer
        if T != self.x.shape[0]:
            raise ValueError("series and time axis differ in length")
        if C == 0 or T == 0:
            return
        if self._dev is None:
            self._dev = DeviceBuffer.from_array(self.x)
        hip().pad_gaps(int(d_ts_ptr), int(itemsize), int(T), int(C), int(C if ld is None else ld), self._dev.ptr,
                       self.max_gap, stream)

    def free(self):
        if self._dev is not None:
            self._dev.free()
            self._dev = None


def make_pad(maxPadLength, time):
    """None when no interpolation was asked for (``if maxPadLength:`` in the reference)"""
    if not maxPadLength:
        return None
    return PadSpec(time, maxPadLength)
