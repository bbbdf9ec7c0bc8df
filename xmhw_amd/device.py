"""Thin object layer over the C ABI: device buffers, plans, and the two
device-side stages of calc_clim() (xmhw/xmhw.py:250-307)."""
import numpy as np

from ._lib import hip
from .exception import XmhwException

KERNELS = {"auto": 0, "ring": 1, "generic": 2}


class DeviceBuffer:
    """Caller-owned HBM allocation (hipMalloc through the C ABI)."""

    def __init__(self, nbytes):
        self._h = hip()
        self.nbytes = int(nbytes)
        self.ptr = self._h.malloc(self.nbytes) if self.nbytes else 0

    @classmethod
    def from_array(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        if a.nbytes:
            b._h.memcpy_h2d(b.ptr, a)
        return b

    def to_array(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        if out.nbytes > self.nbytes:
            raise ValueError("device buffer smaller than requested array")
        if out.nbytes:
            self._h.memcpy_d2h(out, self.ptr)
        return out

    def free(self):
        if self.ptr:
            self._h.free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Plan:
    """Everything derived from the doy labels (xmhw_plan_* in the C ABI)."""

    def __init__(self, doy, window_half_width, kernel="auto", nchunks=0, narrowing=True):
        self._h = hip()
        doy = np.ascontiguousarray(doy, dtype=np.int32)
        try:
            self.handle = self._h.plan_create(doy, int(window_half_width))
        except self._h.InvalidArgument as e:
            raise XmhwException(str(e)) from e
        self._h.plan_set_kernel(self.handle, KERNELS[kernel])
        self._h.plan_set_chunks(self.handle, int(nchunks))
        self._h.plan_set_narrowing(self.handle, int(bool(narrowing)))
        info = self._h.plan_info(self.handle)
        self.D = info["D"]
        self.ntracks = info["ntracks"]
        self.kernel = {v: k for k, v in KERNELS.items()}.get(info["kernel"], "unsupported")
        self.nsteps = info["nsteps"]
        self.step_min = info["step_min"]
        self.doys = self._h.plan_doys(self.handle).astype(np.int64)
        self.T = doy.shape[0]
        self.w = int(window_half_width)

    def narrowed(self):
        """True if the last float64 clim_raw() of this plan ran on the float32 ring kernel (every
        sample float32-representable)."""
        return bool(self._h.plan_narrowed(self.handle))

    def table(self, years_per_lane):
        return self._h.plan_table(self.handle, years_per_lane)

    def destroy(self):
        if getattr(self, "handle", 0):
            self._h.plan_destroy(self.handle)
            self.handle = 0

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def clim_raw(plan, ts_dev, itemsize, C, q, negate, thresh_dev, seas_dev, ld=None, ldo=None, stream=0):
    h = hip()
    try:
        h.clim_raw(plan.handle, ts_dev.ptr if hasattr(ts_dev, "ptr") else int(ts_dev), itemsize, C,
                   C if ld is None else ld, float(q), int(bool(negate)),
                   thresh_dev.ptr if hasattr(thresh_dev, "ptr") else int(thresh_dev),
                   seas_dev.ptr if hasattr(seas_dev, "ptr") else int(seas_dev),
                   C if ldo is None else ldo, stream)
    except h.InvalidArgument as e:
        raise XmhwException(str(e)) from e


def clim_finish(plan, th_in, se_in, C, feb29_fix, smooth, width, th_out, se_out, ldo=None, stream=0):
    h = hip()
    p = lambda b: b.ptr if hasattr(b, "ptr") else int(b)
    try:
        h.clim_finish(plan.handle, p(th_in), p(se_in), C, C if ldo is None else ldo, int(bool(feb29_fix)),
                      int(bool(smooth)), int(width), p(th_out), p(se_out), stream)
    except h.InvalidArgument as e:
        raise XmhwException(str(e)) from e


def calc_clim_device(ts, doy, pctile, windowHalfWidth, smoothPercentile, smoothPercentileWidth,
                     tstep, coldSpells=False, kernel="auto", nchunks=0, max_batch_bytes=32 << 30,
                     narrowing=True):
    """calc_clim() (xmhw/xmhw.py:250-307) for all cells of a dense host (T, C)
    array on the GPU.  Returns (doys[D] int64, thresh[D, C], seas[D, C]).

    Cells are independent, so the array is processed in contiguous cell batches of at
    most ``max_batch_bytes`` of input (one plan, device buffers reused); the result is
    identical to a single call.  PCIe-inclusive: the input is copied to the device.
    float64 input whose samples are all float32-representable runs on the float32 ring kernel
    (decided on the device, see xmhw_plan_set_narrowing); narrowing=False forces the float64 one.
    """
    ts = np.asarray(ts)
    if ts.dtype not in (np.float32, np.float64):
        ts = ts.astype(np.float64)
    if ts.ndim != 2:
        raise XmhwException("calc_clim_device expects a (time, cell) array")
    T, C = ts.shape
    h = hip()
    plan = Plan(doy, windowHalfWidth, kernel=kernel, nchunks=nchunks, narrowing=narrowing)
    bufs = []
    try:
        D = plan.D
        th = np.empty((D, C), dtype=np.float64)
        se = np.empty((D, C), dtype=np.float64)
        if C == 0:
            return plan.doys.copy(), th, se
        isz = ts.dtype.itemsize
        cb = int(max(1, min(C, max_batch_bytes // max(1, T * isz))))
        feb29_fix = tstep is False
        finish = feb29_fix or smoothPercentile
        d_ts = DeviceBuffer(isz * T * cb)
        raw_th, raw_se = DeviceBuffer(8 * D * cb), DeviceBuffer(8 * D * cb)
        bufs += [d_ts, raw_th, raw_se]
        if finish:
            out_th, out_se = DeviceBuffer(8 * D * cb), DeviceBuffer(8 * D * cb)
            bufs += [out_th, out_se]
        else:
            out_th, out_se = raw_th, raw_se
        for lo in range(0, C, cb):
            n = min(cb, C - lo)
            slab = np.ascontiguousarray(ts[:, lo:lo + n])
            h.memcpy_h2d(d_ts.ptr, slab)
            clim_raw(plan, d_ts, isz, n, pctile / 100.0, coldSpells, raw_th, raw_se)
            if finish:
                clim_finish(plan, raw_th, raw_se, n, feb29_fix, smoothPercentile, smoothPercentileWidth,
                            out_th, out_se)
            h.stream_sync(0)
            bt = np.empty((D, n), dtype=np.float64)
            bs = np.empty((D, n), dtype=np.float64)
            h.memcpy_d2h(bt, out_th.ptr)
            h.memcpy_d2h(bs, out_se.ptr)
            th[:, lo:lo + n] = bt
            se[:, lo:lo + n] = bs
        return plan.doys.copy(), th, se
    finally:
        for b in bufs:
            b.free()
        plan.destroy()
