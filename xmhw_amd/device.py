"""Thin object layer over the C ABI: device buffers, plans, and the two
device-side stages of calc_clim() (xmhw/xmhw.py:250-307)."""
import numpy as np

from ._lib import hip
from .exception import XmhwException

KERNELS = {"auto": 0, "ring": 1, "generic": 2}


class PackedArray(np.ndarray):
    """A view of file bytes (netCDF classic: big-endian float32/float64, or CF-packed int16) that is
    decoded ON THE DEVICE: the array carries its decoding recipe through transposes / reshapes /
    slices (xmhw_amd.ingest builds it).  ``decode`` = dict(scale, offset, fill, out) with
    ``out`` the decoded dtype (float32 / float64); scale / fill may be None."""

    def __new__(cls, array, decode):
        obj = np.asarray(array).view(cls)
        obj.decode = dict(decode)
        return obj

    def __array_finalize__(self, obj):
        if obj is not None:
            self.decode = getattr(obj, "decode", None)

    @property
    def decoded_dtype(self):
        return np.dtype(self.decode["out"])


def is_packed(a):
    return isinstance(a, PackedArray) and a.decode is not None


def decode_through_device(a, lo=0, hi=None):
    """A file view (PackedArray: big-endian and / or CF-packed samples) as a decoded host array, decoded by the
    DEVICE kernel the climatology path uses (upload of the raw bytes, xmhw_decode, download) -- for the per-step
    detect() outputs, which compact on the host; [lo, hi) restricts it to a block of columns.  (The numpy restatement of the decoding the tests compare the
    kernel with lives in oracle/ingest_oracle.py; the product has no host decoder.)"""
    T, N = a.shape
    hi = N if hi is None else hi
    if hi <= lo:
        return np.zeros((T, 0), dtype=a.decoded_dtype)
    buf, isz = upload_columns(a, lo, hi)
    try:
        return buf.to_array((T, hi - lo), np.float32 if isz == 4 else np.float64)
    finally:
        buf.free()


def native_float(a):
    """The array as native-endian float32 or float64 (what the kernels read): float32 / float64 of
    either byte order keep their width (netCDF-3 and some GRIB decoders hand over big-endian
    arrays), every other dtype becomes float64."""
    if is_packed(a):
        return a
    a = np.asarray(a)
    if a.dtype.kind == "f" and a.dtype.itemsize in (4, 8):
        return a if a.dtype.isnative else a.astype(a.dtype.newbyteorder("="))
    return a.astype(np.float64)


# Large allocations are kept for the next call instead of being returned to the driver: hipMalloc of
# tens of GB costs about a second (page tables), which is as much as the upload of a global grid and
# far more than the kernels.  At most _POOL_SLOTS buffers of >= _POOL_MIN bytes and _POOL_MAX_BYTES in
# total are held (XMHW_AMD_POOL_GB overrides the total; 0 disables the cache); entries carry the
# device they were ALLOCATED on.  release_device_cache() frees them; every entry point of the C ABI
# that runs out of device memory calls it and retries once (xmhw_amd._lib).
import os as _os

import threading as _threading

_POOL = []
_POOL_LOCK = _threading.Lock()      # the upload thread of the slab pipeline allocates while the main thread frees
_POOL_SLOTS = 4
_POOL_MIN = 1 << 30
_POOL_MAX_BYTES = int(float(_os.environ.get("XMHW_AMD_POOL_GB", "96")) * (1 << 30))
_TRACE = _os.environ.get("XMHW_AMD_TRACE", "0") != "0"


def release_device_cache():
    """Return the cached large device buffers (see DeviceBuffer) to the driver; the number of
    bytes released."""
    h = hip()
    with _POOL_LOCK:
        entries = list(_POOL)
        del _POOL[:]
    freed = 0
    for cap, ptr, _ in entries:
        h.free(ptr)
        freed += cap
    return freed


def device_cache_bytes():
    with _POOL_LOCK:
        return sum(cap for cap, _, _ in _POOL)


class DeviceBuffer:
    """Caller-owned HBM allocation (hipMalloc through the C ABI); buffers of a GB or more are
    recycled through a small, size-capped pool."""

    def __init__(self, nbytes):
        self._h = hip()
        self.nbytes = int(nbytes)
        self.capacity = self.nbytes
        self.ptr = 0
        self.device = -1
        if not self.nbytes:
            return
        self.device = self._h.get_device()      # the device this buffer lives on, recorded at allocation
        if self.nbytes >= _POOL_MIN:
            with _POOL_LOCK:          # look-up and removal are one step: another thread may be in here too
                fit = [i for i, (cap, _, d) in enumerate(_POOL)
                       if d == self.device and self.nbytes <= cap <= 2 * self.nbytes]
                if fit:
                    i = min(fit, key=lambda k: _POOL[k][0])
                    cap, ptr, dev = _POOL.pop(i)
                    if cap >= self.nbytes and dev == self.device:
                        self.capacity, self.ptr = cap, ptr
                        return
                    _POOL.append((cap, ptr, dev))
        if _TRACE and self.nbytes >= _POOL_MIN:
            import time as _time
            _t0 = _time.perf_counter()
            self.ptr = self._h.malloc(self.nbytes)
            _trace(f"hipMalloc {self.nbytes / 1e9:.2f} GB (pool: {[round(c / 1e9, 2) for c, _, _ in _POOL]})", _t0)
            return
        self.ptr = self._h.malloc(self.nbytes)   # out of memory: the pool is drained and the call retried (_lib)

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
            kept = False
            if self.capacity >= _POOL_MIN:
                with _POOL_LOCK:
                    if (len(_POOL) < _POOL_SLOTS
                            and sum(c for c, _, _ in _POOL) + self.capacity <= _POOL_MAX_BYTES):
                        _POOL.append((self.capacity, self.ptr, self.device))
                        kept = True
            if kept:
                pass
            elif _TRACE and self.capacity >= _POOL_MIN:
                import time as _time
                _t0 = _time.perf_counter()
                self._h.free(self.ptr)
                _trace(f"hipFree {self.capacity / 1e9:.2f} GB (pool: {[round(c / 1e9, 2) for c, _, _ in _POOL]})", _t0)
            else:
                self._h.free(self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# include/xmhw_amd.h: XMHW_LAYOUT_*
LAYOUTS = {"auto": -2, "ring1": -1, "ring2_8lane": 8, "ring2_4lane": 10, "ring2_16lane": 12,
           "ring3_8lane": 20, "ring3_4lane": 21, "ring3_2lane": 22, "sorted": 40}


class Plan:
    """Everything derived from the doy labels (xmhw_plan_* in the C ABI)."""

    def __init__(self, doy, window_half_width, kernel="auto", nchunks=0, narrowing=True, ring2=None, layout=None):
        self._h = hip()
        doy = np.ascontiguousarray(doy, dtype=np.int32)
        try:
            self.handle = self._h.plan_create(doy, int(window_half_width))
        except self._h.InvalidArgument as e:
            raise XmhwException(str(e)) from e
        self._h.plan_set_kernel(self.handle, KERNELS[kernel])
        self._h.plan_set_chunks(self.handle, int(nchunks))
        self._h.plan_set_narrowing(self.handle, int(bool(narrowing)))
        if layout is None:
            layout = ring2              # (the rounds 2-3 name of the same argument)
        if layout is not None:          # None: the library default (environment XMHW_RING2)
            self._h.plan_set_layout(self.handle, int(LAYOUTS.get(layout, layout)))
        info = self._h.plan_info(self.handle)
        self.D = info["D"]
        self.ntracks = info["ntracks"]
        self.kernel = {v: k for k, v in KERNELS.items()}.get(info["kernel"], "unsupported")
        self.nsteps = info["nsteps"]
        self.step_min = info["step_min"]
        self.doys = self._h.plan_doys(self.handle).astype(np.int64)
        self.T = doy.shape[0]
        self.w = int(window_half_width)

    def layout_in_use(self):
        """the ring kernel + lane layout float32 input will run on (an XMHW_LAYOUT_* number; -1: round-1 / generic)"""
        return int(self._h.plan_layout_in_use(self.handle))

    ring2_in_use = layout_in_use        # deprecated name (rounds 2-3)

    def chunks_in_use(self, C):
        """chunks of the doy axis a launch over C cells is cut into (every workgroup walks D / chunks + 2w rows)"""
        return int(self._h.plan_chunks_in_use(self.handle, int(C)))

    def f64_mode(self):
        """layout variant of the 64-bit mode float64 samples will run on (-1: generic kernel)"""
        return int(self._h.plan_f64_mode(self.handle))

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


def clim_raw_packed(plan, codes_dev, C, q, negate, thresh_dev, seas_dev, scale_factor=None, add_offset=None, fill=None,
                    decoded="float32", big_endian=False, ld=None, ldo=None, stream=0):
    """xmhw_clim_raw_i16: the raw climatology of an int16-packed series read in place (CF packing attributes as xarray
    would apply them before threshold() sees the data; `decoded` = the dtype it would decode to).  Raises XmhwException
    where the sorted-list kernel does not serve the plan or the quantile: decode first (hip().decode) and use clim_raw."""
    h = hip()
    p = lambda b: b.ptr if hasattr(b, "ptr") else int(b)
    has_scale = scale_factor is not None or add_offset is not None
    try:
        h.clim_raw_i16(plan.handle, p(codes_dev), C, C if ld is None else ld, int(bool(big_endian)), int(has_scale),
                       1.0 if scale_factor is None else float(scale_factor), 0.0 if add_offset is None else float(add_offset),
                       int(fill is not None), 0 if fill is None else int(fill), 8 if str(decoded) in ("float64", "f64", "8") else 4,
                       float(q), int(bool(negate)), p(thresh_dev), p(seas_dev), C if ldo is None else ldo, stream)
    except (h.InvalidArgument, h.HipError) as e:       # (HipError: XMHW_ERR_UNSUPPORTED -- the plan is not the sorted kernel's)
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
                     narrowing=True, pad=None):
    """calc_clim() (xmhw/xmhw.py:250-307) for all cells of a dense host (T, C)
    array on the GPU.  Returns (doys[D] int64, thresh[D, C], seas[D, C]).

    Cells are independent, so the array is processed in contiguous cell batches of at
    most ``max_batch_bytes`` of input (one plan, device buffers reused); the result is
    identical to a single call.  PCIe-inclusive: the input is copied to the device.
    float64 input whose samples are all float32-representable runs on the float32 ring kernel
    (decided on the device, see xmhw_plan_set_narrowing); narrowing=False forces the float64 one.
    ``pad`` (padding.PadSpec): maxPadLength's interpolate_na, applied to the device copy of every batch.
    """
    ts = native_float(ts)
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
            if pad is not None:
                pad.apply(d_ts.ptr, isz, T, n)
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


# ---- pinned, double-buffered upload ------------------------------------------------------------
# A pageable hipMemcpy is synchronous and, from a memory-MAPPED file, pays a page fault per 4 KB on
# one thread (measured 12.9 GB/s end to end on a 60 GB file in /dev/shm).  Here row blocks of the
# slab are copied by a small thread pool into two page-locked staging buffers (the faults and the
# copy spread over the threads) and leave for the device by asynchronous DMA on a copy stream while
# the other buffer is being filled.
_STAGE = {}
_USE_STAGING = _os.environ.get("XMHW_AMD_STAGING", "1") != "0"
_USE_PREAD = _os.environ.get("XMHW_AMD_PREAD", "1") != "0"
_STAGE_BYTES = int(_os.environ.get("XMHW_AMD_STAGE_MB", "256")) << 20   # per buffer, two of them; 128-1024 MB measured alike, the small ones ramp up faster per slab
_STAGE_THREADS = int(_os.environ.get("XMHW_AMD_STAGE_THREADS", "32"))


def _trace(label, t0):
    if _TRACE:
        import sys
        import time
        print(f"[xmhw_amd] {label}: {time.perf_counter() - t0:.3f} s", file=sys.stderr, flush=True)
_STAGE_MIN = 256 << 20


def _stage():
    h = hip()
    dev = h.get_device()
    st = _STAGE.get(dev)
    if st is None:
        from concurrent.futures import ThreadPoolExecutor
        ptrs = [h.host_alloc(_STAGE_BYTES) for _ in range(2)]
        st = dict(ptrs=ptrs, views=[h.host_view(p, _STAGE_BYTES) for p in ptrs],
                  events=[h.event_create(), h.event_create()], stream=h.stream_create(),
                  pool=ThreadPoolExecutor(_STAGE_THREADS), busy=[False, False])
        _STAGE[dev] = st
    return st


def release_staging():
    """free the page-locked staging buffers (kept between calls)"""
    h = hip()
    for st in _STAGE.values():
        st["pool"].shutdown(wait=True)
        for p in st["ptrs"]:
            h.host_free(p)
    _STAGE.clear()


def _file_window(src, file):
    """(fd, offset of src[0, 0] in the file, row pitch) if the 2-D view `src` lies inside the mapped file
    described by `file` (PackedArray.decode['file']) with contiguous rows, else None"""
    if not file or src.ndim != 2 or src.shape[0] == 0 or src.shape[1] == 0:
        return None
    if src.strides[1] != src.dtype.itemsize or src.strides[0] < src.shape[1] * src.dtype.itemsize:
        return None
    off = src.__array_interface__["data"][0] - file["address"]
    last = off + (src.shape[0] - 1) * src.strides[0] + src.shape[1] * src.dtype.itemsize
    if off < 0 or last > file["length"]:
        return None
    return file["fd"], off, src.strides[0]


def _staged_upload(dst_ptr, src, file=None):
    """src: 2-D host array view (rows contiguous, any row pitch) -> dense device array at dst_ptr.
    `file`: the view is a window of a mapped file -- its rows are then read with pread() straight into
    the staging buffers (xmhw_read_rows) instead of being copied out of the mapping."""
    h = hip()
    st = _stage()
    T, n = src.shape
    row_bytes = n * src.dtype.itemsize
    rows_per = max(1, _STAGE_BYTES // max(row_bytes, 1))
    pool = st["pool"]
    win = _file_window(src, file) if _USE_PREAD else None
    for k, r0 in enumerate(range(0, T, rows_per)):
        r1 = min(T, r0 + rows_per)
        b = k % 2
        if st["busy"][b]:
            h.event_sync(st["events"][b])                  # the DMA that last read this buffer is done
        view = st["views"][b][: (r1 - r0) * row_bytes].view(src.dtype).reshape(r1 - r0, n)
        step = max(1, -(-(r1 - r0) // _STAGE_THREADS))
        if win is not None:
            fd, off, pitch = win
            futs = [pool.submit(h.read_rows, fd, off + a * pitch, pitch, row_bytes, min(r1, a + step) - a,
                                st["ptrs"][b] + (a - r0) * row_bytes)
                    for a in range(r0, r1, step)]
        else:
            futs = [pool.submit(np.copyto, view[a - r0:min(r1, a + step) - r0], src[a:min(r1, a + step)])
                    for a in range(r0, r1, step)]
        for f in futs:
            f.result()
        h.memcpy_h2d_async(dst_ptr + r0 * row_bytes, st["ptrs"][b], (r1 - r0) * row_bytes, st["stream"])
        h.event_record(st["events"][b], st["stream"])
        st["busy"][b] = True
    h.stream_sync(st["stream"])
    st["busy"] = [False, False]


def upload_columns(stacked, lo, hi, raw=False):
    """Columns [lo, hi) of a C-contiguous host (T, N) array as a dense device (T, n) array of the
    DECODED dtype: a pitched upload of the raw bytes, then -- for file views (PackedArray: big-endian
    and / or CF-packed samples) -- the decode kernel.  Returns (DeviceBuffer, itemsize).
    ``raw=True``: the stored bytes as they are (int16 codes for the kernels that read them in place)."""
    import time as _time
    h = hip()
    T = stacked.shape[0]
    n = hi - lo
    raw_isz = stacked.dtype.itemsize
    _t0 = _time.perf_counter()
    d_raw = DeviceBuffer(raw_isz * T * n)
    try:
        if raw_isz * T * n >= _STAGE_MIN and _USE_STAGING:
            _staged_upload(d_raw.ptr, np.asarray(stacked)[:, lo:hi],
                           file=stacked.decode.get("file") if is_packed(stacked) else None)
        elif lo == 0 and hi == stacked.shape[1] and stacked.flags.c_contiguous:
            h.memcpy_h2d(d_raw.ptr, np.asarray(stacked))
        else:
            h.memcpy2d_h2d(d_raw.ptr, np.asarray(stacked), lo, n)
        _trace(f"upload columns [{lo},{hi}) {raw_isz * T * n / 1e9:.2f} GB", _t0)
        if raw or not is_packed(stacked):
            out, d_raw = d_raw, None
            return out, raw_isz
        d = stacked.decode
        out_isz = np.dtype(d["out"]).itemsize
        big = stacked.dtype.byteorder == ">" or (stacked.dtype.byteorder == "=" and not np.little_endian)
        kind = stacked.dtype.kind
        if not ((kind == "i" and raw_isz == 2) or (kind == "f" and raw_isz in (4, 8))):
            raise XmhwException(f"stored type {stacked.dtype} is not supported by the device decoder")
        plain = (kind == "f" and not big and d.get("scale") is None and d.get("fill") is None and out_isz == raw_isz)
        if plain:
            out, d_raw = d_raw, None
            return out, raw_isz
        d_out = DeviceBuffer(out_isz * T * n)
        try:
            h.decode(d_raw.ptr, raw_isz, int(big), T, n, n, d_out.ptr, out_isz, n, d.get("scale") is not None,
                     float(d.get("scale") or 1.0), float(d.get("offset") or 0.0), d.get("fill") is not None,
                     float(d.get("fill") or 0.0))
            h.stream_sync(0)
            out, d_out = d_out, None
            return out, out_isz
        finally:
            if d_out is not None:
                d_out.free()
    finally:
        if d_raw is not None:
            d_raw.free()


def mask_compact(d_raw, isz, T, n, anynans):
    """land_check()'s dropna + compaction (xmhw/identify.py:520-525) of a dense device (T, n) array:
    land_mask kernel, gather of the surviving columns.  Consumes d_raw.  Returns (DeviceBuffer holding
    the dense (T, n_keep) array or None if n_keep == 0, keep mask)."""
    h = hip()
    d_mask = d_idx = d_out = None
    try:
        d_mask = DeviceBuffer(n)
        h.land_mask(d_raw.ptr, isz, T, n, n, int(bool(anynans)), d_mask.ptr)
        h.stream_sync(0)
        keep = d_mask.to_array((n,), np.uint8) != 0
        nk = int(keep.sum())
        if nk == n:
            out, d_raw = d_raw, None
            return out, keep
        if nk == 0:
            return None, keep
        d_idx = DeviceBuffer.from_array(np.nonzero(keep)[0].astype(np.int64))
        d_out = DeviceBuffer(isz * T * nk)
        h.gather_cells(d_raw.ptr, isz, T, n, d_idx.ptr, nk, d_out.ptr, nk)
        h.stream_sync(0)
        out, d_out = d_out, None
        return out, keep
    finally:
        for b in (d_raw, d_mask, d_idx, d_out):
            if b is not None:
                b.free()


def packed_recipe(stacked):
    """the CF recipe of an int16 file view as xmhw_clim_raw_i16 / xmhw_land_mask_i16 take it"""
    d = stacked.decode
    fill = d.get("fill")
    # (a NaN or infinite _FillValue on an int16 variable -- int() of it would raise -- equals no code either)
    if fill is not None and not (np.isfinite(float(fill)) and float(fill) == int(fill) and -32768 <= int(fill) <= 32767):
        fill = None                                      # no int16 code equals it: nothing is missing
    big = stacked.dtype.byteorder == ">" or (stacked.dtype.byteorder == "=" and not np.little_endian)
    return dict(scale=d.get("scale"), offset=d.get("offset"), fill=None if fill is None else int(fill),
                decoded=np.dtype(d["out"]).name, big_endian=bool(big))


def mask_compact_codes(d_raw, T, n, anynans, recipe):
    """mask_compact() on int16 codes: a sample is missing when its code is the fill code"""
    h = hip()
    d_mask = d_idx = d_out = None
    try:
        d_mask = DeviceBuffer(n)
        h.land_mask_i16(d_raw.ptr, T, n, n, int(recipe["big_endian"]), int(recipe["fill"] is not None),
                        0 if recipe["fill"] is None else recipe["fill"], int(bool(anynans)), d_mask.ptr)
        h.stream_sync(0)
        keep = d_mask.to_array((n,), np.uint8) != 0
        nk = int(keep.sum())
        if nk == n:
            out, d_raw = d_raw, None
            return out, keep
        if nk == 0:
            return None, keep
        d_idx = DeviceBuffer.from_array(np.nonzero(keep)[0].astype(np.int64))
        d_out = DeviceBuffer(2 * T * nk)
        h.gather_cells(d_raw.ptr, 2, T, n, d_idx.ptr, nk, d_out.ptr, nk)
        h.stream_sync(0)
        out, d_out = d_out, None
        return out, keep
    finally:
        for b in (d_raw, d_mask, d_idx, d_out):
            if b is not None:
                b.free()


def compact_columns(stacked, lo, hi, anynans):
    """upload_columns() + mask_compact(): (DeviceBuffer or None, keep mask of the slab)."""
    d_raw, isz = upload_columns(stacked, lo, hi)
    return mask_compact(d_raw, isz, stacked.shape[0], hi - lo, anynans)


def device_itemsize(stacked):
    """item size of the samples the kernels will see for this host array"""
    return stacked.decoded_dtype.itemsize if is_packed(stacked) else stacked.dtype.itemsize


class ResidentSeries:
    """The compacted device copies of a stacked host series' column slabs, kept between two entry
    points that work on the SAME host array (threshold() then detect(): xmhw_amd.threshold_detect),
    so that the series crosses PCIe once.  An entry is ((lo, hi), DeviceBuffer or None, keep mask,
    item size).  The store only accepts a series that leaves room to work in
    (``XMHW_AMD_RESIDENT_FRACTION`` of the device's HBM, default 0.45: 130 GB on an MI355X); the
    consumer falls back to its own upload when the key does not match or nothing was retained."""

    def __init__(self):
        self.key = None
        self.slabs = []

    @staticmethod
    def key_of(stacked, anynans):
        a = np.asarray(stacked)
        decode = repr(getattr(stacked, "decode", None)) if is_packed(stacked) else ""
        return (a.__array_interface__["data"][0], a.shape, a.strides, a.dtype.str, decode, bool(anynans))

    def accepts(self, stacked):
        try:
            h = hip()
            hbm = h.device_info(h.get_device())["hbm_bytes"]
        except Exception:
            return False
        frac = float(_os.environ.get("XMHW_AMD_RESIDENT_FRACTION", "0.45"))
        return stacked.shape[0] * stacked.shape[1] * device_itemsize(stacked) <= frac * hbm

    def begin(self, key):
        self.free()
        self.key = key

    def add(self, bounds, d_ts, keep, isz):
        self.slabs.append((tuple(bounds), d_ts, keep, isz))

    def matches(self, key, c0, c1):
        return (self.key is not None and self.key == key and bool(self.slabs)
                and self.slabs[0][0][0] == c0 and self.slabs[-1][0][1] == c1)

    def free(self):
        for _, d_ts, _, _ in self.slabs:
            if d_ts is not None:
                d_ts.free()
        self.slabs = []
        self.key = None


class SlabPrefetcher:
    """Iterates over column slabs of a host array with the NEXT slab's upload (and decode) running in
    a background thread while the caller computes on the current one: pageable pitched copies are
    synchronous for the calling thread (about 54 GB/s on this platform), so overlap takes a second
    thread, not a second stream; the bindings release the GIL during copies."""

    def __init__(self, stacked, slabs, raw=False):
        import threading
        self._stacked, self._slabs, self._raw = stacked, list(slabs), bool(raw)
        self._threading = threading
        self._next = None
        self._start(0)

    def _start(self, i):
        if i >= len(self._slabs):
            self._next = None
            return
        box = {}
        dev = hip().get_device()

        def work():
            try:
                hip().set_device(dev)                      # the device is a per-thread setting
                box["out"] = upload_columns(self._stacked, *self._slabs[i], raw=self._raw)
            except BaseException as e:                     # noqa: BLE001 -- re-raised in the consumer
                box["err"] = e

        t = self._threading.Thread(target=work, daemon=True)
        t.start()
        self._next = (i, t, box)

    def __iter__(self):
        while self._next is not None:
            i, t, box = self._next
            t.join()
            if "err" in box:
                self._next = None
                raise box["err"]
            self._start(i + 1)
            yield self._slabs[i], box["out"]

    def close(self):
        """free an upload that was prepared but never consumed"""
        if self._next is not None:
            _, t, box = self._next
            t.join()
            if "out" in box:
                box["out"][0].free()
            self._next = None


def device_budget_bytes(fraction=0.6):
    """Working-set budget of the grid entry points: a fraction of the current device's HBM (a
    single slab - one contiguous upload - whenever the whole grid fits: 173 GB on an MI355X)."""
    h = hip()
    try:
        return int(fraction * h.device_info(h.get_device())["hbm_bytes"])
    except Exception:
        return 64 << 30


def _grid_batch(stacked, max_batch_bytes, per_cell_extra=0, pipeline=True):
    """cells per slab so that raw + decoded + compacted copies (+ per-cell extras) stay below the
    budget; large inputs are cut into at least 8 slabs so that uploads overlap compute"""
    if max_batch_bytes is None:
        max_batch_bytes = device_budget_bytes()
    T, N = stacked.shape
    per_cell = T * (stacked.dtype.itemsize + 2 * device_itemsize(stacked)) + per_cell_extra
    if pipeline:
        per_cell += T * stacked.dtype.itemsize            # the next slab's upload is resident too
    cb = int(max(1, min(N, max_batch_bytes // max(per_cell, 1))))
    if pipeline and T * N * stacked.dtype.itemsize >= (4 << 30):
        # the first slab's upload overlaps nothing: 16 slabs leave 1/16 of the PCIe time exposed
        # (XMHW_AMD_SLABS overrides; a slab still holds tens of thousands of cells)
        nslabs = max(1, int(_os.environ.get("XMHW_AMD_SLABS", "16")))
        cb = min(cb, max(-(-N // nslabs), min(N, 16384)))
    return cb


def _prefault(*arrays):
    """Touch every page of freshly allocated result arrays in a background thread: the kernel zeroes a
    page on its first write (6 GB for a global 0.25 degree grid, 0.23 s inside the first device-to-host
    copy otherwise), and that can happen while the first slab is still on its way to the device.
    The toucher writes zeros, so the caller JOINS the returned thread before the first result is stored."""
    import threading

    def touch():
        for a in arrays:
            flat = a.reshape(-1)
            flat[::512] = 0.0            # one write per 4 KB (numpy releases the GIL for the strided fill)

    t = threading.Thread(target=touch, daemon=True)
    t.start()
    return t


def _grid_block_on_device(plan, stacked, c0, c1, anynans, pctile, coldSpells, feb29_fix, smooth, width, pad=None):
    """One rank's column block: mask, compact, climatology, placement back on the block's grid --
    results stay on the device as a dense (2D, w) block.  Returns (keep, doys, DeviceBuffer, None)."""
    h = hip()
    D, w = plan.D, c1 - c0
    isz = device_itemsize(stacked)     # the DECODED item size (an int16 CF-packed archive arrives as float32)
    if w == 0:                         # more ranks than columns: a zero-byte block, nothing to send
        return np.zeros(0, dtype=bool), plan.doys.copy(), DeviceBuffer(0), None
    block = DeviceBuffer(8 * 2 * D * w)
    bufs = []
    try:
        d_ts, keep = compact_columns(stacked, c0, c1, anynans)
        n = int(keep.sum())
        if d_ts is None:
            h.scatter_cells(0, 2 * D, 1, 0, 0, block.ptr, w)          # all land: a block of NaN
            h.stream_sync(0)
            return keep, plan.doys.copy(), block, None
        bufs.append(d_ts)
        if pad is not None:
            pad.apply(d_ts.ptr, isz, stacked.shape[0], n)
        raw_th, raw_se = DeviceBuffer(8 * D * n), DeviceBuffer(8 * D * n)
        bufs += [raw_th, raw_se]
        clim_raw(plan, d_ts, isz, n, pctile / 100.0, coldSpells, raw_th, raw_se)
        finish = feb29_fix or smooth
        th_ptr, se_ptr = block.ptr, block.ptr + 8 * D * w
        if n == w:
            # no land in the block: the finish kernel (or a copy) writes the block directly
            if finish:
                clim_finish(plan, raw_th, raw_se, n, feb29_fix, smooth, width, th_ptr, se_ptr)
            else:
                idx = DeviceBuffer.from_array(np.arange(n, dtype=np.int64)); bufs.append(idx)
                h.scatter_cells(raw_th.ptr, D, n, idx.ptr, n, th_ptr, w)
                h.scatter_cells(raw_se.ptr, D, n, idx.ptr, n, se_ptr, w)
        else:
            out_th, out_se = raw_th, raw_se
            if finish:
                out_th, out_se = DeviceBuffer(8 * D * n), DeviceBuffer(8 * D * n)
                bufs += [out_th, out_se]
                clim_finish(plan, raw_th, raw_se, n, feb29_fix, smooth, width, out_th, out_se)
            idx = DeviceBuffer.from_array(np.nonzero(keep)[0].astype(np.int64)); bufs.append(idx)
            h.scatter_cells(out_th.ptr, D, n, idx.ptr, n, th_ptr, w)
            h.scatter_cells(out_se.ptr, D, n, idx.ptr, n, se_ptr, w)
        h.stream_sync(0)
        out, block = block, None
        return keep, plan.doys.copy(), out, None
    finally:
        for b in bufs:
            b.free()
        if block is not None:
            block.free()


def calc_clim_grid_device(stacked, doy, anynans, pctile, windowHalfWidth, smoothPercentile, smoothPercentileWidth,
                          tstep, coldSpells=False, kernel="auto", nchunks=0, max_batch_bytes=None,
                          narrowing=True, columns=None, scatter=True, device_block=False, resident=None, pad=None):
    """land_check() + calc_clim() for an UNCOMPACTED stacked host array (T, N): the land mask, the
    compaction and the placement of the results back on the grid (what unstack('cell') does) run on
    the device, so the host only hands the array over (for a global grid numpy's dropna and
    boolean-mask assignment cost tens of times the kernels).  Returns (keep[N] bool, doys[D],
    thresh[D, N], seas[D, N]) with NaN at the dropped cells; scatter=False returns the compact
    (D, C) arrays of the C = keep.sum() surviving cells instead.
    ``columns=(c0, c1)`` restricts the work to that column range (a rank's slab of a sharded run:
    keep and the arrays then cover c1 - c0 columns and an all-land slab is not an error).
    ``device_block=True`` (sharded runs) leaves the result on the device: the third return value is
    a DeviceBuffer holding ONE dense (2D, c1 - c0) float64 block -- the D thresh rows, then the D seas
    rows, on the grid with NaN at the dropped cells -- ready for xmhw_gather_blocks; the fourth is None.
    ``resident`` (a ResidentSeries) keeps every slab's compacted device copy for a later consumer of the
    same host array instead of freeing it.  ``pad`` (padding.PadSpec): maxPadLength's interpolate_na, applied
    to every compacted slab AFTER the mask (the reference interpolates after land_check, xmhw.py:137-160);
    a retained slab is the interpolated one."""
    rkey = ResidentSeries.key_of(stacked, anynans) if resident is not None else None
    if is_packed(stacked):
        if stacked.ndim != 2 or stacked.strides[1] != stacked.dtype.itemsize:
            raise XmhwException("a file view must have contiguous rows (time, cells)")
    elif not (isinstance(stacked, np.ndarray) and stacked.dtype.kind == "f" and stacked.dtype.itemsize in (4, 8)
              and stacked.dtype.isnative and stacked.flags.c_contiguous):
        stacked = np.ascontiguousarray(native_float(stacked))       # (a memmap of floats passes through untouched)
    T, N = stacked.shape
    h = hip()
    plan = Plan(doy, windowHalfWidth, kernel=kernel, nchunks=nchunks, narrowing=narrowing)
    D = plan.D
    isz = device_itemsize(stacked)
    feb29_fix = tstep is False
    finish = feb29_fix or smoothPercentile
    keeps, ths, ses = [], [], []
    pre = None
    touching = None
    try:
        cb = _grid_batch(stacked, max_batch_bytes, per_cell_extra=4 * D * 8)
        c0, c1 = (0, N) if columns is None else (int(columns[0]), int(columns[1]))
        if device_block:
            if c1 - c0 > cb:
                raise XmhwException(f"a block of {c1 - c0} columns does not fit this device in one piece "
                                    f"({cb} columns do): use more ranks")
            return _grid_block_on_device(plan, stacked, c0, c1, anynans, pctile, coldSpells, feb29_fix,
                                         smoothPercentile, smoothPercentileWidth, pad=pad)
        slabs = [(lo, min(c1, lo + cb)) for lo in range(c0, c1, cb)]
        many = len(slabs) > 1
        retain = resident is not None and resident.accepts(stacked)
        if retain:
            resident.begin(rkey)
        th = se = None
        if many and scatter:
            # results go straight to their columns of the full-width host arrays (pitched device-to-host copy)
            th = np.empty((D, c1 - c0))
            se = np.empty((D, c1 - c0))
            touching = _prefault(th, se)
        # int16-packed file views on plans and percentiles the sorted-list kernel serves: the CODES stay what is resident --
        # land mask, compaction and the climatology kernels read them in place (xmhw_clim_raw_i16), no decoded copy, no
        # decode pass.  Not with maxPadLength (interpolation needs decoded samples) nor when detect() will want the decoded
        # series next (resident).  XMHW_PACKED_DIRECT=0 turns it off.
        direct = (is_packed(stacked) and stacked.dtype.kind == "i" and stacked.dtype.itemsize == 2 and pad is None
                  and resident is None and kernel == "auto" and (pctile / 100.0 >= 0.85 or pctile / 100.0 <= 0.15) and plan.layout_in_use() == LAYOUTS["sorted"]
                  and _os.environ.get("XMHW_PACKED_DIRECT", "1") != "0")
        recipe = packed_recipe(stacked) if direct else None
        # slab k+1 is uploaded (and decoded) by a second thread while slab k computes
        pre = SlabPrefetcher(stacked, slabs, raw=direct)
        import time as _time
        _tl = _time.perf_counter()
        for (lo, hi), (d_up, up_isz) in pre:
            _trace(f"wait for slab [{lo},{hi})", _tl)
            _tl = _time.perf_counter()
            if direct:
                d_ts, keep = mask_compact_codes(d_up, T, hi - lo, anynans, recipe)
            else:
                d_ts, keep = mask_compact(d_up, up_isz, T, hi - lo, anynans)
            _trace("mask + compact", _tl)
            _tl = _time.perf_counter()
            keeps.append(keep)
            w = hi - lo
            if retain:
                resident.add((lo, hi), d_ts, keep, isz)
            if d_ts is None:
                if th is not None:
                    if touching is not None:
                        touching.join()            # the page toucher WRITES: it must be done before any result lands
                        touching = None
                    th[:, lo - c0:hi - c0] = np.nan
                    se[:, lo - c0:hi - c0] = np.nan
                continue
            n = int(keep.sum())
            bufs = [] if retain else [d_ts]
            try:
                if pad is not None:
                    pad.apply(d_ts.ptr, isz, T, n)
                raw_th, raw_se = DeviceBuffer(8 * D * n), DeviceBuffer(8 * D * n)
                bufs += [raw_th, raw_se]
                if direct:
                    clim_raw_packed(plan, d_ts, n, pctile / 100.0, coldSpells, raw_th, raw_se, scale_factor=recipe["scale"],
                                    add_offset=recipe["offset"], fill=recipe["fill"], decoded=recipe["decoded"],
                                    big_endian=recipe["big_endian"])
                else:
                    clim_raw(plan, d_ts, isz, n, pctile / 100.0, coldSpells, raw_th, raw_se)
                out_th, out_se = raw_th, raw_se
                if finish:
                    out_th, out_se = DeviceBuffer(8 * D * n), DeviceBuffer(8 * D * n)
                    bufs += [out_th, out_se]
                    clim_finish(plan, raw_th, raw_se, n, feb29_fix, smoothPercentile, smoothPercentileWidth,
                                out_th, out_se)
                h.stream_sync(0)
                if scatter and n != w:
                    d_idx = DeviceBuffer.from_array(np.nonzero(keep)[0].astype(np.int64)); bufs.append(d_idx)
                    full_th, full_se = DeviceBuffer(8 * D * w), DeviceBuffer(8 * D * w)
                    bufs += [full_th, full_se]
                    h.scatter_cells(out_th.ptr, D, n, d_idx.ptr, n, full_th.ptr, w)
                    h.scatter_cells(out_se.ptr, D, n, d_idx.ptr, n, full_se.ptr, w)
                    h.stream_sync(0)
                    out_th, out_se, n = full_th, full_se, w
                _trace("kernels + scatter", _tl)
                _tl = _time.perf_counter()
                if th is not None:
                    if touching is not None:
                        touching.join()            # (see above; normally long finished: it runs during the first upload)
                        touching = None
                    h.memcpy2d_d2h(th, lo - c0, w, out_th.ptr)
                    h.memcpy2d_d2h(se, lo - c0, w, out_se.ptr)
                    _trace("results to host (pitched)", _tl)
                    _tl = _time.perf_counter()
                else:
                    ths.append(((lo, hi), out_th.to_array((D, n), np.float64)))
                    ses.append(((lo, hi), out_se.to_array((D, n), np.float64)))
            finally:
                for b in bufs:
                    b.free()
        keep = np.concatenate(keeps) if keeps else np.zeros(0, dtype=bool)
        if not keep.any() and columns is None:
            raise XmhwException("All points of grid are either land or NaN")     # identify.py:527-528
        if th is not None:
            return keep, plan.doys.copy(), th, se
        if not scatter:
            if not ths:
                return keep, plan.doys.copy(), np.zeros((D, 0)), np.zeros((D, 0))
            return (keep, plan.doys.copy(), np.concatenate([a for _, a in ths], axis=1),
                    np.concatenate([a for _, a in ses], axis=1))
        if len(ths) == 1 and ths[0][0] == (c0, c1):
            return keep, plan.doys.copy(), ths[0][1], ses[0][1]          # one slab: no host copy at all
        th = np.full((D, c1 - c0), np.nan)
        se = np.full((D, c1 - c0), np.nan)
        for ((a, b), x), (_, y) in zip(ths, ses):
            th[:, a - c0:b - c0] = x
            se[:, a - c0:b - c0] = y
        return keep, plan.doys.copy(), th, se
    finally:
        if touching is not None:
            touching.join()
        if pre is not None:
            pre.close()
        plan.destroy()
