"""Multi-GPU threshold() and detect(): cells shard across ranks, one gather at the end.

The reference treats every grid cell as an independent task (xmhw/xmhw.py:184-197) and collects
with ``dask.compute(climls)`` + ``xr.concat(dim='cell')`` (xmhw/xmhw.py:197, :210-211).  Here the
stacked grid columns are cut into ``size`` contiguous blocks, one process per GPU runs the HIP path
on its block (mask, compaction, climatology -- nothing is exchanged during compute), and the
(D, block) float64 results travel once, device to device, to the root: ``xmhw_gather_blocks`` of
the C ABI, i.e. grouped ncclSend / ncclRecv over RCCL / xGMI on the kernels' own output buffers.
The assembled result is bit-identical to the single-GPU one (same per-cell arithmetic).

No PyTorch anywhere: the communicator is the C ABI's (``xmhw_comm_*``); the 128-byte RCCL id is
handed round by ``xmhw_amd.bootstrap`` (a TCP socket).  Everything the sharded entry points need
from the group goes through a small transport object (rank, size, all-gather of an int64 / of byte
masks, gather of float64 blocks, an error agreement), so that the CPU tests can drive the same code
with a stand-in transport (tests/gloo_transport.py) and stand-in device stages.

Only a rank's own column block of ``temp`` is ever touched: pass an ``np.memmap`` (or any ndarray
view of a file) and each process reads 1/size of it.

Failure on one rank between two collectives would leave the others blocked: every local stage runs
under ``transport.agree()``, which all-gathers an error flag and raises on ALL ranks together.
"""
import os

import numpy as np

from . import api
from .detect import INTER_VARIABLES, _detect
from .device import DeviceBuffer, calc_clim_device, calc_clim_grid_device
from .exception import XmhwException
from ._lib import hip


def slab_bounds(ncells, world_size):
    """Contiguous, balanced [lo, hi) per rank (first ``ncells % world`` slabs one larger)."""
    base, extra = divmod(int(ncells), int(world_size))
    bounds, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


class RcclTransport:
    """The product transport: an ``xmhw_comm`` (RCCL) on this process's current device."""

    def __init__(self, rank, size, unique_id, stream=0):
        self._h = hip()
        self.rank, self.size = int(rank), int(size)
        self.stream = stream
        self._comm = self._h.comm_create(self.rank, self.size, unique_id)

    def close(self):
        if self._comm:
            self._h.comm_destroy(self._comm)
            self._comm = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- metadata -------------------------------------------------------------------------
    def allgather_i64(self, value):
        return np.asarray(self._h.comm_allgather_i64(self._comm, int(value), self.stream), dtype=np.int64)

    def allgather_u8(self, arr):
        """variable-length uint8 vectors -> list of the ranks' vectors (every rank gets all)"""
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        counts = self.allgather_i64(arr.shape[0])
        width = int(max(int(counts.max()), 1))
        send = DeviceBuffer(width)
        recv = DeviceBuffer(width * self.size)
        try:
            if arr.shape[0]:
                self._h.memcpy_h2d(send.ptr, arr)
            self._h.comm_allgather_bytes(self._comm, send.ptr, recv.ptr, width, self.stream)
            self._h.stream_sync(self.stream)
            flat = recv.to_array((self.size, width), np.uint8)
        finally:
            send.free()
            recv.free()
        return [flat[r, : int(counts[r])].copy() for r in range(self.size)]

    def agree(self, error=None, value=0):
        """Collective: raise XmhwException on every rank if any rank reports an error; otherwise every rank gets
        every rank's ``value`` (a non-negative int64: a survivor count, say) from the same exchange."""
        vals = self.allgather_i64(-1 if error is not None else int(value))
        if (vals < 0).any():
            bad = [int(r) for r in np.nonzero(vals < 0)[0]]
            if error is not None:
                raise error
            raise XmhwException(f"sharded run aborted: rank(s) {bad} failed in their local stage")
        return vals

    # -- bulk -------------------------------------------------------------------------------
    def gather_columns(self, block, rows, dst=0, counts=None):
        """Per-rank dense (rows, cols_r) float64 blocks -> (rows, sum cols_r) host array on ``dst``
        (blocks side by side in rank order), None elsewhere.  ``block`` is a DeviceBuffer holding
        the dense block (the kernels' output, nothing is copied before it travels) or a host array.
        ``counts``: every rank's block width when all ranks already know them (slab_bounds); without it the
        widths are all-gathered first."""
        own = None
        if isinstance(block, DeviceBuffer):
            cols = (block.nbytes // (8 * rows) if rows else 0) if counts is None else int(counts[self.rank])
            dev = block
        else:
            a = np.ascontiguousarray(block, dtype=np.float64)
            cols = a.shape[1] if a.ndim == 2 else 0
            dev = own = DeviceBuffer.from_array(a) if a.size else None
        counts = self.allgather_i64(cols) if counts is None else np.asarray(counts, dtype=np.int64)
        if int(counts[self.rank]) != cols:
            raise XmhwException(f"rank {self.rank}: block of {cols} columns, {int(counts[self.rank])} announced")
        total = int(counts.sum())
        recv = None
        try:
            if self.rank == dst:
                recv = DeviceBuffer(8 * rows * max(total, 1))
            self._h.gather_blocks(self._comm, dev.ptr if dev is not None else 0, rows, cols, recv.ptr if recv else 0,
                                  counts if self.rank == dst else np.zeros(0, dtype=np.int64), dst, self.stream)
            if self.rank != dst:
                self._h.stream_sync(self.stream)
                return None
            # the blocks leave the device behind the collective on the same stream: one 2-D copy per rank (block r
            # goes straight into columns [c0_r, c0_r + cols_r) of the result), ONE wait at the end
            out = np.empty((rows, total), dtype=np.float64)
            off_bytes, c0 = 0, 0
            for r in range(self.size):
                n = int(counts[r])
                if n and rows:
                    self._h.memcpy2d_d2h_async(out, c0, n, recv.ptr + off_bytes, self.stream)
                off_bytes += 8 * rows * n
                c0 += n
            self._h.stream_sync(self.stream)
            return out
        finally:
            if own is not None:
                own.free()
            if recv is not None:
                recv.free()

    def gather_rows(self, table, dst=0):
        """Per-rank (n_r, ncol) float64 tables -> (sum n_r, ncol) on ``dst`` (rank order), None elsewhere."""
        table = np.ascontiguousarray(table, dtype=np.float64)
        ncol = table.shape[1]
        flat = self.gather_columns(table.reshape(1, -1), 1, dst)
        return None if flat is None else flat.reshape(-1, ncol)


def init_rccl(rank=None, size=None, local_rank=None, addr=None, port=None):
    """One process per GPU: select the device, share the RCCL id, build the communicator.
    Defaults come from the launcher's environment (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR;
    the id travels on XMHW_BOOTSTRAP_PORT, default MASTER_PORT + 17)."""
    from . import bootstrap
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    size = int(os.environ.get("WORLD_SIZE", "1")) if size is None else int(size)
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank))) if local_rank is None else int(local_rank)
    h = hip()
    h.set_device(local_rank)
    uid = bootstrap.share_bytes(rank, size, h.comm_unique_id if rank == 0 else None, addr=addr, port=port)
    return RcclTransport(rank, size, uid)


def _stage(transport, fn, value=None):
    """Run a rank-local stage; if it raises on any rank, all ranks raise together.  ``value`` (a function of
    the stage's result giving a non-negative int) rides the same exchange: returns (result, every rank's value)."""
    err = None
    out = None
    v = 0
    try:
        out = fn()
        if value is not None:
            v = int(value(out))
    except Exception as e:      # noqa: BLE001 -- reported to every rank, then re-raised
        err = e if isinstance(e, XmhwException) else XmhwException(f"rank {transport.rank}: {type(e).__name__}: {e}")
    try:
        vals = transport.agree(err, v)
    except Exception:
        _free_buffers(out)          # another rank failed: what this rank's stage left on the device goes too
        raise
    return out if value is None else (out, vals)


def _free_buffers(obj):
    if isinstance(obj, DeviceBuffer):
        obj.free()
    elif isinstance(obj, dict):
        for v in obj.values():
            _free_buffers(v)
    elif isinstance(obj, (tuple, list)):
        for v in obj:
            _free_buffers(v)


def make_sharded_compute(transport, dst=0, compute=None):
    """A drop-in for device.calc_clim_device that computes only this rank's slab of the COMPACT
    cells and gathers.  ``compute`` defaults to the HIP path; the CPU tests inject a stand-in."""
    inner = compute or calc_clim_device

    def sharded(ts, doy, pctile, windowHalfWidth, smoothPercentile, smoothPercentileWidth,
                tstep, coldSpells=False, **extra):          # extra: e.g. pad= (maxPadLength), per-column work
        C = ts.shape[1]
        lo, hi = slab_bounds(C, transport.size)[transport.rank]
        doys = np.unique(np.asarray(doy, dtype=np.int64))

        def local():
            if hi <= lo:
                return np.empty((doys.shape[0], 0)), np.empty((doys.shape[0], 0))
            _, th, se = inner(np.ascontiguousarray(ts[:, lo:hi]), doy, pctile, windowHalfWidth, smoothPercentile,
                              smoothPercentileWidth, tstep, coldSpells, **extra)
            return th, se

        th_r, se_r = _stage(transport, local)
        D = doys.shape[0]
        both = transport.gather_columns(np.concatenate([th_r, se_r], axis=0), 2 * D, dst)
        if transport.rank != dst:
            th = se = np.full((D, C), np.nan)       # placeholder of the right shape
        else:
            th, se = both[:D], both[D:]
        return doys, th, se

    return sharded


def make_sharded_grid_compute(transport, dst=0, grid_compute=None):
    """A drop-in for device.calc_clim_grid_device: every rank takes a contiguous block of the
    UNCOMPACTED stacked columns, masks / compacts / computes it on its own GPU (no rank runs
    land_check() over the whole grid, no rank touches another rank's columns of ``stacked``).  What is
    exchanged: ONE int64 per rank (the agreement that every local stage succeeded, carrying the rank's number of
    surviving cells, so that an all-land grid raises on every rank alike) and ONE gather of the (2D, block)
    results -- the reference's single collect (xmhw/xmhw.py:197, :210-211).  The block widths are slab_bounds(),
    known to everybody; the surviving cells are read off the gathered block on the root (a dropped cell is an
    all-NaN column, a kept one has at least one pooled sample and therefore a value).  The other ranks get
    ``None`` for the arrays: they have nothing to assemble."""
    inner = grid_compute or calc_clim_grid_device

    def sharded(stacked, doy, anynans, pctile, windowHalfWidth, smoothPercentile, smoothPercentileWidth,
                tstep, coldSpells=False, **extra):
        N = stacked.shape[1]
        bounds = slab_bounds(N, transport.size)
        lo, hi = bounds[transport.rank]

        def local():
            res = inner(stacked, doy, anynans, pctile, windowHalfWidth, smoothPercentile,
                        smoothPercentileWidth, tstep, coldSpells, columns=(lo, hi), **extra)
            # the block must be as wide as the slab every rank expects from this one: checked HERE, inside the stage
            # whose outcome all ranks agree on, so that a mismatch raises everywhere instead of leaving the others
            # inside the gather (gather_columns' own check is local to one rank)
            _, doys_, th_, _ = res
            width = (th_.nbytes // (8 * 2 * doys_.shape[0]) if doys_.shape[0] else 0) if isinstance(th_, DeviceBuffer) \
                else np.asarray(th_).shape[1]
            if width != hi - lo:
                if isinstance(th_, DeviceBuffer):
                    th_.free()
                raise XmhwException(f"rank {transport.rank}: block of {width} columns for a slab of {hi - lo}")
            return res

        th_r = None
        try:
            (keep_r, doys, th_r, se_r), kept = _stage(transport, local, value=lambda res: int(np.count_nonzero(res[0])))
            D = doys.shape[0]
            if int(kept.sum()) == 0:
                raise XmhwException("All points of grid are either land or NaN")     # on every rank alike
            # the blocks come back on the grid (NaN at dropped cells): each has its slab's width
            if isinstance(th_r, DeviceBuffer):
                block = th_r              # device-resident (2D, w) block from the HIP stage
            else:
                block = np.concatenate([th_r, se_r], axis=0)
            both = transport.gather_columns(block, 2 * D, dst, counts=[b - a for a, b in bounds])
        finally:
            if isinstance(th_r, DeviceBuffer):
                th_r.free()
        if transport.rank != dst:
            return None, doys, None, None
        th, se = both[:D], both[D:]
        # the surviving cells, read off the gathered blocks: a dropped cell is NaN in every row of thresh AND seas.
        # (A kept cell can have an all-NaN thresh column -- pools that hold both +inf and -inf interpolate to NaN --
        # so thresh alone would drop its coordinate line on the root and the N-rank grid would differ from the 1-rank
        # one; seas of such a cell is NaN as well only if EVERY pool of the year holds both infinities, which a series
        # with a finite sample somewhere in every 11-day window cannot.)
        keep = ~(np.isnan(th).all(axis=0) & np.isnan(se).all(axis=0))
        return keep, doys, th, se

    return sharded


def _device_grid_compute(stacked, doy, anynans, pctile, windowHalfWidth, smoothPercentile, smoothPercentileWidth,
                         tstep, coldSpells=False, columns=None, **extra):
    """calc_clim_grid_device for one rank's block with the results LEFT ON THE DEVICE as one dense
    (2D, w) float64 block (thresh rows, then seas rows), ready for xmhw_gather_blocks."""
    return calc_clim_grid_device(stacked, doy, anynans, pctile, windowHalfWidth, smoothPercentile,
                                 smoothPercentileWidth, tstep, coldSpells, columns=columns, device_block=True, **extra)


def threshold_sharded(temp, transport, dst=0, _compute=None, _grid_compute=None, **kwargs):
    """threshold() over all ranks of ``transport``; every rank passes the same ``temp`` description
    (only its own column block of the values is read); rank ``dst`` gets the Dataset, the others
    None.  ``_compute`` / ``_grid_compute`` are test hooks (device-stage stand-ins)."""
    if _compute is not None:
        ds = api._threshold(temp, make_sharded_compute(transport, dst, _compute), **kwargs)
    else:
        # a single-point series has no grid to split: the compact-array path handles it
        ds = api._threshold(temp, make_sharded_compute(transport, dst, None),
                            grid_compute=make_sharded_grid_compute(transport, dst, _grid_compute or _device_grid_compute),
                            **kwargs)
    return ds if transport.rank == dst else None


def _gather_planes(transport, planes, T, C, sample_dtype, dst=0):
    """The per-step variables of detect(intermediate=True) (xmhw/identify.py:405-409, xmhw/xmhw.py:354-356): every
    rank hands in its (T, cells of its block) planes -- None for a block without cells -- and rank ``dst`` gets the
    (T, C) planes of the whole grid; the other ranks get zero planes of the same shapes and dtypes."""
    from .detect_front import INTERMEDIATE_U8
    booleans = set(INTERMEDIATE_U8) | {"bthresh"}
    out = {}
    for k in INTER_VARIABLES:
        blk = planes[k] if planes is not None else np.zeros((T, 0))
        full = transport.gather_columns(np.asarray(blk, dtype=np.float64), T, dst)
        if transport.rank != dst:
            out[k] = np.zeros((T, C), dtype=bool if k in booleans else np.float64)
        elif k in booleans:
            out[k] = full != 0
        elif k == "ts":
            out[k] = full.astype(sample_dtype)          # float32 values survive the float64 transport
        else:
            out[k] = full
    return out


def make_sharded_detect(transport, dst=0, compute=None):
    """A drop-in for detect_front.detect_cells that runs only this rank's slab of cells and gathers
    the event tables (variable length per rank) and, if asked for, the per-step columns.  Rank dst
    returns the full result, the others an empty one."""
    from .detect_front import EVENT_COLUMNS, detect_cells

    inner = compute or detect_cells
    ncol = len(EVENT_COLUMNS)

    def sharded(ts, seas, thresh, doy, doys, minDuration=5, joinGaps=True, maxGap=2, coldSpells=False,
                intermediate=False, **extra):
        T, C = ts.shape
        lo, hi = slab_bounds(C, transport.size)[transport.rank]

        def local():
            if hi <= lo:
                return dict(table=np.zeros((0, ncol)), offsets=np.zeros(1, dtype=np.int64), inter=None)
            return inner(np.ascontiguousarray(ts[:, lo:hi]), np.ascontiguousarray(seas[:, lo:hi]),
                         np.ascontiguousarray(thresh[:, lo:hi]), doy, doys, minDuration, joinGaps, maxGap,
                         coldSpells, intermediate, **extra)

        res = _stage(transport, local)
        counts = np.diff(res["offsets"]).astype(np.float64)[None, :]
        all_counts = transport.gather_columns(counts, 1, dst)
        table = transport.gather_rows(res["table"], dst)
        inter = _gather_planes(transport, res["inter"], T, C, ts.dtype, dst) if intermediate else None
        if transport.rank != dst:
            return dict(table=np.zeros((0, ncol)), offsets=np.zeros(C + 1, dtype=np.int64), inter=inter)
        offsets = np.zeros(C + 1, dtype=np.int64)
        np.cumsum(all_counts[0].astype(np.int64), out=offsets[1:])
        return dict(table=table, offsets=offsets, inter=inter)

    return sharded


def make_sharded_detect_grid(transport, dst=0, grid_compute=None):
    """A drop-in for detect_front.detect_grid: every rank masks and compacts its own block of the
    uncompacted series columns on its GPU; the survivor counts are all-gathered (a block's cells
    pair up with the climatology columns at the offset of the blocks before it), then the event
    tables travel as in make_sharded_detect and the keep masks are all-gathered."""
    from .detect_front import EVENT_COLUMNS, detect_grid

    inner = grid_compute or detect_grid
    ncol = len(EVENT_COLUMNS)

    def sharded(stacked, anynans, seas, thresh, doy, doys, minDuration=5, joinGaps=True, maxGap=2, coldSpells=False,
                intermediate=False, clim_stacked=False, **extra):
        T, N = stacked.shape
        lo, hi = slab_bounds(N, transport.size)[transport.rank]

        # the positional pairing with the climatology needs every rank's survivor count in the middle
        # of the local stage: that exchange is itself a collective, so a rank that fails before it
        # still takes part (with a count of 0) and reports its error at the agreement that follows
        state = {"exchanged": False}

        def exchange(n_mine):
            state["exchanged"] = True
            counts = transport.allgather_i64(n_mine)
            return int(counts[: transport.rank].sum()), int(counts.sum())

        def local():
            try:
                return inner(stacked, anynans, seas, thresh, doy, doys, minDuration, joinGaps, maxGap, coldSpells,
                             intermediate, clim_stacked=clim_stacked, columns=(lo, hi), exchange=exchange, **extra)
            except Exception:
                if not state["exchanged"]:
                    exchange(0)
                raise

        res = _stage(transport, local)
        keep = np.concatenate(transport.allgather_u8(res["keep"].astype(np.uint8))) != 0
        if not keep.any():
            raise XmhwException("All points of grid are either land or NaN")
        counts = transport.gather_columns(np.diff(res["offsets"]).astype(np.float64)[None, :], 1, dst)
        table = transport.gather_rows(res["table"], dst)
        C = int(keep.sum())
        # (intermediate=True: the per-step planes travel as the event tables do, one gather per variable)
        from .device import is_packed, native_float
        sample_dtype = stacked.decoded_dtype if is_packed(stacked) else native_float(np.zeros(0, stacked.dtype)).dtype
        inter = _gather_planes(transport, res["inter"], T, C, sample_dtype, dst) if intermediate else None
        if transport.rank != dst:
            return dict(table=np.zeros((0, ncol)), offsets=np.zeros(C + 1, dtype=np.int64), inter=inter, keep=keep)
        offsets = np.zeros(C + 1, dtype=np.int64)
        np.cumsum(counts[0].astype(np.int64), out=offsets[1:])
        return dict(table=table, offsets=offsets, inter=inter, keep=keep)

    return sharded


def detect_sharded(temp, th, se, transport, dst=0, _compute=None, _grid_compute=None, **kwargs):
    """detect() over all ranks of ``transport``; rank ``dst`` gets what detect() returns, the others
    None.  The ranks split the uncompacted grid columns and each masks / compacts its own block on
    its GPU (survivor counts are exchanged so that every block finds its climatology columns);
    ``_compute`` (test hook, compact-array stand-in) selects the older path in which every rank runs
    land_check() on the host and the compact cells are split.  intermediate=True (xmhw/xmhw.py:354-356,
    :479-484): rank ``dst`` gets (mhw, intermediate Dataset), the per-step planes gathered block by block."""
    if _compute is not None:
        out = _detect(temp, th, se, make_sharded_detect(transport, dst, _compute), **kwargs)
    else:
        out = _detect(temp, th, se, make_sharded_detect(transport, dst, None),
                      grid_compute=make_sharded_detect_grid(transport, dst, _grid_compute), **kwargs)
    return out if transport.rank == dst else None
