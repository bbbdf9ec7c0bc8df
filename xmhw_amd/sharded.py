"""Multi-GPU threshold() and detect(): cells shard across ranks, one gather at the end.

The reference treats every grid cell as an independent task
(xmhw/xmhw.py:184-197), so the path shards with no exchange during compute:
the compacted ocean-cell axis is cut into ``world_size`` contiguous slabs, each
rank runs the HIP path on its slab, and the (D, slab) float64 result blocks are
gathered to rank 0 (``torch.distributed.gather``; backend "nccl" = RCCL over
xGMI on MI355X, "gloo" in the CPU tests).  The assembled result is
bit-identical to the single-GPU one: the kernels do the same arithmetic per
cell whatever the slab.

torch is used for the process group and the collective only; the kernels are
called through the C ABI.
"""
import numpy as np

from . import api
from .detect import INTER_VARIABLES, _detect
from .device import calc_clim_device, calc_clim_grid_device
from .exception import XmhwException


def slab_bounds(ncells, world_size):
    """Contiguous, balanced [lo, hi) per rank (first ``ncells % world`` slabs one larger)."""
    base, extra = divmod(int(ncells), int(world_size))
    bounds, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < extra else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def gather_blocks(th, se, ncells, group=None, dst=0, device=None):
    """Gather per-rank (D, slab_r) blocks to ``dst``; returns (th, se) of shape
    (D, ncells) on dst, (None, None) elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bounds = slab_bounds(ncells, world)
    width = max(hi - lo for lo, hi in bounds)
    D = th.shape[0]
    if device is None:
        # RCCL ("nccl") moves device memory only; gloo (CPU tests) takes host tensors
        if dist.get_backend(group) == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
        else:
            device = torch.device("cpu")
    dev = device
    block = torch.full((2, D, width), float("nan"), dtype=torch.float64, device=dev)
    n_r = bounds[rank][1] - bounds[rank][0]
    if n_r:
        block[0, :, :n_r] = torch.as_tensor(th, device=dev)
        block[1, :, :n_r] = torch.as_tensor(se, device=dev)
    out = [torch.empty_like(block) for _ in range(world)] if rank == dst else None
    dist.gather(block, out, dst=dst, group=group)
    if rank != dst:
        return None, None
    full = np.empty((2, D, ncells), dtype=np.float64)
    for r, (lo, hi) in enumerate(bounds):
        full[:, :, lo:hi] = out[r][:, :, : hi - lo].cpu().numpy()
    return full[0], full[1]


def make_sharded_compute(group=None, dst=0, device=None, compute=None):
    """A drop-in for device.calc_clim_device that computes only this rank's slab
    and gathers.  ``compute`` defaults to the HIP path; the CPU tests of the
    sharding logic inject a stand-in."""
    import torch.distributed as dist

    inner = compute or calc_clim_device

    def sharded(ts, doy, pctile, windowHalfWidth, smoothPercentile, smoothPercentileWidth,
                tstep, coldSpells=False):
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        C = ts.shape[1]
        lo, hi = slab_bounds(C, world)[rank]
        slab = np.ascontiguousarray(ts[:, lo:hi])
        if hi > lo:
            doys, th, se = inner(slab, doy, pctile, windowHalfWidth, smoothPercentile,
                                 smoothPercentileWidth, tstep, coldSpells)
        else:
            doys = np.unique(np.asarray(doy, dtype=np.int64))
            th = se = np.empty((doys.shape[0], 0))
        th, se = gather_blocks(th, se, C, group=group, dst=dst, device=device)
        if rank != dst:
            # non-root ranks return a placeholder of the right shape
            th = se = np.full((doys.shape[0], C), np.nan)
        return doys, th, se

    return sharded


def make_sharded_grid_compute(group=None, dst=0, device=None, grid_compute=None):
    """A drop-in for device.calc_clim_grid_device: every rank takes a contiguous block of the
    UNCOMPACTED stacked columns, masks / compacts / computes it on its own GPU (so no rank runs
    land_check() over the whole grid on the host), then the masks are all-gathered (every rank
    needs the surviving cells for the output grid) and the variable-width result blocks gathered
    to ``dst``."""
    import torch
    import torch.distributed as dist

    inner = grid_compute or calc_clim_grid_device

    def sharded(stacked, doy, anynans, pctile, windowHalfWidth, smoothPercentile, smoothPercentileWidth,
                tstep, coldSpells=False):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = _device_for(group, device)
        N = stacked.shape[1]
        bounds = slab_bounds(N, world)
        lo, hi = bounds[rank]
        keep_r, doys, th_r, se_r = inner(stacked, doy, anynans, pctile, windowHalfWidth, smoothPercentile,
                                         smoothPercentileWidth, tstep, coldSpells, columns=(lo, hi))
        D = doys.shape[0]
        width = max(b - a for a, b in bounds)
        mine = torch.zeros(width, dtype=torch.uint8, device=dev)
        mine[: hi - lo] = torch.as_tensor(keep_r.astype(np.uint8), device=dev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        keep = np.concatenate([parts[r][: b - a].cpu().numpy() != 0 for r, (a, b) in enumerate(bounds)])
        if not keep.any():
            raise XmhwException("All points of grid are either land or NaN")     # on every rank alike
        # the blocks come back on the grid (NaN at dropped cells), so they all have their slab's width
        block = torch.full((2, D, width), float("nan"), dtype=torch.float64, device=dev)
        if hi > lo:
            block[0, :, : hi - lo] = torch.as_tensor(th_r, device=dev)
            block[1, :, : hi - lo] = torch.as_tensor(se_r, device=dev)
        out = [torch.empty_like(block) for _ in range(world)] if rank == dst else None
        dist.gather(block, out, dst=dst, group=group)
        if rank != dst:
            th = se = np.full((D, N), np.nan)
        else:
            full = np.concatenate([out[r][:, :, : b - a].cpu().numpy() for r, (a, b) in enumerate(bounds)], axis=2)
            th, se = full[0], full[1]
        return keep, doys, th, se

    return sharded


def threshold_sharded(temp, group=None, dst=0, device=None, _compute=None, _grid_compute=None, **kwargs):
    """threshold() over all ranks of ``group``; every rank passes the same ``temp``; rank ``dst``
    gets the Dataset, the others None.  The ranks split the uncompacted grid columns and each
    masks and compacts its own block on its GPU; ``_compute`` (test hook, compact-array stand-in)
    selects the older path in which every rank runs land_check() on the host first."""
    import torch.distributed as dist

    if _compute is not None:
        ds = api._threshold(temp, make_sharded_compute(group, dst, device, _compute), **kwargs)
    else:
        # a single-point series has no grid to split: the compact-array path handles it
        ds = api._threshold(temp, make_sharded_compute(group, dst, device, None),
                            grid_compute=make_sharded_grid_compute(group, dst, device, _grid_compute), **kwargs)
    return ds if dist.get_rank(group) == dst else None


def _device_for(group, device):
    import torch
    import torch.distributed as dist
    if device is not None:
        return device
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _gather_columns(block, ncells, group, dst, device):
    """Gather per-rank (rows, slab_r) float64 blocks into (rows, ncells) on dst (None elsewhere)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = slab_bounds(ncells, world)
    width = max(hi - lo for lo, hi in bounds)
    rows = block.shape[0]
    pad = torch.zeros((rows, width), dtype=torch.float64, device=device)
    n_r = bounds[rank][1] - bounds[rank][0]
    if n_r:
        pad[:, :n_r] = torch.as_tensor(np.ascontiguousarray(block, dtype=np.float64), device=device)
    out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, out, dst=dst, group=group)
    if rank != dst:
        return None
    full = np.empty((rows, ncells), dtype=np.float64)
    for r, (lo, hi) in enumerate(bounds):
        full[:, lo:hi] = out[r][:, : hi - lo].cpu().numpy()
    return full


def make_sharded_detect(group=None, dst=0, device=None, compute=None):
    """A drop-in for detect_front.detect_cells that runs only this rank's slab of cells and gathers
    the event tables (variable length per rank: sizes first, then one padded gather) and, if asked
    for, the per-step columns.  Rank dst returns the full result, the others an empty one."""
    import torch
    import torch.distributed as dist
    from .detect_front import EVENT_COLUMNS, detect_cells

    from .detect_front import INTERMEDIATE_U8
    inner = compute or detect_cells
    ncol = len(EVENT_COLUMNS)
    BOOL_VARIABLES = set(INTERMEDIATE_U8) | {"bthresh"}

    def sharded(ts, seas, thresh, doy, doys, minDuration=5, joinGaps=True, maxGap=2, coldSpells=False,
                intermediate=False):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = _device_for(group, device)
        T, C = ts.shape
        lo, hi = slab_bounds(C, world)[rank]
        if hi > lo:
            res = inner(np.ascontiguousarray(ts[:, lo:hi]), np.ascontiguousarray(seas[:, lo:hi]),
                        np.ascontiguousarray(thresh[:, lo:hi]), doy, doys, minDuration, joinGaps, maxGap,
                        coldSpells, intermediate)
        else:
            res = dict(table=np.zeros((0, ncol)), offsets=np.zeros(1, dtype=np.int64), inter=None)
        counts = np.diff(res["offsets"]).astype(np.float64)[None, :]
        all_counts = _gather_columns(counts, C, group, dst, dev)
        # table rows: every rank learns the largest table, pads to it, one gather
        n_r = torch.tensor([res["table"].shape[0]], dtype=torch.int64, device=dev)
        sizes = [torch.zeros_like(n_r) for _ in range(world)]
        dist.all_gather(sizes, n_r, group=group)
        sizes = [int(v.item()) for v in sizes]
        nmax = max(max(sizes), 1)
        pad = torch.zeros((nmax, ncol), dtype=torch.float64, device=dev)
        if sizes[rank]:
            pad[: sizes[rank]] = torch.as_tensor(res["table"], device=dev)
        out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, out, dst=dst, group=group)
        inter = None
        if intermediate:
            inter = {}
            for k in INTER_VARIABLES:
                blk = res["inter"][k] if hi > lo else np.zeros((T, 0))
                full = _gather_columns(blk, C, group, dst, dev)
                if rank != dst:
                    continue
                if k in BOOL_VARIABLES:
                    inter[k] = full != 0
                elif k == "ts":
                    inter[k] = full.astype(ts.dtype)        # float32 values survive the float64 transport
                else:
                    inter[k] = full
        if rank != dst:
            return dict(table=np.zeros((0, ncol)), offsets=np.zeros(C + 1, dtype=np.int64),
                        inter=None if not intermediate else
                        {k: np.zeros((T, C), dtype=bool if k in BOOL_VARIABLES else np.float64)
                         for k in INTER_VARIABLES})
        table = np.concatenate([out[r][: sizes[r]].cpu().numpy() for r in range(world)], axis=0)
        offsets = np.zeros(C + 1, dtype=np.int64)
        np.cumsum(all_counts[0].astype(np.int64), out=offsets[1:])
        return dict(table=table, offsets=offsets, inter=inter)

    return sharded


def make_sharded_detect_grid(group=None, dst=0, device=None, grid_compute=None):
    """A drop-in for detect_front.detect_grid: every rank masks and compacts its own block of the
    uncompacted series columns on its GPU; the survivor counts are all-gathered (a block's cells
    pair up with the climatology columns at the offset of the blocks before it), then the event
    tables travel as in make_sharded_detect and the keep masks are all-gathered."""
    import torch
    import torch.distributed as dist
    from .detect_front import EVENT_COLUMNS, detect_grid

    inner = grid_compute or detect_grid
    ncol = len(EVENT_COLUMNS)
    fallback = make_sharded_detect(group, dst, device, None)

    def sharded(stacked, anynans, seas, thresh, doy, doys, minDuration=5, joinGaps=True, maxGap=2, coldSpells=False,
                intermediate=False, clim_stacked=False):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev = _device_for(group, device)
        N = stacked.shape[1]
        bounds = slab_bounds(N, world)
        lo, hi = bounds[rank]

        def exchange(n_mine):
            mine = torch.tensor([n_mine], dtype=torch.int64, device=dev)
            parts = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=group)
            counts = [int(v.item()) for v in parts]
            return sum(counts[:rank]), sum(counts)

        if intermediate:
            raise XmhwException("detect_sharded: intermediate=True is only available through the host "
                                "land_check path (pass _compute)")
        res = inner(stacked, anynans, seas, thresh, doy, doys, minDuration, joinGaps, maxGap, coldSpells, False,
                    clim_stacked=clim_stacked, columns=(lo, hi), exchange=exchange)
        # keep masks to everybody, per-cell event counts and the tables to dst
        width = max(b - a for a, b in bounds)
        mine = torch.zeros(width, dtype=torch.uint8, device=dev)
        mine[: hi - lo] = torch.as_tensor(res["keep"].astype(np.uint8), device=dev)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        keep = np.concatenate([parts[r][: b - a].cpu().numpy() != 0 for r, (a, b) in enumerate(bounds)])
        if not keep.any():
            raise XmhwException("All points of grid are either land or NaN")
        ncells = [int(keep[a:b].sum()) for a, b in bounds]
        cmax = max(max(ncells), 1)
        cnt = torch.zeros(cmax, dtype=torch.int64, device=dev)
        if ncells[rank]:
            cnt[: ncells[rank]] = torch.as_tensor(np.diff(res["offsets"]), device=dev)
        cnts = [torch.zeros_like(cnt) for _ in range(world)] if rank == dst else None
        dist.gather(cnt, cnts, dst=dst, group=group)
        n_r = torch.tensor([res["table"].shape[0]], dtype=torch.int64, device=dev)
        sizes = [torch.zeros_like(n_r) for _ in range(world)]
        dist.all_gather(sizes, n_r, group=group)
        sizes = [int(v.item()) for v in sizes]
        pad = torch.zeros((max(max(sizes), 1), ncol), dtype=torch.float64, device=dev)
        if sizes[rank]:
            pad[: sizes[rank]] = torch.as_tensor(res["table"], device=dev)
        out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, out, dst=dst, group=group)
        C = int(keep.sum())
        if rank != dst:
            return dict(table=np.zeros((0, ncol)), offsets=np.zeros(C + 1, dtype=np.int64), inter=None, keep=keep)
        table = np.concatenate([out[r][: sizes[r]].cpu().numpy() for r in range(world)], axis=0)
        counts = np.concatenate([cnts[r][: ncells[r]].cpu().numpy() for r in range(world)])
        offsets = np.zeros(C + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        return dict(table=table, offsets=offsets, inter=None, keep=keep)

    return sharded


def detect_sharded(temp, th, se, group=None, dst=0, device=None, _compute=None, _grid_compute=None, **kwargs):
    """detect() over all ranks of ``group``; rank ``dst`` gets what detect() returns, the others None.
    The ranks split the uncompacted grid columns and each masks / compacts its own block on its GPU
    (survivor counts are exchanged so that every block finds its climatology columns); ``_compute``
    (test hook, compact-array stand-in; also needed for intermediate=True) selects the older path in
    which every rank runs land_check() on the host and the compact cells are split."""
    import torch.distributed as dist

    if _compute is not None or kwargs.get("intermediate"):
        out = _detect(temp, th, se, make_sharded_detect(group, dst, device, _compute), **kwargs)
    else:
        out = _detect(temp, th, se, make_sharded_detect(group, dst, device, None),
                      grid_compute=make_sharded_detect_grid(group, dst, device, _grid_compute), **kwargs)
    return out if dist.get_rank(group) == dst else None
