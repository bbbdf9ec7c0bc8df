"""Multi-GPU threshold(): cells shard across ranks, one gather at the end.

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
from .device import calc_clim_device


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


def threshold_sharded(temp, group=None, dst=0, device=None, _compute=None, **kwargs):
    """threshold() over all ranks of ``group``; every rank passes the same
    ``temp`` (or at least the same land mask); rank ``dst`` gets the Dataset,
    the others None."""
    import torch.distributed as dist

    ds = api._threshold(temp, make_sharded_compute(group, dst, device, _compute), **kwargs)
    return ds if dist.get_rank(group) == dst else None
