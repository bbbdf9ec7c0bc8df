"""land_check() of the reference (xmhw/identify.py:482-529) on plain arrays:
stack the non-time dims (sorted by name) into 'cell', drop all-NaN cells
(any-NaN with anynans), raise XmhwException on degenerate grids."""
import numpy as np

from .exception import XmhwException


def stack_cells(values, dims, tdim):
    """Return (stacked[T, Ncell] view/copy, stacked dim names, stacked shape)."""
    dims = list(dims)
    rest = [d for d in dims if d != tdim]
    if len(rest) == 0:
        raise XmhwException("Series has only time dimension use point=True option, exiting")
    for d in rest:
        if values.shape[dims.index(d)] == 0:
            raise XmhwException(f"Dimension {d} has 0 lenght, exiting")
    order = sorted(rest)
    perm = [dims.index(tdim)] + [dims.index(d) for d in order]
    v = np.transpose(values, perm)
    sshape = v.shape[1:]
    return v.reshape(v.shape[0], -1), order, tuple(sshape)


def keep_mask(stacked, anynans=False):
    nan = np.isnan(stacked)
    drop = nan.any(axis=0) if anynans else nan.all(axis=0)
    keep = ~drop
    if not keep.any():
        raise XmhwException("All points of grid are either land or NaN")
    return keep


def land_check(values, dims, tdim="time", anynans=False):
    stacked, order, sshape = stack_cells(values, dims, tdim)
    keep = keep_mask(stacked, anynans)
    return np.ascontiguousarray(stacked[:, keep]), keep, order, sshape


def compress_axis(a, mask, axis):
    """np.compress(mask, a, axis) without the copy when the surviving lines form one contiguous run
    (a polar land band, a regional tile's margin): a slice view then.  On a global grid the copy of
    both climatologies cost as much as the kernels."""
    mask = np.asarray(mask, dtype=bool)
    if mask.all():
        return a
    idx = np.nonzero(mask)[0]
    if idx.size and idx[-1] - idx[0] + 1 == idx.size:
        sl = [slice(None)] * a.ndim
        sl[axis] = slice(int(idx[0]), int(idx[-1]) + 1)
        return a[tuple(sl)]
    if a.nbytes < (256 << 20) or axis == 0 or a.shape[0] < 2:
        return np.compress(mask, a, axis=axis)
    # scattered lines (every n-th meridian all land): a copy it is -- for a global grid 2.5 GB per array and
    # 0.27 s on one thread, so the leading axis (doy / time) is cut into blocks that are compressed side by
    # side (numpy releases the GIL in take)
    from concurrent.futures import ThreadPoolExecutor
    shape = list(a.shape)
    shape[axis] = int(idx.size)
    out = np.empty(shape, dtype=a.dtype)
    nblk = min(16, a.shape[0])
    edges = [a.shape[0] * i // nblk for i in range(nblk + 1)]

    def work(i):
        np.take(a[edges[i]:edges[i + 1]], idx, axis=axis, out=out[edges[i]:edges[i + 1]])

    with ThreadPoolExecutor(nblk) as pool:
        list(pool.map(work, range(nblk)))
    return out
