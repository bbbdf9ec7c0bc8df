"""N>1 host logic on CPU: world_size-2/3 gloo process groups behind the transport interface of
xmhw_amd.sharded (tests/gloo_transport.py stands in for the RCCL transport), the device stage
replaced by the oracle (test hook).  The sharded result must be bit-identical to the unsharded one."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from gloo_transport import GlooTransport
    import oracle_fast as fast
    from xmhw_amd import GridSeries
    from xmhw_amd.sharded import threshold_sharded, slab_bounds

    dist.init_process_group("gloo", rank=rank, world_size=world)

    def compute(ts, doy, pctile, w, smooth, width, tstep, cold=False):
        return fast.threshold_cells_fast(ts, doy, pctile=pctile, windowHalfWidth=w, smoothPercentile=smooth,
                                         smoothPercentileWidth=width, tstep=tstep, coldSpells=cold)

    g = np.load(os.path.join(ROOT, "tests", "golden", "oisst_2003_2004.npz"))
    time = np.datetime64("2003-01-01") + g["time"].astype("timedelta64[D]")
    temp = GridSeries(g["sst"], ("time", "lat", "lon"), {"time": time, "lat": g["lat"], "lon": g["lon"]})
    ds = threshold_sharded(temp, GlooTransport(), _compute=compute, smoothPercentileWidth=11)
    assert slab_bounds(12, world)[rank] == ((0, 6), (6, 12))[rank]
    if rank == 0:
        np.savez(os.path.join(outdir, "sharded.npz"), thresh=ds["thresh"], seas=ds["seas"])
    else:
        assert ds is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_equals_single(tmp_path):
    import torch.multiprocessing as mp
    import oracle_fast as fast
    from xmhw_amd import GridSeries
    from xmhw_amd.api import _threshold

    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "sharded.npz")

    def compute(ts, doy, pctile, w, smooth, width, tstep, cold=False):
        return fast.threshold_cells_fast(ts, doy, pctile=pctile, windowHalfWidth=w, smoothPercentile=smooth,
                                         smoothPercentileWidth=width, tstep=tstep, coldSpells=cold)
    g = np.load(os.path.join(ROOT, "tests", "golden", "oisst_2003_2004.npz"))
    time = np.datetime64("2003-01-01") + g["time"].astype("timedelta64[D]")
    temp = GridSeries(g["sst"], ("time", "lat", "lon"), {"time": time, "lat": g["lat"], "lon": g["lon"]})
    ref = _threshold(temp, compute, smoothPercentileWidth=11)
    np.testing.assert_array_equal(got["thresh"], ref["thresh"])
    np.testing.assert_array_equal(got["seas"], ref["seas"])


def test_slab_bounds_cover_and_balance():
    from xmhw_amd.sharded import slab_bounds
    for C in (0, 1, 7, 8, 1036800, 1036801):
        for w in (1, 2, 3, 8):
            b = slab_bounds(C, w)
            assert b[0][0] == 0 and b[-1][1] == C
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _detect_worker(rank, world, port, outdir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from gloo_transport import GlooTransport
    from detect_standin import oracle_detect_cells
    from xmhw_amd.sharded import detect_sharded

    dist.init_process_group("gloo", rank=rank, world_size=world)
    temp, th, se = _detect_inputs()
    out = detect_sharded(temp, th, se, GlooTransport(), _compute=oracle_detect_cells, intermediate=True, minDuration=4, maxGap=1)
    if rank == 0:
        mhw, inter = out
        np.savez(os.path.join(outdir, "detect.npz"), table=mhw.table, offsets=mhw.offsets,
                 **{"inter_" + k: v for k, v in inter.data_vars.items()})
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _detect_inputs():
    import oracle_fast as fast
    from xmhw_amd import GridSeries, climatology_series
    from xmhw_amd.api import _threshold

    def compute(ts, doy, pctile, w, smooth, width, tstep, cold=False):
        return fast.threshold_cells_fast(ts, doy, pctile=pctile, windowHalfWidth=w, smoothPercentile=smooth,
                                         smoothPercentileWidth=width, tstep=tstep, coldSpells=cold)
    g = np.load(os.path.join(ROOT, "tests", "golden", "oisst_2003_2004.npz"))
    time = np.datetime64("2003-01-01") + g["time"].astype("timedelta64[D]")
    temp = GridSeries(g["sst"], ("time", "lat", "lon"), {"time": time, "lat": g["lat"], "lon": g["lon"]})
    clim = _threshold(temp, compute, pctile=80)
    return temp, climatology_series(clim, "thresh"), climatology_series(clim, "seas")


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_detect_equals_single(tmp_path, world):
    """detect() over 2 and 3 ranks (uneven slabs, variable-length event tables) = the single-rank result."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from detect_standin import oracle_detect_cells
    from xmhw_amd.detect import _detect

    mp.spawn(_detect_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "detect.npz")
    temp, th, se = _detect_inputs()
    mhw, inter = _detect(temp, th, se, oracle_detect_cells, intermediate=True, minDuration=4, maxGap=1)
    assert mhw.n_events > 0
    np.testing.assert_array_equal(got["offsets"], mhw.offsets)
    np.testing.assert_array_equal(got["table"], mhw.table)
    for k, v in inter.data_vars.items():
        assert got["inter_" + k].dtype == v.dtype, k
        np.testing.assert_array_equal(got["inter_" + k], v, err_msg=k)


def _grid_standin(stacked, doy, anynans, pctile, w, smooth, width, tstep, cold=False, columns=None):
    """numpy stand-in for device.calc_clim_grid_device (mask + compaction + oracle) on a column range"""
    import oracle_fast as fast
    c0, c1 = columns if columns is not None else (0, stacked.shape[1])
    sub = stacked[:, c0:c1]
    nan = np.isnan(sub)
    keep = ~(nan.any(axis=0) if anynans else nan.all(axis=0))
    doys = np.unique(np.asarray(doy, dtype=np.int64))
    th_full = np.full((doys.shape[0], c1 - c0), np.nan)
    se_full = np.full((doys.shape[0], c1 - c0), np.nan)
    if not keep.any():
        return keep, doys, th_full, se_full
    d, th, se = fast.threshold_cells_fast(np.ascontiguousarray(sub[:, keep]), doy, pctile=pctile, windowHalfWidth=w,
                                          smoothPercentile=smooth, smoothPercentileWidth=width, tstep=tstep,
                                          coldSpells=cold)
    th_full[:, keep] = th
    se_full[:, keep] = se
    return keep, d, th_full, se_full


def _grid_worker(rank, world, port, outdir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from gloo_transport import GlooTransport
    from xmhw_amd import GridSeries, XmhwException
    from xmhw_amd.sharded import threshold_sharded

    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "oisst_2003_2004.npz"))
    time = np.datetime64("2003-01-01") + g["time"].astype("timedelta64[D]")
    temp = GridSeries(g["sst"], ("time", "lat", "lon"), {"time": time, "lat": g["lat"], "lon": g["lon"]})
    tr = GlooTransport()
    ds = threshold_sharded(temp, tr, _grid_compute=_grid_standin, smoothPercentileWidth=11)
    # north_star's "single gather at the end": ONE bulk collective, ONE int64 exchange (the agreement that every
    # local stage succeeded, carrying the survivor counts), no mask exchange
    assert (tr.bulk_collectives, tr.int_collectives, getattr(tr, "mask_collectives", 0)) == (1, 1, 0), \
        (tr.bulk_collectives, tr.int_collectives, getattr(tr, "mask_collectives", 0))
    if rank == 0:
        np.savez(os.path.join(outdir, "grid.npz"), thresh=ds["thresh"], seas=ds["seas"], lat=ds.coords["lat"],
                 lon=ds.coords["lon"])
    else:
        assert ds is None
    # an all-land grid raises on EVERY rank (the survivor counts ride the agreement), nobody hangs
    land = GridSeries(np.full((731, 3, 2), np.nan, np.float32), ("time", "lat", "lon"),
                      {"time": time, "lat": np.arange(3), "lon": np.arange(2)})
    try:
        threshold_sharded(land, GlooTransport(), _grid_compute=_grid_standin)
        raise AssertionError("expected XmhwException")
    except XmhwException:
        pass
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_grid_path_equals_single(tmp_path, world):
    """Every rank masks and compacts only its own block of the uncompacted grid columns (ranks end
    up with different numbers of ocean cells; one block of the fixture is all land for world=3)."""
    import torch.multiprocessing as mp
    import oracle_fast as fast
    from xmhw_amd import GridSeries
    from xmhw_amd.api import _threshold

    mp.spawn(_grid_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "grid.npz")

    def compute(ts, doy, pctile, w, smooth, width, tstep, cold=False):
        return fast.threshold_cells_fast(ts, doy, pctile=pctile, windowHalfWidth=w, smoothPercentile=smooth,
                                         smoothPercentileWidth=width, tstep=tstep, coldSpells=cold)
    g = np.load(os.path.join(ROOT, "tests", "golden", "oisst_2003_2004.npz"))
    time = np.datetime64("2003-01-01") + g["time"].astype("timedelta64[D]")
    temp = GridSeries(g["sst"], ("time", "lat", "lon"), {"time": time, "lat": g["lat"], "lon": g["lon"]})
    ref = _threshold(temp, compute, smoothPercentileWidth=11)
    np.testing.assert_array_equal(got["thresh"], ref["thresh"])
    np.testing.assert_array_equal(got["seas"], ref["seas"])
    np.testing.assert_array_equal(got["lat"], ref.coords["lat"])
    np.testing.assert_array_equal(got["lon"], ref.coords["lon"])


def _detect_grid_standin(stacked, anynans, seas, thresh, doy, doys, minDuration=5, joinGaps=True, maxGap=2,
                         coldSpells=False, intermediate=False, clim_stacked=False, columns=None, exchange=None):
    """numpy / oracle stand-in for detect_front.detect_grid on a block of columns"""
    from detect_standin import oracle_detect_cells

    def compact(a):
        nan = np.isnan(a)
        k = ~(nan.any(axis=0) if anynans else nan.all(axis=0))
        return a[:, k], k
    if clim_stacked:
        seas, thresh = compact(seas)[0], compact(thresh)[0]
    c0, c1 = columns
    sub, keep = compact(stacked[:, c0:c1])
    k0, total = exchange(int(keep.sum()))
    assert total == thresh.shape[1]
    n = int(keep.sum())
    if n == 0:
        return dict(table=np.zeros((0, 31)), offsets=np.zeros(1, dtype=np.int64), inter=None, keep=keep)
    r = oracle_detect_cells(np.ascontiguousarray(sub), np.ascontiguousarray(seas[:, k0:k0 + n]),
                            np.ascontiguousarray(thresh[:, k0:k0 + n]), doy, doys, minDuration, joinGaps, maxGap,
                            coldSpells, intermediate)
    r["keep"] = keep
    return r


def _detect_grid_worker(rank, world, port, outdir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from gloo_transport import GlooTransport
    from xmhw_amd.sharded import detect_sharded

    dist.init_process_group("gloo", rank=rank, world_size=world)
    temp, th, se = _detect_inputs()
    out = detect_sharded(temp, th, se, GlooTransport(), _grid_compute=_detect_grid_standin, minDuration=4, maxGap=1)
    # the same with the per-step planes (xmhw/xmhw.py:354-356): gathered block by block beside the tables
    out_i = detect_sharded(temp, th, se, GlooTransport(), _grid_compute=_detect_grid_standin, minDuration=4, maxGap=1,
                           intermediate=True)
    if rank == 0:
        mhw_i, inter = out_i
        np.testing.assert_array_equal(mhw_i.table, out.table)
        np.savez(os.path.join(outdir, "detect_grid.npz"), table=out.table, offsets=out.offsets, keep=out.keep,
                 **{"inter_" + k: v for k, v in inter.data_vars.items()})
    else:
        assert out is None and out_i is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_detect_grid_path_equals_single(tmp_path, world):
    """detect() with every rank masking its own block of grid columns (blocks with different numbers
    of ocean cells, one all-land block for world=3) = the single-rank result."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from detect_standin import oracle_detect_cells
    from xmhw_amd.detect import _detect

    mp.spawn(_detect_grid_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "detect_grid.npz")
    temp, th, se = _detect_inputs()
    mhw, inter = _detect(temp, th, se, oracle_detect_cells, minDuration=4, maxGap=1, intermediate=True)
    assert mhw.n_events > 0
    np.testing.assert_array_equal(got["keep"], mhw.keep)
    np.testing.assert_array_equal(got["offsets"], mhw.offsets)
    np.testing.assert_array_equal(got["table"], mhw.table)
    for k, v in inter.data_vars.items():
        assert got["inter_" + k].dtype == v.dtype, k
        np.testing.assert_array_equal(got["inter_" + k], v, err_msg=k)


def _failing_worker(rank, world, port, outdir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from gloo_transport import GlooTransport
    from xmhw_amd import GridSeries, XmhwException
    from xmhw_amd.sharded import threshold_sharded, detect_sharded

    dist.init_process_group("gloo", rank=rank, world_size=world)

    def broken(stacked, doy, anynans, *a, columns=None, **k):
        if rank == 1:
            raise MemoryError("simulated hipMalloc failure on rank 1")
        return _grid_standin(stacked, doy, anynans, *a, columns=columns, **k)

    g = np.load(os.path.join(ROOT, "tests", "golden", "oisst_2003_2004.npz"))
    time = np.datetime64("2003-01-01") + g["time"].astype("timedelta64[D]")
    temp = GridSeries(g["sst"], ("time", "lat", "lon"), {"time": time, "lat": g["lat"], "lon": g["lon"]})
    try:
        threshold_sharded(temp, GlooTransport(), _grid_compute=broken)
        raise AssertionError("expected XmhwException on every rank")
    except XmhwException as e:
        msg = str(e)
    assert ("rank 1" in msg) or ("rank(s) [1]" in msg), msg

    def broken_detect(stacked, anynans, *a, columns=None, exchange=None, **k):
        if rank == 0:
            raise RuntimeError("simulated failure before the survivor-count exchange")
        return _detect_grid_standin(stacked, anynans, *a, columns=columns, exchange=exchange, **k)

    t2, th, se = _detect_inputs()
    try:
        detect_sharded(t2, th, se, GlooTransport(), _grid_compute=broken_detect, minDuration=4, maxGap=1)
        raise AssertionError("expected XmhwException on every rank")
    except XmhwException:
        pass
    # the group is still usable afterwards
    ds = threshold_sharded(temp, GlooTransport(), _grid_compute=_grid_standin, smoothPercentileWidth=11)
    assert (ds is None) == (rank != 0)
    open(os.path.join(outdir, f"ok{rank}"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_failure_on_one_rank_raises_on_all_ranks(tmp_path):
    """A rank that fails in its local stage (out of memory, a bad block) must not leave the others
    waiting in the next collective: the error flag is all-gathered and every rank raises."""
    import torch.multiprocessing as mp
    mp.spawn(_failing_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _bootstrap_worker(rank, world, port, outdir):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from xmhw_amd import bootstrap
    got = bootstrap.share_bytes(rank, world, (lambda: bytes(range(128))) if rank == 0 else None,
                                addr="127.0.0.1", port=port)
    assert got == bytes(range(128))
    open(os.path.join(outdir, f"b{rank}"), "w").write("ok")


def test_bootstrap_hands_the_unique_id_to_every_rank(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_bootstrap_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    assert all((tmp_path / f"b{r}").exists() for r in range(3))
