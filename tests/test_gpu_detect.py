"""detect() front end on the GPU vs the reference (golden outputs of its own mhw_filter /
join_gaps) and vs the loop oracle on random gridded data.  Index arrays: bit-exact."""
import os

import numpy as np
import numpy.testing as npt
import pytest

import detect_oracle as det
import xmhw_oracle as ora
import oracle_fast as fast

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "mhw_filter_cases.npz")


@pytest.fixture(scope="module")
def front():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.detect_front as f
    return f


def test_against_reference_outputs_through_the_kernel(front):
    """Every golden case becomes one cell: ts = 1 where the reference's bthresh is True, thresh 0.5."""
    g = np.load(GOLDEN)
    offs, params = g["offsets"], g["params"]
    # group the cases by (T, params) so that each launch carries several cells
    keys = {}
    for i in range(params.shape[0]):
        keys.setdefault((int(offs[i + 1] - offs[i]),) + tuple(int(v) for v in params[i]), []).append(i)
    checked = 0
    for (T, m, jg, gap), idx in keys.items():
        ts = np.stack([g["bthresh"][offs[i]:offs[i + 1]].astype(np.float32) for i in idx], axis=1)
        th = np.full((1, len(idx)), 0.5)
        out = front.mhw_filter_cells(ts, th, np.ones(T, int), np.array([1]), m, bool(jg), gap)
        for k, i in enumerate(idx):
            sl = slice(offs[i], offs[i + 1])
            npt.assert_array_equal(out["bthresh"][:, k], g["bthresh"][sl])
            npt.assert_array_equal(out["start"][:, k], g["start"][sl], err_msg=f"case {i}")
            npt.assert_array_equal(out["end"][:, k], g["end"][sl], err_msg=f"case {i}")
            npt.assert_array_equal(out["events"][:, k], g["events"][sl], err_msg=f"case {i}")
            checked += 1
    assert checked == params.shape[0]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cold", [False, True])
def test_gridded_against_oracle(front, dtype, cold):
    """threshold() climatology of the cells themselves, then the front end; NaN holes included."""
    time = np.arange("2001-01-01", "2007-01-01", dtype="datetime64[D]")
    doy = ora.add_doy(time)
    rng = np.random.default_rng(3)
    T, C = time.shape[0], 37
    t = np.arange(T)[:, None]
    x = 15 + 4 * np.sin(2 * np.pi * (t - rng.uniform(0, 365, C)) / 365.25)
    # persistent anomalies so that events of several days exist
    anom = np.zeros((T, C))
    e = rng.normal(size=(T, C))
    for k in range(1, T):
        anom[k] = 0.9 * anom[k - 1] + e[k]
    x = (x + anom).astype(dtype)
    x[rng.random((T, C)) < 0.01] = np.nan
    doys, th, se = fast.threshold_cells_fast(x, doy, pctile=10 if cold else 90, coldSpells=cold)
    out = front.mhw_filter_cells(x, th, doy, doys, 5, True, 2, coldSpells=cold)
    rows = np.searchsorted(doys, doy)
    nev = 0
    for c in range(C):
        b, s, en, ev = det.detect_front(x[:, c], th[:, c], rows, 5, True, 2, coldSpells=cold)
        npt.assert_array_equal(out["bthresh"][:, c], b)
        npt.assert_array_equal(out["start"][:, c], s)
        npt.assert_array_equal(out["end"][:, c], en)
        npt.assert_array_equal(out["events"][:, c], ev)
        nev += int(np.sum(~np.isnan(s)))
    assert nev > 50          # the test data does contain events


def test_errors(front):
    from xmhw_amd import XmhwException
    x = np.zeros((10, 2), np.float32)
    th = np.zeros((3, 2))
    with pytest.raises(XmhwException):
        front.mhw_filter_cells(x, th, np.arange(10) % 4 + 1, np.array([1, 2, 3]))     # label 4 has no row
    with pytest.raises(XmhwException):
        front.mhw_filter_cells(x, th[:, :1], np.ones(10, int), np.array([1, 2, 3]))   # cells differ
    with pytest.raises(XmhwException):
        front.mhw_filter_cells(x, th, np.ones(10, int), np.array([1, 2, 3]), minDuration=0)


def test_offsets_from_counts_device_prefix_sum():
    """the device prefix sum between the two events_from_bits passes, across block boundaries"""
    from xmhw_amd.device import DeviceBuffer, hip
    h = hip()
    rng = np.random.default_rng(0)
    for n in (0, 1, 1023, 1024, 1025, 5000, 1024 * 1024 + 7, 3_000_001):
        counts = rng.integers(0, 40, size=n).astype(np.int32)
        d_c = DeviceBuffer.from_array(counts if n else np.zeros(1, np.int32))
        d_o = DeviceBuffer(8 * (n + 1))
        h.offsets_from_counts(d_c.ptr, n, d_o.ptr)
        h.stream_sync(0)
        got = d_o.to_array((n + 1,), np.int64)
        want = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(counts, out=want[1:])
        np.testing.assert_array_equal(got, want)
        d_c.free(); d_o.free()


def test_detect_entries_reuse_cached_tables_and_survive_a_release():
    """repeated calls with the same row table (cached on the device), a different one, and a cache
    release in between give the same tables as the first call"""
    import xmhw_amd
    from xmhw_amd import GridSeries, climatology_series
    from xmhw_amd._lib import hip
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "oisst_2003_2004.npz"))
    time = np.datetime64("2003-01-01") + g["time"].astype("timedelta64[D]")
    temp = GridSeries(g["sst"], ("time", "lat", "lon"), {"time": time, "lat": g["lat"], "lon": g["lon"]},
                      time_encoding={"calendar": "proleptic_gregorian"})
    clim = xmhw_amd.threshold(temp)
    th, se = climatology_series(clim, "thresh"), climatology_series(clim, "seas")
    first = xmhw_amd.detect(temp, th, se)
    again = xmhw_amd.detect(temp, th, se)
    np.testing.assert_array_equal(first.table, again.table)
    short = GridSeries(g["sst"][:400], ("time", "lat", "lon"), {"time": time[:400], "lat": g["lat"], "lon": g["lon"]},
                       time_encoding={"calendar": "proleptic_gregorian"})
    xmhw_amd.detect(short, th, se)                       # another row table enters the cache
    hip().release_cached_tables()
    third = xmhw_amd.detect(temp, th, se)
    np.testing.assert_array_equal(first.table, third.table)
    np.testing.assert_array_equal(first.offsets, third.offsets)
