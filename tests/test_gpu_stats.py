"""block_average() on the device (csrc/kernels_stats.hip) against outputs of the reference's own
agg_mhw / agg_ts / agg_cats (tests/golden/block_stats_cases.npz: made by RUNNING xmhw/stats.py) and,
on a gridded detect() result, against the oracle-driven host path."""
import os

import numpy as np
import numpy.testing as npt
import pandas as pd
import pytest

import stats_oracle as so

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gpu():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    from xmhw_amd import stats
    return stats


def test_kernels_against_reference_outputs(gpu):
    g = np.load(os.path.join(GOLD, "mhw_features_cases.npz"))
    b = np.load(os.path.join(GOLD, "block_stats_cases.npz"))
    meta = b["event_meta"]
    k_time = 0
    seen = 0
    for case in range(len(g["offsets"]) - 1):
        sl = slice(g["offsets"][case], g["offsets"][case + 1])
        ts, se, th = g["ts"][sl], g["seas"][sl], g["thresh"][sl]
        T = ts.shape[0]
        tab = g["table"][g["table_offsets"][case]:g["table_offsets"][case + 1]]
        years = pd.date_range("2001-01-01", periods=T).year.to_numpy()
        cats = np.floor(1 + (ts - th) / (th - se))
        offsets = np.array([0, tab.shape[0]], dtype=np.int64)
        for blockLength in (1, 2):
            edges = so.block_bins(int(years[0]), int(years[-1]), blockLength)
            for mt, name in ((0, "time_start"), (1, "time_peak")):
                i = int(np.nonzero((meta[:, 0] == case) & (meta[:, 1] == blockLength) & (meta[:, 2] == mt))[0][0])
                want = b["event_stats"][b["event_offsets"][i]:b["event_offsets"][i + 1]]
                res = gpu.block_stats_device(tab, offsets, years, edges, name, ts[:, None], cats[:, None])
                got = np.stack([res[k][:, 0] for k in so.MHW_STATS], axis=1)
                npt.assert_allclose(got, want, rtol=1e-12, atol=0, equal_nan=True)
                seen += want.shape[0]
            wt = b["time_stats"][b["time_offsets"][k_time]:b["time_offsets"][k_time + 1]]
            k_time += 1
            gt = np.stack([res[k][:, 0] for k in so.TIME_STATS], axis=1)
            npt.assert_allclose(gt, wt, rtol=1e-12, atol=0, equal_nan=True)
            npt.assert_array_equal(res["total_days"][:, 0], wt[:, 3:].sum(axis=1))
    assert seen == b["event_stats"].shape[0]


def test_gridded_block_average_equals_oracle_path(gpu):
    """threshold() -> detect(intermediate) -> block_average() on the OISST fixture grid, float32 series"""
    import xmhw_amd
    from xmhw_amd import GridSeries, climatology_series
    from test_host_stats import oracle_compute
    g = np.load(os.path.join(GOLD, "oisst_2003_2004.npz"))
    time = np.datetime64("2003-01-01") + g["time"].astype("timedelta64[D]")
    temp = GridSeries(g["sst"], ("time", "lat", "lon"), {"time": time, "lat": g["lat"], "lon": g["lon"]},
                      time_encoding={"calendar": "proleptic_gregorian"})
    clim = xmhw_amd.threshold(temp, pctile=80)
    mhw, inter = xmhw_amd.detect(temp, climatology_series(clim, "thresh"), climatology_series(clim, "seas"),
                                 intermediate=True)
    assert mhw.n_events > 20
    for kwargs in (dict(period=[2003, 2004]), dict(dstime=temp, blockLength=2), dict(dstime=inter, mtime="time_peak")):
        got = gpu.block_average(mhw, **kwargs)
        want = gpu.block_average(mhw, _compute=oracle_compute, **kwargs)
        assert got.dims == want.dims == ("years", "lat", "lon") and set(got.data_vars) == set(want.data_vars)
        for k in want.data_vars:
            npt.assert_allclose(got[k], want[k], rtol=1e-12, atol=0, equal_nan=True, err_msg=k)
        npt.assert_array_equal(got.coords["years"], want.coords["years"])
    assert "total_days" in got.data_vars and np.nansum(got["ecount"]) == mhw.n_events
