"""The PUBLIC API on the hot kernel against the oracle ("What's missing" #3): threshold() / threshold_array()
on a 40-year gridded input with scattered land, a land block (whole grid lines dropped), NaN holes inside ocean cells,
climatologyPeriod and coldSpells -- the flow of xmhw/xmhw.py:138-219 from land_check() to the unstacked Dataset -- with
the plan every call creates checked to be the sorted-list layout (40), so that a silent fall-back to the ring kernels,
or a wrong leading dimension between mask -> compaction -> sorted kernel -> recomputation -> finish -> scatter, cannot
stay green.  Every ocean cell is compared with the vectorised oracle (oracle_fast, itself pinned to the per-cell oracle
by tests/test_oracle_fast.py), a block of cells with the per-cell oracle `ora.threshold_grid` itself.
Tolerances: thresh / seas 1e-12 relative (contract: 1e-6); unsmoothed thresh rows other than doy 60 bit for bit;
doy labels and the dropped grid lines exactly."""
import numpy as np
import numpy.testing as npt
import pytest

import oracle_fast as fast
import xmhw_oracle as ora

pytestmark = pytest.mark.gpu

NLAT, NLON = 16, 24
T64 = np.arange("1982-01-01", "2022-01-01", dtype="datetime64[D]")


def _field(seed=2026):
    """40 years of daily SST-like float32 on a 16 x 24 grid: seasonal cycles of 2..12 K amplitude (the steep ones overflow
    row-lists: the recomputation runs too), scattered land, one latitude line and two longitude lines of land, 1 % NaN
    holes and a run of 40 missing days in the ocean cells."""
    rng = np.random.default_rng(seed)
    T = T64.shape[0]
    tt = np.arange(T)[:, None, None]
    amp = rng.uniform(2, 12, (NLAT, NLON))
    ph = rng.uniform(0, 365, (NLAT, NLON))
    x = (15 + amp * np.sin(2 * np.pi * (tt - ph) / 365.25) + rng.normal(size=(T, NLAT, NLON))).astype(np.float32)
    x[rng.random(x.shape) < 0.01] = np.nan
    x[5000:5040, 3, 4] = np.nan
    land = rng.random((NLAT, NLON)) < 0.12
    land[7, :] = True
    land[:, 5] = True
    land[:, 23] = True
    x[:, land] = np.nan
    return x, land


@pytest.fixture(scope="module")
def field():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    return _field()


@pytest.fixture()
def plans(monkeypatch):
    """records the layout of every plan the public API creates"""
    import xmhw_amd.device as dev
    seen = []
    real = dev.Plan

    class Recording(real):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            seen.append((self.ntracks, self.layout_in_use()))

    monkeypatch.setattr(dev, "Plan", Recording)
    return seen


def _oracle_all_cells(x, time64, **kw):
    """every cell of the grid through the vectorised oracle, land = NaN; same keyword meaning as threshold()"""
    period = kw.pop("climatologyPeriod", (None, None))
    if all(period):
        years = time64.astype("datetime64[Y]").astype(int) + 1970
        sel = (years >= period[0]) & (years <= period[1])
        x, time64 = x[sel], time64[sel]
    doy = ora.add_doy(time64)
    flat = x.reshape(x.shape[0], -1)
    ocean = ~np.isnan(flat).all(axis=0)
    doys, th, se = fast.threshold_cells_fast(flat[:, ocean], doy, **kw)
    D = doys.shape[0]
    full_th = np.full((D, flat.shape[1]), np.nan)
    full_se = np.full((D, flat.shape[1]), np.nan)
    full_th[:, ocean] = th
    full_se[:, ocean] = se
    return doys, full_th.reshape(D, NLAT, NLON), full_se.reshape(D, NLAT, NLON), ocean.reshape(NLAT, NLON)


def _series(x):
    from xmhw_amd import GridSeries
    return GridSeries(x, ("time", "lat", "lon"),
                      {"time": T64, "lat": np.linspace(-60, 60, NLAT), "lon": np.linspace(0, 345, NLON)},
                      time_encoding={"calendar": "proleptic_gregorian"})


@pytest.mark.parametrize("kw,tracks", [(dict(), 40), (dict(coldSpells=True), 40),
                                       (dict(climatologyPeriod=[1985, 2014]), 30),
                                       (dict(pctile=95, smoothPercentileWidth=11), 40)])
def test_threshold_on_the_sorted_kernel_equals_the_oracle(field, plans, kw, tracks):
    from xmhw_amd import threshold
    x, land = field
    ds = threshold(_series(x), **kw)
    doys, th, se, ocean = _oracle_all_cells(x, T64, **dict(kw))
    assert plans and all(p == (tracks, 40) for p in plans), plans          # the sorted-list layout served the call
    rows, cols = ocean.any(axis=1), ocean.any(axis=0)
    assert rows.sum() == NLAT - 1 and cols.sum() == NLON - 2                # the all-land grid lines are dropped (xmhw.py:210-219)
    npt.assert_array_equal(ds.coords["doy"], doys)
    npt.assert_array_equal(ds.coords["lat"], np.linspace(-60, 60, NLAT)[rows])
    npt.assert_array_equal(ds.coords["lon"], np.linspace(0, 345, NLON)[cols])
    assert ds["thresh"].shape == (366, NLAT - 1, NLON - 2)
    npt.assert_allclose(ds["thresh"], th[:, rows][:, :, cols], rtol=1e-12, equal_nan=True)
    npt.assert_allclose(ds["seas"], se[:, rows][:, :, cols], rtol=1e-12, equal_nan=True)
    assert np.isnan(np.asarray(ds["thresh"])[:, ~ocean[rows][:, cols]]).all()          # scattered land stays NaN
    assert not np.isnan(np.asarray(ds["thresh"])[:, ocean[rows][:, cols]]).any()


def test_unsmoothed_thresh_rows_are_bit_identical_and_a_block_equals_the_per_cell_oracle(field, plans):
    from xmhw_amd import threshold_array
    x, land = field
    ds = threshold_array(x, T64, dims=("time", "lat", "lon"),
                         coords={"lat": np.linspace(-60, 60, NLAT), "lon": np.linspace(0, 345, NLON)},
                         calendar="proleptic_gregorian", smoothPercentile=False)
    assert plans and all(p == (40, 40) for p in plans), plans
    doys, th, se, ocean = _oracle_all_cells(x, T64, smoothPercentile=False)
    rows, cols = ocean.any(axis=1), ocean.any(axis=0)
    got = np.asarray(ds["thresh"])
    want = th[:, rows][:, :, cols]
    not60 = doys != 60
    npt.assert_array_equal(got[not60], want[not60])                         # raw order statistics + numpy's lerp: every bit
    npt.assert_allclose(got[~not60], want[~not60], rtol=1e-14, equal_nan=True)       # (doy 60: a 3-point mean)
    npt.assert_allclose(ds["seas"], se[:, rows][:, :, cols], rtol=1e-12, equal_nan=True)
    # ... and a 4 x 8 block through the per-cell restatement of the reference itself (xmhw_oracle.threshold_grid), smoothed
    from xmhw_amd import threshold
    blk = x[:, 8:12, 8:16]
    from xmhw_amd import GridSeries
    dsb = threshold(GridSeries(blk, ("time", "lat", "lon"), {"time": T64, "lat": np.arange(4.0), "lon": np.arange(8.0)},
                               time_encoding={"calendar": "proleptic_gregorian"}))
    ref = ora.threshold_grid(blk, T64)
    keep = ref["keep"].reshape(4, 8)
    r2, c2 = keep.any(axis=1), keep.any(axis=0)
    npt.assert_array_equal(dsb.coords["doy"], ref["doy"])
    npt.assert_allclose(dsb["thresh"], ref["thresh"][:, r2][:, :, c2], rtol=1e-12, equal_nan=True)
    npt.assert_allclose(dsb["seas"], ref["seas"][:, r2][:, :, c2], rtol=1e-12, equal_nan=True)
