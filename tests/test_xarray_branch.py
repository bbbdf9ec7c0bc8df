"""The xarray-in / xarray-out branch of threshold() (xmhw/xmhw.py:38-247 return
contract) exercised with a minimal stand-in for the xarray API surface it
touches (xarray is not installable in the build image).  The stand-in checks
that only public, version-stable calls are used: DataArray.dims/.values/.attrs/
.coords/.sizes/[name], Dataset(data_vars, coords), ds[name].attrs, ds.attrs."""
import sys
import types

import numpy as np
import numpy.testing as npt
import pytest

import oracle_fast as fast
import xmhw_oracle as ora


class _Var:
    def __init__(self, dims, values, attrs=None, encoding=None):
        self.dims = tuple(dims)
        self.values = np.asarray(values)
        self.attrs = dict(attrs or {})
        self.encoding = dict(encoding or {})


class FakeDataArray:
    def __init__(self, values, dims, coords, attrs=None, encodings=None):
        self.values = np.asarray(values)
        self.dims = tuple(dims)
        self.attrs = dict(attrs or {})
        self.coords = {k: _Var((k,), v[0], v[1], (encodings or {}).get(k)) for k, v in coords.items()}
        self.sizes = dict(zip(self.dims, self.values.shape))

    def __getitem__(self, k):
        return self.coords[k]


class FakeDataset:
    def __init__(self, data_vars, coords):
        self.attrs = {}
        self._vars = {}
        for k, (dims, vals) in data_vars.items():
            self._vars[k] = _Var(dims, vals)
        for k, v in coords.items():
            if isinstance(v, tuple):
                self._vars[k] = _Var((v[0],), v[1])
            else:
                self._vars[k] = _Var((), v)

    def __getitem__(self, k):
        return self._vars[k]


FakeDataArray.__module__ = "xarray.core.dataarray"


@pytest.fixture()
def fake_xarray(monkeypatch):
    mod = types.ModuleType("xarray")
    mod.Dataset = FakeDataset
    mod.DataArray = FakeDataArray
    monkeypatch.setitem(sys.modules, "xarray", mod)
    return mod


def oracle_compute(ts, doy, pctile, w, smooth, width, tstep, cold=False):
    return fast.threshold_cells_fast(ts, doy, pctile=pctile, windowHalfWidth=w, smoothPercentile=smooth,
                                     smoothPercentileWidth=width, tstep=tstep, coldSpells=cold)


def test_xarray_in_xarray_out(fake_xarray, oisst):
    from xmhw_amd.api import _threshold
    da = FakeDataArray(
        oisst["sst"], ("time", "lat", "lon"),
        {"time": (oisst["time64"], {"long_name": "Center time of the day"}),
         "lat": (oisst["lat"], {"units": "degrees_north"}),
         "lon": (oisst["lon"], {"units": "degrees_east"})},
        attrs={"units": "Celsius"}, encodings={"time": {"calendar": "proleptic_gregorian"}})
    ds = _threshold(da, oracle_compute, smoothPercentile=False)
    assert isinstance(ds, FakeDataset)
    ref = ora.threshold_grid(oisst["sst"], oisst["time64"], smoothPercentile=False)
    keep = ref["keep"].reshape(8, 4)
    rows, cols = keep.any(axis=1), keep.any(axis=0)
    assert ds["thresh"].dims == ("doy", "lat", "lon")
    npt.assert_allclose(ds["thresh"].values, ref["thresh"][:, rows][:, :, cols], rtol=1e-13, equal_nan=True)
    npt.assert_allclose(ds["seas"].values, ref["seas"][:, rows][:, :, cols], rtol=1e-13, equal_nan=True)
    npt.assert_array_equal(ds["doy"].values, np.arange(1, 367))
    npt.assert_array_equal(ds["lat"].values, oisst["lat"][rows])
    assert ds["quantile"].values == 0.9
    assert ds["doy"].attrs == {"units": "1", "long_name": "Day of the year"}
    assert ds["lat"].attrs == {"units": "degrees_north"}
    assert ds["thresh"].attrs["units"] == "degree_C" and ds["seas"].attrs["units"] == "degree_C"
    assert "xmhw_parameters" in ds.attrs and ds.attrs["source"].startswith("xmhw code")
    # the caller's array is not mutated (the reference adds a 'doy' coord to it)
    assert set(da.coords) == {"time", "lat", "lon"}


def test_detect_xarray_in_xarray_out(fake_xarray, oisst):
    """detect(): DataArrays in -> Datasets out, laid out like the reference's (events, lat, lon)."""
    from detect_standin import oracle_detect_cells
    from xmhw_amd.api import _threshold
    from xmhw_amd.detect import EVENT_COLUMNS, _detect
    coords = {"time": (oisst["time64"], {"long_name": "Center time of the day"}),
              "lat": (oisst["lat"], {"units": "degrees_north"}),
              "lon": (oisst["lon"], {"units": "degrees_east"})}
    da = FakeDataArray(oisst["sst"], ("time", "lat", "lon"), coords, attrs={"units": "Celsius"},
                       encodings={"time": {"calendar": "proleptic_gregorian"}})
    clim = _threshold(da, oracle_compute)
    th = FakeDataArray(clim["thresh"].values, clim["thresh"].dims,
                       {d: (clim[d].values, clim[d].attrs) for d in clim["thresh"].dims})
    se = FakeDataArray(clim["seas"].values, clim["seas"].dims,
                       {d: (clim[d].values, clim[d].attrs) for d in clim["seas"].dims})
    mhw, inter = _detect(da, th, se, oracle_detect_cells, intermediate=True)
    assert isinstance(mhw, FakeDataset) and isinstance(inter, FakeDataset)
    assert mhw["intensity_max"].dims == ("events", "lat", "lon")
    n_labels = mhw["events"].values.shape[0]
    assert mhw["duration"].values.shape == (n_labels, clim["lat"].values.shape[0], clim["lon"].values.shape[0])
    npt.assert_array_equal(mhw["lat"].values, clim["lat"].values)
    assert mhw["lat"].attrs == {"units": "degrees_north"}
    assert mhw["events"].attrs["long_name"] == "MHW event identifier: starting index"
    assert mhw["rate_onset"].attrs == {"long_name": "MHW onset rate", "units": "degree_C day-1"}
    assert mhw["time_start"].values.dtype.kind == "M"
    assert mhw.attrs["title"].startswith("Marine heatwave events identified")
    for name in EVENT_COLUMNS:
        assert mhw[name].values.shape[0] == n_labels
    # every event of every cell appears exactly once in the dense layout
    ev = mhw["event"].values
    lab = mhw["events"].values
    ok = ~np.isnan(ev)
    assert ok.sum() > 0
    npt.assert_array_equal(ev[ok], np.broadcast_to(lab[:, None, None], ev.shape)[ok])
    assert inter["relSeas"].dims == ("time", "lat", "lon")
    assert inter["relSeas"].values.shape == (731,) + ev.shape[1:]
