"""xmhw_amd/hdf5min.py -- the netCDF-4 / HDF5 subset reader -- against (a) the reference's OWN netCDF-4 fixtures
(tests/golden/ref_testdata/*.nc: data files copied from the reference's test/testdata/; their contents as arrays are
also in tests/golden/*.npz, extracted independently with h5py by tools/make_golden.py) and (b) small files written with
h5py by tools/make_golden_hdf5.py that cover the other half of the format: old-style groups, version-1 object headers
with continuation blocks, packed int16 of both byte orders, shuffle + deflate + fletcher32, ragged chunks, a
never-written dataset, variable-length strings.  CPU only, no h5py needed."""
import os

import numpy as np
import numpy.testing as npt
import pytest

from xmhw_amd import XmhwException, hdf5min, ingest
from xmhw_amd.device import is_packed
from ingest_oracle import decode_packed

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_reference_oisst_fixture_equals_the_extracted_golden():
    f = hdf5min.File(os.path.join(G, "ref_testdata", "oisst_2003_2004.nc"))
    g = np.load(os.path.join(G, "oisst_2003_2004.npz"))
    assert sorted(f.keys()) == ["lat", "lon", "sst", "time"]
    v = f["sst"]
    assert v.shape == (731, 8, 4) and v.dtype == np.dtype("<f4") and v.layout[0] == "chunked"
    assert [fid for fid, _ in v.filters] == [2, 1]                      # shuffle, deflate
    npt.assert_array_equal(v.read(), g["sst"])
    npt.assert_array_equal(f["lat"].read(), g["lat"])
    npt.assert_array_equal(f["lon"].read(), g["lon"])
    npt.assert_array_equal(f["time"].read(), g["time"])
    # nine attributes on the coordinate variables: netCDF-4 keeps those in a fractal heap ("dense" storage)
    assert f["time"].attrs["units"] == "days since 2003-01-01 12:00:00"
    assert f["time"].attrs["calendar"] == str(g["time_calendar"])
    assert list(f["sst"].attrs["_Netcdf4Coordinates"]) == [2, 0, 1] and f["time"].attrs["_Netcdf4Dimid"] == 2


@pytest.mark.parametrize("name,npz", [("test_clim_oisst.nc", "clim_oisst.npz"), ("test_clim_oisst_nosmooth.nc", "clim_oisst.npz")])
def test_reference_clim_fixtures(name, npz):
    f = hdf5min.File(os.path.join(G, "ref_testdata", name))
    g = np.load(os.path.join(G, npz))
    prefix = "nosmooth_" if "nosmooth" in name else "smooth_"
    for k in ("thresh1", "thresh2", "seas1", "seas2"):
        a = f[k].read()
        assert a.shape == (366,) and a.dtype == np.dtype("<f8")
        npt.assert_array_equal(a, g[prefix + k])


def test_open_series_on_the_reference_fixture():
    gs = ingest.open_series(os.path.join(G, "ref_testdata", "oisst_2003_2004.nc"))
    g = np.load(os.path.join(G, "oisst_2003_2004.npz"))
    assert gs.dims == ("time", "lat", "lon") and is_packed(gs.values)
    assert gs.values.decode["out"] == "float32" and gs.values.decode["fill"] is None      # NaN fill value: nothing to do
    npt.assert_array_equal(np.asarray(gs.values), g["sst"])
    assert gs.coords["time"][0] == np.datetime64("2003-01-01T12:00:00") and gs.time_encoding == {"calendar": "proleptic_gregorian"}
    assert gs.attrs["units"] == "Celsius" and gs.coord_attrs["lat"]["units"] == "degrees_north"
    land = ingest.open_series(os.path.join(G, "ref_testdata", "land.nc"), "sst")
    assert land.values.shape == (731, 40, 80) and np.isnan(np.asarray(land.values)).all()


def test_h5py_written_file_old_style_groups_packed_filters():
    f = hdf5min.File(os.path.join(G, "hdf5", "packed_earliest.h5"))
    e = np.load(os.path.join(G, "hdf5", "expected.npz"))
    assert f.b.mm[8] == 0                                               # superblock 0: symbol-table groups
    be, le = f["sst_be_contig"], f["sst_le_chunked"]
    assert be.dtype == np.dtype(">i2") and be.contiguous_offset() is not None
    assert le.dtype == np.dtype("<i2") and [fid for fid, _ in le.filters] == [2, 1, 3] and le.contiguous_offset() is None
    npt.assert_array_equal(be.read(), e["packed"])
    npt.assert_array_equal(le.read(), e["packed"])                      # chunks (64, 3, 4) over (200, 8, 4): ragged edges
    npt.assert_array_equal(f["f32_chunked"].read(), e["sst"])
    npt.assert_array_equal(f["never_written"].read(), np.full((5, 3), -7.5))
    assert le.attrs["vlen_text"] == "a variable-length string" and le.attrs["note_11"] == "attribute number 11"
    assert be.attrs["scale_factor"].dtype == np.float32 and le.attrs["scale_factor"].dtype == np.float64
    # the series as the ingest sees them: zero-copy window + pread recipe for the contiguous one
    a = ingest.open_series(os.path.join(G, "hdf5", "packed_earliest.h5"), "sst_be_contig")
    b = ingest.open_series(os.path.join(G, "hdf5", "packed_earliest.h5"), "sst_le_chunked")
    assert a.dims == b.dims == ("time", "lat", "lon")
    assert a.values.decode["out"] == "float32" and "file" in a.values.decode and a.values.dtype == np.dtype(">i2")
    assert b.values.decode["out"] == "float64" and "file" not in b.values.decode
    want = e["packed"].astype(np.float32) * np.float32(0.01) + np.float32(10.0)
    want[e["packed"] == -32768] = np.nan
    npt.assert_array_equal(decode_packed(a.values), want)
    assert a.coords["time"][1] == np.datetime64("2003-01-02T12:00:00")


def test_new_file_format_is_refused_by_name():
    f = hdf5min.File(os.path.join(G, "hdf5", "latest_layout4.h5"))
    assert f.b.mm[8] == 3
    npt.assert_array_equal(f["y"].read(), np.arange(6, dtype=np.float32))       # contiguous: the same in both formats
    with pytest.raises(XmhwException, match="fixed array"):
        f["x"]
    with pytest.raises(XmhwException, match="not an HDF5"):
        hdf5min.File(os.path.join(G, "oisst_2003_2004.npz"))


def test_dense_attributes_and_links_of_a_file_edited_in_place():
    """300 attributes on one dataset (fractal heap + a two-level version-2 B-tree name index), three of them deleted,
    one rewritten longer, scale_factor rewritten with another type, a dataset unlinked and another linked -- what
    ncatted / NCO leave behind: free-space gaps and stale messages in the heap.  Every attribute and link must come
    back as h5py reads them (a lost scale_factor or _FillValue decodes packed data as raw counts)."""
    import json
    f = hdf5min.File(os.path.join(G, "hdf5", "dense_rewritten.h5"))
    want = json.load(open(os.path.join(G, "hdf5", "dense_rewritten.json")))
    assert sorted(f.keys()) == want["names"] and "var_07" not in f.keys() and "var_07b" in f.keys()
    got = f["sst"].attrs
    assert set(got) == set(want["attrs"]) and len(got) == 302
    for k, v in want["attrs"].items():
        g = got[k]
        if isinstance(v, str):
            assert g == v, k
        else:
            assert float(g) == v, k
    assert got["scale_factor"].dtype == np.float64 and got["note_150"].startswith("rewritten")
    npt.assert_array_equal(f["var_39"].read(), np.arange(40, dtype=np.float32))
    s = ingest.open_series(os.path.join(G, "hdf5", "dense_rewritten.h5"), "sst")
    assert s.values.decode["scale"] == 0.01 and s.values.decode["fill"] == -32768
