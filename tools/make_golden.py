#!/opt/conda/bin/python3.9
"""Extract golden vectors from the reference's own test fixtures.

Run in the BUILD container only (it reads /root/reference, which does not
exist on the GPU box):

    /opt/conda/bin/python3.9 tools/make_golden.py

The reference's fixtures are netCDF4 (HDF5) files; the system python has no
netCDF reader, the conda python has h5py.  Only DATA is copied (inputs and
expected outputs held by the reference's tests) -- no reference source text.

Sources (all relative to /root/reference):
  test/testdata/oisst_2003_2004.nc      sst(731,8,4) f32, time, lat, lon
  test/testdata/test_clim_oisst.nc      thresh1/2, seas1/2 (366,) f64, smoothed
  test/testdata/test_clim_oisst_nosmooth.nc   same, unsmoothed
  test/testdata/land.nc                 only its shape + "all NaN" is recorded
  test/xmhw_fixtures.py:69-73           oisst_doy literal   (rebuilt as data)
  test/xmhw_fixtures.py:96-98           tstack literal
  test/test_identify.py:52-59           feb29 expected scalar
  test/test_identify.py:62-77           runavg vectors
  test/test_identify.py:132-155         land_check expected shapes
"""
import os
import numpy as np
import h5py

REF = "/root/reference/test/testdata"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def attrs_of(ds):
    out = {}
    for k, v in ds.attrs.items():
        if k in ("DIMENSION_LIST", "REFERENCE_LIST", "CLASS", "NAME",
                 "_Netcdf4Dimid", "_Netcdf4Coordinates", "_FillValue"):
            continue
        out[k] = v.decode() if isinstance(v, bytes) else str(v)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    with h5py.File(os.path.join(REF, "oisst_2003_2004.nc"), "r") as f:
        sst = f["sst"][...].astype(np.float32)
        time = f["time"][...].astype(np.int64)
        lat = f["lat"][...].astype(np.float32)
        lon = f["lon"][...].astype(np.float32)
        tattrs = attrs_of(f["time"])
        sattrs = attrs_of(f["sst"])
        lat_attrs = attrs_of(f["lat"])
        lon_attrs = attrs_of(f["lon"])
    assert sst.shape == (731, 8, 4)
    np.savez_compressed(
        os.path.join(OUT, "oisst_2003_2004.npz"),
        sst=sst, time=time, lat=lat, lon=lon,
        time_units=np.array(tattrs["units"]),
        time_calendar=np.array(tattrs["calendar"]),
        sst_units=np.array(sattrs.get("units", "")),
        sst_long_name=np.array(sattrs.get("long_name", "")),
        lat_units=np.array(lat_attrs.get("units", "")),
        lon_units=np.array(lon_attrs.get("units", "")),
    )
    clim = {}
    for tag, fn in (("smooth", "test_clim_oisst.nc"),
                    ("nosmooth", "test_clim_oisst_nosmooth.nc")):
        with h5py.File(os.path.join(REF, fn), "r") as f:
            for v in ("thresh1", "thresh2", "seas1", "seas2"):
                clim[f"{tag}_{v}"] = f[v][...].astype(np.float64)
    # points the clim files were computed at (test/xmhw_fixtures.py:31-33)
    clim["point1_latlon"] = np.array([-42.625, 148.125])
    clim["point2_latlon"] = np.array([-41.625, 148.375])
    np.savez_compressed(os.path.join(OUT, "clim_oisst.npz"), **clim)

    with h5py.File(os.path.join(REF, "land.nc"), "r") as f:
        land = f["sst"][...]
        land_shape = np.array(land.shape)
        land_all_nan = bool(np.isnan(land).all())
    # literal vectors held by the reference tests, rebuilt as data
    a = np.arange(1, 367)
    oisst_doy = np.concatenate((np.delete(a, [59]), a)).astype(np.int64)
    np.savez_compressed(
        os.path.join(OUT, "literals.npz"),
        oisst_doy=oisst_doy,
        days5_doy=np.concatenate((np.arange(1, 74), np.arange(1, 74))).astype(np.int64),
        mon_doy=np.concatenate((np.arange(1, 13), np.arange(1, 13))).astype(np.int64),
        tstack=np.array([16.99, 17.39, 16.99, 17.39, 17.3, 17.39, 17.3]),
        feb29_expected=np.array([18.13]),          # at [1, 2], decimal=5
        feb29_two_day_alt=np.array([18.2074995]),  # Oliver's 28Feb/1Mar mean
        runavg_in=np.array([1, 2, 2, 4, 3, 2], dtype=np.float64),
        runavg_w3=np.array([1.66667, 1.66667, 2.66667, 3.0, 3.0, 2.0]),
        runavg_w5=np.array([2.0, 2.2, 2.4, 2.6, 2.4, 2.4]),
        land_shape=land_shape,
        land_all_nan=np.array(land_all_nan),
        land_check_ocean_cells=np.array(12),
        land_check_anynans_cells=np.array(11),
        land_check_nan_index=np.array([245, 1, 2]),
    )
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
