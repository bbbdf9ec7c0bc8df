"""Small HDF5 files for tests/test_hdf5min.py, written with h5py (which only /opt/conda/bin/python3.9 of the build
container has; the tests themselves need no h5py).  They cover what the reference's own netCDF-4 fixtures
(tests/golden/ref_testdata/*.nc, copied from the reference's test/testdata/) do not: old-style groups and version-1
object headers (h5py's default), packed int16 of both byte orders with float32 / float64 packing attributes,
contiguous and chunked + shuffle + deflate + fletcher32 layouts, a ragged last chunk, a never-written dataset,
more than eight attributes on one object, variable-length string attributes.

    /opt/conda/bin/python3.9 tools/make_golden_hdf5.py        # writes tests/golden/hdf5/*.h5 and expected.npz
"""
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "hdf5")


def main():
    os.makedirs(OUT, exist_ok=True)
    g = np.load(os.path.join(OUT, "..", "oisst_2003_2004.npz"))
    sst = g["sst"][:200]                                   # (200, 8, 4) float32, land = NaN
    packed = np.where(np.isnan(sst), -32768, np.round((sst - 10.0) / 0.01)).astype(np.int16)
    expected = {}
    for name in ("earliest",):
        fn = os.path.join(OUT, f"packed_{name}.h5")
        with h5py.File(fn, "w", libver=("earliest", "v108")) as f:
            d = f.create_dataset("sst_be_contig", data=packed.astype(">i2"))
            d.attrs["scale_factor"] = np.float32(0.01)
            d.attrs["add_offset"] = np.float32(10.0)
            d.attrs["_FillValue"] = np.array([-32768], dtype=">i2")
            d.attrs["units"] = np.bytes_("degC")
            c = f.create_dataset("sst_le_chunked", data=packed.astype("<i2"), chunks=(64, 3, 4), shuffle=True,
                                 compression="gzip", compression_opts=4, fletcher32=True)
            c.attrs["scale_factor"] = 0.01
            c.attrs["add_offset"] = 10.0
            c.attrs["_FillValue"] = np.int16(-32768)
            for i in range(12):                              # more than eight attributes on one object
                c.attrs[f"note_{i:02d}"] = np.bytes_(f"attribute number {i}")
            c.attrs["vlen_text"] = "a variable-length string"
            f.create_dataset("f32_chunked", data=sst, chunks=(50, 8, 4), compression="gzip")
            f.create_dataset("never_written", shape=(5, 3), dtype="<f8", fillvalue=-7.5)
            t = f.create_dataset("time", data=np.arange(200, dtype="<i8"))
            t.attrs["units"] = np.bytes_("days since 2003-01-01 12:00:00")
            t.attrs["calendar"] = np.bytes_("proleptic_gregorian")
            f.create_dataset("lat", data=g["lat"].astype("<f4"))
            f.create_dataset("lon", data=g["lon"].astype("<f4"))
            for dn, dv in (("time", t), ("lat", f["lat"]), ("lon", f["lon"])):
                dv.make_scale(dn)
            for dname in ("sst_be_contig", "sst_le_chunked", "f32_chunked"):
                for i, dn in enumerate(("time", "lat", "lon")):
                    f[dname].dims[i].attach_scale(f[dn])
    # a file of the NEW format (superblock 3, data layout message version 4): refused by name
    with h5py.File(os.path.join(OUT, "latest_layout4.h5"), "w", libver="latest") as f:
        f.create_dataset("x", data=np.arange(24, dtype="<f4").reshape(2, 3, 4), chunks=(1, 3, 4))
        f.create_dataset("y", data=np.arange(6, dtype="<f4"))
    # dense attributes and links (new-style object headers, fractal heap + version-2 B-tree name index), then EDITED in
    # place as ncatted / NCO do: attributes deleted and rewritten with other sizes leave free-space gaps and stale
    # messages in the heap, and 300 attributes make the name index two levels deep
    import json
    fn = os.path.join(OUT, "dense_rewritten.h5")
    with h5py.File(fn, "w", libver=("v108", "v108")) as f:
        d = f.create_dataset("sst", data=packed[:20].astype("<i2"), chunks=(10, 8, 4))
        d.attrs["scale_factor"] = np.float32(0.01)
        d.attrs["add_offset"] = np.float32(10.0)
        d.attrs["_FillValue"] = np.int16(-32768)
        for i in range(300):
            d.attrs[f"note_{i:03d}"] = np.bytes_(f"attribute number {i}")
        for i in range(40):
            f.create_dataset(f"var_{i:02d}", data=np.arange(i + 1, dtype="<f4"))
    with h5py.File(fn, "r+", libver=("v108", "v108")) as f:
        d = f["sst"]
        for i in (3, 150, 299):
            del d.attrs[f"note_{i:03d}"]
        d.attrs["note_150"] = np.bytes_("rewritten, and a good deal longer than the message it replaces " * 3)
        del d.attrs["scale_factor"]
        d.attrs["scale_factor"] = np.float64(0.01)           # (an edit that changes the type, as ncatted -a does)
        d.attrs["history"] = np.bytes_("edited in place")
        del f["var_07"]
        f.create_dataset("var_07b", data=np.arange(3, dtype="<f4"))
    with h5py.File(fn, "r") as f:
        attrs = {}
        for k, v in f["sst"].attrs.items():
            attrs[k] = v.decode() if isinstance(v, bytes) else float(v)
        json.dump({"attrs": attrs, "names": sorted(f.keys())}, open(os.path.join(OUT, "dense_rewritten.json"), "w"), indent=0, sort_keys=True)
    expected["packed"] = packed
    expected["sst"] = sst
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **expected)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
