"""The multi-slab grid path (pipelined uploads, background page touching of the result arrays, pitched
placement of the results) against the single-slab path: bit-identical, scattered land included."""
import numpy as np
import numpy.testing as npt
import pytest

pytestmark = pytest.mark.gpu


def test_many_slabs_equal_one_slab():
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.device as dev
    from xmhw_amd.calendar import add_doy
    t = np.arange("2001-01-01", "2011-01-01", dtype="datetime64[D]")
    doy = add_doy(t)
    rng = np.random.default_rng(11)
    T, N = t.shape[0], 4096
    x = (15 + 5 * np.sin(2 * np.pi * np.arange(T)[:, None] / 365.25) + rng.normal(size=(T, N))).astype(np.float32)
    x[:, ::7] = np.nan                      # scattered land
    x[:, 1000:1300] = np.nan                # a whole slab of land (NaN block written while the toucher may still run)
    one = dev.calc_clim_grid_device(x, doy, False, 90, 5, True, 31, False)
    for _ in range(3):                      # the race this guards against is timing dependent: a few rounds
        many = dev.calc_clim_grid_device(x, doy, False, 90, 5, True, 31, False, max_batch_bytes=T * 4 * 4 * 300)
        npt.assert_array_equal(many[0], one[0])
        npt.assert_array_equal(many[2], one[2])
        npt.assert_array_equal(many[3], one[3])
    assert np.isnan(one[2][:, 1000:1300]).all() and not np.isnan(one[2][:, 1]).any()
    dev.release_device_cache()
