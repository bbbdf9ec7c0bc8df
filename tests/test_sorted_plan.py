"""The sorted-list kernel's chunks and step table (csrc/plan.cpp: sorted_plan; C ABI xmhw_plan_sorted_table), on the CPU.
The kernel pools, per output row, whatever its table says every track pushed at the row's R = 2w+1 last table rows.  For
every calendar below that must be EXACTLY the pool of window_roll() + groupby("doy") (xmhw/identify.py:184-209, :233) as
the oracle builds it -- doy 60 of the non-leap years, partial first / last years, series edges included -- and every row
of the plan must be served by exactly one chunk."""
import numpy as np
import pytest

import xmhw_oracle as ora
from oracle_fast import pool_index


def _table(doy, pieces=1):
    from xmhw_amd.device import Plan
    import xmhw_amd.device as dev
    plan = Plan(np.asarray(doy), 5)
    try:
        chunks, table, flags = dev.hip().plan_sorted_table(plan.handle, pieces)
        return plan.D, np.asarray(chunks), np.asarray(table), np.asarray(flags)
    finally:
        plan.destroy()


def _check(doy, pieces=1):
    D, chunks, table, flags = _table(doy, pieces)
    doys, pools = pool_index(np.asarray(doy), 5)
    assert D == len(doys)
    served = np.zeros(D, dtype=np.int64)
    T = len(doy)
    for ws, b, e, r0 in chunks:
        assert ws == b - 10 and b < e
        for s in range(b, e):
            rows = table[r0 + (s - ws) - 10: r0 + (s - ws) + 1]
            codes = rows >> 1
            assert not np.any(codes == 0), "a held step inside a chunk"
            got = np.sort(codes[codes >= 2].astype(np.int64) - 2)
            assert np.all(got < T)
            np.testing.assert_array_equal(got, np.sort(pools[s]), err_msg=f"row {s}")
            served[s] += 1
        # flags: SIMPLE = every real track pushes a sample; CONSEC = every track pushes the sample after its last one
        nreal = int((table[r0:r0 + (e - ws)] >> 1 != 1).any(axis=0).sum())
        for v in range(e - ws):
            c = table[r0 + v] >> 1
            simple = bool(np.all(c[:nreal] >= 2)) if nreal else False
            assert bool(flags[r0 + v] & 1) == (simple and nreal == _ntracks(doy))
    np.testing.assert_array_equal(served, 1)
    return chunks


def _ntracks(doy):
    doy = np.asarray(doy)
    return 1 + int(np.sum(doy[1:] <= doy[:-1]))


def _daily(start, stop):
    return ora.add_doy(np.arange(start, stop, dtype="datetime64[D]"))


def test_40_year_daily_axis_three_chunks():
    chunks = _check(_daily("1982-01-01", "2022-01-01"))
    # the calendar asks for two cuts: before and after doy 60 (row 59), where the non-leap years hold
    assert [(int(b), int(e)) for _, b, e, _ in chunks] == [(0, 59), (59, 60), (60, 366)]


def test_pieces_keep_every_pool():
    chunks = _check(_daily("1982-01-01", "2022-01-01"), pieces=12)
    assert len(chunks) > 6


def test_partial_first_and_last_year():
    _check(_daily("1982-09-01", "2021-03-15"))


def test_39_tracks_and_a_leap_day_start():
    _check(_daily("1984-02-29", "2023-01-01"))


def test_no_leap_year_in_the_record_is_one_chunk():
    T = 40 * 365
    doy = np.arange(T) % 365 + 1
    doy = np.where(doy >= 60, doy + 1, doy)
    chunks = _check(doy)
    assert len(chunks) == 1


def test_record_ending_on_feb_29():
    _check(_daily("1981-03-01", "2020-03-01"))


@pytest.mark.parametrize("years,k,lds,waves", [(20, 10, 14080, 8), (30, 12, 17920, 8), (36, 14, 20480, 8), (40, 16, 20480, 8),
                                               (45, 18, 23040, 7)])
def test_keys_per_list_and_lds_bytes_per_wave(years, k, lds, waves):
    """xmhw_plan_sorted_info: keys a cell keeps of every row-list and the LDS a wave of 32 cells takes -- rank-major lists,
    11 x the ranks kept in LDS x 128 bytes, rounded up to the 1,280-byte piece LDS is handed out in on gfx950 (round 6: no
    sentinels, no padding rows; 37..40 tracks keep 16 keys, 14 of them in LDS: 20,480 bytes = 8 waves per CU -- the
    registers' limit, two waves per SIMD)"""
    from xmhw_amd.device import Plan
    import xmhw_amd.device as dev
    time = np.arange("1980-01-01", f"{1980 + years}-01-01", dtype="datetime64[D]")
    plan = Plan(ora.add_doy(time), 5)
    try:
        got_k, got_lds, pieces = dev.hip().plan_sorted_info(plan.handle, 1036800)
        assert (got_k, got_lds) == (k, lds) and pieces >= 1
        assert got_lds % 1280 == 0 and min(8, (160 * 1024) // got_lds) == waves
    finally:
        plan.destroy()
