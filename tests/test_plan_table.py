"""CPU tests of the host-side plan (C ABI, no GPU): the ring kernel's step
table must reproduce window_roll()'s pools (identify.py:184-209) exactly when
replayed with the kernel's ring semantics."""
import ctypes
import os
import re

import numpy as np
import pytest

import xmhw_oracle as ora
from oracle_fast import pool_index

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _plan(doy, w):
    from xmhw_amd.device import Plan
    return Plan(doy, w)


def replay(table, step_min, R, D):
    """Replay the table with the kernel's ring rules; return per row the sorted
    multiset of time indices pooled (counted tracks, valid samples)."""
    nsteps, ntp = table.shape
    ring = -np.ones((ntp, R), dtype=np.int64)
    pools = []
    for i in range(nsteps):
        s = step_min + i
        m = (s - step_min) % R
        counted = (table[i] & 1).astype(bool)
        code = table[i] >> 1
        for k in range(ntp):
            if code[k] == 0:                      # HOLD: rotate, oldest lands on next slot
                ring[k] = np.roll(ring[k], 1)
            else:
                ring[k, m] = code[k] - 2 if code[k] >= 2 else -1
        if s >= 0:
            sel = ring[counted]
            pools.append(np.sort(sel[sel >= 0]))
    assert len(pools) == D
    return pools


def _check(doy, w, yps):
    p = _plan(doy, w)
    tab = p.table(yps)
    assert tab.shape == (p.nsteps, 8 * yps)
    got = replay(tab, p.step_min, 2 * w + 1, p.D)
    doys, want = pool_index(doy, w)
    np.testing.assert_array_equal(p.doys, doys)
    for r in range(p.D):
        np.testing.assert_array_equal(got[r], np.sort(want[r]), err_msg=f"row {r} (doy {doys[r]})")
    return p


def test_daily_with_leap_years():
    time = np.arange("2003-01-01", "2011-01-01", dtype="datetime64[D]")
    p = _check(ora.add_doy(time), 5, 1)
    assert p.D == 366 and p.ntracks == 8 and p.kernel == "ring"


def test_partial_first_and_last_year():
    time = np.arange("2001-07-15", "2006-03-10", dtype="datetime64[D]")
    p = _check(ora.add_doy(time), 5, 1)
    assert p.ntracks == 6


def test_tstep_axis_and_small_windows():
    doy = np.tile(np.arange(1, 74), 7)
    _check(doy, 2, 1)
    _check(doy, 1, 2)
    _check(np.tile(np.arange(1, 13), 3), 1, 1)


def test_no_leap_year_has_365_rows():
    time = np.arange("2001-01-01", "2004-01-01", dtype="datetime64[D]")
    p = _check(ora.add_doy(time), 3, 1)
    assert p.D == 365 and 60 not in p.doys          # quirk Q5


def test_irregular_axis_gap_in_record():
    time = np.concatenate([np.arange("2001-01-01", "2002-05-01", dtype="datetime64[D]"),
                           np.arange("2002-09-01", "2005-01-01", dtype="datetime64[D]")])
    _check(ora.add_doy(time), 4, 1)


def test_too_many_tracks_falls_back_to_generic():
    doy = np.tile(np.arange(1, 4), 100)              # 100 tracks: 32 lanes per cell (97..192 tracks)
    p = _plan(doy, 5)
    assert p.kernel == "ring"
    doy = np.tile(np.arange(1, 4), 200)              # 200 tracks > 32 * max yps
    p = _plan(doy, 5)
    assert p.kernel == "generic"


def test_ring2_layout_choice():
    """which float32 ring kernel a plan runs on (capi.cpp: ring2_resolved): the third-generation kernel on 2 lanes per
    cell (22) for 9..24 tracks, on 4 lanes (21) for 25..48, on 8 lanes (20) for 49..88 -- each measured faster than
    every second-generation layout there (tools/bench_ring2.py --years) -- and the second-generation kernel on 16 lanes
    (12) for 89..96 tracks; outside the round-1 kernel runs (-1)"""
    from xmhw_amd.device import Plan

    def years(n, w=5, ring2=None):
        t = np.arange("1982-01-01", f"{1982 + n}-01-01", dtype="datetime64[D]")
        return Plan(ora.add_doy(t), w, ring2=ring2)

    # round 5: records of 9..48 tracks run on the sorted-list kernel (layout 40); the ring layouts below are what the same
    # plans run on for quantiles below 0.85, for float64 input and when forced (XMHW_SORTED=0 / layout=...)
    assert years(40).ring2_in_use() == 40 and years(39).ring2_in_use() == 40 and years(9).ring2_in_use() == 40
    assert years(48).ring2_in_use() == 40 and years(20).ring2_in_use() == 40
    assert Plan(np.tile(np.arange(1, 1461), 20), 5).ring2_in_use() == 40     # config 5's tstep axis
    assert years(40, ring2=21).ring2_in_use() == 21           # ... 4 x 10 = 40 tracks on ring3 when forced
    assert years(30, ring2=21).ring2_in_use() == 21           # 4 x 8 (2 padded)
    assert years(25, ring2=21).ring2_in_use() == 21 and years(48, ring2=21).ring2_in_use() == 21      # 7 .. 12 tracks per lane
    assert years(24, ring2=22).ring2_in_use() == 22 and years(13, ring2=22).ring2_in_use() == 22      # 2 lanes per cell up to 24 tracks
    assert years(24, ring2=21).ring2_in_use() == 21                                 # (forced)
    assert years(40, ring2=8).ring2_in_use() == 8 and years(40, ring2=20).ring2_in_use() == 20      # forced
    assert years(20, ring2=22).ring2_in_use() == 22           # 2 x 10 = 20 tracks
    assert years(12, ring2=22).ring2_in_use() == 22 and years(9, ring2=22).ring2_in_use() == 22
    for gone in (0, 7, 9, 11, 30):                            # (layouts this build does not have are refused, not silently
        with pytest.raises(Exception):                        # served by a slow fallback: ADVICE r4)
            years(40, ring2=gone)
    assert years(40, ring2=-1).ring2_in_use() == -1           # round-1 kernel
    assert years(10, ring2=22).ring2_in_use() == 22 and years(10, ring2=10).ring2_in_use() == 10
    assert years(16, ring2=22).ring2_in_use() == 22           # 16 tracks: 2 x 8
    assert years(16, ring2=8).ring2_in_use() == 8             # (8 x 2 = 4 x 4 exactly: the second generation's tie goes to 8 lanes)
    assert years(43, ring2=21).ring2_in_use() == 21           # 41..48 tracks (OISST 1982-2024): 4 x 11
    assert years(49).ring2_in_use() == 20 and years(88).ring2_in_use() == 20   # 49..88 tracks: ring3 on 8 lanes per cell
    assert years(89).ring2_in_use() == 12 and years(96).ring2_in_use() == 12   # 89..96 tracks: 16 lanes per cell (ring2)
    assert years(97).ring2_in_use() == -1                     # beyond: round-1 kernel (32 lanes per cell)
    assert years(40, ring2=12).ring2_in_use() == -1           # (that layout's short-record entries are float64-only)
    assert years(8).ring2_in_use() == -1                      # 8 tracks or fewer: round-1 kernel
    assert years(40, w=3).ring2_in_use() == -1                # other windows: round-1 kernel
    # genuinely float64 samples: the 64-bit mode's layout -- the third-generation kernel on 4 lanes per cell for 13..20
    # tracks (21), on 8 lanes otherwise up to 48 tracks (20), the second-generation one on 16 lanes for shorter and longer
    # records (12)
    assert years(30).f64_mode() == 20 and years(12).f64_mode() == 20
    assert years(20).f64_mode() == 21 and years(13).f64_mode() == 21      # 4 / 5 tracks per lane: the 4-lane layout
    assert years(40).f64_mode() == 20 and years(43).f64_mode() == 20 and years(5).f64_mode() == 12
    assert years(49).f64_mode() == 12 and years(96).f64_mode() == 12             # 16 lanes, low words in LDS
    assert years(97).f64_mode() == -1 and years(40, w=3).f64_mode() == -1        # generic kernel
    with pytest.raises(Exception):
        years(40, ring2=14)
    # the round-4 names: xmhw_plan_set_layout / xmhw_plan_layout_in_use with the XMHW_LAYOUT_* constants
    from xmhw_amd.device import LAYOUTS
    assert LAYOUTS == {"auto": -2, "ring1": -1, "ring2_8lane": 8, "ring2_4lane": 10, "ring2_16lane": 12,
                       "ring3_8lane": 20, "ring3_4lane": 21, "ring3_2lane": 22, "sorted": 40}
    t40 = ora.add_doy(np.arange("1982-01-01", "2022-01-01", dtype="datetime64[D]"))
    assert Plan(t40, 5).layout_in_use() == LAYOUTS["sorted"]
    assert Plan(t40, 5, layout="ring3_4lane").layout_in_use() == LAYOUTS["ring3_4lane"]
    assert Plan(t40, 5, layout="ring3_8lane").layout_in_use() == 20
    assert Plan(t40, 5, layout="ring2_8lane").layout_in_use() == 8
    assert Plan(t40, 5, layout="ring2_4lane").layout_in_use() == 10
    assert Plan(t40, 5, layout="ring1").layout_in_use() == -1
    assert Plan(t40, 5, layout="ring3_2lane").layout_in_use() == -1      # 40 tracks do not fit 2 lanes per cell
    with pytest.raises(Exception):
        Plan(t40, 5, layout=14)
    # xmhw_plan_chunks_in_use: the automatic cut of the doy axis (csrc/capi.cpp: auto_chunks) -- one chunk for the
    # 0.25 degree grid, two for the 1 degree grid, more for small batches; a forced count wins
    p = Plan(t40, 5)
    assert p.chunks_in_use(1036800) == 1 and p.chunks_in_use(64800) == 2 and p.chunks_in_use(20000) == 4
    assert Plan(t40, 5, nchunks=3).chunks_in_use(1036800) == 3
    from xmhw_amd._lib import hip
    assert hip().debug_stats_available() in (False, True)        # (False in the product build: no counter twins)


def test_bad_arguments():
    from xmhw_amd.exception import XmhwException
    with pytest.raises(XmhwException):
        _plan(np.arange(1, 10), -1)
    with pytest.raises(XmhwException):
        _plan(np.zeros(0, dtype=np.int32), 5)


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "xmhw_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(xmhw_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 35
    lib = ctypes.CDLL(os.path.join(ROOT, "xmhw_amd", "libxmhw_amd.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.xmhw_arch.restype = ctypes.c_char_p
    assert lib.xmhw_arch() == b"gfx950"
    assert lib.xmhw_version() >= 1
    # error path without touching a GPU: NULL plan
    lib.xmhw_last_error.restype = ctypes.c_char_p
    assert lib.xmhw_plan_info(None, None, None, None, None, None) == 1
    assert b"NULL" in lib.xmhw_last_error()
