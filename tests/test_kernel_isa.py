"""The compiled hot kernel keeps its prefetch: no vmcnt wait inside a comparator block of any clim_sorted_* instantiation
(tools/check_sorted_waits.py says why that is worth a test: one stray wait costs 12 %) -- and its select rounds stay free
of divergent mini-branches (36 of them per round cost 4.8 % in round 6).  Needs hipcc, not a GPU."""
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_no_vmcnt_wait_inside_the_sort_or_the_select_of_any_instantiation():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_sorted_waits as cw
    n, bad = cw.check()
    assert n == 40, n            # 20 float32 + 20 int16 instantiations
    assert not bad, bad


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_no_divergent_mini_branches_in_or_behind_a_select_round():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_sorted_waits as cw
    n, bad = cw.check_branches()
    assert n == 40, n
    assert not bad, bad
