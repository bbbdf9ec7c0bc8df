"""CPU oracle for the detect() front end.  TEST INFRASTRUCTURE ONLY.

Restates, for ONE cell, what the reference does between the climatology and the event
statistics:

* define_events() front part (xmhw/identify.py:366-372): re-expand thresh/seas along time by
  doy label (``th.sel(doy=ts.doy)``) and mark exceedances ``ts > thresh`` (NaN compares False);
* mhw_filter() (identify.py:415-479): runs of exceedances of at least minDuration steps;
* join_gaps()/join_events() (identify.py:273-325, :532-536): merge selected events separated by
  at most maxGap steps.

Written as plain loops (the reference is vectorised pandas); PINNED against outputs of the
reference's own functions (tests/golden/mhw_filter_cases.npz, produced by
tools/make_golden_detect.py by importing /root/reference) and the expectations of the
reference's tests test_mhw_filter / test_join_gaps (test/test_identify.py:88-118).

Quirks kept on purpose:
* a run that starts at index 0 has no preceding non-exceedance; the reference fills that with
  0, so the run's label/start is 1 (not 0), its first step is not part of the event and its
  length counts one less (identify.py:445-449);
* `start` is stored at the END index of the first member of a joined group and `end` at the end
  index of its last member (identify.py:316-321);
* join_events() relabels positions start..end of a joined group, gap steps included.
"""
import numpy as np


def exceedance(ts, thresh_doy, row_of_t):
    """bthresh[t] = ts[t] > thresh[row_of_t[t]] (identify.py:367-372); NaN -> False."""
    ts = np.asarray(ts, dtype=np.float64)
    th = np.asarray(thresh_doy, dtype=np.float64)[np.asarray(row_of_t)]
    with np.errstate(invalid="ignore"):
        return ts > th


def mhw_filter(bthresh, minDuration=5, joinGaps=True, maxGap=2):
    """(start, end, events) float64 arrays of length T with NaN where undefined."""
    b = np.asarray(bthresh, dtype=bool)
    T = b.shape[0]
    start = np.full(T, np.nan)
    end = np.full(T, np.nan)
    events = np.full(T, np.nan)
    selected = []                       # (start label, end position) of runs kept
    prev_nonexc = -1                    # last index with not b; -1: none yet
    t = 0
    while t < T:
        if not b[t]:
            prev_nonexc = t
            t += 1
            continue
        p = prev_nonexc if prev_nonexc >= 0 else 0      # fillna(0), identify.py:445
        te = t
        while te + 1 < T and b[te + 1]:
            te += 1
        length = te - p                                   # events_map at the run's last step
        if length >= minDuration:                         # shifted <= -minDuration, :460
            start[te] = p + 1                             # end - duration + 1, :466
            end[te] = te
            for k in range(t, te + 1):
                if k - p != 0:                            # events_map != 0, :473
                    events[k] = p + 1
            selected.append((p + 1, te))
        t = te + 1
    if joinGaps and len(selected) > 1:                    # join_gaps, :309-323
        groups = [[selected[0]]]
        for (s, e) in selected[1:]:
            prev_e = groups[-1][-1][1]
            if s - prev_e > maxGap + 1:                   # gap longer than maxGap: new group
                groups.append([(s, e)])
            else:
                groups[-1].append((s, e))
        for g in groups:
            if len(g) == 1:
                continue
            gs, ge = g[0][0], g[-1][1]
            for (s, e) in g[1:]:
                start[e] = np.nan                         # only the group's first start stays
            for (s, e) in g[:-1]:
                end[e] = np.nan                           # only the group's last end stays
            events[int(gs):int(ge) + 1] = gs              # join_events, :532-536
    return start, end, events


def detect_front(ts, thresh_doy, row_of_t, minDuration=5, joinGaps=True, maxGap=2, coldSpells=False):
    """Exceedance + event filter for one cell; thresh_doy is indexed by row (distinct doy labels)."""
    ts = np.asarray(ts, dtype=np.float64)
    if coldSpells:
        ts = -1.0 * ts                                    # xmhw.py:413-414
    b = exceedance(ts, thresh_doy, row_of_t)
    return (b,) + mhw_filter(b, minDuration, joinGaps, maxGap)
