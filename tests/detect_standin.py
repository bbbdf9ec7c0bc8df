"""Oracle-based stand-in for detect_front.detect_cells (same signature and return value), used by
the CPU tests of detect()'s host logic and as the checker of the GPU end-to-end tests."""
import numpy as np

import detect_oracle as det
import features_oracle as fo

F64 = ["seas", "thresh", "relSeas", "relThresh", "relThreshNorm", "severity", "cats", "mabs"]
U8 = ["duration_moderate", "duration_strong", "duration_severe", "duration_extreme"]


def oracle_detect_cells(ts, seas, thresh, doy, doys, minDuration=5, joinGaps=True, maxGap=2, coldSpells=False,
                        intermediate=False):
    ts = np.asarray(ts)
    T, C = ts.shape
    rows = np.searchsorted(doys, doy)
    x = -ts.astype(np.float64) if coldSpells else ts.astype(np.float64)
    tables, counts = [], []
    inter = None
    if intermediate:
        inter = {k: np.empty((T, C)) for k in F64 + ["events"]}
        inter.update({k: np.empty((T, C), dtype=bool) for k in U8 + ["bthresh"]})
        inter["ts"] = -ts if coldSpells else ts.copy()
    for c in range(C):
        b, s, e, ev = det.detect_front(x[:, c], thresh[:, c], rows, minDuration, joinGaps, maxGap)
        tab = fo.event_table(x[:, c], seas[rows, c], thresh[rows, c], s, e, ev)
        tables.append(tab)
        counts.append(tab.shape[0])
        if intermediate:
            ic = fo.intermediate_columns(x[:, c], seas[rows, c], thresh[rows, c], ev)
            for k in F64 + U8:
                inter[k][:, c] = ic[k]
            inter["events"][:, c] = ev
            inter["bthresh"][:, c] = b
    offsets = np.zeros(C + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    return dict(table=np.concatenate(tables, axis=0) if tables else np.zeros((0, 31)), offsets=offsets, inter=inter)
