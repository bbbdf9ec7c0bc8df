"""debug: where does layout 40 differ from the generic kernel (rows, cells)"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import xmhw_amd.device as dev
import test_gpu_sorted as T
C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11 + C
nch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
doy = T._daily(1982, 2021)
x = T._series(doy.shape[0], C, seed)
tg, sg, _, _ = T._raw(dev, x, doy, kernel="generic")
t1, s1, st, use = T._raw(dev, x, doy, nchunks=nch, layout="sorted")
bad = np.argwhere(~((t1 == tg) | (np.isnan(t1) & np.isnan(tg))))
print("use", use, "mismatches", len(bad))
for r, c in bad[:20]:
    print("row", r, "cell", c, "sorted", t1[r, c], "generic", tg[r, c], "seas diff", s1[r, c] - sg[r, c])
