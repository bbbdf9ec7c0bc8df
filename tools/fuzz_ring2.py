"""Long randomised cross-check of the ring2 kernel (both lane layouts) against the generic kernel:
the generator and the check of tests/test_gpu_ring2.py::test_ring2_random_cases_equal_generic_kernel,
as many cases as asked for.   python tools/fuzz_ring2.py [--cases 400] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from xmhw_amd._lib import require_gpu
    require_gpu()
    import xmhw_amd.device as dev
    import test_gpu_ring2 as t
    rng = np.random.default_rng(args.seed)
    t0 = time.perf_counter()
    layouts = {0: 0, 7: 0}
    for i in range(args.cases):
        x, doy, pct, tstep, cold, nchunks = t.random_ring2_case(rng)
        seen = t.check_ring2_case(dev, x, doy, pct, tstep, cold, nchunks,
                                  msg=f"seed {args.seed} case {i}: T={x.shape[0]} C={x.shape[1]} pct={pct} tstep={tstep} cold={cold}")
        for v in seen:
            layouts[v] += 1
    print(f"{args.cases} random cases, runs per layout {layouts}: 0 mismatches against the generic kernel "
          f"({time.perf_counter() - t0:.0f} s)")


if __name__ == "__main__":
    main()
