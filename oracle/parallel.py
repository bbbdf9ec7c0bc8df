"""Process-pool front end of the CPU oracles.  TEST / BENCH INFRASTRUCTURE ONLY.

Nothing in the product path imports this module: it exists so that (a) the GPU parity tests can
check thousands of cells against ``oracle_fast`` in seconds on the many host cores of a GPU box,
and (b) ``bench.py``'s ``cpu_baseline`` leg can time the per-cell restatement
(``oracle_percell``, one calc_clim per cell as xmhw/xmhw.py:184-197 does) on every physical core
with the pool start-up and the pool-index construction outside the timed region.

Workers are SPAWNED (never forked): the parent may already hold a HIP context, and a forked child
would inherit its locked runtime state.  The sample travels through POSIX shared memory.
"""
import multiprocessing as mp
import os
import time
from multiprocessing import shared_memory

import numpy as np

_STATE = {}


def physical_cores():
    """(usable physical cores, logical cpus) of this process: distinct (package, core) pairs among
    the cpus in the affinity mask, from /proc/cpuinfo; falls back to the logical count."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    logical = len(allowed)
    try:
        cores = set()
        cpu = pkg = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("processor"):
                    cpu = int(line.split(":")[1])
                    pkg = core = None
                elif line.startswith("physical id"):
                    pkg = int(line.split(":")[1])
                elif line.startswith("core id"):
                    core = int(line.split(":")[1])
                    if cpu in allowed:
                        cores.add((pkg, core))
        if cores:
            return min(len(cores), logical), logical
    except (OSError, ValueError):
        pass
    return logical, logical


def cpu_quota():
    """CPUs' worth of time the cgroup grants this process (cpu.max), None if unlimited / unknown"""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] == "max":
                    return None
                return float(txt[0]) / float(txt[1])
            q = float(txt[0])
            if q <= 0:
                return None
            return q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError, IndexError):
            continue
    return None


def _init(oracle_dir, doy, w):
    import sys
    if oracle_dir not in sys.path:
        sys.path.insert(0, oracle_dir)
    import oracle_percell as opc
    doy = np.asarray(doy, dtype=np.int64)
    _STATE["doy"] = doy
    _STATE["pools"] = opc._pool_index(doy, w)      # built once per worker, outside any timed region
    _STATE["w"] = w


def _attach(name, shape, dtype):
    shm = shared_memory.SharedMemory(name=name)
    return shm, np.ndarray(shape, dtype=dtype, buffer=shm.buf)


def _percell_job(args):
    name, shape, dtype, a, b, kw = args
    import oracle_percell as opc
    shm, x = _attach(name, shape, dtype)
    try:
        cols = np.ascontiguousarray(x[:, a:b])
    finally:
        del x
        shm.close()
    t0, c0 = time.perf_counter(), time.process_time()
    _, th, se = opc.threshold_cells_percell(cols, _STATE["doy"], pools=_STATE["pools"], **kw)
    return a, b, th, se, (time.perf_counter() - t0, time.process_time() - c0)


def _fast_job(args):
    name, shape, dtype, a, b, kw = args
    import oracle_fast as fast
    shm, x = _attach(name, shape, dtype)
    try:
        cols = np.ascontiguousarray(x[:, a:b])
    finally:
        del x
        shm.close()
    _, th, se = fast.threshold_cells_fast(cols, _STATE["doy"], **kw)
    return a, b, th, se, (0.0, 0.0)


def _noop(_):
    return os.getpid()


class OraclePool:
    """A pool of spawned workers holding the doy labels and the pool index of one plan."""

    def __init__(self, doy, windowHalfWidth=5, workers=None):
        phys, logical = physical_cores()
        self.physical, self.logical = phys, logical
        # one worker per physical core, but not more than the cgroup lets run at once: 128 processes on a
        # 16-CPU quota measure the throttle, not the function
        quota = cpu_quota()
        usable = phys if quota is None else max(1, min(phys, int(-(-quota // 1))))
        self.quota = quota
        self.workers = int(workers or usable)
        ctx = mp.get_context("spawn")
        oracle_dir = os.path.dirname(os.path.abspath(__file__))
        # the workers only ever run functions of THIS module: keep multiprocessing from re-importing
        # the parent's __main__ in every child (bench.py, pytest, an interactive script on stdin)
        import sys
        main = sys.modules.get("__main__")
        saved_spec = getattr(main, "__spec__", None)
        saved_file = getattr(main, "__file__", None)
        try:
            if main is not None:
                main.__spec__ = None
                if saved_file is not None:
                    del main.__file__
            self._pool = ctx.Pool(self.workers, initializer=_init,
                                  initargs=(oracle_dir, np.asarray(doy, dtype=np.int64), int(windowHalfWidth)))
        finally:
            if main is not None:
                main.__spec__ = saved_spec
                if saved_file is not None:
                    main.__file__ = saved_file
        # make sure every worker is up (imports done, pool index built) before anything is timed
        self._pool.map(_noop, range(4 * self.workers), chunksize=1)

    def close(self):
        if self._pool is not None:
            self._pool.terminate()
            self._pool.join()
            self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _run(self, job, sample, blocks, kw):
        sample = np.ascontiguousarray(sample)
        T, C = sample.shape
        shm = shared_memory.SharedMemory(create=True, size=max(sample.nbytes, 1))
        try:
            view = np.ndarray(sample.shape, dtype=sample.dtype, buffer=shm.buf)
            view[...] = sample
            del view
            jobs = [(shm.name, sample.shape, sample.dtype, a, b, kw) for a, b in blocks]
            t0 = time.perf_counter()
            res = self._pool.map(job, jobs, chunksize=1)
            wall = time.perf_counter() - t0
        finally:
            shm.close()
            shm.unlink()
        D = res[0][2].shape[0]
        th = np.empty((D, C))
        se = np.empty((D, C))
        busy = cpu = 0.0
        for a, b, t, s, dt in res:
            th[:, a:b] = t
            se[:, a:b] = s
            busy += dt[0]
            cpu += dt[1]
        return th, se, wall, (busy, cpu)

    def blocks(self, C, per_worker_min=16):
        n = max(1, min(self.workers, C // max(1, per_worker_min)))
        edges = [C * i // n for i in range(n + 1)]
        return [(edges[i], edges[i + 1]) for i in range(n) if edges[i + 1] > edges[i]]

    def threshold_fast(self, sample, **kw):
        """oracle_fast.threshold_cells_fast over column blocks: (thresh, seas)."""
        C = sample.shape[1]
        n = max(1, min(self.workers, (C + 127) // 128))
        edges = [C * i // n for i in range(n + 1)]
        blocks = [(edges[i], edges[i + 1]) for i in range(n) if edges[i + 1] > edges[i]]
        th, se, _, _ = self._run(_fast_job, sample, blocks, kw)
        return th, se

    def time_percell(self, sample, per_worker_min=16, **kw):
        """Time oracle_percell over all workers.  Returns a dict with cells/s (wall), the summed
        in-worker compute seconds, and the results."""
        C = sample.shape[1]
        blocks = self.blocks(C, per_worker_min)
        th, se, wall, (busy, cpu) = self._run(_percell_job, sample, blocks, kw)
        # busy = summed in-worker wall time, cpu = summed in-worker CPU time: cpu << busy means the
        # processes did not get a core each (a cgroup CPU quota, other tenants on the box)
        return {"cells": C, "wall_s": wall, "busy_s": busy, "cpu_s": cpu, "processes": len(blocks),
                "cells_per_s": C / wall, "cells_per_s_per_core": C / busy if busy > 0 else float("nan"),
                "cells_per_cpu_s": C / cpu if cpu > 0 else float("nan"), "cpu_quota": cpu_quota(),
                "thresh": th, "seas": se}
