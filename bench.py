#!/usr/bin/env python3
"""bench.py -- grid-cells/s of the threshold() hot path on MI355X.

Workload (BASELINE.json configs[2], the one the metric is quoted on): 0.25 deg
global grid (1440 x 720 = 1,036,800 cells), 40-yr daily SST 1982-2021
(T = 14,610), windowHalfWidth=5, pctile=90, smoothPercentileWidth=31,
float32 input resident in HBM, float64 output.

A step = one pass of the path over the rank's cells: raw climatology (ring
kernel) + Feb-29/smoothing (finish kernel), slab by slab; with N > 1 every
slab's (D, slab) result block is gathered to rank 0 over RCCL while the next
slab computes.  Weak scaling: every rank holds a full 0.25 deg grid of its own.

One JSON line on rank 0 (see the driver contract); the `roofline` object is
for the ring kernel (algorithmic bytes = T*4 + 2*D*8 per cell), the
`cpu_baseline` object times the numpy restatement of the reference (oracle/)
on a bounded sample of the same synthetic input on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def daily_doy(y0, y1):
    from xmhw_amd.calendar import add_doy
    t = np.arange(f"{y0}-01-01", f"{y1 + 1}-01-01", dtype="datetime64[D]")
    return add_doy(t)


def _cpu_cell_worker(args):
    import oracle_percell as opc
    x, doy, kw = args
    return opc.threshold_cells_percell(x, doy, **kw)[1:]


def cpu_baseline(sample, doy, kw, budget_cells):
    """Time the per-cell numpy restatement on all host cores (multiprocessing)."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    ncell = min(sample.shape[1], budget_cells)
    parts = [p for p in np.array_split(np.arange(ncell), cores) if p.size]
    jobs = [(np.ascontiguousarray(sample[:, p]), doy, kw) for p in parts]
    ctx = mp.get_context("fork")
    with ctx.Pool(len(jobs)) as pool:
        pool.map(_cpu_cell_worker, [(j[0][:, :1], doy, kw) for j in jobs])  # warm the workers
        t0 = time.perf_counter()
        res = pool.map(_cpu_cell_worker, jobs)
        dt = time.perf_counter() - t0
    th = np.concatenate([r[0] for r in res], axis=1)
    se = np.concatenate([r[1] for r in res], axis=1)
    return ncell / dt, len(jobs), ncell, th, se


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="0.25deg",
                    choices=["0.25deg", "1deg", "0.25deg_nan", "0.05deg_tstep"],
                    help="BASELINE.json preset: 0.25deg = configs[2] (default, the metric's config); "
                         "1deg = configs[1]; 0.25deg_nan = configs[3] (5%% NaN); "
                         "0.05deg_tstep = configs[4] per-GPU share (810,000 cells, 6-hourly, tstep)")
    ap.add_argument("--cells", type=int, default=0, help="cells per GPU (0: the preset's)")
    ap.add_argument("--years", type=int, nargs=2, default=None)
    ap.add_argument("--slabs", type=int, default=0, help="launches per step (0: 1 at N=1, 8 at N>1)")
    ap.add_argument("--nan-frac", type=float, default=0.0)
    ap.add_argument("--kernel", default="auto")
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64", "f64n"],
                    help="input dtype (output is always f64); f64n = float64 holding float32-representable samples "
                         "(decoded int16/float32 archives): runs on the float32 ring kernel")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--force-dist", action="store_true", help="init RCCL and run the gather even with one rank")
    ap.add_argument("--cpu-cells", type=int, default=512)
    ap.add_argument("--parity-cells", type=int, default=512)
    args = ap.parse_args()

    # stdout carries exactly one JSON line: keep a private handle to it and point fd 1 at stderr, so
    # that banners printed by native libraries (RCCL prints its version to stdout) cannot add lines
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist   # --force-dist: exercise the RCCL path with one rank
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from xmhw_amd._lib import hip
    from xmhw_amd.device import Plan, clim_raw, clim_finish
    h = hip()
    h.set_device(local)

    presets = {
        "0.25deg": dict(cells=1440 * 720, years=(1982, 2021), nan=0.0, tstep=False, name="0.25deg global"),
        "1deg": dict(cells=360 * 180, years=(1991, 2020), nan=0.0, tstep=False, name="1deg global"),
        "0.25deg_nan": dict(cells=1440 * 720, years=(1982, 2021), nan=0.05, tstep=False,
                            name="0.25deg global, 5% NaN"),
        "0.05deg_tstep": dict(cells=810000, years=(2001, 2020), nan=0.0, tstep=True,
                              name="0.05deg tile share, 6-hourly no-leap (tstep)"),
    }
    ps = presets[args.config]
    years = tuple(args.years) if args.years else ps["years"]
    if args.nan_frac == 0.0:
        args.nan_frac = ps["nan"]
    tstep = ps["tstep"]
    if tstep:   # 1460 steps per year, no leap days (docs/frequency.rst:42-50), add_doy tstep branch
        nyr = years[1] - years[0] + 1
        doy = np.tile(np.arange(1, 1461, dtype=np.int64), nyr)
    else:
        doy = daily_doy(*years)
    args.years = list(years)
    T = int(doy.shape[0])
    C = int(args.cells) or ps["cells"]
    w, pctile, width = 5, 90, 31
    q = pctile / 100.0
    plan = Plan(doy, w, kernel=args.kernel, nchunks=args.chunks)
    D = plan.D
    nslab = args.slabs or (8 if use_dist else 1)
    bounds = [C * i // nslab for i in range(nslab + 1)]
    slabs = [(bounds[i], bounds[i + 1]) for i in range(nslab) if bounds[i + 1] > bounds[i]]

    # ---- inputs resident in HBM: synthetic SST generated on the device --------------
    isz = 4 if args.dtype == "f32" else 8
    seed = 20260101 + 2
    stream = torch.cuda.current_stream().cuda_stream
    if args.dtype == "f64n":
        ts32 = torch.empty((T, C), dtype=torch.float32, device=dev)
        h.synth_sst(ts32.data_ptr(), 4, T, C, C, rank * C, seed, args.nan_frac, stream)
        ts = ts32.double()
        del ts32
        torch.cuda.empty_cache()
    else:
        ts = torch.empty((T, C), dtype=torch.float32 if isz == 4 else torch.float64, device=dev)
        h.synth_sst(ts.data_ptr(), isz, T, C, C, rank * C, seed, args.nan_frac, stream)
    raw_th = [torch.empty((D, b - a), dtype=torch.float64, device=dev) for a, b in slabs]
    raw_se = [torch.empty((D, b - a), dtype=torch.float64, device=dev) for a, b in slabs]
    out = [torch.empty((2, D, b - a), dtype=torch.float64, device=dev) for a, b in slabs]
    gathered = None
    if use_dist and rank == 0:
        gathered = [[torch.empty((2, D, b - a), dtype=torch.float64, device=dev) for _ in range(world)]
                    for a, b in slabs]
    torch.cuda.synchronize()

    ev = [(h.event_create(), h.event_create(), h.event_create()) for _ in slabs]
    ring_ms, finish_ms = [], []

    def step(timed):
        works = []
        for i, (a, b) in enumerate(slabs):
            n = b - a
            h.event_record(ev[i][0], stream)
            clim_raw(plan, ts.data_ptr() + isz * a, isz, n, q, False, raw_th[i].data_ptr(),
                     raw_se[i].data_ptr(), ld=C, ldo=n, stream=stream)
            h.event_record(ev[i][1], stream)
            clim_finish(plan, raw_th[i].data_ptr(), raw_se[i].data_ptr(), n, not tstep, True, width,
                        out[i][0].data_ptr(), out[i][1].data_ptr(), ldo=n, stream=stream)
            h.event_record(ev[i][2], stream)
            if use_dist:
                works.append(dist.gather(out[i], gathered[i] if rank == 0 else None, dst=0, async_op=True))
        for wk in works:
            wk.wait()
        if timed:
            torch.cuda.synchronize()
            for i in range(len(slabs)):
                ring_ms.append(h.event_elapsed_ms(ev[i][0], ev[i][1]))
                finish_ms.append(h.event_elapsed_ms(ev[i][1], ev[i][2]))

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    ms_per_step = 1e3 * dt / args.steps
    value = world * C * args.steps / dt
    # ---- roofline of the dominant kernel (ring): algorithmic bytes / launch time --------
    bytes_per_cell = T * isz + 2 * D * 8
    cells_per_launch = float(np.mean([b - a for a, b in slabs]))
    ring_avg_ms = float(np.mean(ring_ms))
    achieved = cells_per_launch * bytes_per_cell / (ring_avg_ms * 1e-3) / 1e9

    # HBM traffic per launch from the committed PMC measurement of this kernel (profiles/),
    # scaled to this run's cells per launch; null if no measurement matches the workload
    traffic = None
    valu = None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        mk = prof["kernels"].get("clim_ring_" + args.dtype)
        if mk and mk.get("T") == T and plan.kernel == "ring":
            traffic = mk["hbm_bytes_per_launch"] * cells_per_launch / mk["cells_per_launch"]
            if "valu_insts_per_wave_row" in mk:
                # the resource that actually binds the kernel: VALU issue.
                # wave-instructions per launch from the committed PMC count, peak = 256 CUs x
                # 4 SIMDs x clock / measured issue cycles of the ops the kernel is made of
                rows = D + 10                              # + ring warm-up steps
                insts = mk["valu_insts_per_wave_row"] * rows * cells_per_launch / mk["cells_per_wave"]
                peak = 256 * 4 * 2.4e9 / prof["valu_issue_cycles_per_inst"]
                ach = insts / (ring_avg_ms * 1e-3)
                valu = {"bound": "valu-issue", "achieved": ach / 1e9, "peak": peak / 1e9,
                        "unit": "G wave-instructions/s", "frac": ach / peak,
                        "valu_insts_per_cell_row": mk["valu_insts_per_wave_row"] / mk["cells_per_wave"],
                        "source": "profiles/r1_pmc_sq.txt, profiles/r1_ubench_valu.txt"}
    except (OSError, KeyError, ValueError):
        traffic = None

    result = {
        "metric": "grid-cells/sec for threshold() on 40yr daily SST",
        "value": value,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": f"{args.dtype} in / f64 out",
        "data": "synthetic",
        "config": {
            "workload": f"{ps['name']} {C} cells/GPU, {args.years[0]}-{args.years[1]} (T={T}), "
                        f"windowHalfWidth={w}, pctile={pctile}, smoothPercentileWidth={width}, "
                        f"nan_frac={args.nan_frac}",
            "cells_per_gpu": C, "T": T, "D": D, "kernel": plan.kernel, "slabs": len(slabs),
            "gather": "rccl gather to rank 0, pipelined per slab" if use_dist else "none",
        },
        "roofline": {
            "bound": "hbm", "kernel": "clim_ring_f32 (float64 samples narrowed on load)" if args.dtype == "f64n"
            else "clim_ring_" + args.dtype, "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": "profiles/hbm_traffic.json (rocprofv3 PMC, FETCH_SIZE x2 + WRITE_SIZE)" if traffic else None,
            "algorithmic_bytes_per_launch": cells_per_launch * bytes_per_cell,
            "algorithmic_bytes_per_cell": bytes_per_cell, "cells_per_launch": cells_per_launch,
            "avg_launch_ms": ring_avg_ms,
        },
        "roofline_binding_resource": valu,
        "finish_kernel_avg_launch_ms": float(np.mean(finish_ms)),
    }
    if use_dist:
        # kernel-only and kernel+gather side by side (rank 0's kernels; the step time is the max over ranks)
        kernels_ms = (float(np.sum(ring_ms)) + float(np.sum(finish_ms))) / args.steps
        result["multi_gpu"] = {"kernels_ms_per_step": kernels_ms, "step_ms": ms_per_step,
                               "exposed_gather_ms_per_step": max(ms_per_step - kernels_ms, 0.0),
                               "gathered_bytes_per_step_at_root": float((world - 1) * 2 * D * C * 8)}

    # ---- parity subset + CPU baseline (rank 0, N=1) ------------------------------------
    if rank == 0:
        import oracle_fast as fast
        npar = min(args.parity_cells, C)
        idx = np.unique(np.linspace(0, C - 1, npar).astype(np.int64))
        sample = ts[:, torch.from_numpy(idx).to(dev)].cpu().numpy()
        got = np.concatenate([o.cpu().numpy() for o in out], axis=2)[:, :, idx]
        _, th0, se0 = fast.threshold_cells_fast(sample, doy, pctile=pctile, windowHalfWidth=w,
                                                smoothPercentileWidth=width, tstep=tstep)
        err_th = float(np.nanmax(np.abs(got[0] - th0) / np.abs(th0)))
        err_se = float(np.nanmax(np.abs(got[1] - se0) / np.abs(se0)))
        result["parity"] = {"cells": int(idx.size), "max_rel_err_thresh": err_th,
                            "max_rel_err_seas": err_se, "tolerance": 1e-6,
                            "ok": bool(err_th < 1e-6 and err_se < 1e-6)}
        if world == 1 and not args.no_cpu:
            ncpu = min(args.cpu_cells, C)
            cs = ts[:, :ncpu].cpu().numpy()
            kw = dict(pctile=pctile, windowHalfWidth=w, smoothPercentileWidth=width, tstep=tstep)
            cps, cores, ncell, th_c, se_c = cpu_baseline(cs, doy, kw, ncpu)
            g = np.concatenate([o.cpu().numpy() for o in out], axis=2)[:, :, :ncell]
            result["cpu_baseline"] = {
                "value": cps, "unit": "cells/s", "cores": cores, "kind": "port",
                "sample": f"first {ncell} cells of the same synthetic input, per-cell numpy "
                          f"restatement (366 x np.quantile + mean per cell), multiprocessing over "
                          f"{cores} processes; excludes xarray/dask per-cell overhead",
                "max_rel_diff_vs_gpu": float(np.nanmax(np.abs(g[0] - th_c) / np.abs(th_c))),
            }
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
