#!/usr/bin/env python3
"""bench.py -- grid-cells/s of the threshold() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 (default): BASELINE.json configs[2], the configuration the metric is quoted on -- 0.25 deg
global grid (1440 x 720 = 1,036,800 cells), 40-yr daily SST 1982-2021 (T = 14,610),
windowHalfWidth=5, pctile=90, smoothPercentileWidth=31, float32 input resident in HBM, float64 out.

N > 1: configs[3] -- ONE 0.25 deg grid with ~5 % NaN (skipna=True), its 1,036,800 cells split into N
contiguous blocks, one process per GPU (STRONG scaling: the total work is fixed), every step's
(2D, block) results gathered to rank 0 over RCCL (xmhw_gather_blocks: grouped ncclSend/ncclRecv on
the kernels' output buffers, issued per slab on a second stream so that it overlaps the next
slab's kernels).  `--scaling weak` keeps a full grid per rank instead.  Launch: either by a process
launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun does), or
directly -- `python bench.py --gpus N` with no WORLD_SIZE starts the N worker processes itself,
before anything in the parent touches a GPU.

A step = one pass of the path over the rank's cells: raw climatology (ring kernel) + Feb-29 /
smoothing (finish kernel).  Timing: barrier + device sync, K steps, device sync + barrier, MAX over
ranks.  One JSON line on rank 0's stdout (see the driver contract):
  roofline      ring kernel: algorithmic bytes (T*4 + 2*D*8 per cell) / average launch time from HIP
                events recorded on the kernels' stream; `traffic` = HBM bytes per launch measured
                LIVE: before this process touches the GPU it runs itself three times under
                `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE, SQ_* in separate passes, the gfx950 x2
                correction of FETCH_SIZE applied) on one step of the same workload; null when
                rocprofv3 is unavailable (N > 1: null).  `binding`: the resource that actually limits
                the kernel -- vector-instruction issue: instructions per wave-row and the share of the
                SIMD's cycles its vector ALU is busy, from the SQ pass
  cpu_baseline  the per-cell numpy restatement of the reference (oracle/oracle_percell.py) on the
                box's PHYSICAL cores (spawned workers, pool start-up and pool-index construction
                outside the timed region), a bounded sample of the same synthetic input
  parity        cells spread over the grid against oracle_fast (contract 1e-6 relative); with N > 1
                also an N-rank == 1-rank bit-identity check on columns of every rank's block

No PyTorch: device memory, streams, events and the collective all go through the C ABI.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0   # same table: what a float4 streaming copy measures (79 % of the spec); SURVEY 8(d) asks for both

PRESETS = {
    "0.25deg": dict(cells=1440 * 720, years=(1982, 2021), nan=0.0, tstep=False, skipna=False, index=2,
                    name="0.25deg global (configs[2])"),
    "1deg": dict(cells=360 * 180, years=(1991, 2020), nan=0.0, tstep=False, skipna=False, index=1,
                 name="1deg global (configs[1])"),
    "0.25deg_nan": dict(cells=1440 * 720, years=(1982, 2021), nan=0.05, tstep=False, skipna=True, index=3,
                        name="0.25deg global, 5% NaN, skipna=True (configs[3])"),
    "0.05deg_tstep": dict(cells=810000, years=(2001, 2020), nan=0.0, tstep=True, skipna=False, index=4,
                          name="0.05deg tile share, 6-hourly no-leap tstep (configs[4], one GPU's share)"),
    # configs[2]'s shape on data that looks like a real archive: values stored at 0.01 K with 10 % of the cells
    # held at -1.8 for 120 days a year (sea ice: scattered cells, each with its own season -- the worst case for 32 cells in
    # lockstep -- and packs of neighbouring cells that freeze together); AR(1) anomalies (rho = 0.9) instead of white noise.  Never the headline.
    "0.25deg_quant_ice": dict(cells=1440 * 720, years=(1982, 2021), nan=0.0, tstep=False, skipna=False, index=2,
                              gen=dict(quant=0.01, ice_frac=0.10, rho=0.0),
                              name="0.25deg global, values at 0.01 K, 10 % of cells at -1.8 for 120 days a year (configs[2] shape)"),
    "0.25deg_quant_icepack": dict(cells=1440 * 720, years=(1982, 2021), nan=0.0, tstep=False, skipna=False, index=2,
                                  gen=dict(quant=0.01, ice_frac=0.10, rho=0.0, ice_patch=4320),
                                  name="0.25deg global, values at 0.01 K, 10 % of cells under ice 120 days a year in packs of 4,320 "
                                       "neighbouring cells that freeze within 15 days of each other (configs[2] shape)"),
    # configs[2]'s shape stored as int16 codes (OISST: scale_factor 0.01, add_offset 0, _FillValue -999) and read IN PLACE by
    # the sorted-list kernel (xmhw_clim_raw_i16): no decoded copy of the series, T * 2 bytes per cell instead of T * 4 / T * 8.
    # dtype "i16>f32" = the float32 decode xarray applies for float32 attributes, "i16>f64" = float64 attributes.
    "0.25deg_packed": dict(cells=1440 * 720, years=(1982, 2021), nan=0.0, tstep=False, skipna=False, index=2,
                           gen=dict(quant=0.01, ice_frac=0.0, rho=0.0), packed=dict(scale=0.01, offset=0.0, fill=-999),
                           name="0.25deg global stored as int16 codes (scale_factor 0.01), read in place (configs[2] shape)"),
    "0.25deg_ar1": dict(cells=1440 * 720, years=(1982, 2021), nan=0.0, tstep=False, skipna=False, index=2,
                        gen=dict(quant=0.0, ice_frac=0.0, rho=0.9),
                        name="0.25deg global, AR(1) anomalies rho = 0.9 (configs[2] shape)"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, choices=sorted(k for k in PRESETS if not PRESETS[k].get("gen")),
                    help="default: 0.25deg (configs[2]) at N=1, 0.25deg_nan (configs[3]) at N>1")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N>1: strong = one grid split over the ranks (default); weak = a full grid per rank")
    ap.add_argument("--cells", type=int, default=0, help="total cells (strong) / cells per GPU (weak, N=1); 0: the preset's")
    ap.add_argument("--slabs", type=int, default=0, help="launches per step (0: 1 at N=1, max(4, N) at N>1)")
    ap.add_argument("--single-gather", action="store_true",
                    help="N>1: ONE launch and ONE xmhw_gather_blocks per step (north_star's single RCCL gather; nothing of it is "
                         "hidden behind compute) instead of a gather per slab behind the later slabs' kernels")
    ap.add_argument("--finish-stream", type=int, default=-1,
                    help="1: the finish kernel of a slab runs on a second stream behind its ring kernel and overlaps the "
                         "next slab's ring kernel (default: on when a step has more than one slab)")
    ap.add_argument("--kernel", default="auto")
    ap.add_argument("--ring2", type=int, default=None, help="ring2 variant override (-2 auto, -1 round-1 kernel, 0..11)")
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 traffic measurement")
    ap.add_argument("--force-dist", action="store_true", help="build the RCCL communicator and gather even with one rank")
    ap.add_argument("--no-other", action="store_true", help="skip the other_configs leg (N = 1: float64 configs[2], float32 configs[1], [3], [4] share)")
    ap.add_argument("--other-cells", type=int, default=0, help="cells of each other_configs run (0: the preset's own grid)")
    ap.add_argument("--cpu-cells", type=int, default=16384)
    ap.add_argument("--parity-cells", type=int, default=4096, help="cells of the headline parity check (SURVEY 8d: 4,096)")
    ap.add_argument("--other-parity-cells", type=int, default=512, help="cells of each other_configs parity check")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_workers(args):
    """`python bench.py --gpus N` without a launcher: start N fresh worker processes (one per GPU).
    The parent never touches a GPU; rank 0 inherits stdout (the JSON line), the other ranks' stdout
    goes to stderr.  Exit code: the largest worker exit code."""
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), XMHW_BOOTSTRAP_PORT=str(_free_port() if r == 0 else 0))
        procs.append((r, env))
    boot = procs[0][1]["XMHW_BOOTSTRAP_PORT"]
    running = []
    for r, env in procs:
        env["XMHW_BOOTSTRAP_PORT"] = boot
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        running.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--worker"], env=env,
                                        stdout=None if r == 0 else sys.stderr))
    # poll all workers: the first one that fails takes the others with it (a rank that died early would leave the
    # rest waiting in the bootstrap accept or inside a collective)
    rc = 0
    alive = list(running)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0:
                rc = max(rc, abs(r))
                for o in alive:
                    o.terminate()
                for o in alive:
                    try:
                        o.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        o.kill()
                alive = []
                break
    return rc


def live_counters(argv, rows_per_wave):
    """Counters of the ring kernel measured on THIS box for THIS workload by rocprofv3 --pmc passes of one
    untimed step, run as child processes before this process touches the GPU (the program directly after `--`):
      FETCH_SIZE, WRITE_SIZE (separate passes: the TCC slots do not fit both) -> HBM bytes per launch.  FETCH_SIZE is
        reported in KB and counts a wide coalesced stream at half its bytes on gfx950 (MI355X_MICROARCH.md, HBM): x2;
      SQ_* (one pass) -> instructions per wave-row and the share of a wave's cycles its vector ALU is busy: the
        binding resource of this kernel is instruction issue, not HBM.
    Returns (traffic bytes per launch or None, its source text, dict of SQ results or None)."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not found", None
    out = {}
    sq = None
    tmp = tempfile.mkdtemp(prefix="xmhw_pmc_", dir="/tmp")
    sq_names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU",
                "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
    try:
        for tag, counters in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("SQ", sq_names)):
            d = os.path.join(tmp, tag)
            cmd = ["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__)] + argv + \
                  ["--steps", "1", "--warmup", "0", "--no-cpu", "--no-pmc", "--parity-cells", "0", "--pmc-child"]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                if tag == "SQ":
                    break           # the traffic stands without the issue counters
                return None, f"rocprofv3 --pmc {tag} failed (rc {r.returncode})", None
            tot, n, name = {}, 0, None
            for row in csv.DictReader(open(files[0])):
                if ("clim_ring" in row["Kernel_Name"] or "clim_sorted" in row["Kernel_Name"]) and row["Counter_Name"] in counters:
                    tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    n += row["Counter_Name"] == counters[0]
                    name = row["Kernel_Name"]
            if n == 0:
                if tag == "SQ":
                    break
                return None, "no ring-kernel dispatch in the counter file", None
            if tag == "SQ":
                waves = max(tot.get("SQ_WAVES", 0.0), 1.0)
                wr = waves * rows_per_wave               # wave-rows of all launches in the file
                wc = max(tot.get("SQ_WAVE_CYCLES", 0.0), 1.0)
                sq = {"kernel": name, "launches": n, "waves_per_launch": waves / n, "rows_per_wave": rows_per_wave,
                      "valu_per_wave_row": tot.get("SQ_INSTS_VALU", 0.0) / wr, "salu_per_wave_row": tot.get("SQ_INSTS_SALU", 0.0) / wr,
                      "lds_per_wave_row": tot.get("SQ_INSTS_LDS", 0.0) / wr,
                      "wave_quad_cycles_per_wave_row": wc / wr,
                      "valu_busy_of_wave_cycles": tot.get("SQ_ACTIVE_INST_VALU", 0.0) / wc,
                      "wait_any_of_wave_cycles": tot.get("SQ_WAIT_ANY", 0.0) / wc,
                      "wait_inst_any_of_wave_cycles": tot.get("SQ_WAIT_INST_ANY", 0.0) / wc}
            else:
                out[tag] = tot[tag] / n * 1024.0        # KB -> bytes, per launch
    except Exception as e:      # noqa: BLE001 -- a profiler problem must not fail the benchmark
        return None, f"{type(e).__name__}: {e}", None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"],
            "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, this run's box and workload", sq)


def _ring_name(variant):
    """the kernel a float32 plan runs on, by its layout number (include/xmhw_amd.h: XMHW_LAYOUT_*)"""
    if variant == 40:
        return "clim_sorted_f32 (sorted row-lists in LDS, 2 lanes per cell, layout 40)"
    if variant >= 20:
        return f"clim_ring3_f32 ({ {20: 8, 21: 4, 22: 2}.get(variant, 8) } lanes per cell, layout {variant})"
    return f"clim_ring2_f32 (layout {variant})"


def _ring_name_f64(layout):
    """the kernel genuinely float64 samples run on (64-bit keys as high / low words), by its layout number"""
    if layout >= 20:
        return f"clim_ring3_f32<double, 64-bit keys> ({ {20: 8, 21: 4}.get(layout, 8) } lanes per cell, layout {layout})"
    return f"clim_ring2_f32<double, 64-bit keys> (layout {layout})"


def _other_config(h, np, fast, cfg, dtype, args, DeviceBuffer, Plan, clim_raw, clim_finish, steps=3, parity_cells=512, cells=0):
    """One more BASELINE config on this GPU: kernel + finish, `steps` timed steps (HIP events around the ring
    kernel), parity of a few cells against the oracle.  Same synthetic generator, seeds and shapes as the
    headline leg."""
    from xmhw_amd.calendar import add_doy
    ps = PRESETS[cfg]
    tstep = ps["tstep"]
    if tstep:
        doy = np.tile(np.arange(1, 1461, dtype=np.int64), ps["years"][1] - ps["years"][0] + 1)
    else:
        doy = add_doy(np.arange(f"{ps['years'][0]}-01-01", f"{ps['years'][1] + 1}-01-01", dtype="datetime64[D]"))
    T, C = int(doy.shape[0]), int(cells) or int(args.other_cells) or ps["cells"]
    packed = ps.get("packed") if dtype.startswith("i16") else None
    isz = 2 if packed else 4 if dtype == "f32" else 8
    w, pctile, width = 5, 90, 31
    plan = Plan(doy, w, kernel=args.kernel, nchunks=args.chunks, ring2=args.ring2 if isz == 4 else None)
    D = plan.D
    bufs = []
    try:
        ts = DeviceBuffer(isz * T * C); bufs.append(ts)
        if packed:
            # the synthetic series, then its codes (the float32 copy is released before anything is timed)
            from xmhw_amd.device import clim_raw_packed
            f32 = DeviceBuffer(4 * T * C)
            h.synth_sst_ex(f32.ptr, T, C, C, 0, 20260101 + ps["index"], ps["nan"], ps["gen"]["quant"], ps["gen"]["ice_frac"],
                           ps["gen"]["rho"], ps["gen"].get("ice_patch", 0), 0)
            h.encode_i16(f32.ptr, T, C, C, ts.ptr, C, packed["scale"], packed["offset"], packed["fill"], 0)
            h.stream_sync(0)
            f32.free()
            dec = "float64" if dtype.endswith("f64") else "float32"
            # (float32 attributes are float32 numbers: the recipe xarray would apply)
            sc, of = (packed["scale"], packed["offset"]) if dec == "float64" else (float(np.float32(packed["scale"])), float(np.float32(packed["offset"])))
        elif ps.get("gen"):
            h.synth_sst_ex(ts.ptr, T, C, C, 0, 20260101 + ps["index"], ps["nan"], ps["gen"]["quant"], ps["gen"]["ice_frac"],
                           ps["gen"]["rho"], ps["gen"].get("ice_patch", 0), 0)
        else:
            h.synth_sst(ts.ptr, isz, T, C, C, 0, 20260101 + ps["index"], ps["nan"], 0)
        th, se = DeviceBuffer(8 * D * C), DeviceBuffer(8 * D * C)
        out = DeviceBuffer(8 * 2 * D * C)
        bufs += [th, se, out]
        # One launch per step.  (Cutting a step into slabs of cells with the memory-bound finish kernel of a slab on a
        # second stream beside the ring kernel of the next one was measured in round 4 and LOSES: the 6-hourly share
        # 93.7 -> 103.5 ms per step at 4 slabs -- the ring launches sum to 101.1 ms against 86.3 for one, four tails
        # and a neighbour that takes CU slots and HBM -- and the headline 57.3 -> 57.4 / 59.8 ms at 2 / 4 slabs:
        # profiles/r4_micro_experiments.txt.  `nslab` is kept for that experiment.)
        nslab = int(os.environ.get("XMHW_BENCH_OTHER_SLABS", "1"))
        edges = [C * i // nslab for i in range(nslab + 1)]
        fin = h.stream_create() if nslab > 1 else 0
        evs = [(h.event_create(), h.event_create()) for _ in range(nslab)]
        ring_ms = []
        main_timed = isz in (2, 4) and plan.kernel == "ring"
        if main_timed:
            h.plan_set_timing(plan.handle, 1)

        def step():
            for i in range(nslab):
                a, n = edges[i], edges[i + 1] - edges[i]
                h.event_record(evs[i][0], 0)
                if packed:
                    clim_raw_packed(plan, ts.ptr + isz * a, n, pctile / 100.0, False, th.ptr + 8 * a, se.ptr + 8 * a, scale_factor=sc,
                                    add_offset=of, fill=packed["fill"], decoded=dec, ld=C, ldo=C)
                else:
                    clim_raw(plan, ts.ptr + isz * a, isz, n, pctile / 100.0, False, th.ptr + 8 * a, se.ptr + 8 * a, ld=C, ldo=C)
                h.event_record(evs[i][1], 0)
                if nslab > 1:
                    h.stream_wait_event(fin, evs[i][1])
                clim_finish(plan, th.ptr + 8 * a, se.ptr + 8 * a, n, not tstep, True, width, out.ptr + 8 * a,
                            out.ptr + 8 * D * C + 8 * a, ldo=C, stream=fin)
            h.stream_sync(0)
            if nslab > 1:
                h.stream_sync(fin)
            if main_timed:
                return sum(h.plan_kernel_ms(plan.handle, i) for i in range(nslab))
            return sum(h.event_elapsed_ms(e0_, e1_) for e0_, e1_ in evs)
        step()
        t0 = time.perf_counter()
        for _ in range(steps):
            ring_ms.append(step())
        ms = 1e3 * (time.perf_counter() - t0) / steps
        idx = np.unique(np.linspace(0, C - 1, parity_cells).astype(np.int64))
        d_idx = DeviceBuffer.from_array(idx); bufs.append(d_idx)
        src, sisz = ts, isz
        if packed:
            # parity against the oracle on the DECODED series (xmhw_decode: what the float paths would have been handed)
            sisz = 8 if dec == "float64" else 4
            src = DeviceBuffer(sisz * T * C); bufs.append(src)
            h.decode(ts.ptr, 2, 0, T, C, C, src.ptr, sisz, C, True, sc, of, True, float(packed["fill"]), 0)
        d_s = DeviceBuffer(sisz * T * idx.size); bufs.append(d_s)
        h.gather_cells(src.ptr, sisz, T, C, d_idx.ptr, idx.size, d_s.ptr, idx.size)
        d_o = DeviceBuffer(8 * 2 * D * idx.size); bufs.append(d_o)
        h.gather_cells(out.ptr, 8, 2 * D, C, d_idx.ptr, idx.size, d_o.ptr, idx.size)
        h.stream_sync(0)
        sample = d_s.to_array((T, idx.size), np.float32 if (sisz if packed else isz) == 4 else np.float64)
        got = d_o.to_array((2 * D, idx.size), np.float64)
        _, th0, se0 = fast.threshold_cells_fast(sample, doy, pctile=pctile, windowHalfWidth=w, smoothPercentileWidth=width,
                                                tstep=tstep)
        with np.errstate(invalid="ignore", divide="ignore"):
            err = max(float(np.nanmax(np.abs(got[:D] - th0) / np.abs(th0))), float(np.nanmax(np.abs(got[D:] - se0) / np.abs(se0))))
        v2 = plan.ring2_in_use() if isz in (2, 4) and plan.kernel == "ring" else -1
        x64 = plan.f64_mode() if isz == 8 and plan.kernel == "ring" else -1
        kname = (_ring_name(v2) if v2 >= 0 else
                 _ring_name_f64(x64) if x64 >= 0 else "clim_generic")
        if packed:
            kname = kname.replace("clim_sorted_f32", "clim_sorted_i16 (int16 codes read in place)")
        bpc = T * isz + 2 * D * 8
        ring_avg = float(np.mean(ring_ms))
        return {"workload": f"{ps['name']}: {C} cells, T={T}, D={D}, nan_frac={ps['nan']}", "dtype": f"{dtype} in / f64 out",
                "ms_per_step": ms, "cells_per_s": C / (ms * 1e-3), "kernel": kname, "kernel_avg_launch_ms": ring_avg,
                "slabs": nslab, "finish": "second stream, beside the next slab's ring kernel" if nslab > 1 else "same stream",
                "algorithmic_bytes_per_cell": bpc, "roofline_frac": C * bpc / (ring_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "parity_cells": int(idx.size), "parity_max_rel_err": err, "parity_ok": bool(err < 1e-6)}
    finally:
        for b in bufs:
            b.free()
        plan.destroy()


def run(args):
    import numpy as np
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # before anything initialises HIP / RCCL
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2

    # stdout carries exactly one JSON line: keep a private handle to it and point fd 1 at stderr, so
    # that banners printed by native libraries (RCCL prints its version) cannot add lines
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    cfg = args.config or ("0.25deg" if world == 1 else "0.25deg_nan")
    ps = PRESETS[cfg]
    strong = world > 1 and args.scaling == "strong"
    tstep = ps["tstep"]
    from xmhw_amd.calendar import add_doy
    if tstep:   # 1460 steps per year, no leap days (docs/frequency.rst:42-50), add_doy tstep branch
        doy = np.tile(np.arange(1, 1461, dtype=np.int64), ps["years"][1] - ps["years"][0] + 1)
    else:
        doy = add_doy(np.arange(f"{ps['years'][0]}-01-01", f"{ps['years'][1] + 1}-01-01", dtype="datetime64[D]"))
    T = int(doy.shape[0])
    w, pctile, width = 5, 90, 31
    q = pctile / 100.0
    kw = dict(pctile=pctile, windowHalfWidth=w, smoothPercentileWidth=width, tstep=tstep, skipna=ps["skipna"])

    # ---- everything that forks / spawns happens BEFORE this process touches the GPU -------------
    pool = None
    if rank == 0 and world == 1 and not args.no_cpu:
        import parallel as opar
        pool = opar.OraclePool(doy, w)
    traffic, traffic_src, sq = None, None, None
    if rank == 0 and world == 1 and not args.no_pmc and not args.pmc_child:
        child_argv = ["--gpus", "1", "--config", cfg, "--dtype", args.dtype, "--kernel", args.kernel,
                      "--chunks", str(args.chunks), "--cells", str(args.cells)]
        if args.ring2 is not None:
            child_argv += ["--ring2", str(args.ring2)]
        # (a wave runs every row of its chunk: the rows with output plus the 2w warm-up rows; one chunk unless --chunks)
        from xmhw_amd.device import Plan as _Plan
        _p = _Plan(doy, w, kernel=args.kernel, nchunks=args.chunks, ring2=args.ring2)      # (host side only: no GPU touched)
        _nch = _p.chunks_in_use(int(args.cells) or ps["cells"])
        _rows_per_wave = float(len(np.unique(doy))) / max(_nch, 1) + 2 * w
        _waves_cu = 8
        if args.dtype == "f32" and _p.layout_in_use() == 40:
            # the sorted-list kernel: a wave runs ONE of the plan's chunks (warm-up rows + rows with output)
            from xmhw_amd._lib import hip as _hip
            _k, _lds, _pieces = _hip().plan_sorted_info(_p.handle, int(args.cells) or ps["cells"])
            _chunks = np.asarray(_hip().plan_sorted_table(_p.handle, int(_pieces))[0])
            _rows_per_wave = float(np.mean(_chunks[:, 2] - _chunks[:, 0]))
            _waves_cu = min(8, (160 * 1024) // (1280 * ((int(_lds) + 1279) // 1280)))      # (LDS is handed out in 1,280-byte pieces)
        _p.destroy()
        traffic, traffic_src, sq = live_counters(child_argv, _rows_per_wave)
        # the issue floor of the sorted-list kernel: its vector instructions by issue class (from the ISA, hipcc
        # -S on this box: no GPU) priced with the measured per-class costs, against the live counters -- tools/issue_mix.py
        if sq is not None and "clim_sorted" in str(sq.get("kernel", "")):
            try:
                r_ = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "issue_mix.py"),
                                     "--measured-valu", str(sq["valu_per_wave_row"]),
                                     "--measured-quad-cycles", str(sq["wave_quad_cycles_per_wave_row"]),
                                     "--waves-per-cu", str(_waves_cu)],
                                    stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
                if r_.returncode == 0:
                    sq["issue_mix"] = json.loads(r_.stdout.decode())
            except Exception:      # noqa: BLE001 -- the floor is a report, not the benchmark
                pass

    from xmhw_amd._lib import hip
    from xmhw_amd.device import DeviceBuffer, Plan, clim_finish, clim_raw, release_device_cache
    from xmhw_amd.sharded import init_rccl, slab_bounds
    h = hip()
    h.set_device(local)
    use_dist = world > 1 or args.force_dist
    tr = init_rccl(rank=rank, size=world, local_rank=local) if use_dist else None

    def barrier():
        h.stream_sync(0)
        if tr is not None:
            tr.allgather_i64(0)

    # ---- this rank's cells --------------------------------------------------------------------
    C_total = int(args.cells) or ps["cells"]
    if strong:
        lo, hi = slab_bounds(C_total, world)[rank]
    else:
        lo, hi = rank * C_total, (rank + 1) * C_total         # weak: a grid of its own per rank
    C = hi - lo
    plan = Plan(doy, w, kernel=args.kernel, nchunks=args.chunks, ring2=args.ring2)
    D = plan.D
    # float32: the library records HIP events on the kernels' stream right around the MAIN kernel of every call (the
    # sorted-list kernel; the recomputation of the cell-rows it flags is counted in the step, not in that launch)
    main_timed = args.dtype == "f32" and plan.kernel == "ring"
    if main_timed:
        h.plan_set_timing(plan.handle, 1)
    # N > 1: the last slab's gather is the part of the exchange nothing hides -- more slabs make it smaller (8 at N = 8:
    # 1/8 of a rank's block instead of 1/4); --single-gather: one launch, one gather, all of it exposed
    nslab = 1 if args.single_gather else (args.slabs or (max(4, world) if use_dist else 1))
    edges = [C * i // nslab for i in range(nslab + 1)]
    slabs = [(edges[i], edges[i + 1]) for i in range(nslab) if edges[i + 1] > edges[i]]

    # ---- inputs resident in HBM: synthetic SST generated on the device ------------------------
    isz = 4 if args.dtype == "f32" else 8
    seed = 20260101 + ps["index"]
    ts = DeviceBuffer(isz * T * C)
    h.synth_sst(ts.ptr, isz, T, C, C, lo, seed, ps["nan"], 0)
    raw_th = [DeviceBuffer(8 * D * (b - a)) for a, b in slabs]
    raw_se = [DeviceBuffer(8 * D * (b - a)) for a, b in slabs]
    out = [DeviceBuffer(8 * 2 * D * (b - a)) for a, b in slabs]          # dense (2D, n): thresh rows, then seas rows
    comm_stream = h.stream_create() if use_dist else 0
    two_streams = (args.finish_stream == 1 or (args.finish_stream < 0 and len(slabs) > 1)) and len(slabs) > 1
    fin_stream = h.stream_create() if two_streams else 0
    recv, cols_of_rank = [], []
    if use_dist:
        for i, (a, b) in enumerate(slabs):
            counts = tr.allgather_i64(b - a)
            cols_of_rank.append(counts)
            recv.append(DeviceBuffer(8 * 2 * D * int(counts.sum())) if rank == 0 else None)
    h.stream_sync(0)
    ev = [(h.event_create(), h.event_create(), h.event_create()) for _ in slabs]
    ring_ms, finish_ms, raw_ms = [], [], []

    def step(timed):
        for i, (a, b) in enumerate(slabs):
            n = b - a
            h.event_record(ev[i][0], 0)
            clim_raw(plan, ts.ptr + isz * a, isz, n, q, False, raw_th[i], raw_se[i], ld=C, ldo=n)
            h.event_record(ev[i][1], 0)
            if two_streams:
                # the memory-bound finish kernel of this slab overlaps the issue-bound ring kernel of the next one
                h.stream_wait_event(fin_stream, ev[i][1])
            clim_finish(plan, raw_th[i], raw_se[i], n, not tstep, True, width, out[i].ptr, out[i].ptr + 8 * D * n, ldo=n,
                        stream=fin_stream)
            h.event_record(ev[i][2], fin_stream)
            if use_dist:
                # the slab's gather rides a second stream behind its finish kernel and overlaps the next slab
                h.stream_wait_event(comm_stream, ev[i][2])
                h.gather_blocks(tr._comm, out[i].ptr, 2 * D, n, recv[i].ptr if rank == 0 else 0,
                                cols_of_rank[i] if rank == 0 else np.zeros(0, dtype=np.int64), 0, comm_stream)
        h.stream_sync(0)
        if two_streams:
            h.stream_sync(fin_stream)
        if use_dist:
            h.stream_sync(comm_stream)
        if timed:
            for i in range(len(slabs)):
                raw_ms.append(h.event_elapsed_ms(ev[i][0], ev[i][1]))
                ring_ms.append(h.plan_kernel_ms(plan.handle, len(slabs) - 1 - i) if main_timed else raw_ms[-1])
                finish_ms.append(h.event_elapsed_ms(ev[i][1], ev[i][2]))

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    dt = time.perf_counter() - t0
    if tr is not None:
        dt = float(tr.allgather_i64(int(dt * 1e9)).max()) * 1e-9

    ms_per_step = 1e3 * dt / args.steps
    cells_per_step = C_total if (strong or world == 1) else world * C_total
    value = cells_per_step * args.steps / dt
    # ---- roofline of the dominant kernel (ring): algorithmic bytes / launch time ----------------
    bytes_per_cell = T * isz + 2 * D * 8
    cells_per_launch = float(np.mean([b - a for a, b in slabs]))
    ring_avg_ms = float(np.mean(ring_ms))
    achieved = cells_per_launch * bytes_per_cell / (ring_avg_ms * 1e-3) / 1e9
    v2 = plan.ring2_in_use() if isz == 4 and plan.kernel == "ring" else -1
    x64 = plan.f64_mode() if isz == 8 and plan.kernel == "ring" else -1
    kname = (_ring_name(v2) if v2 >= 0 else
             _ring_name_f64(x64) if x64 >= 0 else
             "clim_generic" if isz == 8 else
             ("clim_ring_" + args.dtype if plan.kernel == "ring" else "clim_generic"))
    waves_cu = 8
    if v2 == 40:
        # (LDS is handed out in 512-byte pieces; the kernel's registers -- launch bound (64, 2) -- allow 8 waves per CU)
        waves_cu = min(8, (160 * 1024) // (512 * ((int(h.plan_sorted_info(plan.handle, C)[1]) + 511) // 512)))

    result = {
        "metric": "grid-cells/sec for threshold() on 40yr daily SST",
        "value": value,
        "unit": "cells/s",
        "n_gpus": world,
        "ranks": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": f"{args.dtype} in / f64 out",
        "data": "synthetic",
        "config": {
            "workload": f"{ps['name']}: {cells_per_step} cells per step"
                        + (f" split over {world} ranks ({C} on rank 0)" if strong else (f", {C} cells/GPU" if world > 1 else ""))
                        + f", {ps['years'][0]}-{ps['years'][1]} (T={T}), windowHalfWidth={w}, pctile={pctile}, "
                          f"smoothPercentileWidth={width}, nan_frac={ps['nan']}, skipna={ps['skipna']}",
            "cells_per_step": cells_per_step, "cells_rank0": C, "T": T, "D": D, "kernel": kname, "slabs": len(slabs),
            "gather": ("none" if not use_dist else "ONE xmhw_gather_blocks (RCCL send/recv) to rank 0 per step (--single-gather)" if args.single_gather
                       else "xmhw_gather_blocks (RCCL send/recv) to rank 0, per slab on a second stream"),
        },
        "roofline": {
            "bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_rate": achieved / HBM_COPY_GBS,
            "traffic": (traffic * cells_per_launch / C) if traffic else None, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": cells_per_launch * bytes_per_cell,
            "algorithmic_bytes_per_cell": bytes_per_cell, "cells_per_launch": cells_per_launch,
            "avg_launch_ms": ring_avg_ms,
            # the whole raw-climatology call (round 6: the same thing -- flagged cell-rows are recomputed inside the kernel)
            "raw_call_avg_ms": float(np.mean(raw_ms)),
            # what actually binds this kernel: vector-instruction issue.  The waves of a SIMD share one issue port; the
            # share of the SIMD's cycles its vector ALU is busy is (waves per SIMD) x the per-wave share measured by
            # SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES -- 1.0 would be the kernel's ceiling at this instruction count
            "binding": None if sq is None else {
                "resource": ("vector-instruction issue (VALU); LDS holds %d waves per CU (%.2f per SIMD)" % (waves_cu, waves_cu / 4.0)) if v2 == 40
                            else "vector-instruction issue (VALU), 2 waves per SIMD",
                "insts_per_wave_row": {"valu": sq["valu_per_wave_row"], "salu": sq["salu_per_wave_row"],
                                       "lds": sq["lds_per_wave_row"]},
                "cells_per_wave": 16 if v2 in (21, 31) else 32 if v2 in (22, 32, 40) else 8,
                "wave_quad_cycles_per_wave_row": sq["wave_quad_cycles_per_wave_row"],
                "valu_busy_of_wave_cycles": sq["valu_busy_of_wave_cycles"],
                "wait_any_of_wave_cycles": sq["wait_any_of_wave_cycles"],
                "frac_of_issue_peak": min((waves_cu / 4.0 if v2 == 40 else 2.0) * sq["valu_busy_of_wave_cycles"], 1.0),
                # priced per issue class (tools/issue_mix.py): SIMD cycles a wave-row needs at least, and how close the kernel is
                "issue_floor_cycles_per_wave_row": None if "issue_mix" not in sq else sq["issue_mix"]["issue_floor_cycles_per_wave_row"],
                "issue_floor_inputs": None if "issue_mix" not in sq else {k: sq["issue_mix"].get(k) for k in ("cost_cycles", "select_round_equivalents_per_wave_row", "select_round_equivalents_source", "valu_of_blocks_that_run_once_per_row", "valu_of_a_select_round")},
                "class_B_share_of_valu": None if "issue_mix" not in sq else sq["issue_mix"]["class_B_share"],
                "frac_of_issue_floor": None if "issue_mix" not in sq else sq["issue_mix"].get("frac_of_issue_floor_at_this_occupancy"),
                "frac_of_issue_floor_at_two_waves_per_simd": None if "issue_mix" not in sq else sq["issue_mix"].get("frac_of_issue_floor_at_two_waves_per_simd"),
                "kernel": sq["kernel"], "source": "rocprofv3 --pmc SQ_* on one step of this run's box and workload "
                                                   "(the product kernel, no counter twin)"},
        },
        "finish_kernel_avg_launch_ms": float(np.mean(finish_ms)),
    }
    if use_dist:
        # kernel-only and kernel+gather side by side (rank 0's kernels; the step time is the max over ranks)
        kernels_ms = (float(np.sum(ring_ms)) + float(np.sum(finish_ms))) / args.steps
        result["multi_gpu"] = {"kernels_ms_per_step_rank0": kernels_ms, "step_ms": ms_per_step,
                               "exposed_gather_ms_per_step": max(ms_per_step - kernels_ms, 0.0),
                               "kernel_only_cells_per_s": cells_per_step / (kernels_ms * 1e-3),
                               "gathered_bytes_per_step_at_root": float(2 * D * 8 * (cells_per_step - C))}

    # ---- N > 1: the SAME workload on rank 0 alone (the N = 1 default line measures configs[2], a clean grid;
    # the sharded run measures configs[3]: a speed-up must divide like by like) --------------------------------
    if rank == 0 and world > 1 and strong and not args.pmc_child:
        try:
            f_ts = DeviceBuffer(isz * T * C_total)
            h.synth_sst(f_ts.ptr, isz, T, C_total, C_total, 0, seed, ps["nan"], 0)
            f_th, f_se = DeviceBuffer(8 * D * C_total), DeviceBuffer(8 * D * C_total)
            f_out = DeviceBuffer(8 * 2 * D * C_total)

            def one_step():
                clim_raw(plan, f_ts, isz, C_total, q, False, f_th, f_se)
                clim_finish(plan, f_th, f_se, C_total, not tstep, True, width, f_out.ptr, f_out.ptr + 8 * D * C_total)
                h.stream_sync(0)
            one_step()
            t1 = time.perf_counter()
            for _ in range(2):
                one_step()
            single_ms = 1e3 * (time.perf_counter() - t1) / 2
            result["multi_gpu"]["single_rank_ms_same_workload"] = single_ms
            # the step (kernels + whatever of the gather is not hidden behind them) and the kernels alone
            result["multi_gpu"]["speedup_vs_single_rank"] = single_ms / ms_per_step
            result["multi_gpu"]["speedup_step_vs_single_rank"] = single_ms / ms_per_step
            result["multi_gpu"]["speedup_kernel_only_vs_single_rank"] = single_ms / result["multi_gpu"]["kernels_ms_per_step_rank0"]
            for b_ in (f_ts, f_th, f_se, f_out):
                b_.free()
        except Exception as e:      # noqa: BLE001 -- e.g. not enough HBM next to the shard: reported, not fatal
            result["multi_gpu"]["single_rank_ms_same_workload"] = None
            result["multi_gpu"]["single_rank_error"] = f"{type(e).__name__}: {e}"

    # ---- parity (rank 0) ----------------------------------------------------------------------
    def gather_cols(buf, itemsize, rows, ld, idx, dtype, base=0):
        d_idx = DeviceBuffer.from_array(idx.astype(np.int64))
        d_out = DeviceBuffer(itemsize * rows * idx.size)
        try:
            h.gather_cells(buf.ptr + base, itemsize, rows, ld, d_idx.ptr, idx.size, d_out.ptr, idx.size)
            h.stream_sync(0)
            return d_out.to_array((rows, idx.size), dtype)
        finally:
            d_idx.free()
            d_out.free()

    def rank0_columns(idx):
        """columns idx (indices into rank 0's block) of the step's output: (thresh, seas)"""
        th = np.empty((D, idx.size))
        se = np.empty((D, idx.size))
        for i, (a, b) in enumerate(slabs):
            sel = np.nonzero((idx >= a) & (idx < b))[0]
            if sel.size:
                both = gather_cols(out[i], 8, 2 * D, b - a, idx[sel] - a, np.float64)
                th[:, sel], se[:, sel] = both[:D], both[D:]
        return th, se

    if rank == 0 and args.parity_cells > 0:
        import oracle_fast as fast
        idx = np.unique(np.linspace(0, C - 1, min(args.parity_cells, C)).astype(np.int64))
        sample = gather_cols(ts, isz, T, C, idx, np.float32 if isz == 4 else np.float64)
        got_th, got_se = rank0_columns(idx)
        _, th0, se0 = fast.threshold_cells_fast(sample, doy, pctile=pctile, windowHalfWidth=w,
                                                smoothPercentileWidth=width, tstep=tstep)
        err_th = float(np.nanmax(np.abs(got_th - th0) / np.abs(th0)))
        err_se = float(np.nanmax(np.abs(got_se - se0) / np.abs(se0)))
        result["parity"] = {"cells": int(idx.size), "max_rel_err_thresh": err_th, "max_rel_err_seas": err_se,
                            "tolerance": 1e-6, "ok": bool(err_th < 1e-6 and err_se < 1e-6)}
    if rank == 0 and use_dist and args.parity_cells > 0:
        # N-rank == 1-rank: for EVERY rank two probes of k columns each -- the first columns of its first slab and
        # the last columns of its last slab, i.e. always inside one slab -- recomputed on rank 0 alone (the
        # synthetic input is a function of the global cell index) and compared bit for bit with what arrived in
        # the gathered buffers.  The number of columns actually compared is reported and must not be 0.
        k = 32
        bounds = slab_bounds(C_total, world) if strong else [(r * C_total, (r + 1) * C_total) for r in range(world)]
        L = len(slabs) - 1
        probes = []          # (rank, slab, first column inside the slab, columns, global cell index of the first)
        for r, (a, b) in enumerate(bounds):
            w0, wL = int(cols_of_rank[0][r]), int(cols_of_rank[L][r])
            if w0 > 0:
                probes.append((r, 0, 0, min(k, w0), a))
            if wL > 0:
                kk = min(k, wL)
                probes.append((r, L, wL - kk, kk, b - kk))
        ncol = sum(p[3] for p in probes)
        compared, same = 0, True
        if ncol:
            mini = DeviceBuffer(isz * T * ncol)
            c = 0
            for (_, _, _, kk, g0) in probes:
                h.synth_sst(mini.ptr + isz * c, isz, T, kk, ncol, g0, seed, ps["nan"], 0)
                c += kk
            m_th, m_se = DeviceBuffer(8 * D * ncol), DeviceBuffer(8 * D * ncol)
            m_out = DeviceBuffer(8 * 2 * D * ncol)
            clim_raw(plan, mini, isz, ncol, q, False, m_th, m_se)
            clim_finish(plan, m_th, m_se, ncol, not tstep, True, width, m_out.ptr, m_out.ptr + 8 * D * ncol)
            h.stream_sync(0)
            one = m_out.to_array((2 * D, ncol), np.float64)
            c = 0
            for (r, i, c0, kk, _) in probes:
                cr = cols_of_rank[i]
                off = 2 * D * int(cr[:r].sum()) * 8
                got = gather_cols(recv[i], 8, 2 * D, int(cr[r]), np.arange(c0, c0 + kk), np.float64, base=off)
                same = same and bool(np.array_equal(got, one[:, c:c + kk], equal_nan=True))
                compared += kk
                c += kk
            for b_ in (mini, m_th, m_se, m_out):
                b_.free()
        result["multi_gpu"]["compared_columns"] = int(compared)
        result["multi_gpu"]["n_rank_equals_1_rank_bitwise"] = bool(same and compared > 0)

    # ---- CPU baseline (rank 0, N = 1) ----------------------------------------------------------
    if pool is not None:
        ncpu = max(min(args.cpu_cells, C), 1)
        ncpu = min(C, max(ncpu, 16 * pool.workers))          # >= 16 cells per worker (>= 4,096 in all)
        cidx = np.arange(ncpu, dtype=np.int64)
        cs = gather_cols(ts, isz, T, C, cidx, np.float32 if isz == 4 else np.float64)
        r = pool.time_percell(cs, per_worker_min=16, **kw)
        g_th, _ = rank0_columns(cidx)
        result["cpu_baseline"] = {
            "value": r["cells_per_s"], "unit": "cells/s", "cores": r["processes"], "kind": "port",
            "cells_per_s_per_core": r["cells_per_s_per_core"], "cells_per_cpu_second": r["cells_per_cpu_s"],
            "cpu_seconds": r["cpu_s"], "summed_worker_wall_s": r["busy_s"], "cgroup_cpu_quota": r["cpu_quota"],
            "physical_cores": pool.physical, "logical_cpus": pool.logical,
            "sample": f"first {ncpu} cells of the same synthetic input ({ncpu // r['processes']} per process), per-cell numpy "
                      f"restatement (366 x np.quantile + mean per cell, as xmhw/xmhw.py:184-197 does one calc_clim per "
                      f"cell), one spawned process per usable core (physical cores, capped by the cgroup CPU quota); pool start-up and the pool index outside the timed "
                      f"region; excludes xarray/dask per-cell overhead, so it flatters the reference",
            "wall_s": r["wall_s"],
            "max_rel_diff_vs_gpu": float(np.nanmax(np.abs(g_th - r["thresh"]) / np.abs(r["thresh"]))),
        }
        pool.close()
    # ---- the other BASELINE configs and float64 (N = 1): 3 steps each, same contract, a small parity sample ---
    if rank == 0 and world == 1 and not args.no_other and not args.pmc_child and not args.force_dist:
        for b_ in [ts] + raw_th + raw_se + out:
            b_.free()
        release_device_cache()
        import oracle_fast as fast
        others = []
        for ocfg, odt in (("0.25deg", "f64"), ("1deg", "f32"), ("1deg", "f64"), ("0.25deg_nan", "f32"),
                          ("0.05deg_tstep", "f32"), ("0.05deg_tstep", "f64"), ("0.25deg_quant_ice", "f32"),
                          ("0.25deg_quant_icepack", "f32"), ("0.25deg_ar1", "f32"),
                          ("0.25deg_packed", "i16>f32"), ("0.25deg_packed", "i16>f64")):
            if ocfg == cfg and odt == args.dtype:
                continue
            try:
                others.append(_other_config(h, np, fast, ocfg, odt, args, DeviceBuffer, Plan, clim_raw, clim_finish,
                                            parity_cells=args.other_parity_cells))
            except Exception as e:      # noqa: BLE001 -- reported in the line, the headline number stands
                err = f"{type(e).__name__}: {e}"
                release_device_cache()
                try:        # (the float64 share of configs[4] is 227 GB of buffers: half the share if that did not fit)
                    half = PRESETS[ocfg]["cells"] // 2
                    o = _other_config(h, np, fast, ocfg, odt, args, DeviceBuffer, Plan, clim_raw, clim_finish,
                                      parity_cells=args.other_parity_cells, cells=half)
                    o["note"] = f"the full share failed ({err}); half of it measured"
                    others.append(o)
                except Exception as e2:      # noqa: BLE001
                    others.append({"workload": PRESETS[ocfg]["name"], "dtype": odt, "error": err, "error_half": f"{type(e2).__name__}: {e2}"})
            release_device_cache()
        result["other_configs"] = others
    if rank == 0:
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()
    barrier()
    if tr is not None:
        tr.close()
    release_device_cache()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.worker:
        return launch_workers(args)
    return run(args)


if __name__ == "__main__":
    sys.exit(main())
