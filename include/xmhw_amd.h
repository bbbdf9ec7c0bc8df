/*
 * xmhw_amd.h -- C ABI of the MI355X (gfx950) implementation of xmhw's
 * threshold() hot path.
 *
 * The reference (coecms/xmhw v0.9.3) is pure Python and has NO native
 * interface; this ABI is what a binding for the path would call.  Each entry
 * point names the reference code it replaces (paths relative to the reference
 * repository root).  Plain pointers and sizes only; no exceptions cross the
 * ABI: every function returns XMHW_OK or an error code and
 * xmhw_last_error() holds the message for the calling thread.
 *
 * Layout contract (the reference's land_check() output, identify.py:482-529):
 *   ts      [T][ld]  time-major, cell-minor ("cell" = stacked lat x lon);
 *                    element (t, c) at ts[t*ld + c], c < C <= ld
 *   doy     [T]      int32 label of every time step (add_doy(),
 *                    identify.py:28-79); HOST memory
 *   thresh  [D][ldo] float64, row i <-> i-th smallest distinct doy label
 *   seas    [D][ldo] float64
 * "dev" pointers are device (HBM) addresses owned by the caller.
 */
#ifndef XMHW_AMD_H
#define XMHW_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XMHW_OK 0
#define XMHW_ERR_INVALID 1     /* bad argument (the reference raises XmhwException) */
#define XMHW_ERR_HIP 2         /* HIP runtime error / no device                     */
#define XMHW_ERR_UNSUPPORTED 3 /* configuration outside every kernel's limits       */
#define XMHW_ERR_NOMEM 4
#define XMHW_ERR_COMM 5        /* RCCL could not be loaded / a collective failed     */

/* kernel selector for xmhw_plan_set_kernel() / reported by xmhw_plan_info() */
#define XMHW_KERNEL_AUTO 0
#define XMHW_KERNEL_RING 1     /* register-ring sliding-window kernel (fast path)   */
#define XMHW_KERNEL_GENERIC 2  /* one wave per (cell, doy), radix descent (any plan)*/

/* ---- library / device ------------------------------------------------- */
int xmhw_version(void);                 /* 1000*major + minor                     */
const char *xmhw_arch(void);            /* "gfx950"                               */
const char *xmhw_last_error(void);      /* message of the last failure (thread)   */
int xmhw_device_count(int *count);
int xmhw_set_device(int device);
int xmhw_get_device(int *device);
int xmhw_device_info(int device, char *name, int name_len, int *compute_units,
                     uint64_t *hbm_bytes);

/* ---- caller-owned device memory, streams, events ---------------------- */
int xmhw_malloc(void **dev_ptr, size_t bytes);
int xmhw_free(void *dev_ptr);
int xmhw_memcpy_h2d(void *dev_dst, const void *host_src, size_t bytes, void *stream);
int xmhw_memcpy_d2h(void *host_dst, const void *dev_src, size_t bytes, void *stream);
/* a block of columns of a row-major host array into a dense device array (hipMemcpy2D):
 * `height` rows of `width_bytes`, rows `spitch` bytes apart on the host, `dpitch` on the device */
int xmhw_memcpy2d_h2d(void *dev_dst, size_t dpitch, const void *host_src, size_t spitch,
                      size_t width_bytes, size_t height, void *stream);
/* the inverse for results: a dense device array into a block of columns of a row-major host array */
int xmhw_memcpy2d_d2h(void *host_dst, size_t dpitch, const void *dev_src, size_t spitch,
                      size_t width_bytes, size_t height, void *stream);
/* the same without the wait (one per rank of a gathered buffer, then ONE xmhw_stream_sync)   */
int xmhw_memcpy2d_d2h_async(void *dst, size_t dpitch, const void *src_dev, size_t spitch, size_t width,
                            size_t height, void *stream);
int xmhw_memset(void *dev_dst, int value, size_t bytes, void *stream);
/* ---- ingest: file bytes -> samples on the device ------------------------ *
 * The reference leaves reading and CF decoding to xarray (docs/gettingstarted.rst:30-33).  Here the
 * RAW bytes of a column slab cross PCIe (pinned staging: xmhw_host_alloc, asynchronous pitched
 * copy) and are decoded in HBM: byte order (netCDF classic is big-endian), CF packing
 * `raw * scale_factor + add_offset` in the arithmetic of the decoded type (float32 result: two
 * float32 roundings, as xarray computes it for float32 attributes), `_FillValue` -> NaN.
 * raw_itemsize 2 = int16, 4 = float32, 8 = float64; pairs: int16->float32/float64,
 * float32->float32, float64->float64.                                                          */
int xmhw_host_alloc(void **host_ptr, size_t bytes);        /* page-locked host memory */
int xmhw_host_free(void *host_ptr);
int xmhw_memcpy2d_h2d_async(void *dev_dst, size_t dpitch, const void *host_src, size_t spitch,
                            size_t width_bytes, size_t height, void *stream);
int xmhw_memcpy_h2d_async(void *dev_dst, const void *host_src, size_t bytes, void *stream);
int xmhw_memcpy_d2h_async(void *host_dst, const void *dev_src, size_t bytes, void *stream);
int xmhw_event_sync(void *event);                           /* block the host until the event has happened */
int xmhw_decode(const void *raw_dev, int raw_itemsize, int big_endian, int64_t rows, int64_t cols,
                int64_t ld_raw, void *out_dev, int out_itemsize, int64_t ld_out, int has_scale,
                double scale_factor, double add_offset, int has_fill, double fill_value, void *stream);
/* the inverse for float32 -> int16 codes (what writing a packed archive does: xarray's CF encoding): code =
 * rint((x - add_offset) / scale_factor) computed in float64 and clamped to [-32767, 32767]; NaN -> fill_code.  bench.py
 * and the tests make packed input for xmhw_clim_raw_i16 with it.                                              */
int xmhw_encode_i16(const float *in_dev, int64_t rows, int64_t cols, int64_t ld_in, int16_t *out_dev, int64_t ld_out,
                    double scale_factor, double add_offset, int32_t fill_code, void *stream);
/* File bytes -> a (page-locked) host buffer without mapping the file: `rows` strips of row_bytes bytes,
 * row_pitch apart in the file starting at file_offset, are read with pread() into a dense buffer.
 * Thread-safe; the ingest path calls it from many threads at once (copies out of an mmap() of the same
 * file queue up on the process's page-fault path instead).  Replaces the read half of
 * xr.open_dataset(...) (docs/gettingstarted.rst:30-33) for netCDF classic files.                   */
int xmhw_read_rows(int fd, int64_t file_offset, int64_t row_pitch, int64_t row_bytes, int64_t rows,
                   void *dst_host);
/* maxPadLength: `ts.interpolate_na(dim=tdim, max_gap=maxPadLength)` (xmhw/xmhw.py:159-160, :409-410;
 * xarray's linear interpolate_na with use_coordinate=True) on the device copy of a compacted series
 * (T, C), leading dimension ld, IN PLACE.  x_dev[T] = the numeric time coordinate (float64; for
 * datetime axes xarray uses nanoseconds since the first step).  A run of NaN strictly between valid
 * samples at steps a < b is filled iff x[b] - x[a] <= max_gap, with numpy.interp's float64 arithmetic
 * rounded to the sample type; leading / trailing runs and all-NaN cells are left alone.           */
int xmhw_pad_gaps(void *ts_dev, int itemsize, int64_t T, int64_t C, int64_t ld, const double *x_dev,
                  double max_gap, void *stream);
int xmhw_stream_create(void **stream);
int xmhw_stream_destroy(void *stream);
int xmhw_stream_sync(void *stream);     /* NULL = default stream                  */
int xmhw_event_create(void **event);
int xmhw_stream_wait_event(void *stream, void *event);   /* later work on `stream` waits for `event` */
int xmhw_event_destroy(void *event);
int xmhw_event_record(void *event, void *stream);
int xmhw_event_elapsed_ms(void *start, void *stop, float *ms); /* syncs on stop  */

/* ---- plan: everything derived from the doy labels ---------------------- *
 * Replaces the per-cell window_roll() bookkeeping (identify.py:184-209: which
 * samples pool under which doy) and groupby("doy") (identify.py:233,263).
 * Built on the host from doy[T]; device tables are uploaded on first use.   */
typedef struct xmhw_plan xmhw_plan;

int xmhw_plan_create(const int32_t *doy_host, int64_t T, int32_t window_half_width,
                     xmhw_plan **plan);
int xmhw_plan_destroy(xmhw_plan *plan);
/* D = number of distinct doy labels, ntracks = runs of increasing doy ("years"),
 * kernel = XMHW_KERNEL_* that AUTO resolves to for float32 input             */
int xmhw_plan_info(const xmhw_plan *plan, int32_t *D, int32_t *ntracks,
                   int32_t *kernel, int32_t *nsteps, int32_t *step_min);
int xmhw_plan_doys(const xmhw_plan *plan, int32_t *doys_out /* [D] */);
int xmhw_plan_set_kernel(xmhw_plan *plan, int32_t kernel);  /* tests / fallback */
/* float64 input whose samples are all float32-representable (decoded int16 / float32 archives) is
 * run through the float32 ring kernel (same pools, same float64 interpolation and sums of the same
 * values; 2.7x the float64 kernel's rate).  The check happens on the device inside
 * xmhw_clim_raw_f64 (sparse probe, then on every sample the kernel loads; the float64 kernel is
 * queued behind and runs only if a sample failed), so the call stays asynchronous.  Enabled by
 * default; xmhw_plan_narrowed() reports (synchronously) whether the last float64 call of this
 * plan stayed on the float32 kernel.                                                         */
int xmhw_plan_set_narrowing(xmhw_plan *plan, int32_t enable);
/* measurement (bench.py): with timing on, every xmhw_clim_raw_f32 / xmhw_clim_raw_i16 call records a HIP event on its
 * stream right before and right after its MAIN kernel (the sorted-list kernel -- which since round 6 also recomputes the
 * cell-rows its select cannot settle -- or the ring kernel); xmhw_plan_kernel_ms returns the elapsed time of the call
 * `calls_back` calls ago (0 = the last one, up to 15), waiting for it to finish.
 * A plan is used from ONE stream at a time: the timing events and the lazily uploaded tables belong to the plan, not to
 * the call.  Two calls on the same plan may be in flight only if they were issued on the same stream.               */
int xmhw_plan_set_timing(xmhw_plan *plan, int32_t enable);
int xmhw_plan_kernel_ms(xmhw_plan *plan, int32_t calls_back, float *ms);
int xmhw_plan_narrowed(xmhw_plan *plan, int32_t *narrowed_out);
int xmhw_plan_set_chunks(xmhw_plan *plan, int32_t nchunks); /* 0 = auto         */
/* how many chunks of the doy axis a launch over C cells is cut into (the automatic choice or the forced one):
 * every workgroup walks D / nchunks rows with output + 2w warm-up rows (csrc/capi.cpp: auto_chunks)     */
int xmhw_plan_chunks_in_use(const xmhw_plan *plan, int64_t C, int32_t *nchunks);
/* host copy of the ring kernel's step table for inspection:
 * table[nsteps][ntracks_padded] (see csrc/plan.h for the encoding)            */
int xmhw_plan_table(const xmhw_plan *plan, int32_t years_per_lane, uint32_t *table_out,
                    int32_t *ntracks_padded);
/* the sorted-list kernel on this plan: keys a cell keeps of every row-list (for 37..40 tracks 16: 14 ranks in LDS, two in
 * registers), LDS bytes of a wave (32 cells; handed out in 1,280-byte pieces: 20,480 = 8 waves per CU), and the `pieces`
 * a launch over C cells asks xmhw_plan_sorted_table for (the automatic choice or xmhw_plan_set_chunks)            */
int xmhw_plan_sorted_info(const xmhw_plan *plan, int64_t C, int32_t *keys_per_list, int32_t *lds_bytes_per_wave,
                          int32_t *pieces);
/* host copy of the sorted-list kernel's chunks and step table (XMHW_LAYOUT_SORTED; csrc/plan.h: sorted_plan) for
 * inspection: the row axis cut wherever the set of pooled tracks changes (doy 60, the ends of partial years), every
 * chunk with its own table rows -- what window_roll() + groupby("doy") pool (xmhw/identify.py:184-209, :233) restated
 * per chunk.  `pieces` = how many pieces the whole row axis is cut into at least (1: only the cuts the calendar asks
 * for).  Call with NULL outputs for the sizes: nchunks, nrows (table / flag rows), ntp (entries per row, track k at
 * index k); then chunks_out[nchunks][4] = {warm_start, begin, end, trow0}, table_out[nrows][ntp], flags_out[nrows]
 * (in calendar order; a launch runs them longest first).
 * XMHW_ERR_UNSUPPORTED if the kernel is not instantiated for this plan.                                          */
int xmhw_plan_sorted_table(const xmhw_plan *plan, int32_t pieces, int32_t *nchunks, int32_t *nrows, int32_t *ntp,
                           int32_t *chunks_out, uint32_t *table_out, uint32_t *flags_out);

/* debug: ring-kernel pass counters {rows, 32-bit count passes, extractions, cold starts,
 * fast steps, 8-bit probes, code-ring rebuilds} per wave, then count passes summed over CELLS (what
 * each cell needed on its own); all summed since the last read; the third-generation kernel
 * (layouts 20..22) reports its band-path counters in slots 5..7 (csrc/kernels_ring3.hip) and
 * shader-clock ticks per section of its row loop in slots 8..15.
 * The counters live in counter twins of the ring kernels that the PRODUCT build does not contain
 * (make STATS=1 builds them: tools/); xmhw_debug_stats_available() tells, and without them every
 * counter reads 0.
 * enable != 0 allocates the counters.  xmhw_plan_debug_stats_n: `out` receives min(n, 16) values.
 * xmhw_plan_debug_stats (the round-1 entry point): out8 receives the first 8 values.          */
int xmhw_debug_stats_available(void);
int xmhw_plan_debug_stats_n(xmhw_plan *plan, int enable, uint64_t *out, int32_t n);
int xmhw_plan_debug_stats(xmhw_plan *plan, int enable, uint64_t *out8);

/* ---- which ring kernel and lane layout float32 input runs on ------------- *
 * Every layout returns bit-identical thresh; the choice is about speed only.
 * XMHW_LAYOUT_AUTO (the default): the third-generation kernel (xmhw_amd/csrc/kernels_ring3.hip:
 * per-cell histogram in LDS, band compaction, sort across the lanes of a cell) on 2 lanes per cell
 * for 9..24 tracks, on 4 lanes for 25..48, on 8 lanes for 49..88; the second-generation kernel
 * (kernels_ring2.hip) on 16 lanes per cell for 89..96 tracks; the round-1 kernel otherwise.
 * A layout may be forced wherever it is instantiated (RING3_8LANE: 9..88 tracks, RING3_4LANE: 9..48,
 * RING3_2LANE: 9..24, RING2_8LANE / RING2_4LANE: 9..48, RING2_16LANE: 49..96); a plan it is not
 * instantiated for falls back to the round-1 kernel (xmhw_plan_layout_in_use tells).
 * The environment variable XMHW_RING2 sets the default of new plans (the same numbers).        */
enum {
    XMHW_LAYOUT_AUTO = -2,
    XMHW_LAYOUT_RING1 = -1,          /* round-1 kernel (csrc/kernels_ring.hip): other windows, > 96 or < 9 tracks;
                                      * forced on 9..96 tracks at w = 5 it runs on its 32-lane entries, padded   */
    XMHW_LAYOUT_RING2_8LANE = 8,     /* second generation, lists merged into the cell's 8 nearest keys */
    XMHW_LAYOUT_RING2_4LANE = 10,    /* second generation, 4 lanes per cell, 7 merged keys */
    XMHW_LAYOUT_RING2_16LANE = 12,   /* second generation, 16 lanes per cell: 49..96 tracks */
    XMHW_LAYOUT_RING3_8LANE = 20,    /* third generation, 8 cells per wave */
    XMHW_LAYOUT_RING3_4LANE = 21,    /* third generation, 16 cells per wave: the headline layout (40 tracks) */
    XMHW_LAYOUT_RING3_2LANE = 22,    /* third generation, 32 cells per wave */
    XMHW_LAYOUT_SORTED = 40          /* rounds 5-6 (csrc/kernels_sorted.hip): sorted row-lists in LDS + a parallel
                                        merge-select, 32 cells per wave, every row of the plan (chunks with their own
                                        table rows: plan.h); a cell-row whose lists are too short is recomputed inside
                                        the kernel.  float32 (or int16 codes), w = 5, 9..48 tracks.  NOTE: 40 means
                                        "sorted for quantiles >= 0.85 or <= 0.15": a call with a quantile in between runs the same plan
                                        on its ring layout (xmhw_plan_layout_in_use cannot know the call's quantile).
                                        Needs a device whose LDS reads outside the allocation return 0 (gfx950; probed
                                        once per device, otherwise the ring layouts serve the plan) */
};
int xmhw_plan_set_layout(xmhw_plan *plan, int32_t layout);
/* the layout float32 input of this plan will run on (XMHW_LAYOUT_RING1 if the round-1 / generic kernel) */
int xmhw_plan_layout_in_use(const xmhw_plan *plan, int32_t *layout);
/* whether the current device may run the sorted-list kernel: its rank-major lists rely on LDS reads outside a workgroup's
 * allocation returning 0 (gfx950 does: tools/ubench_ldsoob.hip).  Probed on the device the first time it is asked for
 * (seven allocation sizes x 2,048 workgroups, a few microseconds) and remembered per device; *holds = 1 / 0.  Plans on a
 * device where it does not hold run on their ring layout whatever xmhw_plan_layout_in_use says (which needs no device). */
int xmhw_sorted_device_ok(int32_t *holds);
/* DEPRECATED names of the two entries above (rounds 2 and 3, when the numbers meant variants of the
 * second-generation kernel); also accepted: 1..7, 9, 11 = measured-and-rejected alternatives of round 2,
 * built with -DXMHW_RING2_EXPERIMENTS only; 30..32 = the round-4 key-store experiment
 * (csrc/kernels_ring4.hip, built with `make RING4=1` only; profiles/r4_store_experiment.txt)          */
int xmhw_plan_set_ring2(xmhw_plan *plan, int32_t variant);
int xmhw_plan_ring2_in_use(const xmhw_plan *plan, int32_t *variant);
/* genuinely float64 samples (those that do not narrow to float32): the layout of the 64-bit mode
 * (64-bit keys as a high word -- what the selection runs on -- and a low word) this plan will run on:
 * 21 / 20 = the third-generation kernel on 4 lanes per cell (13..20 tracks) / 8 lanes per cell (9..12 and
 * 21..48 tracks; XMHW_RING3_F64=0 turns both off, XMHW_RING3_F64_LANES=8 the 4-lane layout),
 * 8 = the second-generation kernel on 8 lanes per cell (only with XMHW_RING3_F64=0), 12 = on 16 lanes per cell
 * (other records up to 96 tracks), or -1 (generic kernel).  w = 5.                              */
int xmhw_plan_f64_mode(const xmhw_plan *plan, int32_t *variant);

/* ---- the hot path ------------------------------------------------------ *
 * xmhw_clim_raw_*: for every cell, the pooled linear-interpolated quantile
 * and the pooled mean per doy -- calculate_thresh()/calculate_seas() WITHOUT
 * the Feb-29 step (identify.py:233-235, :263) over window_roll()'s pools.
 * NaN samples are dropped from the pools (identify.py:208); an empty pool
 * gives NaN.  negate != 0 computes on -ts (coldSpells, xmhw.py:153-154).
 * q = pctile / 100.0 (identify.py:234).                                      */
int xmhw_clim_raw_f32(xmhw_plan *plan, const float *ts_dev, int64_t C, int64_t ld,
                      double q, int negate, double *thresh_dev, double *seas_dev,
                      int64_t ldo, void *stream);
int xmhw_clim_raw_f64(xmhw_plan *plan, const double *ts_dev, int64_t C, int64_t ld,
                      double q, int negate, double *thresh_dev, double *seas_dev,
                      int64_t ldo, void *stream);
/* The same on an int16-PACKED series read in place (CF packing: value = code * scale_factor + add_offset, `_FillValue`
 * -> NaN; what xr.open_dataset() decodes before threshold() sees it, docs/gettingstarted.rst:30-33): no decoded copy of
 * the series in HBM (2 bytes per sample instead of 4 or 8) and no decode pass.  codes_dev[T][ld] int16 (big_endian != 0:
 * byte-swapped, netCDF classic); decoded_itemsize = the dtype xarray (and xmhw_decode) would decode to:
 *   4  float32 (float32 packing attributes): the result is bit-identical to xmhw_decode(..., float32) +
 *      xmhw_clim_raw_f32 -- the kernel keys and sums float(code) * sf + of, two float32 roundings;
 *   8  float64 (float64 attributes): code -> value is monotone, so the kernel selects on the codes and decodes the
 *      two selected codes and the mean of the codes in float64: thresh bit-identical to xmhw_decode(..., float64) +
 *      xmhw_clim_raw_f64, seas within rounding (<= 1e-12: an exact integer sum instead of 440 rounded additions) --
 *      at the float32 kernel's speed instead of the 64-bit-key kernel's.
 * has_scale == 0: the samples are the codes themselves.  Served by the sorted-list kernel only: w = 5, records of 9..48
 * tracks, q >= 0.85; otherwise XMHW_ERR_UNSUPPORTED (decode, then xmhw_clim_raw_f32 / _f64).                        */
int xmhw_clim_raw_i16(xmhw_plan *plan, const int16_t *codes_dev, int64_t C, int64_t ld, int big_endian,
                      int has_scale, double scale_factor, double add_offset, int has_fill, int32_t fill_code,
                      int decoded_itemsize, double q, int negate, double *thresh_dev, double *seas_dev,
                      int64_t ldo, void *stream);

/* xmhw_clim_finish: the Feb-29 substitution (feb29(), identify.py:137-151,
 * applied at :237-240/:265-268 when feb29_fix != 0, i.e. tstep False) and
 * the circular running mean (runavg(), identify.py:154-181, when smooth != 0;
 * smooth_width must be odd) on both arrays, per cell over the groups PRESENT
 * (non-NaN) for that cell, as the reference's per-cell series are.
 * in/out may not alias.                                                      */
int xmhw_clim_finish(const xmhw_plan *plan, const double *thresh_in_dev,
                     const double *seas_in_dev, int64_t C, int64_t ldo, int feb29_fix,
                     int smooth, int smooth_width, double *thresh_out_dev,
                     double *seas_out_dev, void *stream);

/* One synchronous call = calc_clim() (xmhw.py:250-307) for all cells:
 * raw + finish, device buffers, builds a throw-away plan from doy_host.
 * D must equal the number of distinct labels in doy_host.                    */
int xmhw_clim_f32(const float *ts_dev, const int32_t *doy_host, int64_t T, int64_t C,
                  int32_t D, int32_t window_half_width, double q, int smooth,
                  int smooth_width, int feb29_fix, int negate, double *thresh_dev,
                  double *seas_dev, void *stream);
int xmhw_clim_f64(const double *ts_dev, const int32_t *doy_host, int64_t T, int64_t C,
                  int32_t D, int32_t window_half_width, double q, int smooth,
                  int smooth_width, int feb29_fix, int negate, double *thresh_dev,
                  double *seas_dev, void *stream);
/* Same with HOST buffers (copies in and out; PCIe-inclusive).                */
int xmhw_clim_host_f32(const float *ts_host, const int32_t *doy_host, int64_t T, int64_t C,
                       int32_t D, int32_t window_half_width, double q, int smooth,
                       int smooth_width, int feb29_fix, int negate, double *thresh_host,
                       double *seas_host);
int xmhw_clim_host_f64(const double *ts_host, const int32_t *doy_host, int64_t T, int64_t C,
                       int32_t D, int32_t window_half_width, double q, int smooth,
                       int smooth_width, int feb29_fix, int negate, double *thresh_host,
                       double *seas_host);

/* land_check()'s dropna (identify.py:522-525) on the stacked array:
 * keep[c] = 0 if cell c is all-NaN (anynans != 0: has any NaN), else 1.      */
int xmhw_land_mask_f32(const float *ts_dev, int64_t T, int64_t C, int64_t ld, int anynans,
                       uint8_t *keep_dev, void *stream);
int xmhw_land_mask_f64(const double *ts_dev, int64_t T, int64_t C, int64_t ld, int anynans,
                       uint8_t *keep_dev, void *stream);

/* the same on int16 codes (packed input read in place, xmhw_clim_raw_i16): a sample is missing when its code is
 * fill_code (has_fill == 0: no cell is dropped); big_endian: the codes are byte-swapped                         */
int xmhw_land_mask_i16(const int16_t *codes_dev, int64_t T, int64_t C, int64_t ld, int big_endian, int has_fill,
                       int32_t fill_code, int anynans, uint8_t *keep_dev, void *stream);

/* land_check()'s compaction on resident data: out[r][c] = in[r][index[c]] for the
 * n ocean cells listed in index_dev (ascending stacked-cell numbers, int64), and
 * the inverse for the results (what unstack('cell') does, xmhw.py:210-214):
 * out[r][index[c]] = in[r][c], every other element of out[rows][ld_out] = NaN.  */
int xmhw_gather_cells_f32(const float *in_dev, int64_t rows, int64_t ld_in,
                          const int64_t *index_dev, int64_t n, float *out_dev, int64_t ld_out,
                          void *stream);
int xmhw_gather_cells_f64(const double *in_dev, int64_t rows, int64_t ld_in,
                          const int64_t *index_dev, int64_t n, double *out_dev, int64_t ld_out,
                          void *stream);
int xmhw_gather_cells_i16(const int16_t *in_dev, int64_t rows, int64_t ld_in,
                          const int64_t *index_dev, int64_t n, int16_t *out_dev, int64_t ld_out,
                          void *stream);
int xmhw_scatter_cells_f64(const double *in_dev, int64_t rows, int64_t ld_in,
                           const int64_t *index_dev, int64_t n, double *out_dev, int64_t ld_out,
                           void *stream);

/* ---- detect() front end (next row of the path) ---------------------- *
 * For every cell: bthresh[t] = ts[t] > thresh[row_of_t[t]] (define_events(),
 * identify.py:366-372; NaN compares false; negate != 0 works on -ts, xmhw.py:413-414),
 * then mhw_filter() (identify.py:415-479) with join_gaps()/join_events() (identify.py:273-325,
 * :532-536).  row_of_t_host[T]: index of each step's doy label among the rows of thresh (HOST).
 * Outputs are int32 [T][ldo] with -1 where the reference has NaN: events = label (start
 * position) of the event covering a step; start = start label stored at the END step of the
 * first member of a joined event; end = end step stored at the end step of its last member.
 * bthresh_dev (uint8 [T][ldo]) and nevents_dev (int32 [C], number of joined events per
 * cell) may be NULL.                                                                    */
int xmhw_detect_events_f32(const float *ts_dev, int64_t T, int64_t C, int64_t ld,
                           const double *thresh_dev, int64_t ldt, const int32_t *row_of_t_host,
                           int32_t min_duration, int32_t join_gaps, int32_t max_gap, int32_t negate,
                           int32_t *events_dev, int32_t *start_dev, int32_t *end_dev,
                           uint8_t *bthresh_dev, int64_t ldo, int32_t *nevents_dev, void *stream);
int xmhw_detect_events_f64(const double *ts_dev, int64_t T, int64_t C, int64_t ld,
                           const double *thresh_dev, int64_t ldt, const int32_t *row_of_t_host,
                           int32_t min_duration, int32_t join_gaps, int32_t max_gap, int32_t negate,
                           int32_t *events_dev, int32_t *start_dev, int32_t *end_dev,
                           uint8_t *bthresh_dev, int64_t ldo, int32_t *nevents_dev, void *stream);

/* Number of (joined) events per cell: nevents_dev[C] int32, from the start array of
 * xmhw_detect_events_* (one start per event).                                            */
int xmhw_count_events(const int32_t *start_dev, int64_t T, int64_t C, int64_t ldo,
                      int32_t *nevents_dev, void *stream);

/* Per-event statistics: mhw_df() (xmhw/features.py:22-70) and
 * mhw_features() (features.py:72-315) for every event of every cell, from the labels of
 * xmhw_detect_events_*.  seas/thresh are the (D, C) climatologies (re-expanded by
 * row_of_t_host as in define_events(), identify.py:366-368); offsets_dev[C+1] int64 is the
 * exclusive prefix sum of the per-cell event counts; table_dev[offsets[C]][XMHW_EVENT_COLUMNS]
 * float64 receives one row per event, cells in order, events of a cell in time order.
 * Columns (time stamps as positions along the time axis):
 *  0 event  1 index_start  2 index_end  3 time_start  4 time_end  5 time_peak
 *  6 intensity_max  7 intensity_mean  8 intensity_cumulative  9 severity_max
 * 10 severity_mean 11 severity_cumulative 12 severity_var 13 intensity_mean_relThresh
 * 14 intensity_cumulative_relThresh 15 intensity_mean_abs 16 intensity_cumulative_abs
 * 17 duration_moderate 18 duration_strong 19 duration_severe 20 duration_extreme
 * 21 index_peak 22 intensity_var 23 intensity_max_relThresh 24 intensity_max_abs
 * 25 intensity_var_relThresh 26 intensity_var_abs 27 category 28 duration
 * 29 rate_onset 30 rate_decline                                                          */
#define XMHW_EVENT_COLUMNS 31
int xmhw_event_stats_f32(const float *ts_dev, int64_t T, int64_t C, int64_t ld,
                         const double *seas_dev, const double *thresh_dev, int64_t ldc,
                         const int32_t *row_of_t_host, int32_t negate, const int32_t *events_dev,
                         int64_t ldo, const int64_t *offsets_dev, double *table_dev, void *stream);
int xmhw_event_stats_f64(const double *ts_dev, int64_t T, int64_t C, int64_t ld,
                         const double *seas_dev, const double *thresh_dev, int64_t ldc,
                         const int32_t *row_of_t_host, int32_t negate, const int32_t *events_dev,
                         int64_t ldo, const int64_t *offsets_dev, double *table_dev, void *stream);

/* define_events() (xmhw/identify.py:326-412) when only the event TABLE is wanted — no per-step
 * outputs, the least HBM traffic the result allows.  Three stages:
 *  1. xmhw_exceed_bits_*: ts > thresh[row(t)] (identify.py:366-372) as one bit per sample;
 *     bits_dev [ceil(T/64)][ldb] uint64, bit (t & 63) of word t/64 of column c.  thresh_dev
 *     is (D, ldt); the f32 variant compares against the float32 floor of each threshold
 *     (identical results, half the bytes re-read per step).
 *  2. xmhw_events_from_bits: mhw_filter() + join_gaps() (identify.py:415-479, 273-325) on the
 *     bits.  offsets_dev == NULL: count only, nevents_dev[C] = events per cell.  Otherwise
 *     (offsets_dev[C+1] = exclusive prefix sum of those counts) column 0 (label), 1 (cell
 *     index, scratch), 3 and 4 (first / last labelled step) of every event's row of
 *     table_dev[offsets[C]][XMHW_EVENT_COLUMNS] are written; nevents_dev may be NULL.
 *     Asynchronous on `stream`.
 *  3. xmhw_event_stats_sparse_*: mhw_df() + mhw_features() (xmhw/features.py:22-315) for the
 *     n_events rows prepared by stage 2, one thread per event; fills all 31 columns.
 * Same results as xmhw_detect_events_* + xmhw_event_stats_*.                              */
/* Kernel choice of xmhw_exceed_bits_* (process-wide; tests and measurements): 0 = automatic
 * (tiled kernel - thresholds read once per cell - for calendar-like labels on >= 131072 cells,
 * otherwise the per-step kernel), 1 = per-step kernel, 2 = tiled kernel.                     */
int xmhw_set_exceed_kernel(int32_t mode);
int xmhw_exceed_bits_f32(const float *ts_dev, int64_t T, int64_t C, int64_t ld,
                         const double *thresh_dev, int64_t ldt, int64_t D,
                         const int32_t *row_of_t_host, int32_t negate, uint64_t *bits_dev,
                         int64_t ldb, void *stream);
int xmhw_exceed_bits_f64(const double *ts_dev, int64_t T, int64_t C, int64_t ld,
                         const double *thresh_dev, int64_t ldt, int64_t D,
                         const int32_t *row_of_t_host, int32_t negate, uint64_t *bits_dev,
                         int64_t ldb, void *stream);
int xmhw_events_from_bits(const uint64_t *bits_dev, int64_t T, int64_t C, int64_t ldb,
                          int32_t min_duration, int32_t join_gaps, int32_t max_gap,
                          const int64_t *offsets_dev, int32_t *nevents_dev, double *table_dev,
                          void *stream);
int xmhw_event_stats_sparse_f32(const float *ts_dev, int64_t T, int64_t C, int64_t ld,
                                const double *seas_dev, const double *thresh_dev, int64_t ldc,
                                const int32_t *row_of_t_host, int32_t negate, int64_t n_events,
                                double *table_dev, void *stream);
int xmhw_event_stats_sparse_f64(const double *ts_dev, int64_t T, int64_t C, int64_t ld,
                                const double *seas_dev, const double *thresh_dev, int64_t ldc,
                                const int32_t *row_of_t_host, int32_t negate, int64_t n_events,
                                double *table_dev, void *stream);

/* The `intermediate` Dataset of detect() (xmhw/xmhw.py:354-356; define_events(),
 * identify.py:405-409): the per-step columns mhw_df() adds (xmhw/features.py:36-69).
 * out_dev [8][T][ldv] float64 = seas, thresh (NaN outside events), relSeas, relThresh,
 * relThreshNorm, severity, cats, mabs; dur_dev [4][T][ldv] uint8 = duration_moderate,
 * duration_strong, duration_severe, duration_extreme.                                    */
#define XMHW_INTERMEDIATE_F64_PLANES 8
#define XMHW_INTERMEDIATE_U8_PLANES 4
int xmhw_event_intermediate_f32(const float *ts_dev, int64_t T, int64_t C, int64_t ld,
                                const double *seas_dev, const double *thresh_dev, int64_t ldc,
                                const int32_t *row_of_t_host, int32_t negate,
                                const int32_t *events_dev, int64_t ldo, double *out_dev,
                                int64_t ldv, uint8_t *dur_dev, void *stream);
int xmhw_event_intermediate_f64(const double *ts_dev, int64_t T, int64_t C, int64_t ld,
                                const double *seas_dev, const double *thresh_dev, int64_t ldc,
                                const int32_t *row_of_t_host, int32_t negate,
                                const int32_t *events_dev, int64_t ldo, double *out_dev,
                                int64_t ldv, uint8_t *dur_dev, void *stream);

/* Synthetic SST generated in HBM (bench + large parity runs):
 * x[t,c] = 15 + A_c sin(2 pi (t - phi_c)/365.25) + 5e-4 t beta_c + N(0,1),
 * counter-based on (seed, cell0 + c, t); a sample is NaN with probability
 * nan_frac.                                                                  */
int xmhw_synth_sst_f32(float *ts_dev, int64_t T, int64_t C, int64_t ld, int64_t cell0,
                       uint64_t seed, double nan_frac, void *stream);
int xmhw_synth_sst_f64(double *ts_dev, int64_t T, int64_t C, int64_t ld, int64_t cell0,
                       uint64_t seed, double nan_frac, void *stream);
/* the same with what real archives add (bench legs): values rounded to multiples of `quant` (OISST: 0.01; 0 = off), a
 * share `ice_frac` of the cells held at -1.8 for 120 days of every year, AR(1) anomalies with day-to-day correlation
 * `rho` (unit variance; 0 = the white noise of xmhw_synth_sst_f32, to which this is then bit-identical).  ice_patch:
 * the ice cells come in patches of that many consecutive cells with one season start per patch and +-15 days per cell
 * (an ice pack: neighbours freeze together); <= 1 = every cell decides and freezes on its own (scattered: the worst
 * case for a kernel that runs 32 neighbouring cells in lockstep)                                                     */
int xmhw_synth_sst_ex_f32(float *ts_dev, int64_t T, int64_t C, int64_t ld, int64_t cell0, uint64_t seed,
                          double nan_frac, double quant, double ice_frac, double rho, int64_t ice_patch,
                          void *stream);

/* The detect-side entries above take host row tables (row_of_t).  Their device copies (and the tiled
 * exceedance kernel's chunk tables, and per-stream scratch such as the float32 threshold copy) are
 * CACHED between calls, keyed by content: after the first call with a given table the entries are
 * asynchronous on `stream` -- no allocation, no synchronisation.  xmhw_release_cached_tables() frees
 * the caches (synchronises the device).
 * xmhw_offsets_from_counts: exclusive prefix sum of the per-cell event counts (the count pass of
 * xmhw_events_from_bits) into the int64 offsets (n + 1 entries, the last one = number of events)
 * the fill pass takes -- on the device, asynchronous.                                           */
int xmhw_release_cached_tables(void);
int xmhw_offsets_from_counts(const int32_t *counts_dev, int64_t n, int64_t *offsets_dev, void *stream);

/* ---- block_average() (xmhw/stats.py:27-428) ------------------------------ *
 * The reference runs groupby(pd.cut(years, bins, right=False)).agg(...) per cell (call_groupby
 * :285-319).  Device buffers throughout; bin_of_t[T] (int32, device) = year-bin index of every time
 * step, -1 outside the bins.  Results: out[stat][bin][cell] float64, leading dimension ldo >= C.
 * xmhw_block_events: the 15 statistics of agg_mhw (:344-362: ecount, duration, intensity_max,
 * intensity_max_max, intensity_mean, intensity_cumulative, total_icum, intensity_mean_relThresh,
 * intensity_cumulative_relThresh, severity_mean, severity_cumulative, intensity_mean_abs,
 * intensity_cumulative_abs, rate_onset, rate_decline) from the compact event table of detect()
 * (n_events x 31, events of cell c at rows offsets[c]..offsets[c+1]); an event belongs to the bin
 * of the time step in column mtime_column (3 = time_start, 5 = time_peak).
 * xmhw_block_time_*: ts_mean, ts_max, ts_min per block (agg_ts :421-425) and, when `cats` (T, ldcat)
 * is given, the moderate / strong / severe / extreme day counts (agg_cats :391-400): 7 planes.  */
int xmhw_block_events(const double *table_dev, const int64_t *offsets_dev, int64_t C,
                      const int32_t *bin_of_t_dev, int64_t T, int32_t nbins, int32_t mtime_column,
                      double *out_dev, int64_t ldo, void *stream);
int xmhw_block_time_f32(const float *ts_dev, int64_t T, int64_t C, int64_t ld, const double *cats_dev,
                        int64_t ldcat, const int32_t *bin_of_t_dev, int32_t nbins, double *out_dev,
                        int64_t ldo, void *stream);
int xmhw_block_time_f64(const double *ts_dev, int64_t T, int64_t C, int64_t ld, const double *cats_dev,
                        int64_t ldcat, const int32_t *bin_of_t_dev, int32_t nbins, double *out_dev,
                        int64_t ldo, void *stream);

/* ---- the sharded path: cells split across the GPUs of a node, ONE gather at the end ------- *
 * Replaces the reference's collect, dask.compute(climls) + xr.concat(dim='cell')
 * (xmhw/xmhw.py:197, :210-211).  Cells are independent (xmhw/xmhw.py:184-196), so rank r runs the
 * hot path above on its own contiguous block of columns [c0_r, c0_r + cols_r) -- the caller simply
 * passes that block's device pointer and width to xmhw_clim_raw_* / xmhw_clim_finish -- and the
 * (rows, cols_r) float64 result blocks are gathered device-to-device over RCCL / xGMI.
 * One process per GPU; RCCL is loaded (dlopen) at the first call of this section.             */
typedef struct xmhw_comm xmhw_comm;
#define XMHW_UNIQUE_ID_BYTES 128
/* rank 0 creates the 128-byte id and hands it to the other ranks by any out-of-band means
 * (xmhw_amd/bootstrap.py uses a TCP socket; MPI_Bcast or a shared file work as well)          */
int xmhw_comm_unique_id(void *id_out);
/* collective over all ranks, on each rank's CURRENT device (xmhw_set_device first)             */
int xmhw_comm_create(int rank, int nranks, const void *id, xmhw_comm **comm);
int xmhw_comm_destroy(xmhw_comm *comm);
int xmhw_comm_info(const xmhw_comm *comm, int *rank, int *nranks);
/* metadata: every rank contributes one int64 (cell counts, table sizes, error flags) and gets
 * all of them back on the host; synchronises `stream`                                           */
int xmhw_comm_allgather_i64(xmhw_comm *comm, int64_t value, int64_t *out_host, void *stream);
/* the same in two halves: begin() queues copy-in, collective and copy-out (pinned on both ends) and
 * returns; end() waits for that work only.  One all-gather in flight per communicator.        */
int xmhw_comm_allgather_i64_begin(xmhw_comm *comm, int64_t value, void *stream);
int xmhw_comm_allgather_i64_end(xmhw_comm *comm, int64_t *out_host);
/* equal-sized byte blocks (land-mask slabs): recv_dev holds nranks * bytes_per_rank; asynchronous */
int xmhw_comm_allgather_bytes(xmhw_comm *comm, const void *send_dev, void *recv_dev,
                              size_t bytes_per_rank, void *stream);
/* THE gather: rank r sends its dense (rows, cols) float64 block; on `root`, recv_dev receives the
 * blocks one after the other in rank order, block r being (rows, cols_of_rank[r]) contiguous
 * (cols_of_rank is read on the root only and cols_of_rank[root] must equal cols; empty blocks
 * are allowed).  Grouped ncclSend / ncclRecv on `stream`, asynchronous; the caller places the
 * blocks (xmhw_memcpy2d_d2h writes block r straight into columns [c0_r, c0_r + cols_r) of a
 * row-major host array).                                                                        */
int xmhw_gather_blocks(xmhw_comm *comm, const double *send_dev, int64_t rows, int64_t cols,
                       double *recv_dev, const int64_t *cols_of_rank, int root, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* XMHW_AMD_H */
