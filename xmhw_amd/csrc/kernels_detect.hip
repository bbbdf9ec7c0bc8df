// kernels_detect.hip -- the detect() front end, thread per cell:
//   * th.sel(doy=ts.doy) re-expansion + exceedance  ts > thresh      (xmhw/identify.py:366-372)
//   * mhw_filter(): runs of >= minDuration exceedances                (identify.py:415-479)
//   * join_gaps()/join_events(): merge events <= maxGap steps apart   (identify.py:273-325, 532-536)
// One streaming pass over time per cell (lanes along cells: ts, the thresh row of the step's doy
// and all outputs are coalesced); the reference's vectorised pandas expressions become a small
// state machine.  Outputs are int32 with -1 where the reference has NaN:
//   events[t]  label (= start position) of the event covering step t
//   start[t]   start label, stored at the END step of the first member of a (joined) event
//   end[t]     end step, stored at the end step of the last member
// Quirk kept (identify.py:445-449): a run beginning at step 0 has label 1 and its first step is
// not part of the event.
#include "device_common.h"
#include "event_acc.h"
#include "kernels.h"

namespace xmhw {

template <typename T>
__global__ __launch_bounds__(256) void detect_events(
    const T* __restrict__ ts, int64_t Tn, int64_t C, int64_t ld, const double* __restrict__ thresh,
    int64_t ldt, const int32_t* __restrict__ row_of_t, int32_t min_duration, int32_t join_gaps,
    int32_t max_gap, int32_t negate, int32_t* __restrict__ events, int32_t* __restrict__ start,
    int32_t* __restrict__ end, uint8_t* __restrict__ bthresh, int64_t ldo, int32_t* __restrict__ nevents) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= C) return;
    int32_t count = 0;  // events after joining (= groups)
    int64_t prev_nonexc = -1;  // last step that was not an exceedance (-1: none yet)
    bool in_run = false;
    int64_t p = 0, run_first = 0;
    bool have_prev = false;    // a selected event exists before the current one
    int64_t prev_end = -1;
    int32_t group_start = 0;
    constexpr int U = 8;  // steps loaded ahead of the state machine (memory-level parallelism)
    for (int64_t t0 = 0; t0 < Tn; t0 += U) {
        T xs[U];
        double ths[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t t = t0 + u < Tn ? t0 + u : Tn - 1;
            xs[u] = ts[t * ld + c];
            ths[u] = thresh[static_cast<int64_t>(row_of_t[t]) * ldt + c];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t t = t0 + u;
            if (t >= Tn) break;
            T x = xs[u];
            if (negate) x = -x;
            const bool b = static_cast<double>(x) > ths[u];   // NaN on either side -> false
            if (bthresh) bthresh[t * ldo + c] = b ? 1 : 0;
            start[t * ldo + c] = -1;
            end[t * ldo + c] = -1;
            int32_t ev = -1;
            if (b) {
                if (!in_run) {
                    in_run = true;
                    p = prev_nonexc >= 0 ? prev_nonexc : 0;   // fillna(0)
                    run_first = t;
                }
                if (t - p != 0) ev = static_cast<int32_t>(p + 1);
            }
            events[t * ldo + c] = ev;
            if (in_run && (!b || t == Tn - 1)) {
                const int64_t te = b ? t : t - 1;
                if (te - p >= min_duration) {
                    const int32_t S = static_cast<int32_t>(p + 1);
                    const bool joined = join_gaps && have_prev && (S - prev_end <= max_gap + 1);
                    if (joined) {
                        end[prev_end * ldo + c] = -1;
                        for (int64_t k = prev_end + 1; k <= te; ++k) events[k * ldo + c] = group_start;
                    } else {
                        group_start = S;
                        start[te * ldo + c] = S;
                        ++count;
                    }
                    end[te * ldo + c] = static_cast<int32_t>(te);
                    prev_end = te;
                    have_prev = true;
                } else {
                    for (int64_t k = run_first; k <= te; ++k) events[k * ldo + c] = -1;
                }
                in_run = false;
            }
            if (!b) prev_nonexc = t;
        }
    }
    if (nevents) nevents[c] = count;
}

// ---------------------------------------------------------------------------
// event_stats: per-event statistics: mhw_df() (xmhw/features.py:22-70) and
// mhw_features() (features.py:72-315: agg_df, properties, onset_decline) for every event of every
// cell, from the labels produced by detect_events.  One thread per cell streams over time; the
// steps of an event are contiguous (gap steps of joined events carry the label), so the groupby
// aggregations become running accumulators that are flushed into the cell's slice of a compact
// event table (offsets = exclusive prefix sum of the per-cell event counts).  NaN samples are
// skipped as pandas does; variances (ddof = 1) come from shifted sums, returned as standard deviations.
// Columns: see kEventColumns in kernels.h (same order as oracle/features_oracle.py COLUMNS).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(128) void event_stats(
    const T* __restrict__ ts, int64_t Tn, int64_t C, int64_t ld, const double* __restrict__ seas,
    const double* __restrict__ thresh, int64_t ldc, const int32_t* __restrict__ row_of_t, int32_t negate,
    const int32_t* __restrict__ events, int64_t ldo, const int64_t* __restrict__ offsets,
    double* __restrict__ table) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double* out = table + offsets[c] * kEventColumns;
    const int64_t nmax = offsets[c + 1] - offsets[c];
    int64_t nout = 0;
    EventAcc a;
    a.reset(-1, 0);
    bool in_event = false, prev_in_event = false;
    double anom_prev = make_nan();
    // Labels are read for every step; the sample and the two climatology rows only where they are
    // used: on labelled steps and on their immediate neighbours (anom_plus / anom_minus).  Events
    // cover ~10 % of the steps, so most 32-byte sectors of ts / seas / thresh are never fetched.
    constexpr int U = 8;
    int32_t ev_prev = -1;
    int32_t ev_next = events[c];                         // label of step t0 (carried between batches)
    for (int64_t t0 = 0; t0 < Tn; t0 += U) {
        int32_t evs[U + 1];
        evs[0] = ev_next;
#pragma unroll
        for (int u = 1; u <= U; ++u) evs[u] = t0 + u < Tn ? events[(t0 + u) * ldo + c] : -1;
        ev_next = evs[U];
        double xs[U], ses[U], ths[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t t = t0 + u;
            const bool here = evs[u] >= 0;
            const bool near = here || evs[u + 1] >= 0 || (u ? evs[u - 1] >= 0 : ev_prev >= 0);
            xs[u] = ses[u] = ths[u] = make_nan();
            if (t < Tn && near) {
                const int64_t r = row_of_t[t];
                xs[u] = static_cast<double>(ts[t * ld + c]);
                ses[u] = seas[r * ldc + c];
                if (here) ths[u] = thresh[r * ldc + c];
            }
        }
        ev_prev = evs[U - 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t t = t0 + u;
            if (t >= Tn) break;
            double x = xs[u];
            if (negate) x = -x;
            const double se = ses[u], th = ths[u];
            const int32_t ev = evs[u];
            const double anom = x - se;
            // anom_minus of the previous step (= this step's anomaly) closes the previous step's view
            if (prev_in_event && anom == anom) a.anom_last = anom;
            if (in_event && ev != a.label) {
                if (nout < nmax) flush_event(a, Tn - 1, out + nout * kEventColumns);
                ++nout;
                in_event = false;
            }
            if (ev >= 0) {
                if (!in_event) { a.reset(ev, t); in_event = true; }
                a.last = t;
                if (!a.have_afirst && anom_prev == anom_prev) { a.anom_first = anom_prev; a.have_afirst = true; }
                event_add_step(a, t, x, se, th);
            }
            prev_in_event = in_event;
            anom_prev = anom;
        }
    }
    if (in_event) {
        if (nout < nmax) flush_event(a, Tn - 1, out + nout * kEventColumns);
        ++nout;
    }
}

template <typename T>
hipError_t launch_detect(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* thresh, int64_t ldt,
                         const int32_t* row_of_t, int32_t min_duration, int32_t join_gaps, int32_t max_gap,
                         int32_t negate, int32_t* events, int32_t* start, int32_t* end, uint8_t* bthresh,
                         int64_t ldo, int32_t* nevents, hipStream_t stream) {
    if (C <= 0 || Tn <= 0) return hipSuccess;
    hipLaunchKernelGGL(detect_events<T>, dim3(static_cast<unsigned>((C + 255) / 256)), dim3(256), 0, stream, ts,
                       Tn, C, ld, thresh, ldt, row_of_t, min_duration, join_gaps, max_gap, negate, events,
                       start, end, bthresh, ldo, nevents);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void count_events(const int32_t* __restrict__ start, int64_t Tn, int64_t C,
                                                    int64_t ldo, int32_t* __restrict__ nevents) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= C) return;
    int32_t n = 0;
    for (int64_t t = 0; t < Tn; ++t) n += start[t * ldo + c] >= 0 ? 1 : 0;
    nevents[c] = n;
}

hipError_t launch_count_events(const int32_t* start, int64_t Tn, int64_t C, int64_t ldo, int32_t* nevents,
                               hipStream_t stream) {
    if (C <= 0) return hipSuccess;
    hipLaunchKernelGGL(count_events, dim3(static_cast<unsigned>((C + 255) / 256)), dim3(256), 0, stream, start, Tn,
                       C, ldo, nevents);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_event_stats(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* seas,
                              const double* thresh, int64_t ldc, const int32_t* row_of_t, int32_t negate,
                              const int32_t* events, int64_t ldo, const int64_t* offsets, double* table,
                              hipStream_t stream) {
    if (C <= 0 || Tn <= 0) return hipSuccess;
    hipLaunchKernelGGL(event_stats<T>, dim3(static_cast<unsigned>((C + 127) / 128)), dim3(128), 0, stream, ts, Tn,
                       C, ld, seas, thresh, ldc, row_of_t, negate, events, ldo, offsets, table);
    return hipGetLastError();
}
template hipError_t launch_event_stats<float>(const float*, int64_t, int64_t, int64_t, const double*, const double*,
                                              int64_t, const int32_t*, int32_t, const int32_t*, int64_t,
                                              const int64_t*, double*, hipStream_t);
template hipError_t launch_event_stats<double>(const double*, int64_t, int64_t, int64_t, const double*,
                                               const double*, int64_t, const int32_t*, int32_t, const int32_t*,
                                               int64_t, const int64_t*, double*, hipStream_t);
template hipError_t launch_detect<float>(const float*, int64_t, int64_t, int64_t, const double*, int64_t,
                                         const int32_t*, int32_t, int32_t, int32_t, int32_t, int32_t*, int32_t*,
                                         int32_t*, uint8_t*, int64_t, int32_t*, hipStream_t);
template hipError_t launch_detect<double>(const double*, int64_t, int64_t, int64_t, const double*, int64_t,
                                          const int32_t*, int32_t, int32_t, int32_t, int32_t, int32_t*, int32_t*,
                                          int32_t*, uint8_t*, int64_t, int32_t*, hipStream_t);

// ---------------------------------------------------------------------------
// event_intermediate: the per-step columns mhw_df() adds (xmhw/features.py:36-69), i.e. the
// `intermediate` Dataset of detect() (xmhw/xmhw.py:354-356, identify.py:405-409).  Elementwise:
// out plane k of [8][T][ldv] float64 = seas, thresh (both masked to event steps), relSeas,
// relThresh, relThreshNorm, severity, cats, mabs; dur plane k of [4][T][ldv] u8 =
// duration_moderate / strong / severe / extreme.
// ---------------------------------------------------------------------------
constexpr int kInterRows = 8;   // time steps per thread

template <typename T>
__global__ __launch_bounds__(256) void event_intermediate(
    const T* __restrict__ ts, int64_t Tn, int64_t C, int64_t ld, const double* __restrict__ seas,
    const double* __restrict__ thresh, int64_t ldc, const int32_t* __restrict__ row_of_t, int32_t negate,
    const int32_t* __restrict__ events, int64_t ldo, double* __restrict__ out, int64_t ldv,
    uint8_t* __restrict__ dur) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int64_t t0 = static_cast<int64_t>(blockIdx.y) * kInterRows;
    const int64_t plane = Tn * ldv;
    const double nan = make_nan();
#pragma unroll
    for (int u = 0; u < kInterRows; ++u) {
        const int64_t t = t0 + u;
        if (t >= Tn) break;
        double x = static_cast<double>(ts[t * ld + c]);
        if (negate) x = -x;
        const int64_t r = row_of_t[t];
        const bool is = events[t * ldo + c] >= 0;
        const double mt = is ? x : nan;
        const double ms = is ? seas[r * ldc + c] : nan;
        const double mth = is ? thresh[r * ldc + c] : nan;
        const double relS = mt - ms, relT = mt - mth, ths = mth - ms;
        const double relTN = relT / ths;
        const double sev = relS / -(ths);
        const double cat = floor(1.0 + relTN);
        const int64_t o = t * ldv + c;
        out[o] = ms;
        out[plane + o] = mth;
        out[2 * plane + o] = relS;
        out[3 * plane + o] = relT;
        out[4 * plane + o] = relTN;
        out[5 * plane + o] = sev;
        out[6 * plane + o] = cat;
        out[7 * plane + o] = mt;
        dur[o] = cat == 1.0 ? 1 : 0;
        dur[plane + o] = cat == 2.0 ? 1 : 0;
        dur[2 * plane + o] = cat == 3.0 ? 1 : 0;
        dur[3 * plane + o] = cat >= 4.0 ? 1 : 0;
    }
}

template <typename T>
hipError_t launch_event_intermediate(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* seas,
                                     const double* thresh, int64_t ldc, const int32_t* row_of_t, int32_t negate,
                                     const int32_t* events, int64_t ldo, double* out, int64_t ldv, uint8_t* dur,
                                     hipStream_t stream) {
    if (C <= 0 || Tn <= 0) return hipSuccess;
    const dim3 grid(static_cast<unsigned>((C + 255) / 256), static_cast<unsigned>((Tn + kInterRows - 1) / kInterRows));
    hipLaunchKernelGGL(event_intermediate<T>, grid, dim3(256), 0, stream, ts, Tn, C, ld, seas, thresh, ldc, row_of_t,
                       negate, events, ldo, out, ldv, dur);
    return hipGetLastError();
}
template hipError_t launch_event_intermediate<float>(const float*, int64_t, int64_t, int64_t, const double*,
                                                     const double*, int64_t, const int32_t*, int32_t, const int32_t*,
                                                     int64_t, double*, int64_t, uint8_t*, hipStream_t);
template hipError_t launch_event_intermediate<double>(const double*, int64_t, int64_t, int64_t, const double*,
                                                      const double*, int64_t, const int32_t*, int32_t, const int32_t*,
                                                      int64_t, double*, int64_t, uint8_t*, hipStream_t);

}  // namespace xmhw
