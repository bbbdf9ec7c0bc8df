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
#include "kernels.h"

namespace xmhw {

template <typename T>
__global__ __launch_bounds__(256) void detect_events(
    const T* __restrict__ ts, int64_t Tn, int64_t C, int64_t ld, const double* __restrict__ thresh,
    int64_t ldt, const int32_t* __restrict__ row_of_t, int32_t min_duration, int32_t join_gaps,
    int32_t max_gap, int32_t negate, int32_t* __restrict__ events, int32_t* __restrict__ start,
    int32_t* __restrict__ end, uint8_t* __restrict__ bthresh, int64_t ldo) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= C) return;
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
}

template <typename T>
hipError_t launch_detect(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* thresh, int64_t ldt,
                         const int32_t* row_of_t, int32_t min_duration, int32_t join_gaps, int32_t max_gap,
                         int32_t negate, int32_t* events, int32_t* start, int32_t* end, uint8_t* bthresh,
                         int64_t ldo, hipStream_t stream) {
    if (C <= 0 || Tn <= 0) return hipSuccess;
    hipLaunchKernelGGL(detect_events<T>, dim3(static_cast<unsigned>((C + 255) / 256)), dim3(256), 0, stream, ts,
                       Tn, C, ld, thresh, ldt, row_of_t, min_duration, join_gaps, max_gap, negate, events,
                       start, end, bthresh, ldo);
    return hipGetLastError();
}
template hipError_t launch_detect<float>(const float*, int64_t, int64_t, int64_t, const double*, int64_t,
                                         const int32_t*, int32_t, int32_t, int32_t, int32_t, int32_t*, int32_t*,
                                         int32_t*, uint8_t*, int64_t, hipStream_t);
template hipError_t launch_detect<double>(const double*, int64_t, int64_t, int64_t, const double*, int64_t,
                                          const int32_t*, int32_t, int32_t, int32_t, int32_t, int32_t*, int32_t*,
                                          int32_t*, uint8_t*, int64_t, hipStream_t);

}  // namespace xmhw
