// event_acc.h -- per-event running accumulators shared by the event-statistics kernels
// (mhw_df() + mhw_features(), xmhw/features.py:22-315).  Column order: kEventColumns in kernels.h,
// the same as oracle/features_oracle.py COLUMNS.
#pragma once
#include "device_common.h"
#include "kernels.h"

namespace xmhw {

// Running variance without a division per sample: sums of (x - K) and (x - K)^2 with K = the
// series' first sample in the event (shifted-data algorithm: K lies inside the event's spread, so
// the final subtraction does not cancel catastrophically), one division at the flush.
struct Welford {
    double n = 0.0, k = 0.0, s1 = 0.0, s2 = 0.0;
    __device__ __forceinline__ void add(double x) {
        if (n == 0.0) k = x;
        n += 1.0;
        const double d = x - k;
        s1 += d;
        s2 += d * d;
    }
    __device__ __forceinline__ double sd() const {
        if (n < 2.0) return make_nan();
        const double v = (s2 - s1 * s1 / n) / (n - 1.0);
        return sqrt(v > 0.0 ? v : 0.0);
    }
};

struct EventAcc {
    int32_t label;
    int64_t first, last;          // first / last labelled step
    double s_relS, s_sev, s_relT, s_abs;
    double max_relS, max_sev, max_cat, relT_at_max, abs_at_max;
    int64_t imax;                 // position (within the group) of the first maximum of relSeas
    double n_mod, n_str, n_sev, n_ext;
    double relS_first, relS_last, anom_first, anom_last;
    bool have_first, have_afirst;
    Welford w_relS, w_sev, w_relT, w_abs;
    __device__ void reset(int32_t L, int64_t t) {
        label = L; first = t; last = t;
        s_relS = s_sev = s_relT = s_abs = 0.0;
        max_relS = max_sev = max_cat = relT_at_max = abs_at_max = make_nan();
        imax = -1;
        n_mod = n_str = n_sev = n_ext = 0.0;
        relS_first = relS_last = anom_first = anom_last = make_nan();
        have_first = have_afirst = false;
        w_relS = Welford(); w_sev = Welford(); w_relT = Welford(); w_abs = Welford();
    }
};

__device__ void flush_event(const EventAcc& a, int64_t last_index, double* __restrict__ row) {
    const double nan = make_nan();
    const double L = static_cast<double>(a.label);
    const double i_start = L, i_end = static_cast<double>(a.last);
    const double i_peak = L + static_cast<double>(a.imax);
    row[0] = L;
    row[1] = i_start;
    row[2] = i_end;
    row[3] = static_cast<double>(a.first);
    row[4] = static_cast<double>(a.last);
    row[5] = a.imax >= 0 ? static_cast<double>(a.first + a.imax) : nan;
    row[6] = a.max_relS;
    row[7] = a.w_relS.n > 0 ? a.s_relS / a.w_relS.n : nan;
    row[8] = a.s_relS;
    row[9] = a.max_sev;
    row[10] = a.w_sev.n > 0 ? a.s_sev / a.w_sev.n : nan;
    row[11] = a.s_sev;
    row[12] = a.w_sev.sd();
    row[13] = a.w_relT.n > 0 ? a.s_relT / a.w_relT.n : nan;
    row[14] = a.s_relT;
    row[15] = a.w_abs.n > 0 ? a.s_abs / a.w_abs.n : nan;
    row[16] = a.s_abs;
    row[17] = a.n_mod;
    row[18] = a.n_str;
    row[19] = a.n_sev;
    row[20] = a.n_ext;
    row[21] = i_peak;
    row[22] = a.w_relS.sd();
    row[23] = a.relT_at_max;
    row[24] = a.abs_at_max;
    row[25] = a.w_relT.sd();
    row[26] = a.w_abs.sd();
    row[27] = a.max_cat == a.max_cat ? fmin(a.max_cat, 4.0) : nan;
    row[28] = i_end - i_start + 1.0;
    // onset / decline (features.py:224-295)
    const double peak = i_peak - i_start;
    const double esp = i_end - i_start - peak;
    const double x = peak != 0.0 ? peak : 1.0;
    const double onset_period = i_start == 0.0 ? x : x + 0.5;
    const double y = peak != static_cast<double>(last_index) ? esp : 1.0;
    const double decline_period = i_end == static_cast<double>(last_index) ? y : y + 0.5;
    const double edge0 = 0.5 * (a.relS_first + (i_start == 0.0 ? a.relS_first : a.anom_first));
    const double edge1 = 0.5 * (a.relS_last + (i_end == static_cast<double>(last_index) ? a.relS_last : a.anom_last));
    row[29] = (a.max_relS - edge0) / onset_period;
    row[30] = (a.max_relS - edge1) / decline_period;
}


// One labelled step of an event: every series skips its own NaNs, as pandas' groupby aggregations do.
__device__ __forceinline__ void event_add_step(EventAcc& a, int64_t t, double x, double se, double th) {
    const double relS = x - se, relT = x - th, thse = th - se;
    const double sev = relS / -(thse);
    const double cat = floor(1.0 + relT / thse);
    if (x == x) { a.s_abs += x; a.w_abs.add(x); }
    if (relT == relT) { a.s_relT += relT; a.w_relT.add(relT); }
    if (relS == relS) {
        a.s_relS += relS; a.w_relS.add(relS);
        if (!(a.max_relS >= relS)) {       // first maximum (NaN-initialised)
            a.max_relS = relS; a.imax = t - a.first; a.relT_at_max = relT; a.abs_at_max = x;
        }
        if (!a.have_first) { a.relS_first = relS; a.have_first = true; }
        a.relS_last = relS;
    }
    if (sev == sev) {
        a.s_sev += sev; a.w_sev.add(sev);
        if (!(a.max_sev >= sev)) a.max_sev = sev;
    }
    if (cat == cat) {
        if (!(a.max_cat >= cat)) a.max_cat = cat;
        a.n_mod += cat == 1.0 ? 1.0 : 0.0;
        a.n_str += cat == 2.0 ? 1.0 : 0.0;
        a.n_sev += cat == 3.0 ? 1.0 : 0.0;
        a.n_ext += cat >= 4.0 ? 1.0 : 0.0;
    }
}

}  // namespace xmhw
