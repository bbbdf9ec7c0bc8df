// kernels_stats.hip -- block_average() (xmhw/stats.py:27-428).
//
// The reference converts one cell's events to a DataFrame and runs
// groupby(pd.cut(years, bins, right=False)).agg(...) per cell under dask (call_groupby :285-319,
// agg_mhw :322-364, agg_ts / agg_cats :372-428).  Here the compact event table of detect()
// (n_events x 31, events of a cell contiguous and in time order) is reduced by one thread per cell:
// the events of a cell fall into non-decreasing year bins, so every aggregation is a running
// accumulator flushed when the bin changes -- a segmented reduction keyed by (cell, year bin).
// The time-axis statistics (ts mean / max / min and category day counts per block) stream the
// series once more, lanes along cells (coalesced), the same way.
// pandas semantics: every bin present; count = non-NaN values; mean / max / min / sum skip NaN; an
// empty bin gives NaN for mean / max / min and 0 for count / sum.
// Output layout: out[stat][bin][cell] (cell-minor, coalesced).
#include "device_common.h"
#include "kernels.h"

namespace xmhw {
namespace {

// aggregation dictionary of agg_mhw (stats.py:344-362): source column in the event table
// (detect_front.EVENT_COLUMNS order) and the reduction: 0 count, 1 mean, 2 max, 3 sum
constexpr int kNS = kBlockEventStats;
__constant__ int kSrc[kNS] = {0, 28, 6, 6, 7, 8, 8, 13, 14, 10, 11, 7, 8, 29, 30};
__constant__ int kHow[kNS] = {0, 1, 1, 2, 1, 1, 3, 1, 1, 1, 1, 1, 1, 1, 1};

struct Acc {
    double v[kNS];      // sum (mean, sum) or max
    uint32_t n[kNS];    // non-NaN values seen
    __device__ void reset() {
#pragma unroll
        for (int j = 0; j < kNS; ++j) { v[j] = 0.0; n[j] = 0; }
    }
};

__global__ __launch_bounds__(256) void block_events(const double* __restrict__ table, const int64_t* __restrict__ offsets,
                                                    int64_t C, const int32_t* __restrict__ bin_of_t, int64_t Tn,
                                                    int32_t nbins, int32_t mtime_col, double* __restrict__ out, int64_t ldo) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int64_t plane = static_cast<int64_t>(nbins) * ldo;
    // every bin exists: defaults first (count 0, sum 0, everything else NaN)
    for (int32_t b = 0; b < nbins; ++b)
#pragma unroll
        for (int j = 0; j < kNS; ++j) out[j * plane + b * ldo + c] = (kHow[j] == 0 || kHow[j] == 3) ? 0.0 : make_nan();
    Acc acc;
    acc.reset();
    int32_t cur = -1;
    auto flush = [&]() {
        if (cur < 0) return;
#pragma unroll
        for (int j = 0; j < kNS; ++j) {
            double r;
            if (kHow[j] == 0) r = static_cast<double>(acc.n[j]);
            else if (kHow[j] == 3) r = acc.v[j];
            else if (acc.n[j] == 0) r = make_nan();
            else if (kHow[j] == 1) r = acc.v[j] / static_cast<double>(acc.n[j]);
            else r = acc.v[j];
            out[j * plane + cur * ldo + c] = r;
        }
    };
    for (int64_t e = offsets[c]; e < offsets[c + 1]; ++e) {
        const double* row = table + e * kEventColumns;
        const double pos = row[mtime_col];
        int32_t b = -1;
        if (pos == pos && pos >= 0.0 && pos < static_cast<double>(Tn)) b = bin_of_t[static_cast<int64_t>(pos)];
        if (b < 0 || b >= nbins) continue;          // NaT / outside every bin: in no group
        if (b != cur) {
            flush();
            acc.reset();
            cur = b;
        }
#pragma unroll
        for (int j = 0; j < kNS; ++j) {
            const double x = row[kSrc[j]];
            if (x != x) continue;
            if (kHow[j] == 2) acc.v[j] = (acc.n[j] == 0 || x > acc.v[j]) ? x : acc.v[j];
            else acc.v[j] += x;
            acc.n[j] += 1;
        }
    }
    flush();
}

// ts_mean, ts_max, ts_min (+ moderate / strong / severe / extreme day counts when cats are given)
template <typename T>
__global__ __launch_bounds__(256) void block_time(const T* __restrict__ ts, int64_t Tn, int64_t C, int64_t ld,
                                                  const double* __restrict__ cats, int64_t ldcat,
                                                  const int32_t* __restrict__ bin_of_t, int32_t nbins,
                                                  double* __restrict__ out, int64_t ldo) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int nstat = cats ? 7 : 3;
    const int64_t plane = static_cast<int64_t>(nbins) * ldo;
    for (int32_t b = 0; b < nbins; ++b)
        for (int j = 0; j < nstat; ++j) out[j * plane + b * ldo + c] = j < 3 ? make_nan() : 0.0;
    double sum = 0.0, mx = 0.0, mn = 0.0;
    uint32_t n = 0, days[4] = {0, 0, 0, 0};
    int32_t cur = -1;
    auto flush = [&]() {
        if (cur < 0) return;
        if (n) {
            out[0 * plane + cur * ldo + c] = sum / static_cast<double>(n);
            out[1 * plane + cur * ldo + c] = mx;
            out[2 * plane + cur * ldo + c] = mn;
        }
        if (cats)
            for (int k = 0; k < 4; ++k) out[(3 + k) * plane + cur * ldo + c] = static_cast<double>(days[k]);
    };
    // rows are loaded kAhead at a time (the loads of a one-row-per-iteration loop wait for each other:
    // 1.8 TB/s), then consumed in time order, so the arithmetic and its rounding are unchanged
    constexpr int kAhead = 8;
    for (int64_t t0 = 0; t0 < Tn; t0 += kAhead) {
        T xs[kAhead];
        double ks[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int64_t t = t0 + u < Tn ? t0 + u : Tn - 1;
            xs[u] = ts[t * ld + c];
            ks[u] = cats ? cats[t * ldcat + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int64_t t = t0 + u;
            if (t >= Tn) break;
            const int32_t b = bin_of_t[t];
            if (b < 0 || b >= nbins) continue;
            if (b != cur) {
                flush();
                sum = 0.0; n = 0; days[0] = days[1] = days[2] = days[3] = 0;
                cur = b;
            }
            const double x = static_cast<double>(xs[u]);
            if (x == x) {
                mx = (n == 0 || x > mx) ? x : mx;
                mn = (n == 0 || x < mn) ? x : mn;
                sum += x;
                n += 1;
            }
            if (cats) {
                const double k = ks[u];
                days[0] += k == 1.0; days[1] += k == 2.0; days[2] += k == 3.0; days[3] += k == 4.0;
            }
        }
    }
    flush();
}

}  // namespace

hipError_t launch_block_events(const double* table, const int64_t* offsets, int64_t C, const int32_t* bin_of_t, int64_t Tn,
                               int32_t nbins, int32_t mtime_col, double* out, int64_t ldo, hipStream_t stream) {
    if (C <= 0 || nbins <= 0) return hipSuccess;
    hipLaunchKernelGGL(block_events, dim3(static_cast<unsigned>((C + 255) / 256)), dim3(256), 0, stream, table, offsets, C,
                       bin_of_t, Tn, nbins, mtime_col, out, ldo);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_block_time(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* cats, int64_t ldcat,
                             const int32_t* bin_of_t, int32_t nbins, double* out, int64_t ldo, hipStream_t stream) {
    if (C <= 0 || nbins <= 0) return hipSuccess;
    hipLaunchKernelGGL(block_time<T>, dim3(static_cast<unsigned>((C + 255) / 256)), dim3(256), 0, stream, ts, Tn, C, ld, cats,
                       ldcat, bin_of_t, nbins, out, ldo);
    return hipGetLastError();
}
template hipError_t launch_block_time<float>(const float*, int64_t, int64_t, int64_t, const double*, int64_t, const int32_t*,
                                             int32_t, double*, int64_t, hipStream_t);
template hipError_t launch_block_time<double>(const double*, int64_t, int64_t, int64_t, const double*, int64_t, const int32_t*,
                                              int32_t, double*, int64_t, hipStream_t);

}  // namespace xmhw
