// kernels_events.hip -- define_events() without per-step outputs (SURVEY.md section 8f ranks 1+2):
// the event TABLE of every cell from the series and the two climatologies, moving as few HBM bytes
// as the result needs.  Same semantics as detect_events + event_stats (kernels_detect.hip), which
// stay for callers that want the per-step labels (mhw_filter() frame, `intermediate` Dataset).
//
//   exceed_bits<T,TH>  ts > thresh[row(t)] as ONE BIT per sample, 64 consecutive steps of a cell
//                      per word (xmhw/identify.py:366-372).  bits[w][c], w = t / 64.  For float32
//                      series the threshold rows are first floored to float32 (floor_to_f32),
//                      which halves the bytes re-read per step and changes no result.
//   events_from_bits   mhw_filter() + join_gaps() (identify.py:415-479, 273-325) on those words:
//                      runs of ones are found with count-trailing-zero scans, a few operations
//                      per RUN instead of per step.  Called twice: count (-> prefix sum on the
//                      host) and fill (label, cell, first and last step of each event into its
//                      table row).
//   event_stats_sparse mhw_df() + mhw_features() (xmhw/features.py:22-315), one thread per EVENT:
//                      only the samples and climatology rows of labelled steps and their two
//                      neighbours are read.
#include "device_common.h"
#include "event_acc.h"
#include "kernels.h"

namespace xmhw {

constexpr int kBitsLoadAhead = 8;

// float32 series: (double)x > th  <=>  x > tf with tf = the largest float32 <= th (the next float32
// above tf is > th by construction; NaN and +-inf carry over), so the re-expanded threshold rows
// can be read as 4-byte values without changing a single bit of the result.
// (rows, cols) region of a row-major array with leading dimension ldi -> compact (rows, ldo) floats:
// only the addressed region is read, so a column block of a wider array can be converted in place.
__global__ __launch_bounds__(256) void floor_to_f32(const double* __restrict__ th, int64_t rows, int64_t cols,
                                                    int64_t ldi, float* __restrict__ out, int64_t ldo) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
        const double v = th[r * ldi + c];
        float f = static_cast<float>(v);                       // round to nearest
        if (static_cast<double>(f) > v) f = nextafterf(f, -INFINITY);
        out[r * ldo + c] = f;
    }
}

template <typename T, typename TH>
__global__ __launch_bounds__(256) void exceed_bits(const T* __restrict__ ts, int64_t Tn, int64_t C, int64_t ld,
                                                   const TH* __restrict__ thresh, int64_t ldt,
                                                   const int32_t* __restrict__ row_of_t, int32_t negate,
                                                   uint64_t* __restrict__ bits, int64_t ldb, int64_t words_per_block) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int64_t W = (Tn + 63) / 64;
    const int64_t w0 = static_cast<int64_t>(blockIdx.y) * words_per_block;
    const int64_t w1 = w0 + words_per_block < W ? w0 + words_per_block : W;
    for (int64_t w = w0; w < w1; ++w) {
        uint64_t word = 0;
        for (int j0 = 0; j0 < 64; j0 += kBitsLoadAhead) {
            T xs[kBitsLoadAhead];
            TH ths[kBitsLoadAhead];
#pragma unroll
            for (int u = 0; u < kBitsLoadAhead; ++u) {
                const int64_t t = w * 64 + j0 + u;
                const int64_t tt = t < Tn ? t : Tn - 1;
                xs[u] = ts[tt * ld + c];
                ths[u] = thresh[static_cast<int64_t>(row_of_t[tt]) * ldt + c];
            }
#pragma unroll
            for (int u = 0; u < kBitsLoadAhead; ++u) {
                const int64_t t = w * 64 + j0 + u;
                T x = xs[u];
                if (negate) x = -x;
                const bool b = t < Tn && static_cast<TH>(x) > ths[u];   // NaN on either side -> false
                word |= static_cast<uint64_t>(b) << (j0 + u);
            }
        }
        bits[w * ldb + c] = word;
    }
}


// ---------------------------------------------------------------------------
// exceed_bits_tiled: the same bits with every threshold read ONCE per cell.  The host cuts the time
// axis into chunks of consecutive steps whose climatology rows are consecutive and lie in one tile
// of TILE rows (a calendar year gives one chunk per tile, two around a missing Feb 29) and groups
// them by tile.  A thread keeps the TILE thresholds of its cell in registers (static indices: bit i
// of the unrolled loop <-> row tile*TILE + i) and streams over the chunks of that tile - one visit
// per year - so that the series is read once and the (D, C) thresholds once, instead of once per
// step.  A chunk's bits are shifted to their place in the 64-step words and OR-ed in; only this
// thread touches column c, the buffer is zeroed first.
// ---------------------------------------------------------------------------
template <typename T, typename TH, int TILE>
__global__ __launch_bounds__(128) void exceed_bits_tiled(const T* __restrict__ ts, int64_t C, int64_t ld,
                                                         const TH* __restrict__ thresh, int64_t ldt, int64_t D,
                                                         const int32_t* __restrict__ tile_begin, int32_t ntiles,
                                                         const int32_t* __restrict__ chunk_t0,
                                                         const int32_t* __restrict__ chunk_i0,
                                                         const int32_t* __restrict__ chunk_n, int32_t negate,
                                                         uint64_t* __restrict__ bits, int64_t ldb) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= C) return;
    for (int32_t k = 0; k < ntiles; ++k) {
        const int32_t cb = tile_begin[k], ce = tile_begin[k + 1];
        if (cb == ce) continue;
        TH thr[TILE];
#pragma unroll
        for (int i = 0; i < TILE; ++i) {
            const int64_t r = static_cast<int64_t>(k) * TILE + i;
            thr[i] = r < D ? thresh[r * ldt + c] : static_cast<TH>(0);
        }
        for (int32_t q = cb; q < ce; ++q) {
            const int64_t t0 = chunk_t0[q];
            const int i0 = chunk_i0[q], n = chunk_n[q];          // wave-uniform
            uint64_t m = 0;
            // loads are unconditional (index clamped into the chunk) so that a whole group is in
            // flight before the first compare; positions outside the chunk are masked afterwards
            constexpr int G = 32;
#pragma unroll
            for (int g = 0; g < TILE; g += G) {
                T xs[G];
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    int d = g + u - i0;
                    d = d < 0 ? 0 : (d >= n ? n - 1 : d);
                    xs[u] = ts[(t0 + d) * ld + c];
                }
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const int i = g + u;
                    T x = xs[u];
                    if (negate) x = -x;
                    const bool in = i >= i0 && i < i0 + n;
                    m |= static_cast<uint64_t>(in && static_cast<TH>(x) > thr[i]) << ((i - i0) & 63);
                }
            }
            const int pos = static_cast<int>(t0 & 63);
            const int64_t w = t0 >> 6;
            bits[w * ldb + c] |= m << pos;
            if (pos + n > 64) bits[(w + 1) * ldb + c] |= m >> (64 - pos);
        }
    }
}

template <typename T, typename TH, int TILE>
hipError_t launch_exceed_bits_tiled(const T* ts, int64_t C, int64_t ld, const TH* thresh, int64_t ldt, int64_t D,
                                    const int32_t* tile_begin, int32_t ntiles, const int32_t* chunk_t0,
                                    const int32_t* chunk_i0, const int32_t* chunk_n, int32_t negate, uint64_t* bits,
                                    int64_t ldb, hipStream_t stream) {
    if (C <= 0 || ntiles <= 0) return hipSuccess;
    hipLaunchKernelGGL((exceed_bits_tiled<T, TH, TILE>), dim3(static_cast<unsigned>((C + 127) / 128)), dim3(128), 0,
                       stream, ts, C, ld, thresh, ldt, D, tile_begin, ntiles, chunk_t0, chunk_i0, chunk_n, negate,
                       bits, ldb);
    return hipGetLastError();
}
template hipError_t launch_exceed_bits_tiled<float, float, 64>(const float*, int64_t, int64_t, const float*, int64_t,
                                                               int64_t, const int32_t*, int32_t, const int32_t*,
                                                               const int32_t*, const int32_t*, int32_t, uint64_t*,
                                                               int64_t, hipStream_t);
template hipError_t launch_exceed_bits_tiled<double, double, 32>(const double*, int64_t, int64_t, const double*,
                                                                 int64_t, int64_t, const int32_t*, int32_t,
                                                                 const int32_t*, const int32_t*, const int32_t*,
                                                                 int32_t, uint64_t*, int64_t, hipStream_t);

// ---------------------------------------------------------------------------
// State of mhw_filter() + join_gaps() while walking the runs of one cell.
// A qualified run [s, e] (length test of identify.py:445-449 with the fillna(0) quirk: a run that
// begins at step 0 has p = 0, label 1, and loses its first step) either extends the pending
// event (gap to the previous qualified run <= maxGap) or closes it and opens a new one.
// ---------------------------------------------------------------------------
struct EventWalk {
    int64_t count = 0;
    bool have = false;            // a pending (not yet emitted) event
    int64_t first = 0, last = 0;  // its label (= first labelled step) and end step
    double* rows = nullptr;       // fill mode: the cell's slice of the table
    int64_t nmax = 0;
    int64_t cell = 0;

    __device__ __forceinline__ void emit() {
        if (rows && count < nmax) {
            double* r = rows + count * kEventColumns;
            r[0] = static_cast<double>(first);
            r[1] = static_cast<double>(cell);
            r[3] = static_cast<double>(first);
            r[4] = static_cast<double>(last);
        }
        ++count;
    }
    __device__ __forceinline__ void run(int64_t s, int64_t e, int32_t min_duration, int32_t join_gaps,
                                        int32_t max_gap) {
        const int64_t p = s > 0 ? s - 1 : 0;
        if (e - p < min_duration) return;
        const int64_t S = p + 1;
        if (have && join_gaps && S - last <= max_gap + 1) {
            last = e;
            return;
        }
        if (have) emit();
        have = true;
        first = S;
        last = e;
    }
    __device__ __forceinline__ void finish() {
        if (have) emit();
        have = false;
    }
};

__global__ __launch_bounds__(256) void events_from_bits(const uint64_t* __restrict__ bits, int64_t Tn, int64_t C,
                                                        int64_t ldb, int32_t min_duration, int32_t join_gaps,
                                                        int32_t max_gap, const int64_t* __restrict__ offsets,
                                                        int32_t* __restrict__ nevents, double* __restrict__ table) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int64_t W = (Tn + 63) / 64;
    EventWalk ew;
    ew.cell = c;
    if (offsets) {
        ew.rows = table + offsets[c] * kEventColumns;
        ew.nmax = offsets[c + 1] - offsets[c];
    }
    bool in_run = false;
    int64_t s = 0;
    // Morphological opening by min_duration before the walk: runs shorter than min_duration can
    // never qualify (identify.py:445-449), and white-noise exceedances are mostly such runs.
    // eroded[t] = AND_{j<m} x[t+j] (needs the next word), opened[t] = OR_{j<m} eroded[t-j] (needs the
    // previous eroded word): every run of >= m ones survives unchanged, every shorter run vanishes.
    const int m = min_duration <= 64 ? min_duration : 1;
    constexpr int U = 4;
    uint64_t cur = bits[c];
    uint64_t er_prev = 0;
    for (int64_t w0 = 0; w0 < W; w0 += U) {
        uint64_t ws[U + 1];
        ws[0] = cur;
#pragma unroll
        for (int u = 1; u <= U; ++u) ws[u] = w0 + u < W ? bits[(w0 + u) * ldb + c] : 0;
        cur = ws[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (w0 + u >= W) break;
            const int64_t base = (w0 + u) * 64;
            uint64_t er = ws[u];
            for (int j = 1; j < m; ++j) er &= (ws[u] >> j) | (ws[u + 1] << (64 - j));
            uint64_t x = er;
            for (int j = 1; j < m; ++j) x |= (er << j) | (er_prev >> (64 - j));
            er_prev = er;
            int pos = 0;
            while (pos < 64) {
                const uint64_t rest = x >> pos;
                if (in_run) {
                    const uint64_t z = ~rest;                       // zeros of the remaining bits
                    if (z == 0) break;                              // pos == 0, all ones: continues
                    const int k = __builtin_ctzll(z);               // ones from pos on
                    if (pos + k >= 64) break;                       // the run continues into the next word
                    ew.run(s, base + pos + k - 1, min_duration, join_gaps, max_gap);
                    in_run = false;
                    pos += k;
                } else {
                    if (rest == 0) break;
                    const int k = __builtin_ctzll(rest);
                    s = base + pos + k;
                    in_run = true;
                    pos += k;
                }
            }
        }
    }
    if (in_run) ew.run(s, Tn - 1, min_duration, join_gaps, max_gap);   // bits beyond T-1 are zero
    ew.finish();
    if (nevents) nevents[c] = static_cast<int32_t>(ew.count);
}

template <typename T>
__global__ __launch_bounds__(256) void event_stats_sparse(const T* __restrict__ ts, int64_t Tn, int64_t ld,
                                                          const double* __restrict__ seas,
                                                          const double* __restrict__ thresh, int64_t ldc,
                                                          const int32_t* __restrict__ row_of_t, int32_t negate,
                                                          int64_t n_events, double* __restrict__ table) {
    const int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (e >= n_events) return;
    double* row = table + e * kEventColumns;
    const int64_t c = static_cast<int64_t>(row[1]);
    const int64_t first = static_cast<int64_t>(row[3]), last = static_cast<int64_t>(row[4]);
    EventAcc a;
    a.reset(static_cast<int32_t>(row[0]), first);
    double anom_prev = make_nan();
    if (first > 0) {
        const int64_t t = first - 1;
        double x = static_cast<double>(ts[t * ld + c]);
        if (negate) x = -x;
        anom_prev = x - seas[static_cast<int64_t>(row_of_t[t]) * ldc + c];
    }
    for (int64_t t = first; t <= last; ++t) {
        const int64_t r = row_of_t[t];
        double x = static_cast<double>(ts[t * ld + c]);
        if (negate) x = -x;
        const double se = seas[r * ldc + c], th = thresh[r * ldc + c];
        const double anom = x - se;
        if (t > first && anom == anom) a.anom_last = anom;      // anom_minus of the previous labelled step
        a.last = t;
        if (!a.have_afirst && anom_prev == anom_prev) { a.anom_first = anom_prev; a.have_afirst = true; }
        event_add_step(a, t, x, se, th);
        anom_prev = anom;
    }
    if (last + 1 < Tn) {
        const int64_t t = last + 1;
        double x = static_cast<double>(ts[t * ld + c]);
        if (negate) x = -x;
        const double anom = x - seas[static_cast<int64_t>(row_of_t[t]) * ldc + c];
        if (anom == anom) a.anom_last = anom;
    }
    flush_event(a, Tn - 1, row);
}

hipError_t launch_floor_to_f32(const double* th, int64_t rows, int64_t cols, int64_t ldi, float* out, int64_t ldo,
                               hipStream_t stream) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    hipLaunchKernelGGL(floor_to_f32, dim3(static_cast<unsigned>((cols + 255) / 256), static_cast<unsigned>(rows < 65535 ? rows : 65535)),
                       dim3(256), 0, stream, th, rows, cols, ldi, out, ldo);
    return hipGetLastError();
}

template <typename T, typename TH>
hipError_t launch_exceed_bits(const T* ts, int64_t Tn, int64_t C, int64_t ld, const TH* thresh, int64_t ldt,
                              const int32_t* row_of_t, int32_t negate, uint64_t* bits, int64_t ldb,
                              hipStream_t stream) {
    if (C <= 0 || Tn <= 0) return hipSuccess;
    const int64_t W = (Tn + 63) / 64;
    const int64_t bx = (C + 255) / 256;
    // enough blocks to fill the chip even for small grids: words are independent
    int64_t by = (4096 + bx - 1) / bx;
    if (by > W) by = W;
    if (by < 1) by = 1;
    const int64_t wpb = (W + by - 1) / by;
    by = (W + wpb - 1) / wpb;
    hipLaunchKernelGGL((exceed_bits<T, TH>), dim3(static_cast<unsigned>(bx), static_cast<unsigned>(by)), dim3(256), 0,
                       stream, ts, Tn, C, ld, thresh, ldt, row_of_t, negate, bits, ldb, wpb);
    return hipGetLastError();
}

hipError_t launch_events_from_bits(const uint64_t* bits, int64_t Tn, int64_t C, int64_t ldb, int32_t min_duration,
                                   int32_t join_gaps, int32_t max_gap, const int64_t* offsets, int32_t* nevents,
                                   double* table, hipStream_t stream) {
    if (C <= 0 || Tn <= 0) return hipSuccess;
    hipLaunchKernelGGL(events_from_bits, dim3(static_cast<unsigned>((C + 255) / 256)), dim3(256), 0, stream, bits,
                       Tn, C, ldb, min_duration, join_gaps, max_gap, offsets, nevents, table);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_event_stats_sparse(const T* ts, int64_t Tn, int64_t ld, const double* seas, const double* thresh,
                                     int64_t ldc, const int32_t* row_of_t, int32_t negate, int64_t n_events,
                                     double* table, hipStream_t stream) {
    if (n_events <= 0 || Tn <= 0) return hipSuccess;
    hipLaunchKernelGGL(event_stats_sparse<T>, dim3(static_cast<unsigned>((n_events + 255) / 256)), dim3(256), 0,
                       stream, ts, Tn, ld, seas, thresh, ldc, row_of_t, negate, n_events, table);
    return hipGetLastError();
}

template hipError_t launch_exceed_bits<float, float>(const float*, int64_t, int64_t, int64_t, const float*, int64_t,
                                                     const int32_t*, int32_t, uint64_t*, int64_t, hipStream_t);
template hipError_t launch_exceed_bits<float, double>(const float*, int64_t, int64_t, int64_t, const double*, int64_t,
                                                      const int32_t*, int32_t, uint64_t*, int64_t, hipStream_t);
template hipError_t launch_exceed_bits<double, double>(const double*, int64_t, int64_t, int64_t, const double*,
                                                       int64_t, const int32_t*, int32_t, uint64_t*, int64_t,
                                                       hipStream_t);
template hipError_t launch_event_stats_sparse<float>(const float*, int64_t, int64_t, const double*, const double*,
                                                     int64_t, const int32_t*, int32_t, int64_t, double*,
                                                     hipStream_t);
template hipError_t launch_event_stats_sparse<double>(const double*, int64_t, int64_t, const double*, const double*,
                                                      int64_t, const int32_t*, int32_t, int64_t, double*,
                                                      hipStream_t);

// ---------------------------------------------------------------------------
// offsets_from_counts: exclusive prefix sum of the per-cell event counts (int32) into int64 table
// offsets, on the device -- the count pass and the fill pass of events_from_bits no longer meet on
// the host.  Three small launches: sums of 1024-element blocks, a scan of the block sums by one
// workgroup, the scan inside every block.
// ---------------------------------------------------------------------------
namespace {
constexpr int kScanBlock = 1024;

__device__ __forceinline__ int64_t wave_incl_scan(int64_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
// inclusive scan over the 1024 threads of a workgroup; returns the thread's inclusive value, *total = block sum
__device__ __forceinline__ int64_t block_incl_scan(int64_t v, int64_t* total) {
    __shared__ int64_t wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t inc = wave_incl_scan(v, lane);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        int64_t w = lane < 16 ? wsum[lane] : 0;
        w = wave_incl_scan(w, lane);
        if (lane < 16) wsum[lane] = w;
    }
    __syncthreads();
    const int64_t base = wave ? wsum[wave - 1] : 0;
    *total = wsum[15];
    __syncthreads();
    return inc + base;
}

__global__ __launch_bounds__(kScanBlock) void scan_block_sums(const int32_t* __restrict__ counts, int64_t n,
                                                             int64_t* __restrict__ sums) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kScanBlock + threadIdx.x;
    int64_t total;
    block_incl_scan(i < n ? counts[i] : 0, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanBlock) void scan_of_sums(int64_t* __restrict__ sums, int64_t nblocks) {
    // one workgroup: exclusive scan in place, sums[nblocks] = grand total
    int64_t carry = 0;
    for (int64_t b0 = 0; b0 < nblocks; b0 += kScanBlock) {
        const int64_t i = b0 + threadIdx.x;
        const int64_t v = i < nblocks ? sums[i] : 0;
        int64_t total;
        const int64_t inc = block_incl_scan(v, &total);
        if (i < nblocks) sums[i] = carry + inc - v;
        carry += total;
    }
    if (threadIdx.x == 0) sums[nblocks] = carry;
}
__global__ __launch_bounds__(kScanBlock) void scan_apply(const int32_t* __restrict__ counts, int64_t n,
                                                        const int64_t* __restrict__ sums, int64_t nblocks,
                                                        int64_t* __restrict__ offsets) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kScanBlock + threadIdx.x;
    const int64_t v = i < n ? counts[i] : 0;
    int64_t total;
    const int64_t inc = block_incl_scan(v, &total);
    if (i < n) offsets[i] = sums[blockIdx.x] + inc - v;
    if (i == 0) offsets[n] = sums[nblocks];
}
}  // namespace

hipError_t launch_offsets_from_counts(const int32_t* counts, int64_t n, int64_t* offsets, int64_t* block_sums,
                                      hipStream_t stream) {
    const int64_t nblocks = (n + kScanBlock - 1) / kScanBlock;
    if (n == 0) return hipMemsetAsync(offsets, 0, sizeof(int64_t), stream);
    hipLaunchKernelGGL(scan_block_sums, dim3(static_cast<unsigned>(nblocks)), dim3(kScanBlock), 0, stream, counts, n, block_sums);
    hipLaunchKernelGGL(scan_of_sums, dim3(1), dim3(kScanBlock), 0, stream, block_sums, nblocks);
    hipLaunchKernelGGL(scan_apply, dim3(static_cast<unsigned>(nblocks)), dim3(kScanBlock), 0, stream, counts, n, block_sums,
                       nblocks, offsets);
    return hipGetLastError();
}

}  // namespace xmhw
