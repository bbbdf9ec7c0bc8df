// kernels_redo.hip -- the cell-rows the sorted-list kernel (kernels_sorted.hip) could not settle, recomputed exactly.
//
// The sorted kernel flags a cell-row in a bitmap (bits[row * ldb + (cell >> 5)], bit cell & 31) when a row-list was too
// short for the row; its `seas` is right (sums do not depend on the lists), its `thresh` is not.  Three launches, no
// host round trip:
//   redo_collect   wave per 32 cells: a cell's set bits become consecutive entries (row, cell) of a work list (one atomic
//                  per cell reserves them) and are cleared; bits that do not fit the list stay set;
//   redo_run       ONE WAVE per entry, a fixed grid striding over the list: the wave loads the row's pool (the samples
//                  at centre +- w of every centre of the row: window_roll(), identify.py:184-209) -- nine keys per
//                  lane --, starts from the sorted kernel's own (wrong, but close) answer, counts the keys below it and
//                  steps from key to neighbouring key until order statistic lo is reached (wave-wide counts by ballot +
//                  popcount: the counts and every decision are scalar); numpy's linear interpolation
//                  (identify.py:233-235);
//   clim_generic_flagged (kernels_generic.hip) on whatever is still set: only when the list overflowed.
#include "device_common.h"
#include "kernels.h"
#include "packed_src.h"

namespace xmhw {

// ONE WAVE per bitmap word column (32 cells), lane = row: the flagged rows of a cell become consecutive entries of the work
// list, rows ascending (one atomic per flagged cell reserves them), and their bits are cleared; bits that do not fit the
// list stay set.  A wave that takes a piece of the list in redo_run then meets a cell's neighbouring rows back to back.
// (thread per cell, every thread scanning its column, was 0.66 ms of configs[2]'s step: every wave had a flagged cell)
__global__ __launch_bounds__(256) void redo_collect(uint32_t* __restrict__ bits, int64_t C, int32_t D, int64_t ldb,
                                                    unsigned long long* __restrict__ list, uint32_t* __restrict__ count,
                                                    uint32_t cap) {
    const int64_t wd = static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wd >= ldb) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    // -- pass 1: lane b < 32 counts the flagged rows of cell 32 wd + b
    uint32_t cnt = 0;
    for (int32_t r0 = 0; r0 < D; r0 += 64) {
        const int32_t r = r0 + lane;
        const uint32_t word = r < D ? bits[static_cast<int64_t>(r) * ldb + wd] : 0u;
        if (__builtin_amdgcn_ballot_w64(word != 0u) == 0ull) continue;
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(((word >> b) & 1u) != 0u);
            cnt += lane == b ? static_cast<uint32_t>(__builtin_popcountll(m)) : 0u;
        }
    }
    if (__builtin_amdgcn_ballot_w64(cnt != 0u) == 0ull) return;
    uint32_t next = cnt != 0u ? atomicAdd(count, cnt) : 0u;       // lane b: where cell b's next entry goes
    // -- pass 2: the entries
    for (int32_t r0 = 0; r0 < D; r0 += 64) {
        const int32_t r = r0 + lane;
        const uint32_t word = r < D ? bits[static_cast<int64_t>(r) * ldb + wd] : 0u;
        if (__builtin_amdgcn_ballot_w64(word != 0u) == 0ull) continue;
        uint32_t left = word;
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            const bool mine = ((word >> b) & 1u) != 0u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
            if (m == 0ull) continue;
            const uint32_t base = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(next), b));
            const uint32_t pos = base + static_cast<uint32_t>(__builtin_popcountll(m & below));
            if (mine && pos < cap) {
                list[pos] = (static_cast<unsigned long long>(r) << 40) | static_cast<unsigned long long>(32 * wd + b);
                left &= ~(1u << b);
            }
            next += lane == b ? static_cast<uint32_t>(__builtin_popcountll(m)) : 0u;
        }
        if (left != word) bits[static_cast<int64_t>(r) * ldb + wd] = left;
    }
}

// wave-wide helpers: every lane ends with the result
template <typename K>
__device__ __forceinline__ K wave_min(K v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const K o = static_cast<K>(__shfl_xor(static_cast<unsigned long long>(v), off, 64));
        v = o < v ? o : v;
    }
    return v;
}
template <typename K>
__device__ __forceinline__ K wave_max(K v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const K o = static_cast<K>(__shfl_xor(static_cast<unsigned long long>(v), off, 64));
        v = o > v ? o : v;
    }
    return v;
}

// one row of a run: order statistic lo of the keys whose place in the stretch is k .. k + R - 1
template <typename Src, int KPL>
__device__ __forceinline__ void redo_one(const Src& src, const typename KeyOf<typename Src::sample>::type (&all)[KPL],
                                         const int32_t (&joff)[KPL], int32_t k, int32_t R, int32_t row, int64_t c, double q,
                                         int negate, double* __restrict__ thresh, int64_t ldo, int lane) {
    using T = typename Src::sample;
    using K = typename KeyOf<T>::type;
    K key[KPL];
#pragma unroll
    for (int i = 0; i < KPL; ++i) key[i] = (joff[i] >= k && joff[i] < k + R) ? all[i] : static_cast<K>(0);
    uint32_t n = 0;
#pragma unroll
    for (int i = 0; i < KPL; ++i) n += static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(key[i] != 0)));
    double th = make_nan();
    if (n > 0) {
        const double vi = static_cast<double>(n - 1) * q;
        const double fl = floor(vi);
        const uint32_t lo = static_cast<uint32_t>(fl);
        const double g = vi - fl;
        auto count_lt = [&](K v) -> uint32_t {       // valid keys below v
            uint32_t cnt = 0;
#pragma unroll
            for (int i = 0; i < KPL; ++i)
                cnt += static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(key[i] != 0 && key[i] < v)));
            return cnt;
        };
        auto count_eq = [&](K v) -> uint32_t {
            uint32_t cnt = 0;
#pragma unroll
            for (int i = 0; i < KPL; ++i)
                cnt += static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(key[i] == v)));
            return cnt;
        };
        auto next_above = [&](K v) -> K {            // the smallest key above v (all ones if none)
            K m = ~static_cast<K>(0);
#pragma unroll
            for (int i = 0; i < KPL; ++i) m = (key[i] > v && key[i] < m) ? key[i] : m;
            return wave_min<K>(m);
        };
        auto next_below = [&](K v) -> K {            // the largest valid key below v (0 if none)
            K m = 0;
#pragma unroll
            for (int i = 0; i < KPL; ++i) m = (key[i] < v && key[i] > m) ? key[i] : m;
            return wave_max<K>(m);
        };
        // -- from the sorted kernel's answer: a key near order statistic lo
        const double guess = thresh[static_cast<int64_t>(row) * ldo + c];
        K v = guess == guess ? KeyOf<T>::key(src.guess(guess)) : static_cast<K>(0);
        bool found = false;
        uint32_t cl = 0, ev = 0;                     // keys below v, keys equal to v
        if (v != 0) {
            cl = count_lt(v);
            ev = count_eq(v);
            for (int it = 0; it < 24 && !found; ++it) {
                if (cl <= lo && lo < cl + ev) {
                    found = true;
                } else if (lo < cl) {
                    v = next_below(v);
                    if (v == 0) break;
                    ev = count_eq(v);
                    cl -= ev;
                } else {
                    const K nv = next_above(v);
                    if (nv == ~static_cast<K>(0)) break;
                    cl += ev;
                    v = nv;
                    ev = count_eq(v);
                }
            }
        }
        if (!found) {
            // the whole key, bit by bit: largest v with #{valid keys < v} <= lo   (key 0 = invalid: (0 - 1) wraps high)
            v = 0;
            for (int bit = KeyOf<T>::bits - 1; bit >= 0; --bit) {
                const K cand = v | (static_cast<K>(1) << bit);
                uint32_t cnt = 0;
#pragma unroll
                for (int i = 0; i < KPL; ++i)
                    cnt += static_cast<uint32_t>(__builtin_popcountll(
                        __builtin_amdgcn_ballot_w64(static_cast<K>(key[i] - 1) < static_cast<K>(cand - 1))));
                if (cnt <= lo) v = cand;
            }
            cl = count_lt(v);
            ev = count_eq(v);
        }
        // v = key of a[lo]; a[lo + 1]: v again if it is duplicated past lo, else the smallest key above v
        K vhi = v;
        if (lo + 1 < n && lo + 1 >= cl + ev) vhi = next_above(v);
        th = numpy_lerp(src.value(static_cast<T>(KeyOf<T>::value(v))), src.value(static_cast<T>(KeyOf<T>::value(vhi))), g);
    }
    (void)negate;
    if (lane == 0) thresh[static_cast<int64_t>(row) * ldo + c] = th;
}

// W = the window half width as a compile-time constant (5: the default, every plan the sorted kernel serves) or 0 = taken
// from the argument.  The selection starts from the answer the sorted kernel left in `thresh` (wrong, but a few ranks
// away at most): count the keys below it, then step from key to neighbouring key until order statistic lo is reached;
// the bit-by-bit descent over the whole key is kept for guesses that turn out to be far off.
template <typename Src, int W>
__global__ __launch_bounds__(256) void redo_run(Src src, int64_t Tn, int64_t ld,
                                                const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ centres,
                                                int32_t w_arg, double q, int negate, double* __restrict__ thresh, int64_t ldo,
                                                const unsigned long long* __restrict__ list,
                                                const uint32_t* __restrict__ count, uint32_t cap) {
    using T = typename Src::sample;
    using K = typename KeyOf<T>::type;
    constexpr int KPL = 10;                     // keys per lane: up to 640 samples (48 tracks x 11 = 528; 40 tracks x 16)
    const int lane = threadIdx.x & 63;
    const uint32_t nent = min(*count, cap);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const int32_t w = W > 0 ? W : w_arg;
    const int32_t R = 2 * w + 1;
    // (a wave takes a contiguous piece of the list: the entries of a cell are neighbours in it)
    const uint32_t per = (nent + nwaves - 1) / nwaves;
    const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t e_end = min((wid + 1) * per, nent);
    uint32_t e = wid * per;
    while (e < e_end) {
        const unsigned long long ent = list[e];
        const int32_t row0 = static_cast<int32_t>(ent >> 40);
        const int64_t c = static_cast<int64_t>(ent & ((1ull << 40) - 1ull));
        const int32_t cb = row_ptr[row0], ce = row_ptr[row0 + 1];
        const int32_t ntr = ce - cb;
        // (pools beyond KPL * 64 samples do not occur on plans the sorted kernel serves; such an entry is left alone)
        if (ntr * R > KPL * 64 || ntr <= 0) { ++e; continue; }
        // -- a RUN: the entries that follow, as long as they are the same cell's next rows and every centre of the row is
        //    the centre before plus one (ice edges, steep seasons: a cell's flagged rows come in runs).  The pools of a run
        //    of L rows are windows of ONE set of ntr x (R + L - 1) samples, loaded once: a flagged row costs the memory
        //    (R + L - 1) / (R L) of what it costs alone.
        const int32_t lmax = ntr <= 64 ? (KPL * 64) / ntr - R + 1 : 1;
        int32_t L = 1;
        while (L < lmax && e + L < e_end) {
            const unsigned long long nx = list[e + L];
            if (static_cast<int64_t>(nx & ((1ull << 40) - 1ull)) != c || static_cast<int32_t>(nx >> 40) != row0 + L) break;
            const int32_t cbk = row_ptr[row0 + L];
            if (row_ptr[row0 + L + 1] - cbk != ntr) break;
            const bool same = lane >= ntr || centres[cbk + lane] == centres[cb + lane] + L;
            if (__builtin_amdgcn_ballot_w64(same) != ~0ull) break;
            ++L;
        }
        const int32_t RL = R + L - 1;
        const int32_t nload = ntr * RL;
        K key[KPL];
        int32_t joff[KPL];                      // the sample's place in its track's stretch: row k of the run owns k .. k + R - 1
        {
            // (branch-free, in three sweeps -- centre indices, centres, samples -- so that the loads of a sweep are in
            // flight together: a wave pays two memory latencies per run instead of twenty)
            int32_t cen[KPL];
            bool inb[KPL];
#pragma unroll
            for (int i = 0; i < KPL; ++i) {
                const int32_t p = lane + 64 * i;
                inb[i] = p < nload;
                const int32_t ci = inb[i] ? p / RL : 0;
                joff[i] = inb[i] ? p - ci * RL : -1000000;
                cen[i] = centres[cb + ci];
            }
            T val[KPL];
            bool ok[KPL];
#pragma unroll
            for (int i = 0; i < KPL; ++i) {
                const int64_t t = static_cast<int64_t>(cen[i]) + joff[i] - w;
                ok[i] = inb[i] && t >= 0 && t < Tn;
                const int64_t tc = t < 0 ? 0 : (t >= Tn ? Tn - 1 : t);
                val[i] = src.at(c + tc * ld);
            }
#pragma unroll
            for (int i = 0; i < KPL; ++i) {
                T v = val[i];
                if (negate) v = -v;
                key[i] = ok[i] ? KeyOf<T>::key(v) : static_cast<K>(0);
            }
        }
        for (int32_t k = 0; k < L; ++k) redo_one<Src, KPL>(src, key, joff, k, R, row0 + k, c, q, negate, thresh, ldo, lane);
        e += static_cast<uint32_t>(L);
    }
}

namespace {
template <typename Src>
hipError_t launch_redo_src(Src src, int64_t Tn, int64_t C, int64_t ld, const int32_t* row_ptr, const int32_t* centres, int32_t D,
                           int32_t w, double q, int negate, double* thresh, int64_t ldo, uint32_t* bits, int64_t ldb,
                           unsigned long long* list, uint32_t* count, uint32_t cap, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(redo_collect, dim3(static_cast<unsigned>((ldb + 3) / 4)), dim3(256), 0, stream, bits, C, D, ldb, list,
                       count, cap);
    if (w == 5)
        hipLaunchKernelGGL((redo_run<Src, 5>), dim3(2048), dim3(256), 0, stream, src, Tn, ld, row_ptr, centres, w, q, negate,
                           thresh, ldo, list, count, cap);
    else
        hipLaunchKernelGGL((redo_run<Src, 0>), dim3(2048), dim3(256), 0, stream, src, Tn, ld, row_ptr, centres, w, q, negate,
                           thresh, ldo, list, count, cap);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_redo(const float* ts, int64_t Tn, int64_t C, int64_t ld, const int32_t* row_ptr, const int32_t* centres,
                       int32_t D, int32_t w, double q, int negate, double* thresh, double* seas, int64_t ldo,
                       uint32_t* bits, int64_t ldb, unsigned long long* list, uint32_t* count, uint32_t cap,
                       hipStream_t stream) {
    if (C <= 0 || D <= 0) return hipSuccess;
    hipError_t e = launch_redo_src(PlainSrc<float>{ts}, Tn, C, ld, row_ptr, centres, D, w, q, negate, thresh, ldo, bits, ldb,
                                   list, count, cap, stream);
    if (e != hipSuccess) return e;
    // whatever did not fit the list (bits still set): the thread-per-cell-row kernel
    return launch_generic_flagged<float>(ts, Tn, C, ld, row_ptr, centres, 0, D, w, q, negate, thresh, seas, ldo, bits, ldb,
                                         stream);
}

// the same on int16 codes read in place (negate: what the kernels negate -- pk.key_neg in mode 2)
hipError_t launch_redo_packed(const int16_t* codes, const PackedI16& pk, int64_t Tn, int64_t C, int64_t ld,
                              const int32_t* row_ptr, const int32_t* centres, int32_t D, int32_t w, double q, int negate,
                              double* thresh, double* seas, int64_t ldo, uint32_t* bits, int64_t ldb,
                              unsigned long long* list, uint32_t* count, uint32_t cap, hipStream_t stream) {
    if (C <= 0 || D <= 0) return hipSuccess;
    hipError_t e = launch_redo_src(PackedSrc{codes, pk}, Tn, C, ld, row_ptr, centres, D, w, q, negate, thresh, ldo, bits, ldb,
                                   list, count, cap, stream);
    if (e != hipSuccess) return e;
    return launch_generic_flagged_packed(codes, pk, Tn, C, ld, row_ptr, centres, D, w, q, negate, thresh, seas, ldo, bits, ldb,
                                         stream);
}

}  // namespace xmhw
