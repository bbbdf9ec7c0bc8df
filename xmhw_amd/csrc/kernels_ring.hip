// kernels_ring.hip -- the fast path: sliding-window percentile + mean with the
// pool held in VGPRs ("register ring"), hand-written for gfx950 wave64.
//
// Work decomposition
//   wave  = 8 cells x 8 "subs"; lane = (cell >> 1) * 16 + sub * 2 + (cell & 1), so the 8 subs of a
//   cell sit in one 16-lane DPP row at stride 2 and combine with three row_ror DPP steps (8, 4, 2).
//   A sub owns YPS tracks (years); 8*YPS >= ntracks.  For every owned track
//   the lane keeps the R = 2w+1 samples of that year's current window as
//   order-preserving 32-bit keys in VGPRs: ring[YPS][R].
//   The wave walks the rows (distinct doy labels) in ascending order.  Per
//   row each track PUSHes one new sample into ring slot (step mod R) -- the
//   sample leaving the window sits exactly there -- so every input sample is
//   read from HBM once (plus 2w per track boundary), 32 contiguous bytes per
//   8 lanes, 128 B per 4-wave workgroup row.  No LDS; a wave is self-contained, the
//   only barrier is a rendezvous every 64 rows that keeps the four waves on the same
//   128-byte lines (see the end of the row loop).
//
// Per row, per cell
//   n      = number of valid pooled samples         (running per-track counts)
//   seas   = sum / n                                (running per-track f64 sums;
//                                                    exact for f32 input)
//   thresh = numpy linear quantile of the pool: needs order statistics lo and
//            lo+1.  Found by bracketing in key space: "count" passes give
//            F(p) = #{valid keys <= p} (v_cmp + v_addc per key); the bracket
//            (pl, F(pl) <= lo) / (ph, F(ph) > lo) starts from the previous
//            row's pivot, whose count is kept up to date from the pushed / evicted
//            keys alone, and closes by secant steps; once lo - F(pl) <= J - 2 one
//            "extract" pass returns the J = 5 smallest keys above the pivot
//            (v_sub, J-1 x v_med3, v_min per key; merged across the subs by a bitonic
//            network over DPP), which contain both order statistics; a repair round
//            handles tie-heavy data.
//   Selection is exact for every input (ties, NaN, +-inf); heuristics only
//   choose probe points.
//
// Reference semantics restated: window_roll() (identify.py:184-209),
// calculate_thresh()/calculate_seas() without the Feb-29 step
// (identify.py:233-235, :263), coldSpells negation (xmhw.py:153-154).
#include "device_common.h"
#include "kernels.h"
#include "plan.h"

namespace xmhw {

constexpr int kWavesPerBlock = 4;

// Lane map (SUBS = 8): lane = (c >> 1) * 16 + sub * 2 + (c & 1), c = cell in wave (0..7).
// The 8 subs of a cell sit in ONE 16-lane DPP row at stride 2, so the per-cell
// all-reduce is three full-rate DPP row rotations (row_ror 8, 4, 2) instead of
// LDS-pipe ds_bpermute (measured 24 cycles each vs 4.5, tools/ubench_valu.hip).
// A ts row is still read as 8 x 4 = 32 contiguous bytes per wave.
// SUBS = 16 (records of 49..96 tracks): a cell owns a whole DPP row, lane = c * 16 + sub with 4 cells
// per wave, one more rotation (row_ror 1) per all-reduce; the per-cell logic then serves half as
// many cells per instruction.  SUBS = 32 (97..192 tracks): two DPP rows per cell, 2 cells per wave,
// the rows meet through one ds_swizzle (lane ^ 16) per all-reduce.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
constexpr int kRor8 = 0x128, kRor4 = 0x124, kRor2 = 0x122, kRor1 = 0x121;
// partner lane in the other DPP row of a 32-lane half (lane ^ 16): ds_swizzle BITMASK_PERM, and 0x1F, xor 0x10
__device__ __forceinline__ uint32_t swap16(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), 0x401F));
}

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

template <int SUBS>
__device__ __forceinline__ uint32_t sub_sum(uint32_t v) {
    v += dpp_mov<kRor8>(v);
    v += dpp_mov<kRor4>(v);
    v += dpp_mov<kRor2>(v);
    if constexpr (SUBS >= 16) v += dpp_mov<kRor1>(v);
    if constexpr (SUBS == 32) v += swap16(v);
    return v;
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
    const uint32_t lo = dpp_mov<CTRL>(static_cast<uint32_t>(b));
    const uint32_t hi = dpp_mov<CTRL>(static_cast<uint32_t>(b >> 32));
    return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
}
template <int SUBS>
__device__ __forceinline__ double sub_sum(double v) {
    v += dpp_mov_f64<kRor8>(v);
    v += dpp_mov_f64<kRor4>(v);
    v += dpp_mov_f64<kRor2>(v);
    if constexpr (SUBS >= 16) v += dpp_mov_f64<kRor1>(v);
    if constexpr (SUBS == 32) {
        const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
        const uint32_t lo = swap16(static_cast<uint32_t>(b)), hi = swap16(static_cast<uint32_t>(b >> 32));
        v += __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
    }
    return v;
}
// merge two ascending pairs, keep the two smallest
__device__ __forceinline__ void min2_merge(uint32_t& a1, uint32_t& a2, uint32_t b1, uint32_t b2) {
    const uint32_t hi = umax(a1, b1);
    a1 = umin(a1, b1);
    a2 = umin(hi, umin(a2, b2));
}

__device__ __forceinline__ double key_value(uint32_t k) {
    return k ? static_cast<double>(key_f32(k)) : 0.0;
}

// J smallest (position-wise, ties repeated) distances above a pivot, ascending.
// insert(): one v_med3_u32 per rank + one v_min_u32.
template <int J>
struct TopJ {
    static_assert(J >= 2 && J <= 8, "merge network is built for 2..8 keys");
    uint32_t m[J];
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int i = 0; i < J; ++i) m[i] = 0xFFFFFFFFu;
    }
    __device__ __forceinline__ void insert(uint32_t d) {
#pragma unroll
        for (int i = J - 1; i >= 1; --i) m[i] = umed3(m[i - 1], m[i], d);
        m[0] = umin(m[0], d);
    }
    // keep the J smallest of my list and the partner lane's list: min(a[i], b[J-1-i]) holds
    // exactly those J keys as a bitonic sequence; a half-cleaner network sorts it again
    template <int CTRL>
    __device__ __forceinline__ void merge_dpp() {
        uint32_t b[J];
#pragma unroll
        for (int i = 0; i < J; ++i) b[i] = CTRL ? dpp_mov<CTRL ? CTRL : kRor1>(m[i]) : swap16(m[i]);   // CTRL 0: lane ^ 16
#pragma unroll
        for (int i = 0; i < J; ++i) m[i] = umin(m[i], b[J - 1 - i]);
        // the J keys form an up-down sequence; embedded at offset 8-J of an 8-key bitonic
        // merge network (virtual -inf in front) the compare-exchanges that touch the
        // padding are no-ops and are skipped statically (J=5: 5 compare-exchanges)
        constexpr int OFF = 8 - J;
#pragma unroll
        for (int d = 4; d >= 1; d >>= 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if ((k & d) == 0 && k >= OFF && k + d < 8) {
                    const uint32_t lo_ = umin(m[k - OFF], m[k - OFF + d]);
                    const uint32_t hi_ = umax(m[k - OFF], m[k - OFF + d]);
                    m[k - OFF] = lo_;
                    m[k - OFF + d] = hi_;
                }
            }
        }
    }
    template <int SUBS>
    __device__ __forceinline__ void sub_merge() {
        merge_dpp<kRor8>();
        merge_dpp<kRor4>();
        merge_dpp<kRor2>();
        if constexpr (SUBS >= 16) merge_dpp<kRor1>();
        if constexpr (SUBS == 32) merge_dpp<0>();
    }
    __device__ __forceinline__ uint32_t at(uint32_t j) const {  // m[j], j uniform per cell
        // the empty asm keeps the select chain a select chain (the compiler otherwise builds a dynamically
        // indexed array in LDS: 5,120 B per block and a dependent LDS round trip per row)
        uint32_t r = m[0];
#pragma unroll
        for (int i = 1; i < J; ++i) {
            r = (j == static_cast<uint32_t>(i)) ? m[i] : r;
            asm volatile("" : "+v"(r));
        }
        return r;
    }
    __device__ __forceinline__ uint32_t count_below(uint32_t d) const {  // #{m[i] < d}
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < J; ++i) c += (m[i] < d) ? 1u : 0u;
        return c;
    }
};

constexpr int kTopJ = 5;          // extraction width
constexpr int kCountBudget = 6;   // count passes before the first extraction

// TI = float, or double for float64 input whose values are all float32-representable (decoded
// int16 / float32 archives promoted by the reader): the samples are narrowed on load and the kernel
// runs at the float32 rate.  `narrow_flag` (TI = double only) is set as soon as a sample does not
// survive the round trip; the kernel stops, later blocks do not start, and the float64 kernel
// queued behind it (which runs only if the flag is set) recomputes everything.
template <int W, int YPS, typename TI, int SUBS>
__global__ __launch_bounds__(256) void clim_ring_f32(
    const TI* __restrict__ ts, int64_t C, int64_t ld, const uint32_t* __restrict__ table,
    int32_t step_min, const DevChunk* __restrict__ chunks, double q, int negate,
    double* __restrict__ thresh, double* __restrict__ seas, int64_t ldo,
    unsigned long long* __restrict__ stats, uint32_t* __restrict__ narrow_flag) {
    constexpr bool kNarrow = sizeof(TI) == 8;
    if constexpr (kNarrow) {
        if (*narrow_flag != 0) return;
    }
    bool lossy = false;
    constexpr int R = 2 * W + 1;
    static_assert(SUBS == 8 || SUBS == 16 || SUBS == 32, "lane maps exist for 8, 16 and 32 subs");
    constexpr int NTP = SUBS * YPS;
    constexpr int kCells = 64 / SUBS;   // cells per wave
    constexpr uint32_t NSLOT = static_cast<uint32_t>(NTP) * R;
    constexpr int J = kTopJ;
    constexpr uint32_t SLACK = J - 2;  // a[lo], a[lo+1] are among the J keys above pl iff lo - F(pl) <= J-2
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = SUBS == 8 ? (lane >> 1) & 7 : lane & (SUBS - 1);
    const int cw = SUBS == 8 ? (lane & 1) | ((lane >> 4) << 1) : lane / SUBS;
    const int64_t cell = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave) * kCells + cw;
    const bool cell_ok = cell < C;
    const DevChunk ch = chunks[blockIdx.y];
    const uint32_t* tab = table + sub * YPS;
    const TI* col = ts + (cell_ok ? cell : 0);
    const float fnan = __uint_as_float(0x7FC00000u);

    uint32_t ring[YPS][R];
    double tsum[YPS];
    uint32_t nval[YPS];
#pragma unroll
    for (int y = 0; y < YPS; ++y) {
        tsum[y] = 0.0;
        nval[y] = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) ring[y][k] = 0;
    }

    auto load_entries = [&](int32_t s, uint32_t (&e)[YPS]) {
        const uint32_t* p = tab + static_cast<int64_t>(s - step_min) * NTP;
#pragma unroll
        for (int y = 0; y < YPS; ++y) e[y] = p[y];
    };
    // loads are issued one row ahead and only touched (narrowed, checked) when that row is
    // processed, so that their latency stays hidden behind the selection of the current row
    auto load_samples = [&](const uint32_t (&e)[YPS], TI (&x)[YPS]) {
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            const uint32_t code = e[y] >> 1;
            TI v = static_cast<TI>(fnan);
            if (code >= 2 && cell_ok) v = col[static_cast<int64_t>(code - 2) * ld];
            x[y] = v;
        }
    };

    uint32_t e_cur[YPS], e_nxt[YPS];
    TI x_cur[YPS];
    load_entries(ch.warm_start, e_cur);
    load_samples(e_cur, x_cur);
    if (ch.warm_start + 1 < ch.end) load_entries(ch.warm_start + 1, e_nxt);
    else {
#pragma unroll
        for (int y = 0; y < YPS; ++y) e_nxt[y] = make_entry(kCodeInvalid, false);
    }

    int m = (ch.warm_start - step_min) % R;
    // Carried across rows, uniform over the 8 lanes of a cell:
    //   pc / Fc   pivot with the raw count #{ring keys <= pc over ALL slots}, kept
    //             exact by adding the per-step difference of the pushed / evicted
    //             keys -> the first "probe" of a row costs 2 ops per pushed key
    //   kpr       keys per rank observed on the previous row (local density and
    //             seasonal drift together), used to aim the first real probe
    uint32_t pc = 0, Fc = NSLOT;
    bool have_c = false;
    float kpr = 8192.0f;
    uint32_t st_count = 0, st_extract = 0, st_rows = 0, st_cold = 0;  // debug counters (wave-uniform)

    for (int32_t s = ch.warm_start; s < ch.end; ++s) {
        // ---- prefetch: samples of step s+1, table entries of step s+2 ------------
        TI x_nxt[YPS];
        uint32_t e_nn[YPS];
        load_samples(e_nxt, x_nxt);
        if (s + 2 < ch.end) load_entries(s + 2, e_nn);
        else {
#pragma unroll
            for (int y = 0; y < YPS; ++y) e_nn[y] = make_entry(kCodeInvalid, false);
        }

        // ---- advance the rings -----------------------------------------------------
        uint32_t kin[YPS], kout[YPS];
        float xf[YPS];
        bool hold[YPS], counted[YPS];
        bool any_hold = false, all_counted = true;
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            counted[y] = (e_cur[y] & 1u) != 0;
            hold[y] = (e_cur[y] >> 1) == kCodeHold;
            any_hold |= hold[y];
            all_counted &= counted[y];
            float xv = static_cast<float>(x_cur[y]);
            if constexpr (kNarrow) lossy |= (static_cast<TI>(xv) != x_cur[y]) && (x_cur[y] == x_cur[y]);
            if (negate) xv = -xv;
            xf[y] = xv;
            kin[y] = f32_key(xv);  // NaN / not loaded -> 0 (invalid)
        }
        if constexpr (kNarrow) {
            if (__builtin_amdgcn_ballot_w64(lossy) != 0) break;   // wave-uniform
        }
#define XMHW_RING_CASE(K)                                                  \
    case K:                                                                \
        if constexpr (K < R) {                                             \
            _Pragma("unroll") for (int y = 0; y < YPS; ++y) {              \
                const uint32_t o = ring[y][K < R ? K : 0];                 \
                kout[y] = o;                                               \
                ring[y][K < R ? K : 0] = hold[y] ? o : kin[y];             \
            }                                                              \
        }                                                                  \
        break;
        switch (m) {
            XMHW_RING_CASE(0) XMHW_RING_CASE(1) XMHW_RING_CASE(2) XMHW_RING_CASE(3)
            XMHW_RING_CASE(4) XMHW_RING_CASE(5) XMHW_RING_CASE(6) XMHW_RING_CASE(7)
            XMHW_RING_CASE(8) XMHW_RING_CASE(9) XMHW_RING_CASE(10) XMHW_RING_CASE(11)
            XMHW_RING_CASE(12) XMHW_RING_CASE(13) XMHW_RING_CASE(14) XMHW_RING_CASE(15)
            XMHW_RING_CASE(16) XMHW_RING_CASE(17) XMHW_RING_CASE(18) XMHW_RING_CASE(19)
            XMHW_RING_CASE(20) XMHW_RING_CASE(21) XMHW_RING_CASE(22) XMHW_RING_CASE(23)
            XMHW_RING_CASE(24) XMHW_RING_CASE(25) XMHW_RING_CASE(26) XMHW_RING_CASE(27)
            XMHW_RING_CASE(28) XMHW_RING_CASE(29) XMHW_RING_CASE(30)
            default: break;
        }
#undef XMHW_RING_CASE
        m = (m + 1 == R) ? 0 : m + 1;
        uint32_t dF = 0;  // change of #{keys <= pc} in this lane (two's complement)
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            if (!hold[y]) {
                const float xin = (kin[y] != 0) ? xf[y] : 0.0f;
                tsum[y] += static_cast<double>(xin);
                tsum[y] -= key_value(kout[y]);
                nval[y] += (kin[y] != 0 ? 1u : 0u) - (kout[y] != 0 ? 1u : 0u);
                dF += (kin[y] <= pc ? 1u : 0u) - (kout[y] <= pc ? 1u : 0u);
            }
        }
        const bool wave_hold = __any(any_hold);
        if (wave_hold) {
            // a held track did not advance: rotate its window one slot so that its
            // oldest sample sits where the next step's PUSH will land
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                const uint32_t last = ring[y][R - 1];
#pragma unroll
                for (int k = R - 1; k >= 1; --k) ring[y][k] = hold[y] ? ring[y][k - 1] : ring[y][k];
                ring[y][0] = hold[y] ? last : ring[y][0];
            }
        }

        // ---- select + output (not during warm-up) ---------------------------------
        if (s >= ch.begin) {
            const bool allc = __all(all_counted);  // wave-uniform: no masking needed
            uint32_t nl = 0, ncl = 0;
            double tl = 0.0;
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                nl += counted[y] ? nval[y] : 0u;
                ncl += counted[y] ? 1u : 0u;
                tl += counted[y] ? tsum[y] : 0.0;
            }
            const uint32_t n = sub_sum<SUBS>(nl);
            const uint32_t ninv = (allc ? NSLOT : static_cast<uint32_t>(R) * sub_sum<SUBS>(ncl)) - n;
            double total = sub_sum<SUBS>(tl);
            if (__any(!(fabs(total) <= 1.7976931348623157e308))) {
                // an infinite sample went through a running sum (inf - inf = NaN once it leaves):
                // rebuild the sums from the rings; with the sample still inside the pool the total
                // stays +-inf / NaN, exactly what numpy's mean gives for that pool
                tl = 0.0;
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < R; ++k) t += key_value(ring[y][k]);
                    tsum[y] = t;
                    tl += counted[y] ? t : 0.0;
                }
                total = sub_sum<SUBS>(tl);
            }
            Fc += sub_sum<SUBS>(dF);  // raw count at the carried pivot, now for this row's rings

            const uint32_t nn = n ? n : 1u;
            const double vi = static_cast<double>(nn - 1) * q;
            const double fl = floor(vi);
            const double g = vi - fl;
            const uint32_t lo = static_cast<uint32_t>(fl);
            const bool need2 = lo + 1 < nn;

            // raw count #{counted ring keys <= p}; invalid keys (0) are <= any pivot
            auto count_le = [&](uint32_t p) -> uint32_t {
                uint32_t c = 0;
                if (allc) {
#pragma unroll
                    for (int y = 0; y < YPS; ++y)
#pragma unroll
                        for (int k = 0; k < R; ++k) c += (ring[y][k] <= p) ? 1u : 0u;
                } else {
#pragma unroll
                    for (int y = 0; y < YPS; ++y) {
                        uint32_t cy = 0;
#pragma unroll
                        for (int k = 0; k < R; ++k) cy += (ring[y][k] <= p) ? 1u : 0u;
                        c += counted[y] ? cy : 0u;
                    }
                }
                return sub_sum<SUBS>(c);
            };

            // bracket: F(pl) = Fl <= lo < Fh = F(ph), F = #{valid counted keys <= .}
            uint32_t pl = 0, Fl = 0, ph = 0xFFFFFFFFu, Fh = nn;
            bool lreal = false, hreal = false;
            float grow = 1.0f;
            const bool use_c = have_c && allc;
            uint32_t p0 = pc, F0 = 0;
            if (use_c) F0 = Fc - ninv;  // free probe: the carried pivot
            if (!__all(use_c || n == 0)) {
                // cold start (first row of a chunk, masked row): one real pass at the
                // carried pivot if there is one, else at the pool mean
                uint32_t pm = f32_key(static_cast<float>(total / static_cast<double>(nn)));
                if (pm == 0) pm = 0x80000000u;
                if (!use_c) p0 = have_c ? pc : pm;
                const uint32_t Fr = count_le(p0);
                if (!use_c) F0 = Fr - ninv;
                ++st_cold;
            }
            if (p0 != 0 && p0 != 0xFFFFFFFFu) {
                if (F0 <= lo) { pl = p0; Fl = F0; lreal = true; }
                else { ph = p0; Fh = F0; hreal = true; }
            }
            const uint32_t p_first = p0;
            const int32_t rank_gap = static_cast<int32_t>(lo) - static_cast<int32_t>(F0);
            // probes aim at the middle of the window of acceptable ranks
            // (centre of the window and the 0.75/0.25 blend below: tools/sim_tune.py, 2.71 -> 2.60 passes per wave-row)
            const float aim = static_cast<float>(lo) - 0.5f * static_cast<float>(SLACK);

            bool resolved = (n == 0);
            uint32_t alo = 0, ahi = 0, pe = 0, Fe = 0;
            int budget = kCountBudget;
            for (;;) {
                // ---- count passes until every cell can be settled by one extraction ----
                for (int it = 0;; ++it) {
                    const bool settle = resolved || (lo - Fl <= SLACK) || (ph - pl <= 1u);
                    if (__all(settle) || it >= budget) break;
                    const uint32_t room = ph - pl;
                    // one formula for all cases: step = ranks-to-go x keys-per-rank, taken from
                    // the known end of the bracket; both ends known -> secant slope
                    const bool both = lreal && hreal;
                    const float roomf = static_cast<float>(room);
                    const float slope = both ? roomf * __builtin_amdgcn_rcpf(static_cast<float>(Fh - Fl))
                                             : kpr * grow;
                    const float ranks = lreal ? aim - static_cast<float>(Fl) : static_cast<float>(Fh) - aim;
                    float stf = fminf(fmaxf(ranks * slope, 1.0f), 2.0e9f);
                    stf = lreal ? stf : roomf - stf;
                    stf = fminf(fmaxf(stf, 1.0f), 4.0e9f);
                    uint32_t off = (it < 5) ? static_cast<uint32_t>(stf) : (room >> 1);
                    grow = both ? grow : grow * 2.0f;
                    off = umax(1u, umin(off, room - 1u));
                    const uint32_t p = settle ? pl : pl + off;
                    const uint32_t F = count_le(p) - ninv;
                    ++st_count;
                    if (!settle) {
                        if (F <= lo) { pl = p; Fl = F; lreal = true; }
                        else { ph = p; Fh = F; hreal = true; }
                    }
                }
                // ---- extraction: the J smallest keys above the pivot --------------------
                const bool window = (lo - Fl <= SLACK);
                const bool adjacent = !window && (ph - pl <= 1u);
                const uint32_t px = adjacent ? ph : pl;
                const uint32_t base = px + 1u;
                TopJ<J> top;
                top.reset();
                if (allc) {
#pragma unroll
                    for (int y = 0; y < YPS; ++y)
#pragma unroll
                        for (int k = 0; k < R; ++k) top.insert(ring[y][k] - base);  // keys <= px wrap high
                } else {
#pragma unroll
                    for (int y = 0; y < YPS; ++y)
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            const uint32_t d = ring[y][k] - base;
                            top.insert(counted[y] ? d : 0xFFFFFFFFu);
                        }
                }
                top.template sub_merge<SUBS>();
                ++st_extract;
                if (!resolved) {
                    if (window) {
                        const uint32_t j = lo - Fl;
                        alo = base + top.at(j);
                        ahi = need2 ? base + top.at(j + 1u) : alo;
                        pe = pl; Fe = Fl;
                        resolved = true;
                    } else if (adjacent) {
                        // every key at sorted positions Fl .. Fh-1 equals ph
                        alo = ph;
                        ahi = (need2 && lo + 1u >= Fh) ? base + top.m[0] : ph;
                        pe = ph; Fe = Fh;
                        resolved = true;
                    }
                }
                if (__all(resolved)) break;
                // ---- repair (tie-heavy data): count at the largest extracted key ---------
                const uint32_t dj = top.m[J - 1];
                const uint32_t pj = base + dj;            // an element key, > pl
                const uint32_t Fj = count_le(resolved ? pl : pj) - ninv;
                ++st_count;
                if (!resolved) {
                    if (Fj <= lo) {
                        pl = pj; Fl = Fj; lreal = true;
                    } else {
                        // no key lies strictly between pl and pj except the extracted ones
                        ph = pj; Fh = Fj; hreal = true;
                        Fl = Fl + top.count_below(dj);
                        pl = pj - 1u;
                    }
                }
                budget = 2;
            }

            ++st_rows;
            double th = make_nan(), se = make_nan();
            if (n > 0) {
                th = numpy_lerp(static_cast<double>(key_f32(alo)), static_cast<double>(key_f32(ahi)), g);
                se = total / static_cast<double>(n);
                // calibrate keys-per-rank on what this row needed, carry the pivot
                if (rank_gap > 1 || rank_gap < -1) {
                    const float obs = (static_cast<float>(alo) - static_cast<float>(p_first)) *
                                      __builtin_amdgcn_rcpf(static_cast<float>(rank_gap));
                    if (obs >= 1.0f && obs < 1.0e8f) kpr = 0.75f * kpr + 0.25f * obs;
                }
            }
            if (n > 0 && allc) {
                pc = pe;
                Fc = Fe + ninv;
                have_c = true;
            } else {
                have_c = false;
                pc = 0;
                Fc = 0;
            }
            if (sub == 0 && cell_ok) {
                thresh[static_cast<int64_t>(s) * ldo + cell] = th;
                seas[static_cast<int64_t>(s) * ldo + cell] = se;
            }
        }

        // The 4 waves of a workgroup share every 128-B row segment (32 B each).  Left alone they
        // drift apart until the shared lines have left L2 and are fetched again: measured
        // 1.36x the input bytes at L2-miss level (TCC_EA0_RDREQ_128B), 1.03x with this
        // occasional rendezvous (+1 % time; every 4 rows: +5 %).  Trip counts are workgroup-
        // uniform (same chunk), so the barrier is safe -- except in the narrowing instantiation,
        // whose waves may leave the loop on their own (lossy sample, flag already set): there the
        // rendezvous is compiled out (a wave that has left would make the barrier divergent).
        if constexpr (!kNarrow) {
            if ((s & 63) == 63) __syncthreads();
        }
        // ---- rotate the software pipeline --------------------------------------------
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            e_cur[y] = e_nxt[y];
            e_nxt[y] = e_nn[y];
            x_cur[y] = x_nxt[y];
        }
    }
    if constexpr (kNarrow) {
        if (lossy) atomicOr(narrow_flag, 1u);
    }
    if (stats != nullptr && lane == 0) {
        atomicAdd(&stats[0], static_cast<unsigned long long>(st_rows));
        atomicAdd(&stats[1], static_cast<unsigned long long>(st_count));
        atomicAdd(&stats[2], static_cast<unsigned long long>(st_extract));
        atomicAdd(&stats[3], static_cast<unsigned long long>(st_cold));
    }
}

// ---------------------------------------------------------------------------
// instantiation table
// ---------------------------------------------------------------------------
namespace {
typedef void (*RingKernel)(const float*, int64_t, int64_t, const uint32_t*, int32_t, const DevChunk*,
                           double, int, double*, double*, int64_t, unsigned long long*, uint32_t*);
typedef void (*RingKernelN)(const double*, int64_t, int64_t, const uint32_t*, int32_t, const DevChunk*,
                            double, int, double*, double*, int64_t, unsigned long long*, uint32_t*);
struct RingEntry { int w, yps, subs; RingKernel fn; RingKernelN fn_narrow; };
#define XMHW_RK(W, Y, S) {W, Y, S, clim_ring_f32<W, Y, float, S>, clim_ring_f32<W, Y, double, S>}
// (window half width, tracks per lane, lanes per cell); the ring holds YPS * (2w+1) keys per lane
// (<= 66 VGPRs), so wide windows trade tracks per lane for lanes per cell
const RingEntry kRing[] = {
    // the default window: records of up to 8 tracks on 8 lanes per cell.  (Round 4: the 8- and 16-lane entries for 9..96
    // tracks are gone -- the third- and second-generation kernels serve those records; a plan that is forced onto this
    // kernel there, XMHW_LAYOUT_RING1, runs on the 32-lane entries below, padded.)
    XMHW_RK(5, 1, 8),
    // very long records (e.g. 165-year model runs): 32 lanes per cell, up to 192 tracks
    XMHW_RK(5, 4, 32), XMHW_RK(5, 5, 32), XMHW_RK(5, 6, 32),
    // narrower windows (pentad / coarse-step data)
    XMHW_RK(1, 1, 8), XMHW_RK(1, 2, 8), XMHW_RK(1, 4, 8), XMHW_RK(1, 6, 8),
    XMHW_RK(2, 1, 8), XMHW_RK(2, 2, 8), XMHW_RK(2, 4, 8), XMHW_RK(2, 6, 8),
    XMHW_RK(3, 2, 8), XMHW_RK(3, 4, 8), XMHW_RK(3, 6, 8),
    XMHW_RK(4, 2, 8), XMHW_RK(4, 4, 8), XMHW_RK(4, 6, 8),
    // wider windows
    XMHW_RK(7, 2, 8), XMHW_RK(7, 4, 8), XMHW_RK(7, 4, 16),
    XMHW_RK(10, 3, 8), XMHW_RK(10, 3, 16),
    XMHW_RK(15, 2, 8), XMHW_RK(15, 2, 16),
};
#undef XMHW_RK
const RingEntry* find_ring(int32_t w, int32_t yps, int32_t subs) {
    for (const auto& e : kRing)
        if (e.w == w && e.yps == yps && e.subs == subs) return &e;
    return nullptr;
}

// a sparse look at a float64 series (32 rows spread over the time axis, every cell) before the
// narrowing kernel is tried: genuinely float64 data is recognised here at no cost
__global__ __launch_bounds__(256) void narrow_probe(const double* __restrict__ ts, int64_t Tn, int64_t C,
                                                    int64_t ld, uint32_t* __restrict__ flag) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int64_t stride = Tn / 32 > 0 ? Tn / 32 : 1;
    bool lossy = false;
    for (int64_t t = 0; t < Tn; t += stride) {
        const double raw = ts[t * ld + c];
        lossy |= (static_cast<double>(static_cast<float>(raw)) != raw) && (raw == raw);
    }
    if (lossy) atomicOr(flag, 1u);
}
}  // namespace

bool ring_supported(int32_t w, int32_t yps, int32_t subs, int elem_bytes) {
    return elem_bytes == 4 && find_ring(w, yps, subs) != nullptr;
}

// fewest lanes per cell first (8 subs: twice the cells per wave), then the fewest tracks per lane
int32_t ring_pick(int32_t w, int32_t ntracks, int elem_bytes, int32_t* subs_out) {
    if (subs_out) *subs_out = 0;
    if (elem_bytes != 4) return 0;
    for (int32_t subs : {8, 16, 32}) {
        int32_t best = 0;
        for (const auto& e : kRing)
            if (e.w == w && e.subs == subs && e.yps * subs >= ntracks && (best == 0 || e.yps < best)) best = e.yps;
        if (best) {
            if (subs_out) *subs_out = subs;
            return best;
        }
    }
    return 0;
}

hipError_t launch_ring_f32(const float* ts, int64_t C, int64_t ld, const uint32_t* table,
                           int32_t step_min, const DevChunk* chunks, int32_t nchunks, int32_t w,
                           int32_t yps, int32_t subs, double q, int negate, double* thresh, double* seas,
                           int64_t ldo, hipStream_t stream, unsigned long long* stats) {
    const RingEntry* e = find_ring(w, yps, subs);
    if (!e) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    const int64_t cells_per_block = (64 / subs) * kWavesPerBlock;
    dim3 grid(static_cast<unsigned>((C + cells_per_block - 1) / cells_per_block),
              static_cast<unsigned>(nchunks));
    hipLaunchKernelGGL(e->fn, grid, dim3(64 * kWavesPerBlock), 0, stream, ts, C, ld, table, step_min,
                       chunks, q, negate, thresh, seas, ldo, stats, static_cast<uint32_t*>(nullptr));
    return hipGetLastError();
}

// clears the flag and queues the sparse look at the float64 series (shared with the ring2 narrowing launch)
hipError_t launch_narrow_probe(const double* ts, int64_t Tn, int64_t C, int64_t ld, uint32_t* narrow_flag,
                               hipStream_t stream) {
    if (!narrow_flag) return hipErrorInvalidValue;
    hipError_t err = hipMemsetAsync(narrow_flag, 0, sizeof(uint32_t), stream);
    if (err != hipSuccess) return err;
    if (C <= 0 || Tn <= 0) return hipSuccess;
    hipLaunchKernelGGL(narrow_probe, dim3(static_cast<unsigned>((C + 255) / 256)), dim3(256), 0, stream, ts, Tn, C,
                       ld, narrow_flag);
    return hipGetLastError();
}

hipError_t launch_ring_f32_narrowing(const double* ts, int64_t Tn, int64_t C, int64_t ld, const uint32_t* table,
                                     int32_t step_min, const DevChunk* chunks, int32_t nchunks, int32_t w,
                                     int32_t yps, int32_t subs, double q, int negate, double* thresh, double* seas,
                                     int64_t ldo, hipStream_t stream, unsigned long long* stats,
                                     uint32_t* narrow_flag) {
    const RingEntry* e = find_ring(w, yps, subs);
    if (!e || !narrow_flag) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    hipError_t err = hipMemsetAsync(narrow_flag, 0, sizeof(uint32_t), stream);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(narrow_probe, dim3(static_cast<unsigned>((C + 255) / 256)), dim3(256), 0, stream, ts, Tn, C,
                       ld, narrow_flag);
    const int64_t cells_per_block = (64 / subs) * kWavesPerBlock;
    dim3 grid(static_cast<unsigned>((C + cells_per_block - 1) / cells_per_block),
              static_cast<unsigned>(nchunks));
    hipLaunchKernelGGL(e->fn_narrow, grid, dim3(64 * kWavesPerBlock), 0, stream, ts, C, ld, table, step_min,
                       chunks, q, negate, thresh, seas, ldo, stats, narrow_flag);
    return hipGetLastError();
}

}  // namespace xmhw
