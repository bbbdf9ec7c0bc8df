// kernels_ring64.hip -- float64-input variant of the register-ring kernel
// (see kernels_ring.hip for the design).  Differences:
//   * a cell is spread over SUBS = 16 lanes (one whole DPP row), each lane owns
//     YPS <= 3 tracks, so the 64-bit ring is 2*YPS*R <= 66 VGPRs; 4 cells per wave;
//   * samples stay doubles in the ring and are compared in the float domain
//     (v_cmp_*_f64, v_min/max_f64); an invalid sample is NaN and fails every
//     compare, so no "invalid keys count as small" correction is needed;
//   * the bracket lives in the order-preserving 64-bit key space (device_common.h)
//     so that "no double lies between pl and ph" is an integer test and the
//     bisection fallback terminates; probes are converted key -> double;
//   * extraction filter is inclusive: x >= succ(pivot), which also admits -inf.
// Reference semantics restated: as kernels_ring.hip.
#include "device_common.h"
#include "kernels.h"
#include "plan.h"

namespace xmhw {

namespace r64 {

constexpr int kWavesPerBlock = 4;
constexpr int kSubs = 16;
constexpr int kCells = 4;
constexpr int kTopJ = 6;
constexpr int kCountBudget = 6;
constexpr uint64_t kKeyNegInf = 0x000FFFFFFFFFFFFFull;  // f64_key(-inf): smallest valid key
constexpr uint64_t kKeyPosInf = 0xFFF0000000000000ull;  // f64_key(+inf): largest valid key

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
constexpr int dpp_ror(int n) { return 0x120 | n; }
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
    const uint32_t lo = dpp_mov<CTRL>(static_cast<uint32_t>(b));
    const uint32_t hi = dpp_mov<CTRL>(static_cast<uint32_t>(b >> 32));
    return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
}
// all-reduce over the 16 lanes of a DPP row
__device__ __forceinline__ uint32_t row_sum(uint32_t v) {
    v += dpp_mov<dpp_ror(8)>(v);
    v += dpp_mov<dpp_ror(4)>(v);
    v += dpp_mov<dpp_ror(2)>(v);
    v += dpp_mov<dpp_ror(1)>(v);
    return v;
}
__device__ __forceinline__ double row_sum(double v) {
    v += dpp_mov_f64<dpp_ror(8)>(v);
    v += dpp_mov_f64<dpp_ror(4)>(v);
    v += dpp_mov_f64<dpp_ror(2)>(v);
    v += dpp_mov_f64<dpp_ror(1)>(v);
    return v;
}
__device__ __forceinline__ double dinf() { return __longlong_as_double(0x7FF0000000000000ll); }
__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b) { return a > b ? a : b; }

// J smallest values (position-wise, ties repeated) at or above a bound, ascending
template <int J>
struct TopJ {
    static_assert(J >= 2 && J <= 8, "merge network is built for 2..8 keys");
    double m[J];
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int i = 0; i < J; ++i) m[i] = dinf();
    }
    __device__ __forceinline__ void insert(double d) {  // d is never NaN
#pragma unroll
        for (int i = J - 1; i >= 1; --i) m[i] = fmin(fmax(m[i - 1], d), m[i]);
        m[0] = fmin(m[0], d);
    }
    template <int CTRL>
    __device__ __forceinline__ void merge_dpp() {
        double b[J];
#pragma unroll
        for (int i = 0; i < J; ++i) b[i] = dpp_mov_f64<CTRL>(m[i]);
#pragma unroll
        for (int i = 0; i < J; ++i) m[i] = fmin(m[i], b[J - 1 - i]);
        constexpr int OFF = 8 - J;  // see kernels_ring.hip: front-padded 8-key bitonic merge
#pragma unroll
        for (int d = 4; d >= 1; d >>= 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if ((k & d) == 0 && k >= OFF && k + d < 8) {
                    const double lo_ = fmin(m[k - OFF], m[k - OFF + d]);
                    const double hi_ = fmax(m[k - OFF], m[k - OFF + d]);
                    m[k - OFF] = lo_;
                    m[k - OFF + d] = hi_;
                }
            }
        }
    }
    __device__ __forceinline__ void row_merge() {
        merge_dpp<dpp_ror(8)>();
        merge_dpp<dpp_ror(4)>();
        merge_dpp<dpp_ror(2)>();
        merge_dpp<dpp_ror(1)>();
    }
    __device__ __forceinline__ double at(uint32_t j) const {
        // a chain of selects; the empty asm keeps the compiler from turning it into a dynamically indexed
        // array (which it places in LDS: a store of all J entries and a dependent read on every row --
        // the 12 KB of LDS per block this kernel used to report)
        double r = m[0];
#pragma unroll
        for (int i = 1; i < J; ++i) {
            r = (j == static_cast<uint32_t>(i)) ? m[i] : r;
            asm volatile("" : "+v"(r));
        }
        return r;
    }
    __device__ __forceinline__ uint32_t count_below(double d) const {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < J; ++i) c += (m[i] < d) ? 1u : 0u;
        return c;
    }
};

}  // namespace r64

template <int W, int YPS>
__global__ __launch_bounds__(256) void clim_ring_f64(
    const double* __restrict__ ts, int64_t C, int64_t ld, const uint32_t* __restrict__ table,
    int32_t step_min, const DevChunk* __restrict__ chunks, double q, int negate,
    double* __restrict__ thresh, double* __restrict__ seas, int64_t ldo,
    unsigned long long* __restrict__ stats, const uint32_t* __restrict__ run_flag) {
    using namespace r64;
    // queued behind the narrowing float32 kernel: nothing to do unless that one gave up
    if (run_flag != nullptr && *run_flag == 0) return;
    constexpr int R = 2 * W + 1;
    constexpr int NTP = kSubs * YPS;
    constexpr int J = kTopJ;
    constexpr uint32_t SLACK = J - 2;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane & 15;
    const int cw = lane >> 4;
    const int64_t cell = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave) * kCells + cw;
    const bool cell_ok = cell < C;
    const DevChunk ch = chunks[blockIdx.y];
    const uint32_t* tab = table + sub * YPS;
    const double* col = ts + (cell_ok ? cell : 0);
    const double dnan = make_nan();

    double ring[YPS][R];
    double tsum[YPS];
    uint32_t nval[YPS];
#pragma unroll
    for (int y = 0; y < YPS; ++y) {
        tsum[y] = 0.0;
        nval[y] = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) ring[y][k] = dnan;
    }

    auto load_entries = [&](int32_t s, uint32_t (&e)[YPS]) {
        const uint32_t* p = tab + static_cast<int64_t>(s - step_min) * NTP;
#pragma unroll
        for (int y = 0; y < YPS; ++y) e[y] = p[y];
    };
    auto load_samples = [&](const uint32_t (&e)[YPS], double (&x)[YPS]) {
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            const uint32_t code = e[y] >> 1;
            double v = dnan;
            if (code >= 2 && cell_ok) v = col[static_cast<int64_t>(code - 2) * ld];
            x[y] = v;
        }
    };

    uint32_t e_cur[YPS], e_nxt[YPS];
    double x_cur[YPS];
    load_entries(ch.warm_start, e_cur);
    load_samples(e_cur, x_cur);
    if (ch.warm_start + 1 < ch.end) load_entries(ch.warm_start + 1, e_nxt);
    else {
#pragma unroll
        for (int y = 0; y < YPS; ++y) e_nxt[y] = make_entry(kCodeInvalid, false);
    }

    int m = (ch.warm_start - step_min) % R;
    // carried pivot (value + key) with the count #{valid ring samples <= pc over ALL tracks}
    double pcv = 0.0;
    uint64_t pck = 0;
    uint32_t Fc = 0;
    bool have_c = false;
    float kpr = 4.0e12f;  // keys per rank (2^29 x the float32 kernel's scale)
    uint32_t st_count = 0, st_extract = 0, st_rows = 0, st_cold = 0;

    for (int32_t s = ch.warm_start; s < ch.end; ++s) {
        double x_nxt[YPS];
        uint32_t e_nn[YPS];
        load_samples(e_nxt, x_nxt);
        if (s + 2 < ch.end) load_entries(s + 2, e_nn);
        else {
#pragma unroll
            for (int y = 0; y < YPS; ++y) e_nn[y] = make_entry(kCodeInvalid, false);
        }

        // ---- advance the rings -----------------------------------------------------
        double xin[YPS], xout[YPS];
        bool hold[YPS], counted[YPS];
        bool any_hold = false, all_counted = true;
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            counted[y] = (e_cur[y] & 1u) != 0;
            hold[y] = (e_cur[y] >> 1) == kCodeHold;
            any_hold |= hold[y];
            all_counted &= counted[y];
            double xv = x_cur[y];
            if (negate) xv = -xv;
            // one zero only: this kernel compares samples as doubles (-0.0 == +0.0) but brackets in key space
            // (-0.0 < +0.0); with both zeros in a pool the two disagree about counts.  -0.0 + 0.0 = +0.0.
            // (The second-generation kernel compares keys throughout and keeps both.)
            xv = xv + 0.0;
            xin[y] = xv;
        }
#define XMHW_RING_CASE(K)                                                  \
    case K:                                                                \
        if constexpr (K < R) {                                             \
            _Pragma("unroll") for (int y = 0; y < YPS; ++y) {              \
                const double o = ring[y][K < R ? K : 0];                   \
                xout[y] = o;                                               \
                ring[y][K < R ? K : 0] = hold[y] ? o : xin[y];             \
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
        uint32_t dF = 0;
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            if (!hold[y]) {
                const bool vin = xin[y] == xin[y], vout = xout[y] == xout[y];
                tsum[y] += vin ? xin[y] : 0.0;
                tsum[y] -= vout ? xout[y] : 0.0;
                nval[y] += (vin ? 1u : 0u) - (vout ? 1u : 0u);
                dF += (xin[y] <= pcv ? 1u : 0u) - (xout[y] <= pcv ? 1u : 0u);
            }
        }
        if (__any(any_hold)) {
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                const double last = ring[y][R - 1];
#pragma unroll
                for (int k = R - 1; k >= 1; --k) ring[y][k] = hold[y] ? ring[y][k - 1] : ring[y][k];
                ring[y][0] = hold[y] ? last : ring[y][0];
            }
        }

        // ---- select + output (not during warm-up) ---------------------------------
        if (s >= ch.begin) {
            const bool allc = __all(all_counted);
            uint32_t nl = 0;
            double tl = 0.0;
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                nl += counted[y] ? nval[y] : 0u;
                tl += counted[y] ? tsum[y] : 0.0;
            }
            const uint32_t n = row_sum(nl);
            double total = row_sum(tl);
            if (__any(!(fabs(total) <= 1.7976931348623157e308))) {
                // an infinite sample went through a running sum: rebuild the sums from the rings
                // (see kernels_ring.hip); NaN slots are the invalid ones
                tl = 0.0;
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    double t = 0.0;
#pragma unroll
                    for (int k = 0; k < R; ++k) t += ring[y][k] == ring[y][k] ? ring[y][k] : 0.0;
                    tsum[y] = t;
                    tl += counted[y] ? t : 0.0;
                }
                total = row_sum(tl);
            }
            Fc += row_sum(dF);

            const uint32_t nn = n ? n : 1u;
            const double vi = static_cast<double>(nn - 1) * q;
            const double fl = floor(vi);
            const double g = vi - fl;
            const uint32_t lo = static_cast<uint32_t>(fl);
            const bool need2 = lo + 1 < nn;

            // F(p) = #{valid counted samples <= p}; NaN (invalid) never compares true
            auto count_le = [&](double p) -> uint32_t {
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
                return row_sum(c);
            };

            // bracket in key space: F(kl) = Fl <= lo < Fh = F(kh)
            uint64_t kl = 0, kh = ~0ull;
            uint32_t Fl = 0, Fh = nn;
            bool lreal = false, hreal = false;
            float grow = 1.0f;
            const bool use_c = have_c && allc;
            uint64_t k0 = pck;
            uint32_t F0 = 0;
            if (use_c) F0 = Fc;
            if (!__all(use_c || n == 0)) {
                const double mean = total / static_cast<double>(nn);
                uint64_t km = f64_key(mean);
                if (km == 0) km = 0x8000000000000000ull;
                if (!use_c) k0 = have_c ? pck : km;
                const uint32_t Fr = count_le(key_f64(k0));
                if (!use_c) F0 = Fr;
                ++st_cold;
            }
            if (k0 != 0 && k0 != ~0ull) {
                if (F0 <= lo) { kl = k0; Fl = F0; lreal = true; }
                else { kh = k0; Fh = F0; hreal = true; }
            }
            const uint64_t k_first = k0;
            const int32_t rank_gap = static_cast<int32_t>(lo) - static_cast<int32_t>(F0);
            const float aim = static_cast<float>(lo) - 0.5f * static_cast<float>(SLACK);   // window centre (tools/sim_tune.py)

            bool resolved = (n == 0);
            double alo = 0.0, ahi = 0.0;
            uint64_t ke = 0;
            uint32_t Fe = 0;
            int budget = kCountBudget;
            for (;;) {
                for (int it = 0;; ++it) {
                    const bool settle = resolved || (lo - Fl <= SLACK) || (kh - kl <= 1ull);
                    if (__all(settle) || it >= budget) break;
                    const uint64_t room = kh - kl;
                    const bool both = lreal && hreal;
                    const float roomf = static_cast<float>(room);
                    const float slope = both ? roomf * __builtin_amdgcn_rcpf(static_cast<float>(Fh - Fl))
                                             : kpr * grow;
                    const float ranks = lreal ? aim - static_cast<float>(Fl) : static_cast<float>(Fh) - aim;
                    float stf = fminf(fmaxf(ranks * slope, 1.0f), 9.0e18f);
                    stf = lreal ? stf : roomf - stf;
                    stf = fminf(fmaxf(stf, 1.0f), 9.0e18f);
                    uint64_t off = (it < 5) ? static_cast<uint64_t>(stf) : (room >> 1);
                    grow = both ? grow : grow * 2.0f;
                    off = umax64(1ull, umin64(off, room - 1ull));
                    // keep the pivot among the keys of real numbers: the key space continues into the NaN
                    // patterns beyond +-inf, and a pivot there compares false with everything (count 0 where it
                    // should be n) -- a one-sided secant step on clustered data could overshoot into them
                    // (found by tools/fuzz_ring2.py --dtype f64 in round 2)
                    const uint64_t kp = settle ? kl : umin64(umax64(kl + off, kKeyNegInf), kKeyPosInf);
                    const uint32_t F = count_le(key_f64(kp));
                    ++st_count;
                    if (!settle) {
                        if (F <= lo) { kl = kp; Fl = F; lreal = true; }
                        else { kh = kp; Fh = F; hreal = true; }
                    }
                }
                // ---- extraction: the J smallest samples above the pivot ------------------
                const bool window = (lo - Fl <= SLACK);
                const bool adjacent = !window && (kh - kl <= 1ull);
                const uint64_t kx = adjacent ? kh : kl;
                // inclusive bound: x > value(kx)  <=>  x >= value(kx + 1); -inf is the floor
                const double qx = key_f64(umax64(kx + 1ull, kKeyNegInf));
                TopJ<J> top;
                top.reset();
#pragma unroll
                for (int y = 0; y < YPS; ++y)
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const double x = ring[y][k];
                        const bool in = allc ? (x >= qx) : (counted[y] && x >= qx);
                        top.insert(in ? x : dinf());
                    }
                top.row_merge();
                ++st_extract;
                if (!resolved) {
                    if (window) {
                        const uint32_t j = lo - Fl;
                        alo = top.at(j);
                        ahi = need2 ? top.at(j + 1u) : alo;
                        ke = kl; Fe = Fl;
                        resolved = true;
                    } else if (adjacent) {
                        alo = key_f64(kh);
                        ahi = (need2 && lo + 1u >= Fh) ? top.m[0] : alo;
                        ke = kh; Fe = Fh;
                        resolved = true;
                    }
                }
                if (__all(resolved)) break;
                // ---- repair (tie-heavy data): count at the largest extracted sample -------
                const double vj = top.m[J - 1];
                const uint64_t kj = f64_key(vj);
                const uint32_t Fj = count_le(resolved ? key_f64(kl) : vj);
                ++st_count;
                if (!resolved) {
                    if (Fj <= lo) {
                        kl = kj; Fl = Fj; lreal = true;
                    } else {
                        kh = kj; Fh = Fj; hreal = true;
                        Fl = Fl + top.count_below(vj);
                        kl = kj - 1ull;
                    }
                }
                budget = 2;
            }

            ++st_rows;
            double th = make_nan(), se = make_nan();
            if (n > 0) {
                th = numpy_lerp(alo, ahi, g);
                se = total / static_cast<double>(n);
                if (rank_gap > 1 || rank_gap < -1) {
                    const uint64_t ka = f64_key(alo);
                    const float dk = ka >= k_first ? static_cast<float>(ka - k_first)
                                                   : -static_cast<float>(k_first - ka);
                    const float obs = dk * __builtin_amdgcn_rcpf(static_cast<float>(rank_gap));
                    if (obs >= 1.0f && obs < 1.0e30f) kpr = 0.75f * kpr + 0.25f * obs;
                }
            }
            if (n > 0 && allc && ke >= kKeyNegInf) {
                pck = ke;
                pcv = key_f64(ke);
                Fc = Fe;
                have_c = true;
            } else {
                have_c = false;
                pck = 0;
                pcv = make_nan();  // compares false: dF stays 0 until a pivot is carried again
                Fc = 0;
            }
            if (sub == 0 && cell_ok) {
                thresh[static_cast<int64_t>(s) * ldo + cell] = th;
                seas[static_cast<int64_t>(s) * ldo + cell] = se;
            }
        }

        // occasional rendezvous of the workgroup's waves, which share every 128-B row segment
        // (see kernels_ring.hip: keeps the L2-miss traffic at the input size)
        if ((s & 63) == 63) __syncthreads();
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            e_cur[y] = e_nxt[y];
            e_nxt[y] = e_nn[y];
            x_cur[y] = x_nxt[y];
        }
    }
    if (stats != nullptr && lane == 0) {
        atomicAdd(&stats[0], static_cast<unsigned long long>(st_rows));
        atomicAdd(&stats[1], static_cast<unsigned long long>(st_count));
        atomicAdd(&stats[2], static_cast<unsigned long long>(st_extract));
        atomicAdd(&stats[3], static_cast<unsigned long long>(st_cold));
    }
}

namespace {
typedef void (*Ring64Kernel)(const double*, int64_t, int64_t, const uint32_t*, int32_t, const DevChunk*,
                             double, int, double*, double*, int64_t, unsigned long long*, const uint32_t*);
struct Ring64Entry { int w, yps; Ring64Kernel fn; };
#define XMHW_RK(W, Y) {W, Y, clim_ring_f64<W, Y>}
const Ring64Entry kRing64[] = {
    // (window half width, tracks per lane) on 16 lanes per cell; the 64-bit ring is 2 * YPS * (2w+1) <= 66 VGPRs
    XMHW_RK(5, 1), XMHW_RK(5, 2), XMHW_RK(5, 3),
    XMHW_RK(1, 1), XMHW_RK(1, 2), XMHW_RK(1, 3), XMHW_RK(2, 1), XMHW_RK(2, 2), XMHW_RK(2, 3),
    XMHW_RK(3, 1), XMHW_RK(3, 2), XMHW_RK(3, 3), XMHW_RK(4, 1), XMHW_RK(4, 2), XMHW_RK(4, 3),
    XMHW_RK(7, 1), XMHW_RK(7, 2), XMHW_RK(10, 1), XMHW_RK(15, 1),
};
#undef XMHW_RK
}  // namespace

int32_t ring64_pick_yps(int32_t w, int32_t ntracks) {
    int32_t best = 0;
    for (const auto& e : kRing64)
        if (e.w == w && e.yps * r64::kSubs >= ntracks && (best == 0 || e.yps < best)) best = e.yps;
    return best;
}

hipError_t launch_ring_f64(const double* ts, int64_t C, int64_t ld, const uint32_t* table,
                           int32_t step_min, const DevChunk* chunks, int32_t nchunks, int32_t w,
                           int32_t yps, double q, int negate, double* thresh, double* seas,
                           int64_t ldo, hipStream_t stream, unsigned long long* stats,
                           const uint32_t* run_flag) {
    Ring64Kernel fn = nullptr;
    for (const auto& e : kRing64)
        if (e.w == w && e.yps == yps) fn = e.fn;
    if (!fn) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    const int64_t cells_per_block = r64::kCells * r64::kWavesPerBlock;
    dim3 grid(static_cast<unsigned>((C + cells_per_block - 1) / cells_per_block),
              static_cast<unsigned>(nchunks));
    hipLaunchKernelGGL(fn, grid, dim3(64 * r64::kWavesPerBlock), 0, stream, ts, C, ld, table, step_min,
                       chunks, q, negate, thresh, seas, ldo, stats, run_flag);
    return hipGetLastError();
}

}  // namespace xmhw
