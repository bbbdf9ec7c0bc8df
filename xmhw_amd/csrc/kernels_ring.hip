// kernels_ring.hip -- the fast path: sliding-window percentile + mean with the
// pool held in VGPRs ("register ring"), hand-written for gfx950 wave64.
//
// Work decomposition
//   wave  = 8 cells x 8 "subs"; lane = sub*8 + cell_in_wave.
//   A sub owns YPS tracks (years); 8*YPS >= ntracks.  For every owned track
//   the lane keeps the R = 2w+1 samples of that year's current window as
//   order-preserving 32-bit keys in VGPRs: ring[YPS][R].
//   The wave walks the rows (distinct doy labels) in ascending order.  Per
//   row each track PUSHes one new sample into ring slot (step mod R) -- the
//   sample leaving the window sits exactly there -- so every input sample is
//   read from HBM once (plus 2w per track boundary), 32 contiguous bytes per
//   8 lanes, 128 B per 4-wave workgroup row.  No LDS, no barriers: a wave is
//   self-contained; the 8 subs of a cell combine through cross-lane shuffles.
//
// Per row, per cell
//   n      = number of valid pooled samples         (running per-track counts)
//   seas   = sum / n                                (running per-track f64 sums;
//                                                    exact for f32 input)
//   thresh = numpy linear quantile of the pool: needs order statistics lo and
//            lo+1.  Found by bracketing in key space: "count" passes give
//            F(p) = #{valid keys <= p} (v_cmp + v_addc per key); the bracket
//            (pl, F(pl) <= lo) / (ph, F(ph) > lo) starts from the previous
//            row's answer and closes by secant steps on the counts; once
//            F(pl) == lo (or ph == pl+1) one "extract" pass returns the two
//            smallest keys above the pivot (v_sub, v_med3, v_min per key).
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

constexpr int kSubs = 8;
constexpr int kCellsPerWave = 8;
constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// all-reduce over the 8 subs of a cell: lanes l ^ {8,16,32}
__device__ __forceinline__ uint32_t sub_sum(uint32_t v) {
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ double sub_sum(double v) {
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
// merge two ascending pairs, keep the two smallest
__device__ __forceinline__ void min2_merge(uint32_t& a1, uint32_t& a2, uint32_t b1, uint32_t b2) {
    const uint32_t hi = umax(a1, b1);
    a1 = umin(a1, b1);
    a2 = umin(hi, umin(a2, b2));
}
__device__ __forceinline__ void sub_min2(uint32_t& m1, uint32_t& m2) {
#pragma unroll
    for (int x = 8; x <= 32; x <<= 1) {
        const uint32_t b1 = __shfl_xor(m1, x);
        const uint32_t b2 = __shfl_xor(m2, x);
        min2_merge(m1, m2, b1, b2);
    }
}

__device__ __forceinline__ double key_value(uint32_t k) {
    return k ? static_cast<double>(key_f32(k)) : 0.0;
}

template <int W, int YPS>
__global__ __launch_bounds__(256) void clim_ring_f32(
    const float* __restrict__ ts, int64_t C, int64_t ld, const uint32_t* __restrict__ table,
    int32_t step_min, const DevChunk* __restrict__ chunks, double q, int negate,
    double* __restrict__ thresh, double* __restrict__ seas, int64_t ldo) {
    constexpr int R = 2 * W + 1;
    constexpr int NTP = kSubs * YPS;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane >> 3;
    const int64_t cell =
        (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave) * kCellsPerWave + (lane & 7);
    const bool cell_ok = cell < C;
    const DevChunk ch = chunks[blockIdx.y];
    const uint32_t* tab = table + sub * YPS;
    const float* col = ts + (cell_ok ? cell : 0);
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
    auto load_samples = [&](const uint32_t (&e)[YPS], float (&x)[YPS]) {
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            const uint32_t code = e[y] >> 1;
            float v = fnan;
            if (code >= 2 && cell_ok) v = col[static_cast<int64_t>(code - 2) * ld];
            x[y] = v;
        }
    };

    uint32_t e_cur[YPS], e_nxt[YPS];
    float x_cur[YPS];
    load_entries(ch.warm_start, e_cur);
    load_samples(e_cur, x_cur);
    if (ch.warm_start + 1 < ch.end) load_entries(ch.warm_start + 1, e_nxt);
    else {
#pragma unroll
        for (int y = 0; y < YPS; ++y) e_nxt[y] = make_entry(kCodeInvalid, false);
    }

    int m = (ch.warm_start - step_min) % R;
    uint32_t p_prev = 0;
    bool warm = false;
    float rho = 8192.0f;

    for (int32_t s = ch.warm_start; s < ch.end; ++s) {
        // ---- prefetch: samples of step s+1, table entries of step s+2 ------------
        float x_nxt[YPS];
        uint32_t e_nn[YPS];
        load_samples(e_nxt, x_nxt);
        if (s + 2 < ch.end) load_entries(s + 2, e_nn);
        else {
#pragma unroll
            for (int y = 0; y < YPS; ++y) e_nn[y] = make_entry(kCodeInvalid, false);
        }

        // ---- advance the rings -----------------------------------------------------
        uint32_t kin[YPS], kout[YPS];
        bool hold[YPS], counted[YPS];
        bool any_hold = false;
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            counted[y] = (e_cur[y] & 1u) != 0;
            hold[y] = (e_cur[y] >> 1) == kCodeHold;
            any_hold |= hold[y];
            float xv = x_cur[y];
            if (negate) xv = -xv;
            kin[y] = f32_key(xv);  // NaN / not loaded -> 0 (invalid)
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
            XMHW_RING_CASE(12) XMHW_RING_CASE(13) XMHW_RING_CASE(14)
            default: break;
        }
#undef XMHW_RING_CASE
        m = (m + 1 == R) ? 0 : m + 1;
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            if (!hold[y]) {
                tsum[y] += key_value(kin[y]);
                tsum[y] -= key_value(kout[y]);
                nval[y] += (kin[y] != 0 ? 1u : 0u) - (kout[y] != 0 ? 1u : 0u);
            }
        }
        if (__any(any_hold)) {
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
            uint32_t nl = 0, ncl = 0;
            double tl = 0.0;
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                nl += counted[y] ? nval[y] : 0u;
                ncl += counted[y] ? 1u : 0u;
                tl += counted[y] ? tsum[y] : 0.0;
            }
            const uint32_t n = sub_sum(nl);
            const uint32_t ninv = static_cast<uint32_t>(R) * sub_sum(ncl) - n;  // counted invalid keys
            const double total = sub_sum(tl);

            const uint32_t nn = n ? n : 1u;
            const double vi = static_cast<double>(nn - 1) * q;
            const double fl = floor(vi);
            const double g = vi - fl;
            const uint32_t lo = static_cast<uint32_t>(fl);
            const bool need2 = lo + 1 < nn;

            // F(p) = #{valid counted keys <= p}
            auto count_le = [&](uint32_t p) -> uint32_t {
                uint32_t c = 0;
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    uint32_t cy = 0;
#pragma unroll
                    for (int k = 0; k < R; ++k) cy += (ring[y][k] <= p) ? 1u : 0u;
                    c += counted[y] ? cy : 0u;
                }
                return sub_sum(c) - ninv;
            };

            if (!warm) {  // cold start of a chunk: begin at the pool mean
                p_prev = f32_key(static_cast<float>(total / static_cast<double>(nn)));
                if (p_prev == 0) p_prev = 0x80000000u;
            }
            uint32_t pl = 0, Fl = 0, ph = 0xFFFFFFFFu, Fh = nn;
            bool lreal = false, hreal = false;
            bool done = (n == 0);
            float grow = 1.0f;
            for (int it = 0;; ++it) {
                if (!done && (Fl == lo || ph - pl <= 1u)) done = true;
                if (__all(done)) break;
                const uint32_t room = ph - pl;  // >= 2 for lanes not done
                uint32_t off;
                if (it == 0) {
                    off = (p_prev - 1u) - pl;  // pl == 0 here
                } else if (it < 7 && lreal && hreal) {
                    const float frac = static_cast<float>(lo - Fl) / static_cast<float>(Fh - Fl);
                    off = static_cast<uint32_t>(static_cast<float>(room) * frac);
                } else if (it < 7 && lreal) {
                    const float st = static_cast<float>(lo - Fl) * rho * grow;
                    off = st < 2.0e9f ? static_cast<uint32_t>(st) : 2000000000u;
                    grow *= 2.0f;
                } else if (it < 7 && hreal) {
                    const float st = static_cast<float>(Fh - lo) * rho * grow;
                    const uint32_t back = st < 2.0e9f ? static_cast<uint32_t>(st) : 2000000000u;
                    off = room > back ? room - back : 1u;
                    grow *= 2.0f;
                } else {
                    off = room >> 1;
                }
                off = umax(1u, umin(off, room - 1u));
                const uint32_t p = done ? pl : pl + off;
                const uint32_t F = count_le(p);
                if (!done) {
                    if (F <= lo) {
                        if (!lreal) grow = 1.0f;
                        pl = p; Fl = F; lreal = true;
                    } else {
                        if (!hreal) grow = 1.0f;
                        ph = p; Fh = F; hreal = true;
                    }
                }
            }

            // extract the two smallest keys above the pivot
            const bool type_a = (Fl == lo);
            const uint32_t pe = type_a ? pl : ph;
            const uint32_t base = pe + 1u;
            uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                uint32_t a1 = 0xFFFFFFFFu, a2 = 0xFFFFFFFFu;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const uint32_t d = ring[y][k] - base;  // keys <= pe wrap to huge distances
                    a2 = umed3(a1, a2, d);
                    a1 = umin(a1, d);
                }
                if (!counted[y]) { a1 = 0xFFFFFFFFu; a2 = 0xFFFFFFFFu; }
                min2_merge(m1, m2, a1, a2);
            }
            sub_min2(m1, m2);
            const uint32_t k1 = base + m1, k2 = base + m2;
            uint32_t alo, ahi;
            if (type_a) {
                alo = k1;
                ahi = need2 ? k2 : k1;
            } else {
                alo = ph;
                ahi = (need2 && lo + 1u >= Fh) ? k1 : ph;
            }
            double th = make_nan(), se = make_nan();
            if (n > 0) {
                th = numpy_lerp(static_cast<double>(key_f32(alo)), static_cast<double>(key_f32(ahi)), g);
                se = total / static_cast<double>(n);
                p_prev = alo;
                warm = true;
                if (ahi > alo) {
                    const uint32_t gap = ahi - alo;
                    rho = 0.75f * rho + 0.25f * static_cast<float>(gap < (1u << 24) ? gap : (1u << 24));
                    rho = rho < 1.0f ? 1.0f : rho;
                }
            }
            if (sub == 0 && cell_ok) {
                thresh[static_cast<int64_t>(s) * ldo + cell] = th;
                seas[static_cast<int64_t>(s) * ldo + cell] = se;
            }
        }

        // ---- rotate the software pipeline --------------------------------------------
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            e_cur[y] = e_nxt[y];
            e_nxt[y] = e_nn[y];
            x_cur[y] = x_nxt[y];
        }
    }
}

// ---------------------------------------------------------------------------
// instantiation table
// ---------------------------------------------------------------------------
namespace {
typedef void (*RingKernel)(const float*, int64_t, int64_t, const uint32_t*, int32_t, const DevChunk*,
                           double, int, double*, double*, int64_t);
struct RingEntry { int w, yps; RingKernel fn; };
#define XMHW_RK(W, Y) {W, Y, clim_ring_f32<W, Y>}
const RingEntry kRing[] = {
    XMHW_RK(5, 1), XMHW_RK(5, 2), XMHW_RK(5, 3), XMHW_RK(5, 4), XMHW_RK(5, 5), XMHW_RK(5, 6),
    XMHW_RK(1, 1), XMHW_RK(1, 5), XMHW_RK(2, 3), XMHW_RK(2, 5), XMHW_RK(3, 4),
};
#undef XMHW_RK
RingKernel find_ring(int32_t w, int32_t yps) {
    for (const auto& e : kRing)
        if (e.w == w && e.yps == yps) return e.fn;
    return nullptr;
}
}  // namespace

bool ring_supported(int32_t w, int32_t yps, int elem_bytes) {
    return elem_bytes == 4 && find_ring(w, yps) != nullptr;
}

int32_t ring_pick_yps(int32_t w, int32_t ntracks, int elem_bytes) {
    if (elem_bytes != 4) return 0;
    int32_t best = 0;
    for (const auto& e : kRing)
        if (e.w == w && e.yps * kSubs >= ntracks && (best == 0 || e.yps < best)) best = e.yps;
    return best;
}

hipError_t launch_ring_f32(const float* ts, int64_t C, int64_t ld, const uint32_t* table,
                           int32_t step_min, const DevChunk* chunks, int32_t nchunks, int32_t w,
                           int32_t yps, double q, int negate, double* thresh, double* seas,
                           int64_t ldo, hipStream_t stream) {
    RingKernel fn = find_ring(w, yps);
    if (!fn) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    const int64_t cells_per_block = kCellsPerWave * kWavesPerBlock;
    dim3 grid(static_cast<unsigned>((C + cells_per_block - 1) / cells_per_block),
              static_cast<unsigned>(nchunks));
    hipLaunchKernelGGL(fn, grid, dim3(64 * kWavesPerBlock), 0, stream, ts, C, ld, table, step_min,
                       chunks, q, negate, thresh, seas, ldo);
    return hipGetLastError();
}

}  // namespace xmhw
