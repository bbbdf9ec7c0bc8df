// kernels_ring2.hip -- second-generation float32 ring kernel (round 2).
//
// Same decomposition as kernels_ring.hip (wave = 8 cells x 8 subs, a sub owns YPS tracks, the
// R = 2w+1 samples of every owned track's window live in VGPRs as order-preserving 32-bit keys,
// every input sample is read from HBM once) and the same exact selection of the two order
// statistics numpy's linear quantile needs.  What changed, and why (profiles/r1_pmc_sq.txt: the
// round-1 kernel is VALU-issue bound at 1,372 instructions per wave-row, 47 % of them per-row
// fixed cost):
//
//  * INVALID keys are 0xFFFFFFFF (above every real key) instead of 0: counts F(p) = #{keys <= p}
//    see valid samples only, the `ninv` bookkeeping of every probe is gone, and a padded track is
//    made invalid with one OR.
//  * FAST steps: the host marks the steps at which every real track pushes a valid sample and is
//    part of the pool (all but ~25 of the 376 steps of a daily axis).  On such a step, when no
//    NaN was loaded and no invalid key sits in the rings of the wave, nothing is decoded per
//    track: no hold / counted selects, no valid-count updates, one running float64 sum per lane
//    instead of one per track.  Everything else (calendar edges, Feb 29, NaN samples, chunk
//    warm-up) takes the GENERAL step, which is the round-1 logic.
//  * tracks are dealt to lanes y-major (track k -> sub k % 8, slot k / 8), so padding only ever
//    sits in the last slot of a lane.
//  * optional (template switches, measured separately -- DESIGN.md 3.1):
//      PROBE8  the bracket is closed on an 8-bit code ring (code = clamp((key - base) >> shift)):
//              one probe = 2 x v_sad_u8 per 4 keys (#{code < L} = (SAD(L) - SAD(L-1) + N) / 2)
//              instead of v_cmp + v_addc per key; the 32-bit count pass remains as the fallback.
//      SKIPX   the extraction pass skips the insertion network at ring positions where no lane
//              of the wave holds a key inside the target band (one v_cmp + scalar branch).
//
// Reference semantics restated: window_roll() (identify.py:184-209),
// calculate_thresh()/calculate_seas() without the Feb-29 step (identify.py:233-235, :263),
// coldSpells negation (xmhw.py:153-154).
#include "device_common.h"
#include "kernels.h"
#include "plan.h"

namespace xmhw {
namespace {

constexpr int kWaves2 = 4;
constexpr uint32_t kInv = 0xFFFFFFFFu;

template <int CTRL>
__device__ __forceinline__ uint32_t dpp2(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
constexpr int kR8 = 0x128, kR4 = 0x124, kR2 = 0x122;

__device__ __forceinline__ uint32_t cell_sum(uint32_t v) {
    v += dpp2<kR8>(v);
    v += dpp2<kR4>(v);
    v += dpp2<kR2>(v);
    return v;
}
template <int CTRL>
__device__ __forceinline__ double dpp2_f64(double v) {
    const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
    const uint32_t lo = dpp2<CTRL>(static_cast<uint32_t>(b));
    const uint32_t hi = dpp2<CTRL>(static_cast<uint32_t>(b >> 32));
    return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
}
__device__ __forceinline__ double cell_sum(double v) {
    v += dpp2_f64<kR8>(v);
    v += dpp2_f64<kR4>(v);
    v += dpp2_f64<kR2>(v);
    return v;
}
__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t minu(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t maxu(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t sad_u8(uint32_t a, uint32_t b, uint32_t acc) {
    return __builtin_amdgcn_sad_u8(a, b, acc);
}
__device__ __forceinline__ uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) {
    return __builtin_amdgcn_perm(hi, lo, sel);
}

// key of a non-NaN float: negmask = 0 (heat waves) or 0xFFFFFFFF (cold spells: key(-x) = ~key(x))
__device__ __forceinline__ uint32_t key_of_bits(uint32_t b, uint32_t negmask) {
    return b ^ (static_cast<uint32_t>(static_cast<int32_t>(b) >> 31) | 0x80000000u) ^ negmask;
}
// bits of the float a VALID key stands for (the negated sample under coldSpells)
__device__ __forceinline__ uint32_t bits_of_key(uint32_t k) {
    return k ^ (~static_cast<uint32_t>(static_cast<int32_t>(k) >> 31) | 0x80000000u);
}
// identity the optimiser cannot see through: keeps the rare paths (masked rows, infinite samples)
// from being merged with, or hoisted above, the per-row code
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ double value_of_key(uint32_t k) {   // 0 for an invalid key
    const float f = __uint_as_float(bits_of_key(k));
    return k == kInv ? 0.0 : static_cast<double>(f);
}

// J smallest (position-wise, ties repeated) distances above a pivot, ascending.
template <int J>
struct Top2 {
    uint32_t m[J];
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int i = 0; i < J; ++i) m[i] = 0xFFFFFFFFu;
    }
    __device__ __forceinline__ void insert(uint32_t d) {
#pragma unroll
        for (int i = J - 1; i >= 1; --i) m[i] = med3u(m[i - 1], m[i], d);
        m[0] = minu(m[0], d);
    }
    template <int CTRL>
    __device__ __forceinline__ void merge() {
        uint32_t b[J];
#pragma unroll
        for (int i = 0; i < J; ++i) b[i] = dpp2<CTRL>(m[i]);
#pragma unroll
        for (int i = 0; i < J; ++i) m[i] = minu(m[i], b[J - 1 - i]);
        constexpr int OFF = 8 - J;
#pragma unroll
        for (int d = 4; d >= 1; d >>= 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if ((k & d) == 0 && k >= OFF && k + d < 8) {
                    const uint32_t lo_ = minu(m[k - OFF], m[k - OFF + d]);
                    const uint32_t hi_ = maxu(m[k - OFF], m[k - OFF + d]);
                    m[k - OFF] = lo_;
                    m[k - OFF + d] = hi_;
                }
            }
        }
    }
    __device__ __forceinline__ void merge_cell() {
        merge<kR8>();
        merge<kR4>();
        merge<kR2>();
    }
    __device__ __forceinline__ uint32_t at(uint32_t j) const {
        // a chain of selects; the empty asm keeps the compiler from turning it into a dynamically
        // indexed array (which it would place in LDS: a store of all J entries + a dependent read)
        uint32_t r = m[0];
#pragma unroll
        for (int i = 1; i < J; ++i) {
            r = (j == static_cast<uint32_t>(i)) ? m[i] : r;
            asm volatile("" : "+v"(r));
        }
        return r;
    }
    __device__ __forceinline__ uint32_t count_below(uint32_t d) const {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < J; ++i) c += (m[i] < d) ? 1u : 0u;
        return c;
    }
};

constexpr int kJ2 = 5;
constexpr int kBudget2 = 6;

}  // namespace

// sflags[step]: bit 0 = SIMPLE (every real track pushes a valid sample and is counted; padded
// tracks push invalid).  ntracks = real tracks (tracks >= ntracks are padding).
template <int W, int YPS, bool PROBE8, bool SKIPX, bool STATS>
__global__ __launch_bounds__(256) void clim_ring2_f32(
    const float* __restrict__ ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* __restrict__ table,
    const uint32_t* __restrict__ sflags, int32_t step_min, const DevChunk* __restrict__ chunks, double q,
    int negate, int32_t ntracks, double* __restrict__ thresh, double* __restrict__ seas, int64_t ldo,
    unsigned long long* __restrict__ stats) {
    constexpr int R = 2 * W + 1;
    constexpr int SUBS = 8;
    constexpr int NTP = SUBS * YPS;
    constexpr int J = kJ2;
    constexpr uint32_t SLACK = J - 2;
    constexpr int NK = YPS * R;                 // keys per lane
    constexpr int NW = (NK + 3) / 4;            // code words per lane
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = (lane >> 1) & 7;
    const int cw = (lane & 1) | ((lane >> 4) << 1);
    const int64_t cell = (static_cast<int64_t>(blockIdx.x) * kWaves2 + wave) * 8 + cw;
    const bool cell_ok = cell < C;
    const DevChunk ch = chunks[blockIdx.y];
    const uint32_t* tab = table + sub;           // y-major: entry of slot y at tab[step * NTP + y * 8]
    const float* col = ts + (cell_ok ? cell : C - 1);
    const uint32_t negmask = negate ? 0xFFFFFFFFu : 0u;
    const uint32_t tmax = static_cast<uint32_t>(Tn - 1);
    // padding: only the last slot of a lane can be a padded track
    const bool padded_last = (YPS - 1) * SUBS + sub >= ntracks;
    const uint32_t padmask = padded_last ? 0xFFFFFFFFu : 0u;
    const uint32_t full_valid = static_cast<uint32_t>((padded_last ? YPS - 1 : YPS) * R);

    uint32_t ring[YPS][R];
#pragma unroll
    for (int y = 0; y < YPS; ++y)
#pragma unroll
        for (int k = 0; k < R; ++k) ring[y][k] = kInv;
    double lsum = 0.0;        // sum of the valid samples in this lane's rings (all tracks)
    uint32_t nval = 0;        // number of valid keys in this lane's rings (all tracks)

    auto load_entries = [&](int32_t s, uint32_t (&e)[YPS]) {
        const uint32_t* p = tab + static_cast<int64_t>(s - step_min) * NTP;
#pragma unroll
        for (int y = 0; y < YPS; ++y) e[y] = p[y * SUBS];
    };
    // unconditional loads: hold / invalid codes wrap to a huge index and are clamped to the last
    // row (a wasted but harmless read); what the sample means is decided when it is consumed
    auto load_samples = [&](const uint32_t (&e)[YPS], float (&x)[YPS]) {
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            const uint32_t t = minu((e[y] >> 1) - 2u, tmax);
            x[y] = col[static_cast<int64_t>(t) * ld];
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
    uint32_t pc = 0, Fc = 0;
    bool have_c = false;
    float kpr = 8192.0f;
    bool clean = false;       // wave-uniform: every lane's rings hold valid keys only
    uint32_t st_count = 0, st_extract = 0, st_rows = 0, st_cold = 0, st_fast = 0, st_probe8 = 0, st_rebase = 0;

    // PROBE8 state: 8-bit codes of the ring keys relative to (cbase, cshift); valid while have_code
    uint32_t codes[PROBE8 ? NW : 1];
    uint32_t cbase = 0, cshift = 0;
    bool have_code = false;
    float lpr = 2.0f;         // levels per rank near the target (carried)
    uint32_t Lc = 0;

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
        const uint32_t sf = __builtin_amdgcn_readfirstlane(sflags[s - step_min]);

        // ---- advance the rings -----------------------------------------------------
        uint32_t kin[YPS], kout[YPS];
        uint32_t cmask = (1u << YPS) - 1u;     // bit y: track y is part of this row's pool (per lane)
        bool allc = true;
        uint32_t dF = 0;
        // NaN among the loaded samples?  (the sum propagates NaN; inf - inf also lands here and
        // merely takes the general step)
        float xs = x_cur[0];
#pragma unroll
        for (int y = 1; y < YPS; ++y) xs += x_cur[y];
        const bool fast = (sf & 1u) && clean && !__any(xs != xs);
        if (fast) {
            if constexpr (STATS) ++st_fast;
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                kin[y] = key_of_bits(__float_as_uint(x_cur[y]), negmask);
            }
            kin[YPS - 1] |= padmask;
#define XMHW_R2_FAST(K)                                                    \
    case K:                                                                \
        if constexpr (K < R) {                                             \
            _Pragma("unroll") for (int y = 0; y < YPS; ++y) {              \
                kout[y] = ring[y][K < R ? K : 0];                          \
                ring[y][K < R ? K : 0] = kin[y];                           \
            }                                                              \
        }                                                                  \
        break;
            switch (m) {
                XMHW_R2_FAST(0) XMHW_R2_FAST(1) XMHW_R2_FAST(2) XMHW_R2_FAST(3) XMHW_R2_FAST(4)
                XMHW_R2_FAST(5) XMHW_R2_FAST(6) XMHW_R2_FAST(7) XMHW_R2_FAST(8) XMHW_R2_FAST(9)
                XMHW_R2_FAST(10) XMHW_R2_FAST(11) XMHW_R2_FAST(12) XMHW_R2_FAST(13) XMHW_R2_FAST(14)
                XMHW_R2_FAST(15) XMHW_R2_FAST(16) XMHW_R2_FAST(17) XMHW_R2_FAST(18) XMHW_R2_FAST(19)
                XMHW_R2_FAST(20) XMHW_R2_FAST(21) XMHW_R2_FAST(22) XMHW_R2_FAST(23) XMHW_R2_FAST(24)
                XMHW_R2_FAST(25) XMHW_R2_FAST(26) XMHW_R2_FAST(27) XMHW_R2_FAST(28) XMHW_R2_FAST(29)
                XMHW_R2_FAST(30)
                default: break;
            }
#undef XMHW_R2_FAST
            // running sum: + new samples - evicted samples (padded slot: both are masked to +0.0)
            double din = 0.0, dout = 0.0;
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                uint32_t bi = __float_as_uint(x_cur[y]) ^ (negmask & 0x80000000u);
                uint32_t bo = bits_of_key(kout[y]);
                if (y == YPS - 1) {
                    bi &= ~padmask;
                    bo &= ~padmask;
                }
                din += static_cast<double>(__uint_as_float(bi));
                dout += static_cast<double>(__uint_as_float(bo));
                if constexpr (!PROBE8) dF += (kin[y] <= pc ? 1u : 0u) - (kout[y] <= pc ? 1u : 0u);
            }
            lsum += din - dout;
        } else {
            bool hold[YPS];
            bool any_hold = false;
            cmask = 0;
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                const uint32_t code = e_cur[y] >> 1;
                cmask |= (e_cur[y] & 1u) << y;
                hold[y] = code == kCodeHold;
                any_hold |= hold[y];
                const float xv = x_cur[y];
                const bool ok = code >= 2u && cell_ok && xv == xv;
                kin[y] = ok ? key_of_bits(__float_as_uint(xv), negmask) : kInv;
            }
#define XMHW_R2_GEN(K)                                                     \
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
                XMHW_R2_GEN(0) XMHW_R2_GEN(1) XMHW_R2_GEN(2) XMHW_R2_GEN(3) XMHW_R2_GEN(4)
                XMHW_R2_GEN(5) XMHW_R2_GEN(6) XMHW_R2_GEN(7) XMHW_R2_GEN(8) XMHW_R2_GEN(9)
                XMHW_R2_GEN(10) XMHW_R2_GEN(11) XMHW_R2_GEN(12) XMHW_R2_GEN(13) XMHW_R2_GEN(14)
                XMHW_R2_GEN(15) XMHW_R2_GEN(16) XMHW_R2_GEN(17) XMHW_R2_GEN(18) XMHW_R2_GEN(19)
                XMHW_R2_GEN(20) XMHW_R2_GEN(21) XMHW_R2_GEN(22) XMHW_R2_GEN(23) XMHW_R2_GEN(24)
                XMHW_R2_GEN(25) XMHW_R2_GEN(26) XMHW_R2_GEN(27) XMHW_R2_GEN(28) XMHW_R2_GEN(29)
                XMHW_R2_GEN(30)
                default: break;
            }
#undef XMHW_R2_GEN
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                if (!hold[y]) {
                    lsum += value_of_key(kin[y]);
                    lsum -= value_of_key(kout[y]);
                    nval += (kin[y] != kInv ? 1u : 0u) - (kout[y] != kInv ? 1u : 0u);
                    dF += (kin[y] <= pc ? 1u : 0u) - (kout[y] <= pc ? 1u : 0u);
                } else {
                    kin[y] = kout[y];      // nothing changed in this track (code ring update below)
                }
            }
            if (__any(any_hold)) {
                // a held track did not advance: rotate its window one slot so that its oldest
                // sample sits where the next step's PUSH will land
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    const uint32_t last = ring[y][R - 1];
#pragma unroll
                    for (int k = R - 1; k >= 1; --k) ring[y][k] = hold[y] ? ring[y][k - 1] : ring[y][k];
                    ring[y][0] = hold[y] ? last : ring[y][0];
                }
                have_code = false;     // byte positions moved: rebuild the code ring
            }
            allc = cmask == (1u << YPS) - 1u;
            clean = !__any(nval != full_valid);
        }

        // ---- 8-bit code ring: the slot written this step ------------------------------
        if constexpr (PROBE8) {
            if (have_code) {
                // code = min((key -sat base) >> shift, 254); the byte of slot (y, m) is byte
                // (y*R + m) & 3 of word (y*R + m) >> 2 -- m is wave-uniform, so this is a scalar switch
#define XMHW_R2_CODE(K)                                                                          \
    case K:                                                                                      \
        if constexpr (K < R) {                                                                   \
            _Pragma("unroll") for (int y = 0; y < YPS; ++y) {                                    \
                const int pos = y * R + (K < R ? K : 0);                                          \
                const uint32_t c = minu(__builtin_elementwise_sub_sat(kin[y], cbase) >> cshift, 254u); \
                const uint32_t sel = pos % 4 == 0 ? 0x07060500u : pos % 4 == 1 ? 0x07060004u    \
                                         : pos % 4 == 2 ? 0x07000504u : 0x00060504u;             \
                codes[pos / 4] = perm_b32(codes[pos / 4], c, sel);                               \
            }                                                                                    \
        }                                                                                        \
        break;
                // m was not advanced yet: it still names the slot written above
                switch (m) {
                    XMHW_R2_CODE(0) XMHW_R2_CODE(1) XMHW_R2_CODE(2) XMHW_R2_CODE(3) XMHW_R2_CODE(4)
                    XMHW_R2_CODE(5) XMHW_R2_CODE(6) XMHW_R2_CODE(7) XMHW_R2_CODE(8) XMHW_R2_CODE(9)
                    XMHW_R2_CODE(10) XMHW_R2_CODE(11) XMHW_R2_CODE(12) XMHW_R2_CODE(13) XMHW_R2_CODE(14)
                    XMHW_R2_CODE(15) XMHW_R2_CODE(16) XMHW_R2_CODE(17) XMHW_R2_CODE(18) XMHW_R2_CODE(19)
                    XMHW_R2_CODE(20) XMHW_R2_CODE(21) XMHW_R2_CODE(22) XMHW_R2_CODE(23) XMHW_R2_CODE(24)
                    XMHW_R2_CODE(25) XMHW_R2_CODE(26) XMHW_R2_CODE(27) XMHW_R2_CODE(28) XMHW_R2_CODE(29)
                    XMHW_R2_CODE(30)
                    default: break;
                }
#undef XMHW_R2_CODE
            }
        }
        m = (m + 1 == R) ? 0 : m + 1;

        // ---- select + output (not during warm-up) ---------------------------------
        if (s >= ch.begin) {
            const bool wallc = __all(allc);
            uint32_t n;
            double total;
            if (wallc) {
                n = cell_sum(nval);
                total = cell_sum(lsum);
            } else {
                // Feb-29 style rows: only the counted tracks are pooled; count and sum them afresh
                uint32_t nl = 0;
                double tl = 0.0;
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    uint32_t cy = 0;
                    double ty = 0.0;
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const uint32_t key = opaque(ring[y][k]);
                        cy += key != kInv ? 1u : 0u;
                        ty += value_of_key(key);
                    }
                    const bool cnt = (cmask >> y) & 1u;
                    nl += cnt ? cy : 0u;
                    tl += cnt ? ty : 0.0;
                }
                n = cell_sum(nl);
                total = cell_sum(tl);
            }
            if (__any(!(fabs(total) <= 1.7976931348623157e308))) {
                // an infinite sample went through the running sum (inf - inf = NaN once it leaves):
                // rebuild it from the rings
                double t = 0.0, tl = 0.0;
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    double ty = 0.0;
#pragma unroll
                    for (int k = 0; k < R; ++k) ty += value_of_key(opaque(ring[y][k]));
                    t += ty;
                    tl += ((cmask >> y) & 1u) ? ty : 0.0;
                }
                lsum = t;
                total = cell_sum(tl);
            }
            Fc += cell_sum(dF);

            const uint32_t nn = n ? n : 1u;
            const double vi = static_cast<double>(nn - 1) * q;
            const double fl = floor(vi);
            const double g = vi - fl;
            const uint32_t lo = static_cast<uint32_t>(fl);
            const bool need2 = lo + 1 < nn;

            // F(p) = #{counted ring keys <= p}; p < 0xFFFFFFFF never counts an invalid key
            auto count_le = [&](uint32_t p) -> uint32_t {
                uint32_t c = 0;
                if (wallc) {
#pragma unroll
                    for (int y = 0; y < YPS; ++y)
#pragma unroll
                        for (int k = 0; k < R; ++k) c += (ring[y][k] <= p) ? 1u : 0u;
                } else {
#pragma unroll
                    for (int y = 0; y < YPS; ++y) {
                        uint32_t cy = 0;
#pragma unroll
                        for (int k = 0; k < R; ++k) cy += (opaque(ring[y][k]) <= p) ? 1u : 0u;
                        c += ((cmask >> y) & 1u) ? cy : 0u;
                    }
                }
                return cell_sum(c);
            };

            uint32_t pl = 0, Fl = 0, ph = 0xFFFFFFFFu, Fh = nn;
            bool lreal = false, hreal = false;
            float grow = 1.0f;
            bool resolved = (n == 0);
            uint32_t p_first = 0;
            int32_t rank_gap = 0;
            bool settled8 = false;     // PROBE8: the bracket came out of the code ring, ready for extraction

            if constexpr (PROBE8) {
                // ---------- close the bracket on the 8-bit code ring ----------------------
                // (re)build the code ring when there is none, or when the previous row's level came
                // close to an edge of the 254-level window
                const bool usable = wallc;     // masked rows take the 32-bit path
                if (usable) {
                    const bool want = have_c;                    // an estimate of the target exists
                    bool rebase = want && (!have_code || Lc < 24u || Lc > 230u);
                    if (__any(rebase) && __all(want || n == 0)) {
                        // window: 254 levels of 2^shift keys; aim at ~2 levels per rank, the carried
                        // pivot at level 96 (the target drifts either way by ~13 ranks a day)
                        const float lw = fmaxf(kpr * 0.5f, 1.0f);
                        uint32_t sh = 31u - static_cast<uint32_t>(__builtin_clz(static_cast<uint32_t>(lw)));
                        sh = minu(sh, 23u);
                        const uint32_t span = 96u << sh;
                        uint32_t nb = pc > span ? pc - span : 0u;
                        nb = minu(nb, 0xFFFFFFFFu - (256u << sh));
                        cbase = nb;
                        cshift = sh;
#pragma unroll
                        for (int wd = 0; wd < NW; ++wd) {
                            uint32_t word = 0;
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                const int pos = wd * 4 + b;
                                uint32_t c = 255u;
                                if (pos < NK)
                                    c = minu(__builtin_elementwise_sub_sat(ring[pos / R][pos % R], cbase) >> cshift, 254u);
                                word |= c << (8 * b);
                            }
                            codes[wd] = word;
                        }
                        have_code = true;
                        Lc = 96u;
                        if constexpr (STATS) ++st_rebase;
                    }
                    if (__all(have_code || n == 0)) {
                        // cum(L) = #{code < L} = #{key < cbase + (L << cshift)}, exact for 1 <= L <= 254
                        auto cum8 = [&](uint32_t L) -> uint32_t {
                            const uint32_t b1 = L * 0x01010101u, b0 = b1 - 0x01010101u;
                            uint32_t a1 = 0, a0 = 0;
#pragma unroll
                            for (int wd = 0; wd < NW; ++wd) {
                                a1 = sad_u8(codes[wd], b1, a1);
                                a0 = sad_u8(codes[wd], b0, a0);
                            }
                            const uint32_t d = cell_sum(a1 - a0);
                            return (d + static_cast<uint32_t>(8 * 4 * NW)) >> 1;
                        };
                        uint32_t Ll = 0, Cl = 0, Lh = 255u, Ch = nn;     // cum(Ll) <= lo < cum(Lh); ends virtual
                        uint32_t L = minu(maxu(Lc, 1u), 254u);
                        bool done = (n == 0), fail = false;
                        const float aim = static_cast<float>(lo) - 0.5f * static_cast<float>(SLACK);
                        for (int it = 0; it < 12; ++it) {
                            const uint32_t cu = cum8(done ? 1u : L);
                            if constexpr (STATS) ++st_probe8;
                            if (!done) {
                                if (cu <= lo) { Ll = L; Cl = cu; } else { Lh = L; Ch = cu; }
                                if (Ll >= 1u && lo - Cl <= SLACK) done = true;          // window hit
                                else if (Lh - Ll <= 1u) { done = true; fail = true; }   // a level holds > J-1 keys
                                else {
                                    // secant in level space from the probed end, with the carried slope
                                    const bool both = Ll >= 1u && Lh <= 254u;
                                    const float slope = both ? static_cast<float>(Lh - Ll) *
                                                                   __builtin_amdgcn_rcpf(static_cast<float>(Ch - Cl))
                                                             : lpr;
                                    const bool from_l = cu <= lo;
                                    const float ranks = from_l ? aim - static_cast<float>(Cl) : static_cast<float>(Ch) - aim;
                                    float stf = fmaxf(ranks * slope, 1.0f);
                                    uint32_t st = static_cast<uint32_t>(fminf(stf, 255.0f));
                                    if (it >= 6) st = maxu((Lh - Ll) >> 1, 1u);
                                    uint32_t Ln = from_l ? Ll + st : (Lh > st ? Lh - st : 0u);
                                    Ln = minu(maxu(Ln, Ll + 1u), Lh - 1u);
                                    L = Ln;
                                }
                            }
                            if (__all(done)) break;
                        }
                        // outside the window (Ll == 0: target below level 1; Lh == 255 with Ll == 254:
                        // above it) or an overfull level: the 32-bit path finishes from this bracket
                        if (n != 0) {
                            if (Ll >= 1u) {
                                pl = cbase + (Ll << cshift) - 1u; Fl = Cl; lreal = true;
                            }
                            if (Lh <= 254u) {
                                ph = cbase + (Lh << cshift) - 1u; Fh = Ch; hreal = true;
                            }
                            settled8 = done && !fail && Ll >= 1u;
                            if (settled8) {
                                // carry: levels per rank seen on this row, and the level to start from
                                lpr = 0.75f * lpr + 0.25f * fminf(fmaxf(static_cast<float>(Ll > Lc ? Ll - Lc : Lc - Ll) *
                                                                         __builtin_amdgcn_rcpf(13.0f), 0.25f), 8.0f);
                                Lc = Ll;
                            } else {
                                have_code = false;     // next row rebuilds the window around the new answer
                            }
                        }
                        (void)fail;
                    }
                }
                p_first = pc;
            } else {
                const bool use_c = have_c && wallc;
                uint32_t p0 = pc, F0 = 0;
                if (use_c) F0 = Fc;
                if (!__all(use_c || n == 0)) {
                    uint32_t pm = key_of_bits(__float_as_uint(static_cast<float>(total / static_cast<double>(nn))), 0u);
                    if (!use_c) p0 = have_c ? pc : pm;
                    const uint32_t Fr = count_le(minu(p0, 0xFFFFFFFEu));
                    if (!use_c) F0 = Fr;
                    if constexpr (STATS) ++st_cold;
                }
                if (p0 != 0 && p0 < 0xFFFFFFFEu) {
                    if (F0 <= lo) { pl = p0; Fl = F0; lreal = true; }
                    else { ph = p0; Fh = F0; hreal = true; }
                }
                p_first = p0;
                rank_gap = static_cast<int32_t>(lo) - static_cast<int32_t>(F0);
            }
            const float aim = static_cast<float>(lo) - 0.5f * static_cast<float>(SLACK);

            uint32_t alo = 0, ahi = 0, pe = 0, Fe = 0;
            int budget = kBudget2;
            for (;;) {
                // ---- 32-bit count passes until every cell can be settled by one extraction ----
                for (int it = 0;; ++it) {
                    const bool settle = resolved || (lreal && lo - Fl <= SLACK) || (ph - pl <= 1u) ||
                                        (!lreal && lo <= SLACK && Fl == 0);
                    if (__all(settle) || it >= budget) break;
                    const uint32_t room = ph - pl;
                    const bool both = lreal && hreal;
                    const float roomf = static_cast<float>(room);
                    const float slope = both ? roomf * __builtin_amdgcn_rcpf(static_cast<float>(Fh - Fl))
                                             : kpr * grow;
                    const float ranks = (lreal || !hreal) ? aim - static_cast<float>(Fl) : static_cast<float>(Fh) - aim;
                    float stf = fminf(fmaxf(ranks * slope, 1.0f), 2.0e9f);
                    stf = (lreal || !hreal) ? stf : roomf - stf;
                    stf = fminf(fmaxf(stf, 1.0f), 4.0e9f);
                    uint32_t off = (it < 5) ? static_cast<uint32_t>(stf) : (room >> 1);
                    grow = both ? grow : grow * 2.0f;
                    off = maxu(1u, minu(off, room - 1u));
                    const uint32_t p = settle ? pl : pl + off;
                    const uint32_t F = count_le(p);
                    if constexpr (STATS) ++st_count;
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
                Top2<J> top;
                top.reset();
                if (wallc) {
                    if constexpr (SKIPX) {
                        // band: keys within (px, px + width] can be among the J smallest; positions at
                        // which no lane of the wave holds such a key skip the insertion network.
                        // width is a guess (8 ranks' worth of keys); the result is checked below.
                        const uint32_t width = static_cast<uint32_t>(fminf(kpr * 8.0f, 1.0e9f));
#pragma unroll
                        for (int y = 0; y < YPS; ++y)
#pragma unroll
                            for (int k = 0; k < R; ++k) {
                                const uint32_t d = ring[y][k] - base;
                                if (__any(d <= width)) top.insert(d);
                            }
                        top.merge_cell();
                        // valid iff the entries that will be read lie inside the band (every key of
                        // the band was inserted); otherwise redo the pass in full
                        const uint32_t jn = window ? (lo - Fl) + (need2 ? 1u : 0u) : 0u;
                        const bool bad = !resolved && (top.at(minu(jn, J - 1u)) > width || !window);
                        if (__any(bad)) {
                            top.reset();
#pragma unroll
                            for (int y = 0; y < YPS; ++y)
#pragma unroll
                                for (int k = 0; k < R; ++k) top.insert(ring[y][k] - base);
                            top.merge_cell();
                            if constexpr (STATS) ++st_extract;
                        }
                    } else {
#pragma unroll
                        for (int y = 0; y < YPS; ++y)
#pragma unroll
                            for (int k = 0; k < R; ++k) top.insert(ring[y][k] - base);
                        top.merge_cell();
                    }
                } else {
#pragma unroll
                    for (int y = 0; y < YPS; ++y)
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            const uint32_t d = opaque(ring[y][k]) - base;
                            top.insert(((cmask >> y) & 1u) ? d : 0xFFFFFFFFu);
                        }
                    top.merge_cell();
                }
                if constexpr (STATS) ++st_extract;
                if (!resolved) {
                    if (window) {
                        const uint32_t j = lo - Fl;
                        alo = base + top.at(j);
                        ahi = need2 ? base + top.at(j + 1u) : alo;
                        pe = pl; Fe = Fl;
                        resolved = true;
                    } else if (adjacent) {
                        alo = ph;
                        ahi = (need2 && lo + 1u >= Fh) ? base + top.m[0] : ph;
                        pe = ph; Fe = Fh;
                        resolved = true;
                    }
                }
                if (__all(resolved)) break;
                // ---- repair (tie-heavy data): count at the largest extracted key ---------
                const uint32_t dj = top.m[J - 1];
                const uint32_t pj = base + dj;
                const uint32_t Fj = count_le(resolved ? pl : pj);
                if constexpr (STATS) ++st_count;
                if (!resolved) {
                    if (Fj <= lo) {
                        pl = pj; Fl = Fj; lreal = true;
                    } else {
                        ph = pj; Fh = Fj; hreal = true;
                        Fl = Fl + top.count_below(dj);
                        pl = pj - 1u;
                        lreal = true;
                    }
                }
                budget = 2;
            }
            (void)settled8;

            if constexpr (STATS) ++st_rows;
            double th = make_nan(), se = make_nan();
            if (n > 0) {
                th = numpy_lerp(static_cast<double>(__uint_as_float(bits_of_key(alo))),
                                static_cast<double>(__uint_as_float(bits_of_key(ahi))), g);
                se = total / static_cast<double>(n);
                if constexpr (!PROBE8) {
                    if (rank_gap > 1 || rank_gap < -1) {
                        const float obs = (static_cast<float>(alo) - static_cast<float>(p_first)) *
                                          __builtin_amdgcn_rcpf(static_cast<float>(rank_gap));
                        if (obs >= 1.0f && obs < 1.0e8f) kpr = 0.75f * kpr + 0.25f * obs;
                    }
                } else {
                    // keys per rank from the extracted neighbours (local density at the target)
                    const float obs = static_cast<float>(ahi - alo);
                    if (need2 && obs >= 1.0f && obs < 1.0e8f) kpr = 0.9f * kpr + 0.1f * obs;
                }
            }
            if (n > 0 && wallc) {
                pc = pe;
                Fc = Fe;
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

        if ((s & 63) == 63) __syncthreads();
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            e_cur[y] = e_nxt[y];
            e_nxt[y] = e_nn[y];
            x_cur[y] = x_nxt[y];
        }
    }
    if (STATS && stats != nullptr && lane == 0) {
        atomicAdd(&stats[0], static_cast<unsigned long long>(st_rows));
        atomicAdd(&stats[1], static_cast<unsigned long long>(st_count));
        atomicAdd(&stats[2], static_cast<unsigned long long>(st_extract));
        atomicAdd(&stats[3], static_cast<unsigned long long>(st_cold));
        atomicAdd(&stats[4], static_cast<unsigned long long>(st_fast));
        atomicAdd(&stats[5], static_cast<unsigned long long>(st_probe8));
        atomicAdd(&stats[6], static_cast<unsigned long long>(st_rebase));
    }
}

// ---------------------------------------------------------------------------
namespace {
typedef void (*Ring2Kernel)(const float*, int64_t, int64_t, int64_t, const uint32_t*, const uint32_t*, int32_t,
                            const DevChunk*, double, int, int32_t, double*, double*, int64_t, unsigned long long*);
struct Ring2Entry { int w, yps, variant; Ring2Kernel fn, fn_stats; };
// variant: bit 0 = PROBE8, bit 1 = SKIPX; the _stats twin carries the debug pass counters
#define XMHW_R2V(W, Y, V) {W, Y, V, clim_ring2_f32<W, Y, ((V) & 1) != 0, ((V) & 2) != 0, false>, \
                           clim_ring2_f32<W, Y, ((V) & 1) != 0, ((V) & 2) != 0, true>}
#define XMHW_R2(W, Y) XMHW_R2V(W, Y, 0), XMHW_R2V(W, Y, 1), XMHW_R2V(W, Y, 2), XMHW_R2V(W, Y, 3)
const Ring2Entry kRing2[] = {
    XMHW_R2(5, 3), XMHW_R2(5, 4), XMHW_R2(5, 5),
};
#undef XMHW_R2
#undef XMHW_R2V
const Ring2Entry* find_ring2(int32_t w, int32_t yps, int32_t variant) {
    for (const auto& e : kRing2)
        if (e.w == w && e.yps == yps && e.variant == variant) return &e;
    return nullptr;
}
}  // namespace

int32_t ring2_pick_yps(int32_t w, int32_t ntracks) {
    int32_t best = 0;
    for (const auto& e : kRing2)
        if (e.w == w && e.variant == 0 && e.yps * 8 >= ntracks && (best == 0 || e.yps < best)) best = e.yps;
    // padding may only sit in the last slot of a lane
    if (best && (best - 1) * 8 >= ntracks) return 0;
    return best;
}

hipError_t launch_ring2_f32(const float* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                            const uint32_t* sflags, int32_t step_min, const DevChunk* chunks, int32_t nchunks,
                            int32_t w, int32_t yps, int32_t ntracks, int32_t variant, double q, int negate,
                            double* thresh, double* seas, int64_t ldo, hipStream_t stream,
                            unsigned long long* stats) {
    const Ring2Entry* e = find_ring2(w, yps, variant);
    if (!e) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    const int64_t cells_per_block = 8 * kWaves2;
    dim3 grid(static_cast<unsigned>((C + cells_per_block - 1) / cells_per_block), static_cast<unsigned>(nchunks));
    hipLaunchKernelGGL(stats ? e->fn_stats : e->fn, grid, dim3(64 * kWaves2), 0, stream, ts, C, ld, Tn, table, sflags, step_min, chunks, q,
                       negate, ntracks, thresh, seas, ldo, stats);
    return hipGetLastError();
}

}  // namespace xmhw
