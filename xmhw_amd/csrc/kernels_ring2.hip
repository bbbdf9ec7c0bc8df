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
//    (a second switch, an extraction pass that skips the insertion network at ring positions where
//    no lane of the wave holds a key inside the target band, was measured and removed: 158 ms
//    against 141 ms without it, profiles/r2_ring2_first_variants.jsonl -- with 64 lanes sharing an
//    instruction a position is skipped less than half of the time and the test costs as much as it saves)
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
constexpr int kR8 = 0x128, kR4 = 0x124, kR2 = 0x122, kR1 = 0x121;      // row_ror:8 / 4 / 2 / 1

// lanes of a cell sit in one 16-lane DPP row: 16 subs (one cell per row, 4 cells per wave), 8 subs at
// stride 2 (8 cells per wave) or 4 subs at stride 4 (16 cells per wave); the all-reduce is 4, 3 or 2 row
// rotations
template <int SUBS>
__device__ __forceinline__ uint32_t cell_sum(uint32_t v) {
    v += dpp2<kR8>(v);
    v += dpp2<kR4>(v);
    if constexpr (SUBS >= 8) v += dpp2<kR2>(v);
    if constexpr (SUBS == 16) v += dpp2<kR1>(v);
    return v;
}
template <int CTRL>
__device__ __forceinline__ double dpp2_f64(double v) {
    const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
    const uint32_t lo = dpp2<CTRL>(static_cast<uint32_t>(b));
    const uint32_t hi = dpp2<CTRL>(static_cast<uint32_t>(b >> 32));
    return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
}
template <int SUBS>
__device__ __forceinline__ double cell_sum(double v) {
    v += dpp2_f64<kR8>(v);
    v += dpp2_f64<kR4>(v);
    if constexpr (SUBS >= 8) v += dpp2_f64<kR2>(v);
    if constexpr (SUBS == 16) v += dpp2_f64<kR1>(v);
    return v;
}
template <int SUBS>
__device__ __forceinline__ uint32_t cell_max(uint32_t v) {
    uint32_t o = dpp2<kR8>(v); v = v > o ? v : o;
    o = dpp2<kR4>(v); v = v > o ? v : o;
    if constexpr (SUBS >= 8) { o = dpp2<kR2>(v); v = v > o ? v : o; }
    if constexpr (SUBS == 16) { o = dpp2<kR1>(v); v = v > o ? v : o; }
    return v;
}
template <int SUBS>
__device__ __forceinline__ uint32_t cell_min(uint32_t v) {
    uint32_t o = dpp2<kR8>(v); v = v < o ? v : o;
    o = dpp2<kR4>(v); v = v < o ? v : o;
    if constexpr (SUBS >= 8) { o = dpp2<kR2>(v); v = v < o ? v : o; }
    if constexpr (SUBS == 16) { o = dpp2<kR1>(v); v = v < o ? v : o; }
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
__device__ __forceinline__ uint32_t sad_u16(uint32_t a, uint32_t b, uint32_t acc) {
    return __builtin_amdgcn_sad_u16(a, b, acc);
}
__device__ __forceinline__ uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) {
    return __builtin_amdgcn_perm(hi, lo, sel);
}

// key of a non-NaN float: negmask = 0 (heat waves) or 0xFFFFFFFF (cold spells: key(-x) = ~key(x))
__device__ __forceinline__ uint32_t key_of_bits(uint32_t b, uint32_t negmask) {
    return b ^ (static_cast<uint32_t>(static_cast<int32_t>(b) >> 31) | 0x80000000u) ^ negmask;
}
// Ring registers are only ever modified through these two: the output is tied to the input register,
// so the register allocator keeps every ring element in ONE register for the whole kernel.  (With
// plain assignments it split the live ranges around the rare rotation path and copied all 55 ring
// registers to a second set and back on every row.)
__device__ __forceinline__ void ring_set(uint32_t& slot, uint32_t v) {
    asm volatile("v_mov_b32 %0, %1" : "+v"(slot) : "v"(v));
}
__device__ __forceinline__ void ring_sel(uint32_t& slot, uint32_t other, unsigned long long take_other) {
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(slot) : "v"(other), "s"(take_other));
}
// arithmetic shift kept as a shift (the compiler turns `x >> 31` feeding a bitwise op into
// v_cmp + v_cndmask, two quarter-rate instructions with a wait state between them)
__device__ __forceinline__ uint32_t ashr31(uint32_t v) {
    uint32_t r;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(v));
    return r;
}
// bits of the float a VALID key stands for (the negated sample under coldSpells)
__device__ __forceinline__ uint32_t bits_of_key(uint32_t k) {
    return k ^ (~ashr31(k) | 0x80000000u);
}
// c + #{r[i] <= p}, 11 keys.  Hand-scheduled: a v_cmp that writes an SGPR pair needs two wait states
// before a VALU instruction may read it; the compiler pads every pair with s_nop (167 issue slots
// for 55 keys), here three SGPR pairs rotate so that compare i is consumed four slots later
// (22 slots per 11 keys, no s_nop).
template <class RingT>
__device__ __forceinline__ uint32_t count_le11(const RingT& r, uint32_t p, uint32_t& c, uint32_t d) {
    // two accumulators (c, d) so that consecutive v_addc do not depend on each other
    unsigned long long s0, s1, s2, sd;
    asm("v_cmp_le_u32_e64 %[s0], %[k0], %[p]\n\t"
        "v_cmp_le_u32_e64 %[s1], %[k1], %[p]\n\t"
        "v_cmp_le_u32_e64 %[s2], %[k2], %[p]\n\t"
        "v_addc_co_u32_e64 %[c], %[sd], %[c], 0, %[s0]\n\t"
        "v_cmp_le_u32_e64 %[s0], %[k3], %[p]\n\t"
        "v_addc_co_u32_e64 %[d], %[sd], %[d], 0, %[s1]\n\t"
        "v_cmp_le_u32_e64 %[s1], %[k4], %[p]\n\t"
        "v_addc_co_u32_e64 %[c], %[sd], %[c], 0, %[s2]\n\t"
        "v_cmp_le_u32_e64 %[s2], %[k5], %[p]\n\t"
        "v_addc_co_u32_e64 %[d], %[sd], %[d], 0, %[s0]\n\t"
        "v_cmp_le_u32_e64 %[s0], %[k6], %[p]\n\t"
        "v_addc_co_u32_e64 %[c], %[sd], %[c], 0, %[s1]\n\t"
        "v_cmp_le_u32_e64 %[s1], %[k7], %[p]\n\t"
        "v_addc_co_u32_e64 %[d], %[sd], %[d], 0, %[s2]\n\t"
        "v_cmp_le_u32_e64 %[s2], %[k8], %[p]\n\t"
        "v_addc_co_u32_e64 %[c], %[sd], %[c], 0, %[s0]\n\t"
        "v_cmp_le_u32_e64 %[s0], %[k9], %[p]\n\t"
        "v_addc_co_u32_e64 %[d], %[sd], %[d], 0, %[s1]\n\t"
        "v_cmp_le_u32_e64 %[s1], %[k10], %[p]\n\t"
        "v_addc_co_u32_e64 %[c], %[sd], %[c], 0, %[s2]\n\t"
        "v_addc_co_u32_e64 %[d], %[sd], %[d], 0, %[s0]\n\t"
        "v_addc_co_u32_e64 %[c], %[sd], %[c], 0, %[s1]"
        : [c] "+v"(c), [d] "+v"(d), [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [sd] "=&s"(sd)
        : [k0] "v"(r[0]), [k1] "v"(r[1]), [k2] "v"(r[2]), [k3] "v"(r[3]), [k4] "v"(r[4]), [k5] "v"(r[5]),
          [k6] "v"(r[6]), [k7] "v"(r[7]), [k8] "v"(r[8]), [k9] "v"(r[9]), [k10] "v"(r[10]), [p] "v"(p));
    return d;
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

// float64 samples as 64-bit order-preserving keys, split into a HIGH word (sign, exponent, 20 mantissa
// bits: what the selection works on) and a LOW word (the remaining 32 mantissa bits: looked at only when
// the two order statistics have been found).  negmask = 0 or 0xFFFFFFFF as for the float32 keys.
__device__ __forceinline__ void key64_of(double v, uint32_t negmask, uint32_t& hi, uint32_t& lo) {
    const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
    const uint32_t bh = static_cast<uint32_t>(b >> 32), bl = static_cast<uint32_t>(b);
    const uint32_t sgn = ashr31(bh);                         // all ones for a negative sample
    hi = bh ^ (sgn | 0x80000000u) ^ negmask;
    lo = bl ^ sgn ^ negmask;
}
// the double a VALID key pair stands for (the negated sample under coldSpells)
__device__ __forceinline__ double double_of_key64(uint32_t hi, uint32_t lo) {
    const uint32_t sgn = ~ashr31(hi);                        // all ones if the value is negative
    const uint32_t bh = hi ^ (sgn | 0x80000000u), bl = lo ^ sgn;
    return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(bh) << 32) | bl));
}
__device__ __forceinline__ double value_of_key64(uint32_t hi, uint32_t lo) {      // 0 for an invalid key
    return hi == kInv ? 0.0 : double_of_key64(hi, lo);
}

// The smallest distances above a pivot, ascending (ties repeated).  Every lane keeps the J smallest of ITS
// keys (insert: J instructions per key); the lanes of a cell then merge into the JM >= J smallest of the
// cell.  With JM > J the merged list can be longer than what one lane contributes: an entry is EXACT (the
// true order statistic of the whole pool) as long as it does not exceed the cell's horizon = the smallest
// J-th entry of any lane, because every key a lane did not list is at least that lane's J-th one.  A
// lane's J-th smallest key above the pivot is on average the cell's (J x lanes)-th, so the horizon lies far
// beyond the JM-th entry except when one lane happens to hold J of the cell's JM nearest keys (about 1 % of
// the rows for J = 5, JM = 8, 8 lanes); the caller checks and takes the repair path then.
template <int J, int JM = J>
struct Top2 {
    static_assert(JM >= J && JM <= 8, "merged width: J..8");
    uint32_t m[JM];
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int i = 0; i < JM; ++i) m[i] = 0xFFFFFFFFu;
    }
    __device__ __forceinline__ void insert(uint32_t d) {
#pragma unroll
        for (int i = J - 1; i >= 1; --i) m[i] = med3u(m[i - 1], m[i], d);
        m[0] = minu(m[0], d);
    }
    // the cell's horizon (call BEFORE merge_cell): min over the cell's lanes of the J-th listed distance
    template <int SUBS>
    __device__ __forceinline__ uint32_t horizon() const {
        uint32_t g = m[J - 1];
        g = minu(g, dpp2<kR8>(g));
        g = minu(g, dpp2<kR4>(g));
        if constexpr (SUBS >= 8) g = minu(g, dpp2<kR2>(g));
        if constexpr (SUBS == 16) g = minu(g, dpp2<kR1>(g));
        return g;
    }
    // FIRST: both lists still have their entries J..JM-1 at "infinity" (nothing merged yet), which the
    // compiler cannot know: those compares and lane exchanges are left out by hand
    template <int CTRL, bool FIRST = false>
    __device__ __forceinline__ void merge() {
        constexpr int N = FIRST ? J : JM;        // live entries of the partner's list
        uint32_t b[N];
#pragma unroll
        for (int i = 0; i < N; ++i) b[i] = dpp2<CTRL>(m[i]);
#pragma unroll
        for (int i = 0; i < JM; ++i) {
            const int k = JM - 1 - i;            // partner entry in the bitonic half-cleaner
            if (i < N && k < N) m[i] = minu(m[i], b[k]);
            else if (k < N) m[i] = b[k];         // own entry is infinite
        }                                        // (partner entry infinite: own entry stays)
        constexpr int OFF = 8 - JM;
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
    template <int SUBS>
    __device__ __forceinline__ void merge_cell() {
        merge<kR8, true>();
        merge<kR4>();
        if constexpr (SUBS >= 8) merge<kR2>();
        if constexpr (SUBS == 16) merge<kR1>();
    }
    // entries j and j + 1 with one chain of compares
    __device__ __forceinline__ void at2(uint32_t j, uint32_t& a, uint32_t& b) const {
        a = m[0];
        b = m[1];
#pragma unroll
        for (int i = 1; i < JM; ++i) {
            const bool hit = j == static_cast<uint32_t>(i);
            a = hit ? m[i] : a;
            b = hit ? m[i + 1 < JM ? i + 1 : i] : b;
            asm volatile("" : "+v"(a), "+v"(b));
        }
    }
    __device__ __forceinline__ uint32_t at(uint32_t j) const {
        // a chain of selects; the empty asm keeps the compiler from turning it into a dynamically
        // indexed array (which it would place in LDS: a store of all entries + a dependent read)
        uint32_t r = m[0];
#pragma unroll
        for (int i = 1; i < JM; ++i) {
            r = (j == static_cast<uint32_t>(i)) ? m[i] : r;
            asm volatile("" : "+v"(r));
        }
        return r;
    }
    __device__ __forceinline__ uint32_t count_below(uint32_t d) const {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < JM; ++i) c += (m[i] < d) ? 1u : 0u;
        return c;
    }
};

constexpr int kBudget2 = 6;

}  // namespace

// sflags[step]: bit 0 = SIMPLE (every real track pushes a valid sample and is counted; padded
// tracks push invalid).  ntracks = real tracks (tracks >= ntracks are padding).
// TI = float, or double for float64 input whose samples are float32-representable (decoded int16 /
// float32 archives promoted by a reader): the samples are narrowed on load; `narrow_flag` (TI = double
// only) is set as soon as a sample does not survive the round trip, the kernel gives up and the float64
// kernel queued behind it (which looks at the same flag) does the work instead.
// X64 (TI = double): genuinely float64 samples.  The rings hold the HIGH words of the 64-bit keys (the
// whole selection runs on them exactly as on float32 keys) and, in a second set of tuples, the LOW words;
// once the two order statistics are known by their high words, one pass over the rings fetches the low
// words (a tie of high words -- two distinct doubles within 2^-20 relative of each other at the target
// rank, or repeated values -- is settled there by successive minima of the low words).  `narrow_flag` is
// then the RUN flag: the kernel is queued behind the narrowing one and returns unless that one gave up.
template <int W, int YPS, int PB, int JX, int JMX, int SUBS, bool STATS, typename TI = float, int X64 = 0>
__global__ __launch_bounds__(256, 2) void clim_ring2_f32(
    const TI* __restrict__ ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* __restrict__ table,
    const uint32_t* __restrict__ sflags, int32_t step_min, const DevChunk* __restrict__ chunks, double q,
    int negate, int32_t ntracks, double* __restrict__ thresh, double* __restrict__ seas, int64_t ldo,
    unsigned long long* __restrict__ stats, uint32_t* __restrict__ narrow_flag) {
    static_assert(W == 5, "count_le11 is written for an 11-sample window");
    static_assert(X64 == 0 || (sizeof(TI) == 8 && PB == 0), "the 64-bit mode takes double input and no code ring");
    constexpr bool kX64 = X64 != 0;
    // X64 == 2: the low words live in LDS (one column of NK words per thread, conflict-free) instead of a
    // second set of register tuples -- for plans whose two rings would not fit the register file (5 and
    // 6 tracks per lane at 8 lanes per cell), at the price of 2 workgroups per CU
    constexpr bool kLoLds = X64 == 2;
    constexpr bool kNarrow = sizeof(TI) == 8 && X64 == 0;
    if constexpr (kNarrow) {
        if (*narrow_flag != 0) return;           // the probe (or another workgroup) already found a lossy sample
    }
    if constexpr (kX64) {
        if (narrow_flag != nullptr && *narrow_flag == 0) return;     // the narrowing kernel did the work
    }
    bool lossy = false;
    constexpr int R = 2 * W + 1;
    static_assert(SUBS == 16 || SUBS == 8 || SUBS == 4, "16, 8 or 4 lanes per cell (4, 8 or 16 cells per wave)");
    constexpr int NTP = SUBS * YPS;
    constexpr int CPWAVE = 64 / SUBS;            // cells per wave
    // PB: width of the code ring the bracket is closed on (0: none, 32-bit count passes only; 8; 16);
    // JX: extraction width (the window of acceptable ranks is JX - 1 wide)
    static_assert(PB == 0 || PB == 8 || PB == 16, "code ring of 8 or 16 bits");
    constexpr bool PROBE8 = PB != 0;             // (name kept: "closes the bracket on a code ring")
    constexpr int J = JX;
    constexpr int JM = JMX;                      // width of the merged list (the window of acceptable ranks is JM - 1 wide)
    static_assert(JM >= J, "merged list at least as long as a lane's");
    constexpr int CPW = PB == 16 ? 2 : 4;        // codes per 32-bit word
    constexpr uint32_t LMAX = PB == 16 ? 65534u : 254u;   // levels 1..LMAX are exact thresholds; LMAX + 1 = padding
    constexpr uint32_t SLACK = JM - 2;
    constexpr int NK = YPS * R;                 // keys per lane
    constexpr int NW = PB == 16 ? (NK + 1) / 2 : (NK + 3) / 4;            // code words per lane
    constexpr uint32_t ALLC = (1u << YPS) - 1u;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // lane = (c / k) * 16 + sub * k + (c % k) with k = 16 / SUBS cells per DPP row
    const int sub = SUBS == 16 ? lane & 15 : SUBS == 8 ? (lane >> 1) & 7 : (lane >> 2) & 3;
    const int cw = SUBS == 16 ? lane >> 4 : SUBS == 8 ? (lane & 1) | ((lane >> 4) << 1) : (lane & 3) | ((lane >> 4) << 2);
    const int64_t cell = (static_cast<int64_t>(blockIdx.x) * kWaves2 + wave) * CPWAVE + cw;
    const bool cell_ok = cell < C;
    const DevChunk ch = chunks[blockIdx.y];
    const uint32_t* tab = table + sub;           // y-major: entry of slot y at tab[step * NTP + y * SUBS]
    // lanes beyond the last cell work on a copy of the last cell (and store nothing): a partly filled
    // wave then takes the same fast steps as a full one
    const TI* col = ts + (cell_ok ? cell : C - 1);
    const uint32_t negmask = negate ? 0xFFFFFFFFu : 0u;
    const uint32_t tmax = static_cast<uint32_t>(Tn - 1);
    // padding: only the last slot of a lane can be a padded track
    const bool padded_last = (YPS - 1) * SUBS + sub >= ntracks;
    const uint32_t padmask = padded_last ? 0xFFFFFFFFu : 0u;
    const uint32_t full_valid = static_cast<uint32_t>((padded_last ? YPS - 1 : YPS) * R);

    // one 11-register tuple per track: slot m (wave-uniform) is read and written through the VGPR index
    // register (s_set_gpr_idx_on + v_mov), every other access names its register statically
    typedef uint32_t RingT __attribute__((ext_vector_type(R)));
    RingT ring[YPS];
#pragma unroll
    for (int y = 0; y < YPS; ++y) ring[y] = kInv;
    RingT ringlo[(kX64 && !kLoLds) ? YPS : 1];     // X64: low words of the keys, slot for slot (register version)
#pragma unroll
    for (int y = 0; y < ((kX64 && !kLoLds) ? YPS : 1); ++y) ringlo[y] = kInv;
    __shared__ uint32_t lo_lds[kLoLds ? YPS * R * 256 : 1];
    uint32_t* const lo_col = lo_lds + threadIdx.x;               // slot (y, k) of this thread: lo_col[(y * R + k) * 256]
    if constexpr (kLoLds) {
#pragma unroll
        for (int i = 0; i < YPS * R; ++i) lo_col[i * 256] = kInv;
    }
    auto lo_at = [&](int y, int k) -> uint32_t {                 // static slot
        if constexpr (kLoLds) return lo_col[(y * R + k) * 256];
        else return ringlo[y][k];
    };
    double lsum = 0.0;        // sum of the valid samples in this lane's rings (all tracks)
    uint32_t nval = 0;        // number of valid keys in this lane's rings (all tracks)

    // Sample addressing.  The step table (plan.h) names the time step every track pushes at every step;
    // on the steps the host flags CONSEC every real track simply pushes the sample after the one it
    // pushed at the previous step, so the per-track pointers advance by one row and the table is not
    // read at all (all but ~30 of the 376 steps of a daily axis).  Otherwise the step's entries are read
    // on the spot (a dependent load, exposed, but rare).  Loads are unconditional: hold / invalid codes
    // wrap to a huge index and are clamped to the last row (a wasted but harmless read); what the
    // sample means is decided when it is consumed.  A padded slot keeps its clamped address (stride 0).
    uint32_t tix[YPS];        // time index of the sample most recently requested for each track
    const uint32_t last_step = padded_last ? 0u : 1u;
    auto entries_of = [&](int32_t step, uint32_t (&e)[YPS]) {
        const uint32_t* p = tab + static_cast<int64_t>(step - step_min) * NTP;
#pragma unroll
        for (int y = 0; y < YPS; ++y) e[y] = p[y * SUBS];
    };
    auto point_at = [&](int32_t step) {
        uint32_t e[YPS];
        entries_of(step, e);
#pragma unroll
        for (int y = 0; y < YPS; ++y) tix[y] = minu((e[y] >> 1) - 2u, tmax);
    };
    auto advance = [&]() {
#pragma unroll
        for (int y = 0; y < YPS; ++y) tix[y] += (y == YPS - 1) ? last_step : 1u;
    };
    auto request = [&](TI (&x)[YPS]) {
#pragma unroll
        for (int y = 0; y < YPS; ++y) x[y] = col[static_cast<int64_t>(tix[y]) * ld];
    };

    TI x_raw[YPS];            // as loaded (consumed one row after the request)
    point_at(ch.warm_start);
    request(x_raw);

    int m = (ch.warm_start - step_min) % R;
    // carried across rows, uniform over the 8 lanes of a cell (kept as integers, not as lane masks)
    uint32_t pc = 0, Fc = 0;  // !PROBE8: pivot with its exact count #{keys <= pc}, updated from the pushed / evicted keys
    uint32_t have_c = 0;
    float kpr = 8192.0f;      // keys per rank near the target
    bool clean = false;       // wave-uniform: every lane's rings hold valid keys only
    uint32_t st_count = 0, st_extract = 0, st_rows = 0, st_cold = 0, st_fast = 0, st_probe8 = 0, st_rebase = 0;
    uint32_t st_cell = 0;     // per lane: count passes in which this lane's cell was not yet settled
    uint32_t st_k0 = 0, st_k1 = 0, st_k2 = 0, st_k3 = 0;     // per lane: rows that needed 0 / 1 / 2 / >= 3 passes

    // PROBE8 state: 8-bit codes of the ring keys relative to (cbase, cshift); valid while have_code
    uint32_t codes[PROBE8 ? NW : 1];
    uint32_t cbase = 0, cshift = 0, have_code = 0;
    uint32_t Lc = 0;          // level of the previous row's pivot
    float drift = 0.0f;       // levels per row the pivot moved lately (signed)

    // The row loop is cut into segments that END with a step at which some track holds: the rotation
    // that realigns a held track rewrites all of its ring registers, and with that code inside the
    // row loop the register allocator copied the whole ring to a second register set and back on
    // every row.  Between two segments it costs nothing.
    uint32_t hmask = 0;        // bit y: track y held at the last step of the segment
    int32_t s = ch.warm_start;
    // step flags are read two steps ahead (a scalar load consumed on the spot costs its full latency
    // on every row)
    uint32_t sf_cur = __builtin_amdgcn_readfirstlane(sflags[s - step_min]);
    uint32_t sf_nxt = s + 1 < ch.end ? __builtin_amdgcn_readfirstlane(sflags[s + 1 - step_min]) : 0u;
    while (s < ch.end) {
    bool rotate = false;
    for (; s < ch.end && !rotate; ++s) {
        // ---- prefetch: the samples of step s+1 (consumed one row later) ------------------
        TI x_nxt[YPS];
        const uint32_t sf_nn = s + 2 < ch.end ? __builtin_amdgcn_readfirstlane(sflags[s + 2 - step_min]) : 0u;
        if (s + 1 < ch.end) {
            const uint32_t sfn = sf_nxt;
            if (sfn & 2u) advance();
            else point_at(s + 1);
            request(x_nxt);
        } else {
#pragma unroll
            for (int y = 0; y < YPS; ++y) x_nxt[y] = static_cast<TI>(0);
        }
        const uint32_t sf = sf_cur;

        // ---- what this step pushes ------------------------------------------------------
        uint32_t kin[YPS], kout[YPS];
        uint32_t cmask = ALLC;     // bit y: track y is part of this row's pool (per lane)
        hmask = 0;                 // bit y: track y holds (does not advance) at this step
        bool wave_hold = false;
        // NaN among the loaded samples?  (the sum propagates NaN; inf - inf also lands here and
        // merely takes the general step)
        // the samples of this row as float32 (narrowed and checked for float64 input)
        float x_cur[kX64 ? 1 : YPS];
        uint32_t kin_lo[kX64 ? YPS : 1], kout_lo[kX64 ? YPS : 1];      // X64: low words
        uint32_t khi[kX64 ? YPS : 1];                                   // X64: high words of this row's samples
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            if constexpr (!kX64) x_cur[y] = static_cast<float>(x_raw[y]);
            if constexpr (kNarrow) lossy |= (static_cast<TI>(x_cur[y]) != x_raw[y]) && (x_raw[y] == x_raw[y]);
            if constexpr (kX64) key64_of(static_cast<double>(x_raw[y]), negmask, khi[y], kin_lo[y]);
        }
        bool row_nan;
        if constexpr (kX64) {
            bool nn_ = false;
#pragma unroll
            for (int y = 0; y < YPS; ++y) nn_ |= x_raw[y] != x_raw[y];
            row_nan = nn_;
        } else {
            float xs = x_cur[0];
#pragma unroll
            for (int y = 1; y < YPS; ++y) xs += x_cur[y];
            row_nan = xs != xs;
        }
        const bool fast = (sf & 1u) && clean && !__any(row_nan);
        auto key_in = [&](int y) -> uint32_t {
            if constexpr (kX64) return khi[y];
            else return key_of_bits(__float_as_uint(x_cur[y]), negmask);
        };
        auto is_nan = [&](int y) -> bool { return x_raw[y] != x_raw[y]; };
        if (fast) {
            if constexpr (STATS) ++st_fast;
#pragma unroll
            for (int y = 0; y < YPS; ++y) kin[y] = key_in(y);
            kin[YPS - 1] |= padmask;
            if constexpr (kX64) kin_lo[YPS - 1] |= padmask;
        } else if (sf & 1u) {
            // a simple step by the table, but a NaN was loaded or invalid keys sit in the rings: every
            // real track pushes and is pooled, only the samples need looking at
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                const bool ok = !is_nan(y);
                kin[y] = ok ? key_in(y) : kInv;
                if constexpr (kX64) kin_lo[y] = ok ? kin_lo[y] : kInv;
            }
            kin[YPS - 1] |= padmask;
            if constexpr (kX64) kin_lo[YPS - 1] |= padmask;
        } else {
            // calendar edges, Feb 29, chunk warm-up: decode the step's table entries
            uint32_t e_cur[YPS];
            entries_of(s, e_cur);
            cmask = 0;
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                const uint32_t code = e_cur[y] >> 1;
                cmask |= (e_cur[y] & 1u) << y;
                hmask |= (code == kCodeHold ? 1u : 0u) << y;
                const bool ok = code >= 2u && !is_nan(y);
                kin[y] = ok ? key_in(y) : kInv;
                if constexpr (kX64) kin_lo[y] = ok ? kin_lo[y] : kInv;
            }
            wave_hold = __any(hmask != 0);
        }
        // ---- the one place where the rings are written (slot m of every track) ------------
#pragma unroll
        for (int y = 0; y < YPS; ++y) kout[y] = ring[y][m];
        if constexpr (kX64) {
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                if constexpr (kLoLds) kout_lo[y] = lo_col[(y * R + m) * 256];
                else kout_lo[y] = ringlo[y][m];
            }
        }
        if (wave_hold) {
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                kin[y] = ((hmask >> y) & 1u) ? kout[y] : kin[y];
                if constexpr (kX64) kin_lo[y] = ((hmask >> y) & 1u) ? kout_lo[y] : kin_lo[y];
            }
        }
#pragma unroll
        for (int y = 0; y < YPS; ++y) ring[y][m] = kin[y];
        if constexpr (kX64) {
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                if constexpr (kLoLds) lo_col[(y * R + m) * 256] = kin_lo[y];
                else ringlo[y][m] = kin_lo[y];
            }
        }
        uint32_t dF = 0;
        if (fast) {
            // running sum: + new samples - evicted samples (padded slot: both are masked to +0.0)
            double din, dout;
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                double di, dq;
                if constexpr (kX64) {
                    // (a padded slot holds the invalid key: value 0 on both sides)
                    di = value_of_key64(kin[y], kin_lo[y]);
                    dq = value_of_key64(kout[y], kout_lo[y]);
                } else {
                    uint32_t bi = __float_as_uint(x_cur[y]) ^ (negmask & 0x80000000u);
                    uint32_t bo = bits_of_key(kout[y]);
                    if (y == YPS - 1) {
                        bi &= ~padmask;
                        bo &= ~padmask;
                    }
                    di = static_cast<double>(__uint_as_float(bi));
                    dq = static_cast<double>(__uint_as_float(bo));
                }
                din = y == 0 ? di : din + di;
                dout = y == 0 ? dq : dout + dq;
                if constexpr (!PROBE8) dF += (kin[y] <= pc ? 1u : 0u) - (kout[y] <= pc ? 1u : 0u);
            }
            lsum += din - dout;
        } else {
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                // a held track contributes kin == kout: nothing changes
                if constexpr (kX64) {
                    lsum += value_of_key64(kin[y], kin_lo[y]);
                    lsum -= value_of_key64(kout[y], kout_lo[y]);
                } else {
                    lsum += value_of_key(kin[y]);
                    lsum -= value_of_key(kout[y]);
                }
                nval += (kin[y] != kInv ? 1u : 0u) - (kout[y] != kInv ? 1u : 0u);
                if constexpr (!PROBE8) dF += (kin[y] <= pc ? 1u : 0u) - (kout[y] <= pc ? 1u : 0u);
            }
            rotate = wave_hold;
            clean = !__any(nval != full_valid);
        }

        // ---- 8-bit code ring: the slot written this step ------------------------------
        if constexpr (PROBE8) {
            if (__any(have_code != 0)) {
                // code = min((key -sat base) >> shift, 254); the byte of slot (y, m) is byte
                // (y*R + m) & 3 of word (y*R + m) >> 2 -- m is wave-uniform, so this is a scalar switch
                // (lanes without a code ring write garbage codes that nothing reads)
                uint32_t cin[YPS];
#pragma unroll
                for (int y = 0; y < YPS; ++y)
                    cin[y] = minu(__builtin_elementwise_sub_sat(kin[y], cbase) >> cshift, LMAX);
#define XMHW_R2_CODE(K)                                                                          \
    case K:                                                                                      \
        if constexpr (K < R) {                                                                   \
            _Pragma("unroll") for (int y = 0; y < YPS; ++y) {                                    \
                const int pos = y * R + (K < R ? K : 0);                                         \
                const uint32_t sel = PB == 16 ? (pos % 2 == 0 ? 0x07060100u : 0x01000504u)       \
                                   : pos % 4 == 0 ? 0x07060500u : pos % 4 == 1 ? 0x07060004u    \
                                   : pos % 4 == 2 ? 0x07000504u : 0x00060504u;                   \
                codes[pos / CPW] = perm_b32(codes[pos / CPW], cin[y], sel);                      \
            }                                                                                    \
        }                                                                                        \
        break;
                // m was not advanced yet: it still names the slot written above
                switch (m) {
                    XMHW_R2_CODE(0) XMHW_R2_CODE(1) XMHW_R2_CODE(2) XMHW_R2_CODE(3) XMHW_R2_CODE(4)
                    XMHW_R2_CODE(5) XMHW_R2_CODE(6) XMHW_R2_CODE(7) XMHW_R2_CODE(8) XMHW_R2_CODE(9)
                    XMHW_R2_CODE(10)
                    default: break;
                }
#undef XMHW_R2_CODE
            }
        }
        m = (m + 1 == R) ? 0 : m + 1;

        // ---- select + output (not during warm-up) ---------------------------------
        if (s >= ch.begin) {
            const bool wallc = __all(cmask == ALLC);
            uint32_t n;
            double total;
            if (wallc) {
                n = cell_sum<SUBS>(nval);
                total = cell_sum<SUBS>(lsum);
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
                        if constexpr (kX64) ty += value_of_key64(key, opaque(lo_at(y, k)));
                        else ty += value_of_key(key);
                    }
                    const bool cnt = (cmask >> y) & 1u;
                    nl += cnt ? cy : 0u;
                    tl += cnt ? ty : 0.0;
                }
                n = cell_sum<SUBS>(nl);
                total = cell_sum<SUBS>(tl);
            }
            if (__any(!(fabs(total) <= 1.7976931348623157e308))) {
                // an infinite sample went through the running sum (inf - inf = NaN once it leaves):
                // rebuild it from the rings
                double t = 0.0, tl = 0.0;
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    double ty = 0.0;
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        if constexpr (kX64) ty += value_of_key64(opaque(ring[y][k]), opaque(lo_at(y, k)));
                        else ty += value_of_key(opaque(ring[y][k]));
                    }
                    t += ty;
                    tl += ((cmask >> y) & 1u) ? ty : 0.0;
                }
                lsum = t;
                total = cell_sum<SUBS>(tl);
            }
            if constexpr (!PROBE8) Fc += cell_sum<SUBS>(dF);

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
                    uint32_t c2 = 0;
#pragma unroll
                    for (int y = 0; y < YPS; ++y) c2 = count_le11(ring[y], p, c, c2);
                    c += c2;
                } else {
#pragma unroll
                    for (int y = 0; y < YPS; ++y) {
                        uint32_t cy = 0;
#pragma unroll
                        for (int k = 0; k < R; ++k) cy += (opaque(ring[y][k]) <= p) ? 1u : 0u;
                        c += ((cmask >> y) & 1u) ? cy : 0u;
                    }
                }
                return cell_sum<SUBS>(c);
            };

            uint32_t pl = 0, Fl = 0, ph = 0xFFFFFFFFu, Fh = nn;
            uint32_t lreal = 0, hreal = 0;
            float grow = 1.0f;
            bool resolved = (n == 0);
            uint32_t p_first = 0;
            int32_t rank_gap = 0;

            if constexpr (PROBE8) {
                // ---------- close the bracket on the 8-bit code ring ----------------------
                // masked rows (not all tracks pooled) take the 32-bit path
                if (wallc) {
                    // (re)build the code ring when there is none, or when the previous row's level came
                    // close to an edge of the 254-level window
                    const bool want = have_c != 0 && n != 0;
                    // 8 bits: 254 levels of about one rank, rebuilt when the target nears an edge (every few
                    // rows: it drifts ~13 ranks a day); 16 bits: 65534 levels of about a quarter rank cover the
                    // whole year of a cell, rebuilt only when the local key density has changed a lot
                    const float lpr_now = kpr * __builtin_ldexpf(1.0f, -static_cast<int>(cshift));
                    const bool edge = PB == 16 ? (Lc < 4096u || Lc > 61000u || lpr_now < 1.5f || lpr_now > 24.0f)
                                               : (Lc < 20u || Lc > 234u);
                    const bool rebase = want && (have_code == 0 || edge);
                    if (__any(rebase)) {
                        const float lw = fmaxf(PB == 16 ? kpr * 0.25f : kpr, 1.0f);
                        uint32_t sh = 31u - static_cast<uint32_t>(__builtin_clz(static_cast<uint32_t>(fminf(lw, 8.0e6f))));
                        sh = minu(sh, PB == 16 ? 15u : 23u);
                        const uint32_t at = PB == 16 ? 32768u : (drift >= 0.0f ? 72u : 182u);
                        const uint32_t span = at << sh;
                        uint32_t nb = pc > span ? pc - span : 0u;
                        nb = minu(nb, 0xFFFFFFFFu - ((LMAX + 2u) << sh));
                        if (rebase) {
                            cbase = nb;
                            cshift = sh;
                            have_code = 1;
                            Lc = minu(static_cast<uint32_t>((pc - nb) >> sh) + 1u, LMAX);
                            drift = 0.0f;
                        }
                        // every lane of the wave converts (cells that keep their window recompute the same codes)
#pragma unroll
                        for (int wd = 0; wd < NW; ++wd) {
                            uint32_t word = 0;
#pragma unroll
                            for (int b = 0; b < CPW; ++b) {
                                const int pos = wd * CPW + b;
                                uint32_t c = LMAX + 1u;
                                if (pos < NK)
                                    c = minu(__builtin_elementwise_sub_sat(ring[pos / R][pos % R], cbase) >> cshift, LMAX);
                                word |= c << (PB * b);
                            }
                            codes[wd] = word;
                        }
                        if constexpr (STATS) ++st_rebase;
                    }
                    if (__any(have_code != 0 && n != 0)) {
                        // cum(L) = #{code < L} = #{key < cbase + (L << cshift)}, exact for 1 <= L <= LMAX:
                        // sum |c - L| - sum |c - (L-1)| = 2 #{c < L} - N over the N code slots of the cell
                        auto cum8 = [&](uint32_t L) -> uint32_t {
                            const uint32_t one = PB == 16 ? 0x00010001u : 0x01010101u;
                            const uint32_t b1 = L * one, b0 = b1 - one;
                            uint32_t a1 = 0, a0 = 0, c1 = 0, c0 = 0;       // two chains each: the SADs of a chain are dependent
#pragma unroll
                            for (int wd = 0; wd < NW; ++wd) {
                                if (wd & 1) {
                                    c1 = PB == 16 ? sad_u16(codes[wd], b1, c1) : sad_u8(codes[wd], b1, c1);
                                    c0 = PB == 16 ? sad_u16(codes[wd], b0, c0) : sad_u8(codes[wd], b0, c0);
                                } else {
                                    a1 = PB == 16 ? sad_u16(codes[wd], b1, a1) : sad_u8(codes[wd], b1, a1);
                                    a0 = PB == 16 ? sad_u16(codes[wd], b0, a0) : sad_u8(codes[wd], b0, a0);
                                }
                            }
                            const uint32_t d = cell_sum<SUBS>((a1 + c1) - (a0 + c0));
                            return (d + static_cast<uint32_t>(SUBS * CPW * NW)) >> 1;
                        };
                        const bool active = have_code != 0 && n != 0;
                        // levels per rank near the target
                        const float lpr = fminf(fmaxf(kpr * __builtin_ldexpf(1.0f, -static_cast<int>(cshift)), 0.25f), 64.0f);
                        const float LMAXF = static_cast<float>(LMAX);
                        uint32_t Ll = 0, Cl = 0, Lh = LMAX + 1u, Ch = nn;     // cum(Ll) <= lo < cum(Lh); ends virtual
                        const float start = static_cast<float>(Lc) + drift;
                        uint32_t L = static_cast<uint32_t>(fminf(fmaxf(start, 1.0f), LMAXF));
                        bool done = !active;
                        const float aim = static_cast<float>(lo) - 0.5f * static_cast<float>(SLACK);
                        for (int it = 0; it < (PB == 16 ? 24 : 14); ++it) {
                            const uint32_t cu = cum8(done ? 1u : L);
                            if constexpr (STATS) ++st_probe8;
                            if (!done) {
                                if (cu <= lo) { Ll = L; Cl = cu; } else { Lh = L; Ch = cu; }
                                if (Ll >= 1u && lo - Cl <= SLACK) done = true;          // window hit
                                else if (Lh - Ll <= 1u) done = true;                     // a level holds > J-1 keys, or off the window
                                else {
                                    // secant in level space from the end just probed; one-sided: carried slope
                                    const bool both = Ll >= 1u && Lh <= LMAX;
                                    const float slope = both ? static_cast<float>(Lh - Ll) *
                                                                   __builtin_amdgcn_rcpf(static_cast<float>(Ch - Cl))
                                                             : lpr;
                                    const bool from_l = Ll >= 1u && (cu <= lo || Lh > LMAX);
                                    const float ranks = from_l ? aim - static_cast<float>(Cl) : static_cast<float>(Ch) - aim;
                                    const float stf = fminf(fmaxf(ranks * slope, 1.0f), LMAXF);
                                    uint32_t st = static_cast<uint32_t>(stf);
                                    if (it >= 5) st = maxu((Lh - Ll) >> 1, 1u);
                                    uint32_t Ln = from_l ? Ll + st : (Lh > st ? Lh - st : 0u);
                                    L = minu(maxu(Ln, Ll + 1u), Lh - 1u);
                                }
                            }
                            if (__all(done)) break;
                        }
                        if (active) {
                            // hand the bracket to the 32-bit code below: inside the window it is settled
                            // (one extraction); off the window or on an overfull level count passes finish it
                            if (Ll >= 1u) {
                                pl = cbase + (Ll << cshift) - 1u; Fl = Cl; lreal = 1;
                            }
                            if (Lh <= LMAX) {
                                ph = cbase + (Lh << cshift) - 1u; Fh = Ch; hreal = 1;
                            }
                            if (Ll >= 1u && lo - Cl <= SLACK) {
                                const float moved = static_cast<float>(Ll) - static_cast<float>(Lc);
                                const float cap = PB == 16 ? 1024.0f : 64.0f;
                                drift = 0.5f * drift + 0.5f * fminf(fmaxf(moved, -cap), cap);
                                Lc = Ll;
                            } else if (Ll >= 1u && Lh <= LMAX) {
                                // an overfull level (ties, a dense spot): the window itself is fine, the 32-bit
                                // passes below finish this row
                                Lc = Ll;
                            } else {
                                have_code = 0;     // off the window: next row rebuilds it around the new answer
                                drift = 0.0f;
                            }
                        }
                    }
                }
                p_first = pc;
            } else {
                const bool use_c = have_c != 0 && wallc;
                uint32_t p0 = pc, F0 = 0;
                if (use_c) F0 = Fc;
                if (!__all(use_c || n == 0)) {
                    uint32_t pm = key_of_bits(__float_as_uint(static_cast<float>(total / static_cast<double>(nn))), 0u);
                    if (!use_c) p0 = have_c != 0 ? pc : pm;
                    const uint32_t Fr = count_le(minu(p0, 0xFFFFFFFEu));
                    if (!use_c) F0 = Fr;
                    if constexpr (STATS) ++st_cold;
                }
                if (p0 != 0 && p0 < 0xFFFFFFFEu) {
                    if (F0 <= lo) { pl = p0; Fl = F0; lreal = 1; }
                    else { ph = p0; Fh = F0; hreal = 1; }
                }
                p_first = p0;
                rank_gap = static_cast<int32_t>(lo) - static_cast<int32_t>(F0);
            }
            const float aim = static_cast<float>(lo) - 0.5f * static_cast<float>(SLACK);

            if constexpr (PROBE8) {
                // no bracket at all (first row of a chunk, masked row, no code ring yet): one 32-bit
                // pass at the pool mean gives the secant loop below something to start from
                if (!__all(lreal != 0 || hreal != 0 || n == 0)) {
                    uint32_t pm = key_of_bits(__float_as_uint(static_cast<float>(total / static_cast<double>(nn))), 0u);
                    pm = minu(maxu(pm, 1u), 0xFFFFFFFEu);
                    const bool cold = lreal == 0 && hreal == 0 && n != 0;
                    const uint32_t Fm = count_le(cold ? pm : pl);
                    if (cold) {
                        if (Fm <= lo) { pl = pm; Fl = Fm; lreal = 1; }
                        else { ph = pm; Fh = Fm; hreal = 1; }
                    }
                    if constexpr (STATS) ++st_cold;
                }
            }
            uint32_t alo = 0, ahi = 0, pe = 0, Fe = 0, top_span = 0;
            // the wide window (JM > J) rests on the horizon check below; a cell that fails it once falls back
            // to the window of the lanes' own list length for the rest of the row, where every merged entry is
            // exact by construction (the cell's J nearest keys are each among their lane's J nearest), so the
            // loop ends exactly as it does for JM == J
            uint32_t slack = SLACK;
            uint32_t k_row = 0;       // STATS: count passes this cell needed for this row
            int budget = kBudget2;
            for (;;) {
                // ---- 32-bit count passes until every cell can be settled by one extraction ----
                for (int it = 0;; ++it) {
                    const bool settle = resolved || (lo - Fl <= slack) || (ph - pl <= 1u);
                    if (__all(settle) || it >= budget) break;
                    const uint32_t room = ph - pl;
                    const bool both = lreal != 0 && hreal != 0;
                    const bool from_l = lreal != 0 || hreal == 0;
                    const float roomf = static_cast<float>(room);
                    const float slope = both ? roomf * __builtin_amdgcn_rcpf(static_cast<float>(Fh - Fl))
                                             : kpr * grow;
                    const float ranks = from_l ? aim - static_cast<float>(Fl) : static_cast<float>(Fh) - aim;
                    float stf = fminf(fmaxf(ranks * slope, 1.0f), 2.0e9f);
                    stf = from_l ? stf : roomf - stf;
                    stf = fminf(fmaxf(stf, 1.0f), 4.0e9f);
                    uint32_t off = (it < 5) ? static_cast<uint32_t>(stf) : (room >> 1);
                    grow = both ? grow : grow * 2.0f;
                    off = maxu(1u, minu(off, room - 1u));
                    const uint32_t p = settle ? pl : pl + off;
                    const uint32_t F = count_le(p);
                    if constexpr (STATS) {
                        ++st_count;
                        st_cell += settle ? 0u : 1u;      // passes THIS cell needed (the wave runs the maximum)
                        k_row += settle ? 0u : 1u;
                    }
                    if (!settle) {
                        if (F <= lo) { pl = p; Fl = F; lreal = 1; }
                        else { ph = p; Fh = F; hreal = 1; }
                    }
                }
                // ---- extraction: the J smallest keys above the pivot --------------------
                const bool window = (lo - Fl <= slack);
                const bool adjacent = !window && (ph - pl <= 1u);
                const uint32_t px = adjacent ? ph : pl;
                const uint32_t base = px + 1u;
                Top2<J, JM> top;
                top.reset();
                if (wallc) {
#pragma unroll
                    for (int y = 0; y < YPS; ++y)
#pragma unroll
                        for (int k = 0; k < R; ++k) top.insert(ring[y][k] - base);
                } else {
#pragma unroll
                    for (int y = 0; y < YPS; ++y)
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            const uint32_t d = opaque(ring[y][k]) - base;
                            top.insert(((cmask >> y) & 1u) ? d : 0xFFFFFFFFu);
                        }
                }
                // entries of the merged list up to the horizon are exact order statistics (see Top2)
                const uint32_t horizon = JM > J ? top.template horizon<SUBS>() : 0xFFFFFFFFu;
                top.template merge_cell<SUBS>();
                if constexpr (STATS) ++st_extract;
                if (!resolved) {
                    const uint32_t j = window ? lo - Fl : 0u;
                    uint32_t d_lo, d_nx;
                    top.at2(j, d_lo, d_nx);      // (j + 1 <= JM - 1 inside the window)
                    const uint32_t d_hi = need2 ? d_nx : d_lo;
                    const bool exact = d_hi <= horizon || j + (need2 ? 1u : 0u) < static_cast<uint32_t>(J);
                    if (window && !exact) slack = J - 2;
                    if (window && exact) {
                        alo = base + d_lo;
                        ahi = base + d_hi;
                        pe = pl; Fe = Fl;
                        top_span = top.m[J - 1] - top.m[0];      // (code-ring variants: JM == J)
                        resolved = true;
                    } else if (adjacent && !window) {
                        alo = ph;
                        ahi = (need2 && lo + 1u >= Fh) ? base + top.m[0] : ph;
                        pe = ph; Fe = Fh;
                        resolved = true;
                    }
                }
                if (__all(resolved)) break;
                // ---- repair (tie-heavy data): count at the largest extracted key ---------
                // (every key below base + dj is in the merged list: dj is within the horizon and within the list)
                const uint32_t dj = minu(top.m[JM - 1], horizon);
                const uint32_t pj = base + dj;
                const uint32_t Fj = count_le(resolved ? pl : pj);
                if constexpr (STATS) ++st_count;
                if (!resolved) {
                    if (Fj <= lo) {
                        pl = pj; Fl = Fj; lreal = 1;
                    } else {
                        ph = pj; Fh = Fj; hreal = 1;
                        Fl = Fl + top.count_below(dj);
                        pl = pj - 1u;
                        lreal = 1;
                    }
                }
                budget = 2;
            }

            if constexpr (STATS) {
                ++st_rows;
                if constexpr (!PROBE8) {      // histogram of the per-cell pass count (slots 5 and 6 are free without a code ring)
                    st_k0 += k_row == 0 ? 1u : 0u;
                    st_k1 += k_row == 1 ? 1u : 0u;
                    st_k2 += k_row == 2 ? 1u : 0u;
                    st_k3 += k_row >= 3 ? 1u : 0u;
                }
            }
            double th = make_nan(), se = make_nan();
            double v_lo = 0.0, v_hi = 0.0;      // the two order statistics as values
            if constexpr (kX64) {
                // ---- low words of the two order statistics (alo, ahi are HIGH words here) ----------
                // one pass: how many pooled keys carry each high word, and the low word of one of them
                uint32_t ca = 0, cb = 0, la = 0, lb = 0;
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    const bool cnt = wallc || ((cmask >> y) & 1u);
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const uint32_t key = ring[y][k], l = lo_at(y, k);
                        const bool ea = cnt && key == alo, eb = cnt && key == ahi;
                        ca += ea ? 1u : 0u;
                        cb += eb ? 1u : 0u;
                        la = ea ? l : la;
                        lb = eb ? l : lb;
                    }
                    // (LDS version: one track's 11 reads in flight at a time, not all of the lane's at once)
                    if constexpr (kLoLds) asm volatile("" ::: "memory");
                }
                ca = cell_sum<SUBS>(ca);
                cb = cell_sum<SUBS>(cb);
                la = cell_max<SUBS>(la);        // (exact when the cell holds ONE such key: the other lanes offer 0)
                lb = cell_max<SUBS>(lb);
                // ties of high words (repeated values, or distinct doubles within 2^-20 of each other at the target
                // rank): the r-th smallest low word of the group, by successive minima with their multiplicities
                const bool tie_a = n > 0 && ca > 1u, tie_b = n > 0 && need2 && cb > 1u;
                if (__any(tie_a || tie_b)) {
                    // keys below group A: lo itself when A is a single key, otherwise counted
                    uint32_t below_a = lo;
                    if (__any(tie_a)) {
                        const uint32_t c = count_le(tie_a ? alo - 1u : 0u);
                        below_a = tie_a ? c : lo;
                    }
                    const uint32_t ra = lo - below_a;                                   // rank inside group A
                    const uint32_t rb = ahi == alo ? ra + 1u : lo + 1u - (below_a + ca);  // rank inside group B
                    auto nth_low = [&](uint32_t H, uint32_t r, bool want) -> uint32_t {
                        uint32_t prev = 0, rem = r, ans = 0;
                        bool have_prev = false, open = want;
                        while (__any(open)) {
                            uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
                            for (int y = 0; y < YPS; ++y) {
                                const bool cnt = wallc || ((cmask >> y) & 1u);
#pragma unroll
                                for (int k = 0; k < R; ++k) {
                                    const uint32_t key = opaque(ring[y][k]), l = opaque(lo_at(y, k));
                                    const bool in = cnt && key == H && (!have_prev || l > prev);
                                    mn = in ? minu(mn, l) : mn;
                                }
                            }
                            mn = cell_min<SUBS>(mn);
                            uint32_t c = 0;
#pragma unroll
                            for (int y = 0; y < YPS; ++y) {
                                const bool cnt = wallc || ((cmask >> y) & 1u);
#pragma unroll
                                for (int k = 0; k < R; ++k)
                                    c += (cnt && opaque(ring[y][k]) == H && opaque(lo_at(y, k)) == mn) ? 1u : 0u;
                            }
                            c = cell_sum<SUBS>(c);
                            if (open) {
                                if (rem < c || c == 0u) {       // (c == 0 cannot happen for r inside the group; it ends the loop)
                                    ans = mn;
                                    open = false;
                                } else {
                                    rem -= c;
                                    prev = mn;
                                    have_prev = true;
                                }
                            }
                        }
                        return ans;
                    };
                    const uint32_t xa = nth_low(alo, ra, tie_a);
                    const uint32_t xb = nth_low(ahi, rb, tie_b);
                    la = tie_a ? xa : la;
                    lb = tie_b ? xb : lb;
                }
                v_lo = double_of_key64(alo, la);
                v_hi = need2 ? double_of_key64(ahi, lb) : v_lo;
            } else {
                v_lo = static_cast<double>(__uint_as_float(bits_of_key(alo)));
                v_hi = static_cast<double>(__uint_as_float(bits_of_key(ahi)));
            }
            if (n > 0) {
                th = numpy_lerp(v_lo, v_hi, g);
                se = total / static_cast<double>(n);
                if constexpr (!PROBE8) {
                    if (rank_gap > 1 || rank_gap < -1) {
                        const float obs = (static_cast<float>(alo) - static_cast<float>(p_first)) *
                                          __builtin_amdgcn_rcpf(static_cast<float>(rank_gap));
                        if (obs >= 1.0f && obs < 1.0e8f) kpr = 0.75f * kpr + 0.25f * obs;
                    }
                } else {
                    // keys per rank: spread of the extracted list (J - 1 gaps above the pivot)
                    const float obs = static_cast<float>(top_span) * (1.0f / static_cast<float>(J - 1));
                    if (obs >= 1.0f && obs < 1.0e8f) kpr = 0.875f * kpr + 0.125f * obs;
                }
            }
            if (n > 0 && wallc) {
                pc = pe;
                Fc = Fe;
                have_c = 1;
            } else {
                have_c = 0;
                pc = 0;
                Fc = 0;
            }
            if (sub == 0 && cell_ok) {
                thresh[static_cast<int64_t>(s) * ldo + cell] = th;
                seas[static_cast<int64_t>(s) * ldo + cell] = se;
            }
        }

        if constexpr (kNarrow) {
            // (no rendezvous here: a wave that has seen a lossy sample leaves, and the others must not wait for it)
            if ((s & 63) == 63 && __any(lossy)) {
                if (lossy) atomicOr(narrow_flag, 1u);
                return;
            }
        } else {
            if ((s & 63) == 63) __syncthreads();
        }
#pragma unroll
        for (int y = 0; y < YPS; ++y) x_raw[y] = x_nxt[y];
        sf_cur = sf_nxt;
        sf_nxt = sf_nn;
    }
    if (rotate) {
        // a held track did not advance: rotate its window one slot so that its oldest sample sits
        // where the next step's PUSH will land
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            const unsigned long long hy = __builtin_amdgcn_ballot_w64(((hmask >> y) & 1u) != 0);
            const uint32_t last = opaque(ring[y][R - 1]);
            asm volatile("s_nop 1");     // hy may come straight from a v_cmp: two wait states before a VALU read
#pragma unroll
            for (int k = R - 1; k >= 1; --k) {
                uint32_t e = ring[y][k];
                ring_sel(e, ring[y][k - 1], hy);
                ring[y][k] = e;
            }
            uint32_t e0 = ring[y][0];
            ring_sel(e0, last, hy);
            ring[y][0] = e0;
            if constexpr (kX64 && !kLoLds) {
                const uint32_t last_lo = opaque(ringlo[y][R - 1]);
#pragma unroll
                for (int k = R - 1; k >= 1; --k) {
                    uint32_t e = ringlo[y][k];
                    ring_sel(e, ringlo[y][k - 1], hy);
                    ringlo[y][k] = e;
                }
                uint32_t l0 = ringlo[y][0];
                ring_sel(l0, last_lo, hy);
                ringlo[y][0] = l0;
            }
            if constexpr (kLoLds) {
                if ((hmask >> y) & 1u) {         // (per lane: only the holding tracks' columns move)
                    uint32_t t[R];
#pragma unroll
                    for (int k = 0; k < R; ++k) t[k] = lo_col[(y * R + k) * 256];
#pragma unroll
                    for (int k = 0; k < R; ++k) lo_col[(y * R + k) * 256] = t[(k + R - 1) % R];
                }
            }
        }
        have_code = 0;         // byte positions moved: rebuild the code ring
    }
    }
    if constexpr (kNarrow) {
        if (lossy) atomicOr(narrow_flag, 1u);
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
    // [7]: count passes summed over CELLS (one lane per cell reports), to set against [1] x cells per wave
    if (STATS && stats != nullptr && sub == 0 && cell_ok) {
        atomicAdd(&stats[7], static_cast<unsigned long long>(st_cell));
        if constexpr (!PROBE8) {
            // without a code ring slots 5 and 6 carry the per-cell histogram: {k = 0 | k = 1 << 32}, {k = 2 | k >= 3 << 32}
            atomicAdd(&stats[5], static_cast<unsigned long long>(st_k0) | (static_cast<unsigned long long>(st_k1) << 32));
            atomicAdd(&stats[6], static_cast<unsigned long long>(st_k2) | (static_cast<unsigned long long>(st_k3) << 32));
        }
    }
}

// ---------------------------------------------------------------------------
namespace {
typedef void (*Ring2Kernel)(const float*, int64_t, int64_t, int64_t, const uint32_t*, const uint32_t*, int32_t,
                            const DevChunk*, double, int, int32_t, double*, double*, int64_t, unsigned long long*,
                            uint32_t*);
typedef void (*Ring2KernelN)(const double*, int64_t, int64_t, int64_t, const uint32_t*, const uint32_t*, int32_t,
                             const DevChunk*, double, int, int32_t, double*, double*, int64_t, unsigned long long*,
                             uint32_t*);
struct Ring2Entry { int w, yps, subs, variant; Ring2Kernel fn, fn_stats; Ring2KernelN fn_narrow, fn_x64; };
// variant -> (code-ring bits, extraction width, lanes per cell); the _stats twin carries the debug pass counters
//   0: 32-bit count passes, J = 5        1: 8-bit probes, J = 5      2: 16-bit probes, J = 5
//   3: 16-bit probes, J = 4              4: 16-bit probes, J = 3
//   5: 32-bit count passes, J = 4        6: 32-bit count passes, J = 6
//   7: as 0 with 4 lanes per cell (16 cells per wave, twice the tracks per lane)
//   8 / 9: as 0 / 7 with the lanes' J = 5 lists merged into the cell's 8 nearest keys (window of 7 ranks)
//   10 / 11: as 9 with 7 / 6 merged keys (4 lanes per cell: a lane holds 5 of the cell's 8 nearest more often)
//   12: 16 lanes per cell (4 cells per wave), wide merge: the 64-bit mode's layout for short records, and the
//       float32 layout of 49..96-track records
// (the counter twins are built with -DXMHW_RING_STATS only: tools/, not the product)
#ifdef XMHW_RING_STATS
#define XMHW_R2S(W, Y, PB, JX, JM, S) clim_ring2_f32<W, Y, PB, JX, JM, S, true>
#else
#define XMHW_R2S(W, Y, PB, JX, JM, S) nullptr
#endif
#define XMHW_R2V(W, Y, S, V, PB, JX, JM) {W, Y, S, V, clim_ring2_f32<W, Y, PB, JX, JM, S, false>, XMHW_R2S(W, Y, PB, JX, JM, S), nullptr, nullptr}
// the shipped layouts also exist for float64 input: narrowing to float32 where that is lossless, and the
// 64-bit mode (high / low key words) for genuinely float64 samples
#define XMHW_R2N(W, Y, S, V, PB, JX, JM) {W, Y, S, V, clim_ring2_f32<W, Y, PB, JX, JM, S, false>, XMHW_R2S(W, Y, PB, JX, JM, S), \
                                          clim_ring2_f32<W, Y, PB, JX, JM, S, false, double>,             \
                                          clim_ring2_f32<W, Y, PB, JX, JM, S, false, double, 1>}
// (narrowing only: the 64-bit mode would spill heavily at 66 or more keys per lane; those plans keep the
// round-1 float64 kernel)
#define XMHW_R2M(W, Y, S, V, PB, JX, JM) {W, Y, S, V, clim_ring2_f32<W, Y, PB, JX, JM, S, false>, XMHW_R2S(W, Y, PB, JX, JM, S), \
                                          clim_ring2_f32<W, Y, PB, JX, JM, S, false, double>, nullptr}
// 16 lanes per cell, 64-bit mode only (variant 12): genuinely float64 samples of plans with more keys per lane
// than the 8- and 4-lane layouts can hold in registers next to the low words, and of short records
#define XMHW_R2X(W, Y, S, V, PB, JX, JM) {W, Y, S, V, nullptr, nullptr, nullptr, clim_ring2_f32<W, Y, PB, JX, JM, S, false, double, 1>}
// narrowing + the 64-bit mode with the low words in LDS (5 and 6 tracks per lane at 8 lanes per cell)
#define XMHW_R2L(W, Y, S, V, PB, JX, JM) {W, Y, S, V, clim_ring2_f32<W, Y, PB, JX, JM, S, false>, XMHW_R2S(W, Y, PB, JX, JM, S), \
                                          clim_ring2_f32<W, Y, PB, JX, JM, S, false, double>,             \
                                          clim_ring2_f32<W, Y, PB, JX, JM, S, false, double, 2>}
// Default build: the layouts the library can pick on its own or for float64 input -- variant 8 (8 lanes, wide
// merge: float32, narrowing float64, 64-bit mode), 10 (4 lanes), 12 (16 lanes) -- plus their plain counterparts 0 / 7
// on the headline shapes (tests compare them).  The measured-and-rejected experiments of round 2 (code rings 1-4,
// extraction widths 5 / 6, merged widths 9 / 11) are compiled with -DXMHW_RING2_EXPERIMENTS only
// (tools/fuzz_ring2.py and tests/test_gpu_ring2.py skip variants that are not built).
#ifdef XMHW_RING2_EXPERIMENTS
#define XMHW_R2(W, Y) XMHW_R2V(W, Y, 8, 0, 0, 5, 5), XMHW_R2V(W, Y, 8, 1, 8, 5, 5), XMHW_R2V(W, Y, 8, 2, 16, 5, 5), \
                      XMHW_R2V(W, Y, 8, 3, 16, 4, 4), XMHW_R2V(W, Y, 8, 4, 16, 3, 3), XMHW_R2V(W, Y, 8, 5, 0, 4, 4), \
                      XMHW_R2V(W, Y, 8, 6, 0, 6, 6),
#define XMHW_R2E(...) __VA_ARGS__,
#else
#define XMHW_R2(W, Y)
#define XMHW_R2E(...)
#endif
// (round 4: the plain layouts 0 and 7 -- the shipped 8 and 10 without the merged lists -- left the default build with
// the other experiments)
const Ring2Entry kRing2[] = {
    XMHW_R2(5, 3) XMHW_R2(5, 4) XMHW_R2(5, 5)
    XMHW_R2L(5, 3, 8, 8, 0, 5, 8), XMHW_R2N(5, 4, 8, 8, 0, 5, 8), XMHW_R2L(5, 5, 8, 8, 0, 5, 8),
    XMHW_R2E(XMHW_R2V(5, 5, 4, 7, 0, 5, 5), XMHW_R2V(5, 8, 4, 7, 0, 5, 5), XMHW_R2V(5, 10, 4, 7, 0, 5, 5))
    XMHW_R2E(XMHW_R2V(5, 5, 4, 9, 0, 5, 8), XMHW_R2V(5, 8, 4, 9, 0, 5, 8), XMHW_R2V(5, 10, 4, 9, 0, 5, 8))
    XMHW_R2N(5, 5, 4, 10, 0, 5, 7), XMHW_R2M(5, 8, 4, 10, 0, 5, 7), XMHW_R2M(5, 10, 4, 10, 0, 5, 7),
    XMHW_R2E(XMHW_R2V(5, 5, 4, 11, 0, 5, 6), XMHW_R2V(5, 8, 4, 11, 0, 5, 6), XMHW_R2V(5, 10, 4, 11, 0, 5, 6))
    // shorter and longer records (9..16 and 41..48 tracks: 10-year series, OISST 1982-today), shipped layouts only
    XMHW_R2E(XMHW_R2V(5, 2, 8, 0, 0, 5, 5), XMHW_R2V(5, 6, 8, 0, 0, 5, 5), XMHW_R2V(5, 3, 4, 7, 0, 5, 5), XMHW_R2V(5, 4, 4, 7, 0, 5, 5))
    XMHW_R2N(5, 2, 8, 8, 0, 5, 8), XMHW_R2L(5, 6, 8, 8, 0, 5, 8),
    XMHW_R2N(5, 3, 4, 10, 0, 5, 7), XMHW_R2N(5, 4, 4, 10, 0, 5, 7),
    XMHW_R2X(5, 1, 16, 12, 0, 5, 8), XMHW_R2X(5, 2, 16, 12, 0, 5, 8), XMHW_R2X(5, 3, 16, 12, 0, 5, 8),
    // long records (49..96 tracks: reanalyses, model runs) on 16 lanes per cell: float32, narrowing float64 and the
    // 64-bit mode with its low words in LDS
    XMHW_R2L(5, 4, 16, 12, 0, 5, 8), XMHW_R2L(5, 5, 16, 12, 0, 5, 8), XMHW_R2L(5, 6, 16, 12, 0, 5, 8),
};
#undef XMHW_R2
#undef XMHW_R2S
#undef XMHW_R2E
#undef XMHW_R2V
#undef XMHW_R2N
#undef XMHW_R2M
#undef XMHW_R2X
#undef XMHW_R2L
const Ring2Entry* find_ring2(int32_t w, int32_t yps, int32_t subs, int32_t variant) {
    for (const auto& e : kRing2)
        if (e.w == w && e.yps == yps && e.subs == subs && e.variant == variant) return &e;
    return nullptr;
}
}  // namespace

// variants 20 / 21 / 22: the third-generation kernel (kernels_ring3.hip) on 8 / 4 / 2 lanes per cell; 30 / 31 / 32: the
// round-4 key-store experiment (kernels_ring4.hip, built with `make RING4=1` only: profiles/r4_store_experiment.txt)
int32_t ring2_subs(int32_t variant) {
    if (variant >= 20) return variant % 10 == 2 ? 2 : variant % 10 == 1 ? 4 : 8;
    return variant == 12 ? 16 : (variant == 7 || variant >= 9) ? 4 : 8;
}

int32_t ring2_pick_yps(int32_t w, int32_t ntracks, int32_t variant) {
    const int32_t subs = ring2_subs(variant);
#ifdef XMHW_RING4
    if (variant >= 30) return ring4_pick_yps(w, ntracks, subs);
#else
    if (variant >= 30) return 0;
#endif
    if (variant >= 20) return ring3_pick_yps(w, ntracks, subs);
    int32_t best = 0;
    for (const auto& e : kRing2)
        if (e.w == w && e.variant == (variant < 0 ? 8 : variant) && e.subs == subs && e.yps * subs >= ntracks &&
            (best == 0 || e.yps < best))
            best = e.yps;
    // padding may only sit in the last slot of a lane
    if (best && (best - 1) * subs >= ntracks) return 0;
    return best;
}

hipError_t launch_ring2_f32(const float* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                            const uint32_t* sflags, int32_t step_min, const DevChunk* chunks, int32_t nchunks,
                            int32_t w, int32_t yps, int32_t ntracks, int32_t variant, double q, int negate,
                            double* thresh, double* seas, int64_t ldo, hipStream_t stream,
                            unsigned long long* stats) {
    const int32_t subs = ring2_subs(variant);
#ifdef XMHW_RING4
    if (variant >= 30)
        return launch_ring4_f32(ts, C, ld, Tn, table, sflags, step_min, chunks, nchunks, w, yps, subs, ntracks, q, negate,
                                thresh, seas, ldo, stream, stats);
#else
    if (variant >= 30) return hipErrorInvalidValue;
#endif
    if (variant >= 20)
        return launch_ring3_f32(ts, C, ld, Tn, table, sflags, step_min, chunks, nchunks, w, yps, subs, ntracks, q, negate,
                                thresh, seas, ldo, stream, stats);
    const Ring2Entry* e = find_ring2(w, yps, subs, variant);
    if (!e || !e->fn) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    const int64_t cells_per_block = (64 / subs) * kWaves2;
    dim3 grid(static_cast<unsigned>((C + cells_per_block - 1) / cells_per_block), static_cast<unsigned>(nchunks));
    const bool twin = stats != nullptr && e->fn_stats != nullptr;
    hipLaunchKernelGGL(twin ? e->fn_stats : e->fn, grid, dim3(64 * kWaves2), 0, stream, ts, C, ld, Tn, table, sflags, step_min,
                       chunks, q, negate, ntracks, thresh, seas, ldo, twin ? stats : nullptr,
                       static_cast<uint32_t*>(nullptr));
    return hipGetLastError();
}

bool ring2_f32_supported(int32_t w, int32_t yps, int32_t variant) {
#ifdef XMHW_RING4
    if (variant >= 30) return ring4_supported(w, yps, ring2_subs(variant));
#else
    if (variant >= 30) return false;
#endif
    if (variant >= 20) return ring3_supported(w, yps, ring2_subs(variant));
    const Ring2Entry* e = find_ring2(w, yps, ring2_subs(variant), variant);
    return e != nullptr && e->fn != nullptr;
}

bool ring2_x64_supported(int32_t w, int32_t yps, int32_t variant) {
    const Ring2Entry* e = find_ring2(w, yps, ring2_subs(variant), variant);
    return e != nullptr && e->fn_x64 != nullptr;
}

// genuinely float64 samples on the second-generation kernel (64-bit keys as high / low words); run_flag: device
// flag of the narrowing launch queued before it (nullptr: always run; 0 at run time: nothing to do)
hipError_t launch_ring2_f64(const double* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                            const uint32_t* sflags, int32_t step_min, const DevChunk* chunks, int32_t nchunks,
                            int32_t w, int32_t yps, int32_t ntracks, int32_t variant, double q, int negate,
                            double* thresh, double* seas, int64_t ldo, hipStream_t stream, const uint32_t* run_flag) {
    const int32_t subs = ring2_subs(variant);
    if (variant >= 30) return hipErrorInvalidValue;
    if (variant >= 20)
        return launch_ring3_f64(ts, C, ld, Tn, table, sflags, step_min, chunks, nchunks, w, yps, subs, ntracks, q, negate,
                                thresh, seas, ldo, stream, run_flag);
    const Ring2Entry* e = find_ring2(w, yps, subs, variant);
    if (!e || !e->fn_x64) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    const int64_t cells_per_block = (64 / subs) * kWaves2;
    dim3 grid(static_cast<unsigned>((C + cells_per_block - 1) / cells_per_block), static_cast<unsigned>(nchunks));
    hipLaunchKernelGGL(e->fn_x64, grid, dim3(64 * kWaves2), 0, stream, ts, C, ld, Tn, table, sflags, step_min, chunks, q,
                       negate, ntracks, thresh, seas, ldo, static_cast<unsigned long long*>(nullptr),
                       const_cast<uint32_t*>(run_flag));
    return hipGetLastError();
}

bool ring2_narrowing_supported(int32_t w, int32_t yps, int32_t variant) {
    if (variant >= 30) return false;
    if (variant >= 20) return ring3_narrowing_supported(w, yps, ring2_subs(variant));
    const Ring2Entry* e = find_ring2(w, yps, ring2_subs(variant), variant);
    return e != nullptr && e->fn_narrow != nullptr;
}

// float64 input on the float32 kernel: the flag must have been cleared and the sparse probe queued by the
// caller (launch_narrow_probe, kernels_ring.hip); the kernel leaves as soon as the flag is set
hipError_t launch_ring2_f32_narrowing(const double* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                                      const uint32_t* sflags, int32_t step_min, const DevChunk* chunks,
                                      int32_t nchunks, int32_t w, int32_t yps, int32_t ntracks, int32_t variant,
                                      double q, int negate, double* thresh, double* seas, int64_t ldo,
                                      hipStream_t stream, uint32_t* narrow_flag) {
    const int32_t subs = ring2_subs(variant);
    if (variant >= 30) return hipErrorInvalidValue;
    if (variant >= 20)
        return launch_ring3_f32_narrowing(ts, C, ld, Tn, table, sflags, step_min, chunks, nchunks, w, yps, subs, ntracks, q,
                                          negate, thresh, seas, ldo, stream, narrow_flag);
    const Ring2Entry* e = find_ring2(w, yps, subs, variant);
    if (!e || !e->fn_narrow || !narrow_flag) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    const int64_t cells_per_block = (64 / subs) * kWaves2;
    dim3 grid(static_cast<unsigned>((C + cells_per_block - 1) / cells_per_block), static_cast<unsigned>(nchunks));
    hipLaunchKernelGGL(e->fn_narrow, grid, dim3(64 * kWaves2), 0, stream, ts, C, ld, Tn, table, sflags, step_min, chunks, q,
                       negate, ntracks, thresh, seas, ldo, static_cast<unsigned long long*>(nullptr), narrow_flag);
    return hipGetLastError();
}

}  // namespace xmhw
