// kernels_ring4.hip -- fourth-generation ring kernel (round 4): selection from a KEY STORE in LDS.
//
// The ring (R = 2w+1 samples of every owned track as order-preserving 32-bit keys in VGPRs), the step table, the push
// logic, the running sums and the slow path (count passes + extraction) are those of kernels_ring3.hip.  What changed
// is how the two order statistics of numpy's linear quantile are found on an ordinary row.  Round 3 mirrored the ring
// in a per-cell histogram, bracketed the two ranks with it and then made ONE PASS OVER ALL POOLED KEYS (110 per lane:
// v_sub, v_cmp, EXEC switch, masked ds_write) to collect the handful of keys of the band: a third of the kernel, and
// 110 of its 141 LDS instructions per row.  Here the keys near the target are KEPT in LDS, where the histogram was:
//
//   * a per-cell WINDOW of NS rows of 2^shift keys each, [wbase, wbase + (NS << shift)), wbase a multiple of the row
//     width.  Row p (p = key bits [shift, shift + log2 NS): the physical row of a key does not change when the window
//     moves) is a ring of CAPB keys with one header word, push count << 16 | keys alive.  A pushed key inside the
//     window is stored: ds_add_rtn(header, 0x10001) returns its slot, one ds_write puts it there.  An evicted key inside
//     the window only decrements the header.  Keys outside the window touch a per-lane dummy word.
//     Every key lives for exactly R rows, so the keys of a row leave it in the order they entered: the keys alive are
//     always the last `alive` ones written, no search, no tombstones (rows on which a track is held -- Feb 29 -- break
//     that order: the window is rebuilt after them).
//   * the exact number Fw of pooled keys below wbase is carried by the free probe (the borrow of key - wbase, which the
//     window test needs anyway).
//   * per row: the lanes of a cell read the NS headers, a prefix sum finds the row that holds order statistic lo, its
//     rank j inside the row and the row's population; the lanes read the row (CAPB keys: ds_read_b128), blank the dead
//     slots, sort the CAPB slots across the lanes of the cell (kernels_ring3.hip's bitonic network) and pick entries j
//     and j + 1 (or the minimum of the next populated row).  Nothing is counted and no key of the ring is touched.
//   * when a target nears an end of its window (the seasonal drift: 4 ranks a row on average, 13 at the 99th
//     percentile on the synthetic SST) or the population of its row leaves [MU_LO, MU_HI] (the row width then changes by
//     powers of two), the wave makes ONE pass over the ring in AGE order that re-places every cell's window around its
//     target and stores the keys of the rows that are new (fill pass; it recounts Fw).
//
// What can fail is capacity (a row with CAPB or more keys alive: ties, constant cells) or the target leaving the window;
// those rows -- and rows that do not pool every track (Feb 29), and the first row of a chunk -- take the slow path.
//
// Lane layout, workgroup shape, step table and chunks: kernels_ring3.hip.
//
// Reference semantics restated: window_roll() (identify.py:184-209),
// calculate_thresh()/calculate_seas() without the Feb-29 step (identify.py:233-235, :263),
// coldSpells negation (xmhw.py:153-154).
#include "device_common.h"
#include "kernels.h"
#include "plan.h"

#include "ring3_helpers.h"

namespace xmhw {
namespace {

// NS rows per window, CAPB keys per row, JM merged slow-path list; MU_*: population of the target row (running mean)
// the row width is steered to; PLACE: rows kept behind the target when a window is placed (the rest lies ahead, in the
// direction the target has been moving).
// 4 lanes per cell: a cell takes NS * (CAPB + 4) + 2 * SUBS words = 296, a workgroup of two waves 37,888 bytes: four
// per CU.
template <int SUBS> struct Cfg4;
template <> struct Cfg4<4> {
    static constexpr int NS = 8, CAPB = 32, JM = 7, MU_TARGET = 10, MU_LO = 5, MU_HI = 18, PLACE = 2, EDGE = 1;
};
template <> struct Cfg4<8> {
    static constexpr int NS = 8, CAPB = 32, JM = 8, MU_TARGET = 10, MU_LO = 5, MU_HI = 18, PLACE = 2, EDGE = 1;
};
template <> struct Cfg4<2> {
    static constexpr int NS = 8, CAPB = 16, JM = 7, MU_TARGET = 5, MU_LO = 5, MU_HI = 18, PLACE = 2, EDGE = 1;
};

typedef __attribute__((address_space(3))) uint32_t* lds_u32_ptr;
__device__ __forceinline__ lds_u32_ptr lds_at(uint32_t byte_addr) {
    return reinterpret_cast<lds_u32_ptr>(static_cast<uintptr_t>(byte_addr));
}
__device__ __forceinline__ uint32_t bfe_u(uint32_t v, uint32_t off, uint32_t width) {
    return __builtin_amdgcn_ubfe(v, off, width);
}
__device__ __forceinline__ uint32_t bfe_s(uint32_t v, uint32_t off, uint32_t width) {
    return static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int32_t>(v), off, width));
}

}  // namespace

// sflags[step]: bit 0 = SIMPLE, bit 1 = CONSEC (plan.h).  ntracks = real tracks (tracks >= ntracks are padding).
// stats (STATS builds): [0] wave-rows, [1] count passes, [2] extractions, [3] cold passes, [4] fast steps,
// [5] low word: wave-rows settled by the store alone, high word: fill passes (wave level),
// [6] low word: cell-rows that tried the store, high word: cell-rows it failed on, [7] low word: of those, target
// outside the window, high word: a row with CAPB or more keys; [1] high word: cells asking for a fill because they had no
// window, [2] high word: target near an end, [4] high word: row population out of range; [8..15] ticks per section.
// TI = float, or double for float64 input whose samples are float32-representable (narrowed on load: the protocol of
// kernels_ring2.hip).
template <int YPS, int SUBS, bool STATS, typename TI = float>
__global__ __launch_bounds__(64 * waves3(SUBS, 4), 2) void clim_ring4_f32(
    const TI* __restrict__ ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* __restrict__ table,
    const uint32_t* __restrict__ sflags, int32_t step_min, const DevChunk* __restrict__ chunks, double q,
    int negate, int32_t ntracks, double* __restrict__ thresh, double* __restrict__ seas, int64_t ldo,
    unsigned long long* __restrict__ stats, uint32_t* __restrict__ narrow_flag) {
    constexpr bool kNarrow = sizeof(TI) == 8;
    if constexpr (kNarrow) {
        if (*narrow_flag != 0) return;           // the probe (or another workgroup) already found a lossy sample
    }
    bool lossy = false;
    constexpr int W = 5;
    constexpr int R = 2 * W + 1;
    static_assert(SUBS == 8 || SUBS == 4 || SUBS == 2, "8, 4 or 2 lanes per cell");
    constexpr int kWaves3 = waves3(SUBS, 4);
    constexpr int NTP = SUBS * YPS;
    constexpr int CPWAVE = 64 / SUBS;
    constexpr int NS = Cfg4<SUBS>::NS;           // rows per window
    constexpr int LNS = NS == 8 ? 3 : 4;         // log2
    static_assert((1 << LNS) == NS, "NS is 8 or 16");
    constexpr int CAPB = Cfg4<SUBS>::CAPB;       // keys per row
    static_assert(CAPB == 32 || CAPB == 16, "rows of 32 or 16 keys");
    constexpr int LCAP = CAPB == 32 ? 5 : 4;
    constexpr int RS = CAPB + 4;                 // words per row: header at word 3, keys from word 4 (16-byte aligned)
    constexpr int CSTR = NS * RS + 2 * SUBS;     // words per cell: rows, then a dummy header + dummy slot per lane
    constexpr int LPC = NS / SUBS > 0 ? NS / SUBS : 1;    // headers a lane reads in the walk
    static_assert(LPC * SUBS == NS, "every lane reads the same number of headers");
    constexpr int EPL = CAPB / SUBS;             // keys of a row a lane reads
    static_assert(EPL == 4 || EPL == 8, "4 or 8 keys per lane");
    constexpr int J = 5;
    constexpr int JM = Cfg4<SUBS>::JM;
    constexpr uint32_t SLACK = JM - 2;
    constexpr uint32_t ALLC = (1u << YPS) - 1u;
    constexpr int32_t M16_TARGET = Cfg4<SUBS>::MU_TARGET * 16, M16_HI = Cfg4<SUBS>::MU_HI * 16,
                      M16_LO = Cfg4<SUBS>::MU_LO * 16;
    constexpr uint32_t PLACE = Cfg4<SUBS>::PLACE, EDGE = Cfg4<SUBS>::EDGE;

    __shared__ __attribute__((aligned(16))) uint32_t lds[kWaves3 * CPWAVE * CSTR];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane & (SUBS - 1);
    const int cw = lane / SUBS;
    const int64_t cell = (static_cast<int64_t>(blockIdx.x) * kWaves3 + wave) * CPWAVE + cw;
    const bool cell_ok = cell < C;
    const DevChunk ch = chunks[blockIdx.y];
    const uint32_t* tab = table + sub;           // y-major: entry of slot y at tab[step * NTP + y * SUBS]
    const TI* col = ts + (cell_ok ? cell : C - 1);
    const uint32_t negmask = negate ? 0xFFFFFFFFu : 0u;
    const uint32_t tmax = static_cast<uint32_t>(Tn - 1);
    const bool padded_last = (YPS - 1) * SUBS + sub >= ntracks;
    const uint32_t padmask = padded_last ? 0xFFFFFFFFu : 0u;
    const uint32_t full_valid = static_cast<uint32_t>((padded_last ? YPS - 1 : YPS) * R);

    uint32_t* const cellw = lds + (wave * CPWAVE + cw) * CSTR;
    // LDS byte addresses (the low 32 bits of a generic LDS pointer are the LDS offset): the cell's rows, this lane's dummy
    const uint32_t cell_addr =
        static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint32_t*)cellw));
    const uint32_t hdr0_addr = cell_addr + 12u;                              // header of physical row 0
    const uint32_t dump_addr = cell_addr + (NS * RS + 2 * sub) * 4u;         // header-like word (stays 0) + one slot
    // lane constants of the sort (lower / upper lane of a pair) and of the scan over the lanes of a cell
    const uint32_t bnd1 = (sub & 1) ? 0xFFFFFFFFu : 0u, bnd2 = (sub & 2) ? 0xFFFFFFFFu : 0u,
                   bnd4 = (sub & 4) ? 0xFFFFFFFFu : 0u;
    const uint32_t mk1 = sub >= 1 ? 0xFFFFFFFFu : 0u, mk2 = sub >= 2 ? 0xFFFFFFFFu : 0u,
                   mk4 = sub >= 4 ? 0xFFFFFFFFu : 0u;

    // the store starts empty
#pragma unroll
    for (int i = 0; i < LPC; ++i) cellw[(sub * LPC + i) * RS + 3] = 0u;
    cellw[NS * RS + 2 * sub] = 0u;

    typedef uint32_t RingT __attribute__((ext_vector_type(R)));
    RingT ring[YPS];
#pragma unroll
    for (int y = 0; y < YPS; ++y) ring[y] = kInv3;
    auto val_at = [&](int y, int k) __attribute__((always_inline)) -> double {
        return value_of_key3(opaque3(ring[y][k]));
    };
    double lsum = 0.0;
    uint32_t nval = 0;

    uint32_t tix[YPS];
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
        for (int y = 0; y < YPS; ++y) tix[y] = minu3((e[y] >> 1) - 2u, tmax);
    };
    auto advance = [&]() {
#pragma unroll
        for (int y = 0; y < YPS; ++y) tix[y] += (y == YPS - 1) ? last_step : 1u;
    };
    // (the row stride in BYTES as a 32-bit number -- the launcher refuses ld >= 2^30 -- so that a sample address is ONE
    // v_mad_u64_u32 with the column pointer as its addend)
    const uint32_t ld4 = static_cast<uint32_t>(ld) * static_cast<uint32_t>(sizeof(TI));
    auto request = [&](TI (&x)[YPS]) {
#pragma unroll
        for (int y = 0; y < YPS; ++y)
            x[y] = *reinterpret_cast<const TI*>(reinterpret_cast<const char*>(col) + static_cast<uint64_t>(tix[y]) * ld4);
    };

    TI x_in[YPS];           // the samples as requested (one row ahead)
    point_at(ch.warm_start);
    request(x_in);

    int m = (ch.warm_start - step_min) % R;
    // ---- the window of the cell (uniform over its lanes) ----
    uint32_t wbase = 0;       // lowest key of the window, a multiple of the row width
    uint32_t wspan = 0;       // NS << wshift; 0: no window
    uint32_t wshift = 0;      // log2 of the row width
    uint32_t Fw = 0;          // pooled keys below wbase, exact while wspan != 0
    uint32_t wrow0 = 0;       // physical row of the lowest row of the window
    uint32_t kfill = 0;       // the answer the window was last placed around (tells which way the target is moving)
    uint32_t wbuilt = 0;      // the cell has had a window before (its row width is then adjusted, not re-derived)
    int32_t m16 = M16_TARGET; // running mean of the target row's population, x 16
    float kpr = 8192.0f;      // keys per rank near the target (from the slow path)
    bool kpr_seen = false;
    bool clean = false;
    uint32_t st_count = 0, st_extract = 0, st_rows = 0, st_cold = 0, st_fast = 0, st_cell = 0;
    uint32_t st_band = 0, st_fill = 0, st_try = 0, st_fail = 0, st_lost = 0, st_cap = 0;
    uint32_t st_rb_inv = 0, st_rb_edge = 0, st_rb_m = 0;

    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = 0;
    if constexpr (STATS) tlast = __builtin_amdgcn_s_memtime();
    auto tick = [&](int idx) {
        if constexpr (STATS) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tacc[idx] += now - tlast;
            tlast = now;
        }
    };

    // inputs of the epilogue of the row this lane finishes (see below)
    uint32_t e_alo = 0, e_ahi = 0, e_n = 0;
    double e_total = 0.0, e_g = 0.0;

    // the row of the store a key belongs to: LDS address of its header, or of this lane's dummy if the key is outside
    // the window (below: the subtraction wraps) -- `below` is that borrow, the free probe
    auto row_of = [&](uint32_t key, uint32_t& below) -> uint32_t {
        const uint32_t d = key - wbase;
        below = key < wbase ? 1u : 0u;
        const uint32_t a = hdr0_addr + bfe_u(key, wshift, LNS) * static_cast<uint32_t>(RS * 4);
        return d < wspan ? a : dump_addr;
    };

    uint32_t hmask = 0;
    bool refill = false;      // the store lost its order (a held track): rebuild every window at the next row
    int32_t s = ch.warm_start;
    uint32_t sf_cur = __builtin_amdgcn_readfirstlane(sflags[s - step_min]);
    uint32_t sf_nxt = s + 1 < ch.end ? __builtin_amdgcn_readfirstlane(sflags[s + 1 - step_min]) : 0u;
    while (s < ch.end) {
    bool rotate = false;
    for (; s < ch.end && !rotate; ++s) {
        const uint32_t sf_nn = s + 2 < ch.end ? __builtin_amdgcn_readfirstlane(sflags[s + 2 - step_min]) : 0u;
        const uint32_t sf = sf_cur;

        // ---- what this step pushes ------------------------------------------------------
        uint32_t kin[YPS], kout[YPS];
        uint32_t cmask = ALLC;
        hmask = 0;
        bool wave_hold = false;
        float x_raw[YPS];
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            x_raw[y] = static_cast<float>(x_in[y]);
            if constexpr (kNarrow) lossy |= (static_cast<TI>(x_raw[y]) != x_in[y]) && (x_in[y] == x_in[y]);
        }
        bool row_nan;
        {
            float xs = x_raw[0];
#pragma unroll
            for (int y = 1; y < YPS; ++y) xs += x_raw[y];
            row_nan = xs != xs;
        }
        const bool fast = (sf & 1u) && clean && !__any(row_nan);
        if (fast) {
            if constexpr (STATS) ++st_fast;
            if (negate) {
#pragma unroll
                for (int y = 0; y < YPS; ++y) kin[y] = key_of_bits3_fast<true>(__float_as_uint(x_raw[y]));
            } else {
#pragma unroll
                for (int y = 0; y < YPS; ++y) kin[y] = key_of_bits3_fast<false>(__float_as_uint(x_raw[y]));
            }
            kin[YPS - 1] |= padmask;
        } else if (sf & 1u) {
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                const bool ok = x_raw[y] == x_raw[y];
                kin[y] = ok ? key_of_bits3(__float_as_uint(x_raw[y]), negmask) : kInv3;
            }
            kin[YPS - 1] |= padmask;
        } else {
            uint32_t e_cur[YPS];
            entries_of(s, e_cur);
            cmask = 0;
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                const uint32_t code = e_cur[y] >> 1;
                cmask |= (e_cur[y] & 1u) << y;
                hmask |= (code == kCodeHold ? 1u : 0u) << y;
                const bool ok = code >= 2u && x_raw[y] == x_raw[y];
                kin[y] = ok ? key_of_bits3(__float_as_uint(x_raw[y]), negmask) : kInv3;
            }
            wave_hold = __any(hmask != 0);
        }
        // ---- the one place where the rings are written (slot m of every track) ------------
#pragma unroll
        for (int y = 0; y < YPS; ++y) kout[y] = ring[y][m];
        if (wave_hold) {
#pragma unroll
            for (int y = 0; y < YPS; ++y) kin[y] = ((hmask >> y) & 1u) ? kout[y] : kin[y];
        }
#pragma unroll
        for (int y = 0; y < YPS; ++y) ring[y][m] = kin[y];
        // ---- the store mirrors the ring inside the window; the probe counts the keys below it ------
        // (a held track pushes the key it evicts: neither touches the store -- and the order of the store is lost, see
        // `refill`)
        uint32_t ha[YPS], hw[YPS];
        uint32_t dF = 0;
        if (wave_hold) {
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                uint32_t b;
                const uint32_t a = row_of(kin[y], b);
                const bool live = ((hmask >> y) & 1u) == 0 && a != dump_addr;
                ha[y] = live ? a : dump_addr;
                dF += b;
                hw[y] = __hip_atomic_fetch_add(lds_at(ha[y]), live ? 0x10001u : 0u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                uint32_t b;
                const uint32_t a = row_of(kout[y], b);
                const bool live = ((hmask >> y) & 1u) == 0 && a != dump_addr;
                dF -= b;
                __hip_atomic_fetch_add(lds_at(live ? a : dump_addr), live ? 0xFFFFFFFFu : 0u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            refill = true;
        } else {
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                uint32_t b;
                ha[y] = row_of(kin[y], b);
                dF += b;
                hw[y] = __hip_atomic_fetch_add(lds_at(ha[y]), ha[y] != dump_addr ? 0x10001u : 0u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                uint32_t b;
                const uint32_t a = row_of(kout[y], b);
                dF -= b;
                __hip_atomic_fetch_add(lds_at(a), a != dump_addr ? 0xFFFFFFFFu : 0u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (fast) {
            // (the pushed samples are summed as they are, the sum changes sign for cold spells: one instruction
            // instead of one per sample)
            double din, dout;
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                uint32_t bi = __float_as_uint(x_raw[y]);
                uint32_t bo = bits_of_key3(kout[y]);
                if (y == YPS - 1) {
                    bi &= ~padmask;
                    bo &= ~padmask;
                }
                const double di = static_cast<double>(__uint_as_float(bi));
                const double dq = static_cast<double>(__uint_as_float(bo));
                din = y == 0 ? di : din + di;
                dout = y == 0 ? dq : dout + dq;
            }
            lsum += (negate ? -din : din) - dout;
        } else {
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                lsum += value_of_key3(kin[y]);
                lsum -= value_of_key3(kout[y]);
                nval += (kin[y] != kInv3 ? 1u : 0u) - (kout[y] != kInv3 ? 1u : 0u);
            }
            rotate = wave_hold;
            clean = !__any(nval != full_valid);
        }
        m = (m + 1 == R) ? 0 : m + 1;
        // ---- prefetch: the samples of step s+1 are requested as soon as this row's are used up, into the SAME
        // registers (no second buffer, no copies; they have the rest of the row -- the selection -- to arrive)
        if (s + 1 < ch.end) {
            if (sf_nxt & 2u) advance();
            else point_at(s + 1);
            request(x_in);
        }
        // ---- the pushed keys go to the slots their headers handed out ----
#pragma unroll
        for (int y = 0; y < YPS; ++y)
            *lds_at(ha[y] + 4u + (bfe_u(hw[y], 16, LCAP) << 2)) = kin[y];
        tick(0);

        // ---- select + output (not during warm-up) ---------------------------------
        if (s >= ch.begin) {
            const bool wallc = __all(cmask == ALLC);
            uint32_t n;
            double total;
            if (wallc) {
                n = csum<SUBS>(nval);
                total = csum<SUBS>(lsum);
            } else {
                uint32_t nl = 0;
                double tl = 0.0;
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    uint32_t cy = 0;
                    double ty = 0.0;
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const uint32_t key = opaque3(ring[y][k]);
                        cy += key != kInv3 ? 1u : 0u;
                        ty = opaque3d(ty + val_at(y, k));
                    }
                    const bool cnt = (cmask >> y) & 1u;
                    nl += cnt ? cy : 0u;
                    tl += cnt ? ty : 0.0;
                }
                n = csum<SUBS>(nl);
                total = csum<SUBS>(tl);
            }
            if (__any(!(fabs(total) <= 1.7976931348623157e308))) {
                double t = 0.0, tl = 0.0;
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    double ty = 0.0;
#pragma unroll
                    for (int k = 0; k < R; ++k) ty = opaque3d(ty + val_at(y, k));
                    t += ty;
                    tl += ((cmask >> y) & 1u) ? ty : 0.0;
                }
                lsum = t;
                total = csum<SUBS>(tl);
            }
            Fw += csum<SUBS>(dF);

            const uint32_t nn = n ? n : 1u;
            const double vi = static_cast<double>(nn - 1) * q;
            const double fl = floor(vi);
            const double g = vi - fl;
            const uint32_t lo = static_cast<uint32_t>(fl);
            const bool need2 = lo + 1 < nn;

            auto count_le = [&](uint32_t p) -> uint32_t {
                uint32_t c = 0;
                if (wallc) {
                    uint32_t c2 = 0;
#pragma unroll
                    for (int y = 0; y < YPS; ++y) c2 = count_le11_3(ring[y], p, c, c2);
                    c += c2;
                } else {
#pragma unroll
                    for (int y = 0; y < YPS; ++y) {
                        uint32_t cy = 0;
#pragma unroll
                        for (int k = 0; k < R; ++k) cy += (opaque3(ring[y][k]) <= p) ? 1u : 0u;
                        c += ((cmask >> y) & 1u) ? cy : 0u;
                    }
                }
                return csum<SUBS>(c);
            };

            bool resolved = (n == 0);
            uint32_t alo = 0, ahi = 0;
            bool lost = false;            // the target is outside this cell's window
            uint32_t rl = PLACE;          // logical row of the target (cells the store settled)

            tick(1);
            // ================= the store =====================================================
            const bool btry = wallc && wspan != 0 && n != 0 && !refill;
            if (__any(btry)) {
                // ---- 1. walk: the headers of the window's rows, bottom up -------------------
                uint32_t hd[LPC], pr[LPC], pf[LPC];
#pragma unroll
                for (int i = 0; i < LPC; ++i) {
                    pr[i] = (wrow0 + static_cast<uint32_t>(sub * LPC + i)) & static_cast<uint32_t>(NS - 1);
                    hd[i] = cellw[pr[i] * RS + 3];
                }
                pf[0] = hd[0] & 0xFFFFu;
#pragma unroll
                for (int i = 1; i < LPC; ++i) pf[i] = pf[i - 1] + (hd[i] & 0xFFFFu);
                const uint32_t T = pf[LPC - 1];
                uint32_t incl = T;
                incl += dpp3<kShr1>(incl) & mk1;
                if constexpr (SUBS >= 4) incl += dpp3<kShr2>(incl) & mk2;
                if constexpr (SUBS == 8) incl += dpp3<kShr4>(incl) & mk4;
                const uint32_t excl = incl - T;
                const uint32_t t0 = lo - Fw, t1 = t0 + 1u;      // ranks inside the window (wrap: below it)
                // the row that holds rank t: found << 20 | rank inside the row << 14 | keys alive (<= 63) << 8 |
                // push count << 3 | physical row
                uint32_t s0 = 0, s1 = 0;
#pragma unroll
                for (int i = 0; i < LPC; ++i) {
                    const uint32_t P = excl + (i ? pf[i - 1] : 0u);
                    const uint32_t c = hd[i] & 0xFFFFu;
                    const uint32_t info = pr[i] | (bfe_u(hd[i], 16, LCAP) << 3) | (minu3(c, 63u) << 8) | (1u << 20);
                    s0 = (t0 - P) < c ? info | ((t0 - P) << 14) : s0;
                    s1 = (t1 - P) < c ? info | ((t1 - P) << 14) : s1;
                }
                s0 = btry ? s0 : 0u;
                const uint32_t i0 = csum<SUBS>(s0), i1 = csum<SUBS>(s1);
                const uint32_t prow = i0 & static_cast<uint32_t>(NS - 1), wp = bfe_u(i0, 3, 5), cR = bfe_u(i0, 8, 6),
                               j = bfe_u(i0, 14, 6);
                const uint32_t prow1 = i1 & static_cast<uint32_t>(NS - 1), wp1 = bfe_u(i1, 3, 5), cR1 = bfe_u(i1, 8, 6);
                const bool same = prow1 == prow;
                const bool found = (i0 >> 20) != 0 && (!need2 || (i1 >> 20) != 0);
                lost = btry && !found;
                const bool capf = btry && found && (cR >= static_cast<uint32_t>(CAPB) ||
                                                    (need2 && !same && cR1 >= static_cast<uint32_t>(CAPB)));
                const bool bok = btry && found && !capf;
                tick(2);
                // ---- 2. the keys of the target's row: dead slots blanked ------------------------
                // (alive: the last cR slots written, wp - cR .. wp - 1 modulo CAPB)
                auto read_row = [&](uint32_t p, uint32_t wpp, uint32_t cc, uint32_t (&c)[EPL]) {
                    const uint32_t* rowp = cellw + p * RS + 4 + sub * EPL;
#pragma unroll
                    for (int i = 0; i < EPL; i += 4) {
                        const uint4 v = *reinterpret_cast<const uint4*>(rowp + i);
                        c[i] = v.x; c[i + 1] = v.y; c[i + 2] = v.z; c[i + 3] = v.w;
                    }
                    uint32_t M = __builtin_amdgcn_ubfe(0xFFFFFFFFu, 0u, cc);          // cc low bits (cc < 32)
                    const uint32_t r = (wpp - cc) & static_cast<uint32_t>(CAPB - 1);
                    if constexpr (CAPB == 32) {
                        M = __builtin_amdgcn_alignbit(M, M, (32u - r) & 31u);          // rotate left by r
                    } else {
                        M <<= r;
                        M |= M >> CAPB;
                    }
                    const uint32_t dead = ~(M >> (sub * EPL));
#pragma unroll
                    for (int i = 0; i < EPL; ++i) c[i] |= bfe_s(dead, i, 1);
                };
                uint32_t c[EPL];
                read_row(prow, wp, bok ? cR : 0u, c);
                tick(3);
                // ---- 3. sort the row across the lanes of the cell, pick entries j and j + 1 -------
                sort_cell<SUBS, EPL>(c, bnd1, bnd2, bnd4);
                const uint32_t j1 = j + 1u;
                const uint32_t va = pick_reg<EPL>(c, j & (EPL - 1)), vb = pick_reg<EPL>(c, j1 & (EPL - 1));
                const uint32_t xa = cmax<SUBS>((static_cast<uint32_t>(sub) == j / EPL) ? va : 0u);
                uint32_t xb = cmax<SUBS>((static_cast<uint32_t>(sub) == j1 / EPL) ? vb : 0u);
                // ... or the smallest key of the next populated row
                const bool other = bok && need2 && !same;
                if (__any(other)) {
                    uint32_t d[EPL];
                    read_row(prow1, wp1, other ? cR1 : 0u, d);
                    uint32_t mn = d[0];
#pragma unroll
                    for (int i = 1; i < EPL; ++i) mn = minu3(mn, d[i]);
                    mn = cmin<SUBS>(mn);
                    xb = other ? mn : xb;
                }
                if constexpr (STATS) {
                    st_try += btry ? 1u : 0u;
                    st_fail += (btry && !bok) ? 1u : 0u;
                    st_lost += lost ? 1u : 0u;
                    st_cap += capf ? 1u : 0u;
                }
                if (bok) {
                    alo = xa;
                    ahi = need2 ? xb : xa;
                    resolved = true;
                    rl = (prow - wrow0) & static_cast<uint32_t>(NS - 1);
                    m16 += (static_cast<int32_t>(cR << 4) - m16) >> 3;
                } else if (capf) {
                    m16 += (static_cast<int32_t>(cR << 4) - m16) >> 2;      // overfull rows: narrow them
                }
            }
            if constexpr (STATS) st_band += __all(resolved) ? 1u : 0u;
            tick(4);

            // ================= slow path: the round-2 selection ==================================
            uint32_t top_span = 0;
            if (!__all(resolved)) {
                uint32_t pl = 0, Fl = 0, ph = 0xFFFFFFFFu, Fh = nn;
                uint32_t lreal = 0, hreal = 0;
                float grow = 1.0f;
                uint32_t p_first = 0;
                int32_t rank_gap = 0;
                {
                    // (the window's lower edge with its exact count is a bracket end for free)
                    const bool use_c = wspan != 0 && wallc && wbase > 1u;
                    uint32_t p0 = wbase - 1u, F0 = 0;
                    if (use_c) F0 = Fw;
                    if (!__all(use_c || n == 0)) {
                        uint32_t pm = key_of_bits3(__float_as_uint(static_cast<float>(total / static_cast<double>(nn))), 0u);
                        if (!use_c) p0 = (wspan != 0 && wbase > 1u) ? wbase - 1u : pm;
                        const uint32_t Fr = count_le(minu3(p0, 0xFFFFFFFEu));
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
                uint32_t slack = SLACK;
                int budget = kBudget3;
                uint32_t s_alo = 0, s_ahi = 0;
                bool sres = resolved;         // settled (by the store, or n == 0)
                for (;;) {
                    for (int it = 0;; ++it) {
                        const bool settle = sres || (lo - Fl <= slack) || (ph - pl <= 1u);
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
                        off = maxu3(1u, minu3(off, room - 1u));
                        const uint32_t p = settle ? pl : pl + off;
                        const uint32_t F = count_le(p);
                        if constexpr (STATS) {
                            ++st_count;
                            st_cell += settle ? 0u : 1u;
                        }
                        if (!settle) {
                            if (F <= lo) { pl = p; Fl = F; lreal = 1; }
                            else { ph = p; Fh = F; hreal = 1; }
                        }
                    }
                    const bool window = (lo - Fl <= slack);
                    const bool adjacent = !window && (ph - pl <= 1u);
                    const uint32_t px = adjacent ? ph : pl;
                    const uint32_t base = px + 1u;
                    Top3<J, JM> top;
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
                                const uint32_t d = opaque3(ring[y][k]) - base;
                                top.insert(((cmask >> y) & 1u) ? d : 0xFFFFFFFFu);
                            }
                    }
                    const uint32_t horizon = JM > J ? top.template horizon<SUBS>() : 0xFFFFFFFFu;
                    top.template merge_cell<SUBS>();
                    if constexpr (STATS) ++st_extract;
                    if (!sres) {
                        const uint32_t jj = window ? lo - Fl : 0u;
                        uint32_t d_lo, d_nx;
                        top.at2(jj, d_lo, d_nx);
                        const uint32_t d_hi = need2 ? d_nx : d_lo;
                        const bool exact = d_hi <= horizon || jj + (need2 ? 1u : 0u) < static_cast<uint32_t>(J);
                        if (window && !exact) slack = J - 2;
                        if (window && exact) {
                            s_alo = base + d_lo;
                            s_ahi = base + d_hi;
                            top_span = top.m[J - 1] - top.m[0];
                            sres = true;
                        } else if (adjacent && !window) {
                            s_alo = ph;
                            s_ahi = (need2 && lo + 1u >= Fh) ? base + top.m[0] : ph;
                            sres = true;
                        }
                    }
                    if (__all(sres)) break;
                    const uint32_t dj = minu3(top.m[JM - 1], horizon);
                    const uint32_t pj = base + dj;
                    const uint32_t Fj = count_le(sres ? pl : pj);
                    if constexpr (STATS) ++st_count;
                    if (!sres) {
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
                if (!resolved) {
                    alo = s_alo; ahi = s_ahi;
                    resolved = true;
                    if (n > 0) {
                        if (rank_gap > 1 || rank_gap < -1) {
                            const float obs = (static_cast<float>(alo) - static_cast<float>(p_first)) *
                                              __builtin_amdgcn_rcpf(static_cast<float>(rank_gap));
                            if (obs >= 1.0f && obs < 1.0e8f) kpr = 0.75f * kpr + 0.25f * obs;
                        }
                        if (top_span != 0) {
                            // local spacing of the keys just above the pivot: what sizes the rows
                            const float obs = static_cast<float>(top_span) * (1.0f / static_cast<float>(J - 1));
                            if (obs >= 1.0f && obs < 1.0e8f) {
                                kpr = kpr_seen ? 0.5f * kpr + 0.5f * obs : obs;
                                kpr_seen = true;
                            }
                        }
                    }
                }
            }

            tick(5);
            if constexpr (STATS) ++st_rows;
            // The epilogue (key -> value, numpy's lerp, the float64 division, the two stores) is the same ~50
            // instructions for every lane of a cell: the lanes take turns -- lane `sub` keeps the inputs of the row
            // whose number is sub modulo SUBS, and once per SUBS rows (and at the end of the chunk) every lane
            // finishes ITS row.  Same arithmetic per cell-row, a quarter (an eighth) of the instructions.
            const uint32_t eph = static_cast<uint32_t>(s - ch.begin) & static_cast<uint32_t>(SUBS - 1);
            if (static_cast<uint32_t>(sub) == eph) {
                e_alo = alo;
                e_ahi = ahi;
                e_n = n;
                e_total = total;
                e_g = g;
            }
            if (eph == static_cast<uint32_t>(SUBS - 1) || s + 1 == ch.end) {
                double th = make_nan(), se = make_nan();
                if (e_n > 0) {
                    const double v_lo = static_cast<double>(__uint_as_float(bits_of_key3(e_alo)));
                    const double v_hi = static_cast<double>(__uint_as_float(bits_of_key3(e_ahi)));
                    th = numpy_lerp(v_lo, v_hi, e_g);
                    se = e_total / static_cast<double>(e_n);
                }
                if (static_cast<uint32_t>(sub) <= eph && cell_ok) {
                    const int64_t row = static_cast<int64_t>(s) - static_cast<int64_t>(eph) + sub;
                    thresh[row * ldo + cell] = th;
                    seas[row * ldo + cell] = se;
                }
            }

            tick(6);
            // ================= window placement + fill pass ======================================
            // (rows that do not pool every track leave the windows alone: the store mirrors ALL keys of the ring)
            const bool can = wallc && n > 0;
            if (wallc && n == 0) wspan = 0;
            const bool w_inv = can && (wspan == 0 || lost || refill);
            // (the width changes only if it can: a cell of ties sits at width 1 with overfull rows)
            const bool w_pop = can && !w_inv && ((m16 > M16_HI && wshift > 0u) || (m16 < M16_LO && wshift < 26u));
            const bool w_edge = can && !w_inv && !w_pop && (rl < EDGE || rl >= static_cast<uint32_t>(NS) - EDGE);
            // (not on a row that held a track: its ring slots are out of age order until they are rotated below; the
            // row after it finds `refill` still set)
            if (!rotate && __any(w_inv || w_pop || w_edge)) {
                if constexpr (STATS) {
                    ++st_fill;
                    st_rb_inv += w_inv ? 1u : 0u;
                    st_rb_edge += w_edge ? 1u : 0u;
                    st_rb_m += w_pop ? 1u : 0u;
                }
                // every cell of the wave that has an answer is given a new place: PLACE rows behind its target, the
                // rest ahead in the direction the target moved since the last pass
                uint32_t nshift = wshift, nbase = wbase, fl0 = 0, fn = 0;
                if (can) {
                    const bool fresh = wspan == 0 || lost || refill || w_pop;     // every row is filled
                    if (wspan == 0 && wbuilt == 0) {
                        const float bw = fminf(fmaxf(kpr * static_cast<float>(Cfg4<SUBS>::MU_TARGET), 1.0f), 6.0e7f);
                        nshift = 31u - static_cast<uint32_t>(__builtin_clz(static_cast<uint32_t>(bw)));
                        m16 = M16_TARGET;
                    } else {
#pragma unroll
                        for (int it = 0; it < 3; ++it) {
                            const bool dn = m16 > M16_HI && nshift > 0u, upw = m16 < M16_LO && nshift < 26u;
                            nshift = dn ? nshift - 1u : upw ? nshift + 1u : nshift;
                            m16 = dn ? m16 >> 1 : upw ? m16 << 1 : m16;
                        }
                    }
                    nshift = minu3(nshift, 26u);
                    const bool upward = alo >= kfill;
                    const uint32_t behind = upward ? PLACE : static_cast<uint32_t>(NS - 1) - PLACE;
                    uint32_t rb = alo >> nshift;                              // the target's row, in rows from key 0
                    rb = rb > behind + 1u ? rb - behind : 1u;                 // (row 0 is never inside a window)
                    rb = minu3(rb, (0xFFFFFFFFu >> nshift) - static_cast<uint32_t>(NS));   // window inside the key space
                    nbase = rb << nshift;
                    // rows of the new window (bottom up) that are not rows of the old one
                    const int32_t k = static_cast<int32_t>(rb) - static_cast<int32_t>(wbase >> wshift);
                    if (fresh || nshift != wshift || k >= NS || k <= -NS) {
                        fl0 = 0;
                        fn = NS;
                    } else if (k > 0) {
                        fl0 = static_cast<uint32_t>(NS - k);
                        fn = static_cast<uint32_t>(k);
                    } else {
                        fl0 = 0;
                        fn = static_cast<uint32_t>(-k);
                    }
                    kfill = alo;
                    wbuilt = 1;
                }
                const uint32_t nrow0 = (nbase >> nshift) & static_cast<uint32_t>(NS - 1);
                // the headers of the new rows start at 0
#pragma unroll
                for (int i = 0; i < LPC; ++i) {
                    const uint32_t l = static_cast<uint32_t>(sub * LPC + i);
                    if (l - fl0 < fn) cellw[((nrow0 + l) & static_cast<uint32_t>(NS - 1)) * RS + 3] = 0u;
                }
                // one pass over the ring, oldest slot first: the keys of the new rows are stored, the keys below the
                // new window counted
                const uint32_t fbase = nbase + (fl0 << nshift), fspan = fn << nshift;
                uint32_t below = 0;
                int k_age = m;
                for (int a = 0; a < R; ++a) {
                    uint32_t fa[YPS], fw[YPS], fk[YPS];
#pragma unroll
                    for (int y = 0; y < YPS; ++y) {
                        fk[y] = ring[y][k_age];
                        below += fk[y] < nbase ? 1u : 0u;
                        const bool in = (fk[y] - fbase) < fspan;
                        fa[y] = in ? hdr0_addr + bfe_u(fk[y], nshift, LNS) * static_cast<uint32_t>(RS * 4) : dump_addr;
                        fw[y] = __hip_atomic_fetch_add(lds_at(fa[y]), in ? 0x10001u : 0u, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
#pragma unroll
                    for (int y = 0; y < YPS; ++y) *lds_at(fa[y] + 4u + (bfe_u(fw[y], 16, LCAP) << 2)) = fk[y];
                    k_age = (k_age + 1 == R) ? 0 : k_age + 1;
                }
                below = csum<SUBS>(below);
                if (can) {
                    wbase = nbase;
                    wshift = nshift;
                    wspan = static_cast<uint32_t>(NS) << nshift;
                    wrow0 = nrow0;
                    Fw = below;
                }
                refill = false;
            }
        }

        tick(7);
        // The waves of a workgroup read the two halves of the same 128-byte lines: they are marched in step every 32
        // rows (kernels_ring3.hip, profiles/r3_rendezvous.txt).
        if constexpr (kNarrow) {
            // (no rendezvous here: a wave that has seen a lossy sample leaves, and the others must not wait for it)
            if ((s & 63) == 63 && __any(lossy)) {
                if (lossy) atomicOr(narrow_flag, 1u);
                return;
            }
        } else if constexpr (kWaves3 > 1) {
            if ((s & 31) == 31) __syncthreads();
        }
        sf_cur = sf_nxt;
        sf_nxt = sf_nn;
    }
    if (rotate) {
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            const unsigned long long hy = __builtin_amdgcn_ballot_w64(((hmask >> y) & 1u) != 0);
            const uint32_t last = opaque3(ring[y][R - 1]);
            asm volatile("s_nop 1");
#pragma unroll
            for (int k = R - 1; k >= 1; --k) {
                uint32_t e = ring[y][k];
                ring_sel3(e, ring[y][k - 1], hy);
                ring[y][k] = e;
            }
            uint32_t e0 = ring[y][0];
            ring_sel3(e0, last, hy);
            ring[y][0] = e0;
        }
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
        atomicAdd(&stats[5], static_cast<unsigned long long>(st_band) | (static_cast<unsigned long long>(st_fill) << 32));
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&stats[8 + i], tacc[i]);
    }
    if (STATS && stats != nullptr && sub == 0 && cell_ok) {
        atomicAdd(&stats[6], static_cast<unsigned long long>(st_try) | (static_cast<unsigned long long>(st_fail) << 32));
        atomicAdd(&stats[7], static_cast<unsigned long long>(st_lost) | (static_cast<unsigned long long>(st_cap) << 32));
        atomicAdd(&stats[1], static_cast<unsigned long long>(st_rb_inv) << 32);
        atomicAdd(&stats[2], static_cast<unsigned long long>(st_rb_edge) << 32);
        atomicAdd(&stats[4], static_cast<unsigned long long>(st_rb_m) << 32);
    }
}

// ---------------------------------------------------------------------------
namespace {
typedef void (*Ring4Kernel)(const float*, int64_t, int64_t, int64_t, const uint32_t*, const uint32_t*, int32_t,
                            const DevChunk*, double, int, int32_t, double*, double*, int64_t, unsigned long long*,
                            uint32_t*);
typedef void (*Ring4KernelN)(const double*, int64_t, int64_t, int64_t, const uint32_t*, const uint32_t*, int32_t,
                             const DevChunk*, double, int, int32_t, double*, double*, int64_t, unsigned long long*,
                             uint32_t*);
struct Ring4Entry { int yps, subs; Ring4Kernel fn, fn_stats; Ring4KernelN fn_narrow; };
#ifdef XMHW_RING_STATS
#define XMHW_R4S(Y, S) clim_ring4_f32<Y, S, true>
#else
#define XMHW_R4S(Y, S) nullptr
#endif
#define XMHW_R4(Y, S) {Y, S, clim_ring4_f32<Y, S, false>, XMHW_R4S(Y, S), nullptr}
const Ring4Entry kRing4[] = {
    XMHW_R4(7, 4), XMHW_R4(8, 4), XMHW_R4(9, 4), XMHW_R4(10, 4), XMHW_R4(11, 4), XMHW_R4(12, 4),
};
#undef XMHW_R4
const Ring4Entry* find_ring4(int32_t yps, int32_t subs) {
    for (const auto& e : kRing4)
        if (e.yps == yps && e.subs == subs) return &e;
    return nullptr;
}
}  // namespace

int32_t ring4_pick_yps(int32_t w, int32_t ntracks, int32_t subs) {
    if (w != 5) return 0;
    int32_t best = 0;
    for (const auto& e : kRing4)
        if (e.subs == subs && e.yps * subs >= ntracks && (best == 0 || e.yps < best)) best = e.yps;
    if (best && (best - 1) * subs >= ntracks) return 0;      // padding may only sit in the last slot of a lane
    return best;
}

bool ring4_supported(int32_t w, int32_t yps, int32_t subs) { return w == 5 && find_ring4(yps, subs) != nullptr; }

bool ring4_stats_built() {
#ifdef XMHW_RING_STATS
    return true;
#else
    return false;
#endif
}

hipError_t launch_ring4_f32(const float* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                            const uint32_t* sflags, int32_t step_min, const DevChunk* chunks, int32_t nchunks,
                            int32_t w, int32_t yps, int32_t subs, int32_t ntracks, double q, int negate,
                            double* thresh, double* seas, int64_t ldo, hipStream_t stream,
                            unsigned long long* stats) {
    const Ring4Entry* e = w == 5 ? find_ring4(yps, subs) : nullptr;
    if (!e || ld >= (int64_t(1) << 30)) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    const int kWaves3 = waves3(subs, 4);
    const int64_t cells_per_block = (64 / subs) * kWaves3;
    dim3 grid(static_cast<unsigned>((C + cells_per_block - 1) / cells_per_block), static_cast<unsigned>(nchunks));
    Ring4Kernel fn = (stats && e->fn_stats) ? e->fn_stats : e->fn;
    hipLaunchKernelGGL(fn, grid, dim3(64 * kWaves3), 0, stream, ts, C, ld, Tn, table, sflags, step_min, chunks, q,
                       negate, ntracks, thresh, seas, ldo, (stats && e->fn_stats) ? stats : nullptr,
                       static_cast<uint32_t*>(nullptr));
    return hipGetLastError();
}

}  // namespace xmhw
