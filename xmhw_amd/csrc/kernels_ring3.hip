// kernels_ring3.hip -- third-generation ring kernel (round 3): selection by BAND COMPACTION.
//
// The ring (R = 2w+1 samples of every owned track as order-preserving 32-bit keys in VGPRs), the step
// table, the push logic, the running sums and the carried pivot with its exact count ("free probe")
// are those of kernels_ring2.hip.  What changed is how the two order statistics of numpy's linear
// quantile are found.  Round 2 closed a bracket with count passes over ALL pooled keys (2 instructions
// per key and pass, 1.92 passes per wave-row because a wave iterates until the slowest of its cells has
// settled) and then ran an insertion network over ALL keys (6 instructions per key): 75 % of the kernel.
// Here, per row:
//
//   1. a per-cell HISTOGRAM in LDS (NB buckets of 2^shift keys around the target, one ds_add per pushed
//      and per evicted key) mirrors the ring.  The carried pivot is the lower edge of a bucket, so the
//      free probe gives the exact number of keys below that edge; a prefix sum over the 16 buckets next
//      to it (one LDS read per lane, a scan over the lanes of the cell) finds the buckets B0..B1 that
//      hold order statistics lo and lo+1, the exact count Cb of keys below B0 and the exact number m of
//      keys inside B0..B1 -- no pass over the keys, no iteration, nothing a slow cell could hold the
//      wave up with;
//   2. ONE pass over the keys (v_sub, v_cmp, and under the resulting EXEC mask ds_write + v_add: three
//      vector instructions per key) appends the m keys of the band to per-lane lists in LDS;
//   3. the lanes of the cell read their lists back (<= CAP keys each), sort the CAP x lanes slots with a
//      bitonic network across the lanes (DPP) and pick entries lo - Cb and lo - Cb + 1.
//
// Every step is exact (the histogram is an exact mirror of the ring, band edges are bucket edges in key
// space); what can go wrong is only capacity (m > list space, more than CAP band keys in one lane: ties,
// constant cells), the target leaving the 16 buckets looked at, or the window.  Those rows -- and rows
// that do not pool every track (Feb 29), and the first row of a chunk -- take the round-2 selection
// (count passes + extraction), which is kept verbatim as the slow path (a cell that overflows keeps its
// anchor: pivot and count stay exact, so nothing has to be rebuilt for it).
// The window is rebuilt (re-centred, bucket width re-derived) for all cells of the wave when any cell's
// target comes near an end of its window or its band population drifts out of range.
//
// Lane layout: the lanes of a cell are ADJACENT (lane = cell_in_wave * SUBS + sub), so that every
// exchange inside a cell is one quad_perm / row_half_mirror DPP operation.  A workgroup covers one 128-byte
// line of a sample row (two waves of 16 float32 cells at 4 lanes per cell).
//
// Three instantiations per layout: float32 input; float64 input whose samples are float32-representable
// (narrowed on load, gives up at the first lossy sample); genuinely float64 samples (64-bit mode: the rings
// and everything above hold the HIGH words of the 64-bit keys, the low words ride in a second set of tuples
// and in the band lists -- 8 lanes per cell up to 6 tracks per lane, 4 lanes per cell at 4 and 5).
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


// NB buckets per cell, CAP sorted list slots per lane, LW words per lane list (= most keys a band may hold: the bound
// is the cell's total, all of it may fall into one lane), JM merged slow-path list.
// 4 lanes: a workgroup (2 waves) takes 2 * 16 * (NB + 1) + 128 * LW words = 40,064 bytes of LDS, four of them a CU's
// 160 KB.  Lists of 24 instead of 16 words (and 216 instead of 224 buckets to pay for them) send 60 % fewer bands to
// the slow path for being too populous: 56.9 -> 54.7 ms; 28 words / 200 buckets is no better, 24 words with 224
// buckets costs the fourth workgroup (70.7 ms).
template <int SUBS> struct Cfg3;
template <> struct Cfg3<4> {   // 16 cells per wave
    // (round 4: 200 buckets instead of 216 pay for the per-cell ring of row sums, 11 float64 per cell -- kRowSum below)
    static constexpr int NB = 200, CAP = 8, LW = 24, JM = 7, PS = 16, EDGE_LO = 20, EDGE_HI = 36;
};
// 2 lanes: 32 cells per wave, a workgroup is ONE wave (32 float32 cells = one 128-byte line): 32 * (NB + 1) + 64 * LW
// words = 19.6 KB, eight of them per CU.  Records of up to 24 tracks (264 pooled keys: 120 buckets cover them all).
template <> struct Cfg3<2> {
    // (bands three quarters as populous as on 4 lanes -- PS = 12 -- because a lane holds half of a cell's band and sorts
    // at most 8 keys: 6-hourly share of 405,000 cells 56.9 ms at PS = 16, 48.1 at 12, 49.8 at 11, 56.7 at 10)
    // (round 4: 96 buckets instead of 120 pay for the ring of row sums: eight workgroups per CU still fit)
    static constexpr int NB = 96, CAP = 8, LW = 16, JM = 7, PS = 12, EDGE_LO = 14, EDGE_HI = 22;
};
template <> struct Cfg3<8> {   // 8 cells per wave
    // (208 buckets and lists of 20 entries instead of 128 / 16: float64 configs[2] 103.8 -> 97.7 ms, fewer window
    // rebuilds and over-populated bands; a workgroup of two float64 waves takes 34 KB of LDS)
    static constexpr int NB = 208, CAP = 4, LW = 20, JM = 8, PS = 16, EDGE_LO = 20, EDGE_HI = 36;
};
// buckets are sized to hold about this many pooled keys near the target
constexpr float kBucketRanks = 3.5f;

}  // namespace

// sflags[step]: bit 0 = SIMPLE, bit 1 = CONSEC (plan.h).  ntracks = real tracks (tracks >= ntracks are padding).
// stats (STATS builds): [0] wave-rows, [1] count passes, [2] extractions, [3] cold passes, [4] fast steps,
// [5] low word: wave-rows settled by the band path alone, high word: window rebuilds (wave level),
// [6] low word: cell-rows that tried the band path, high word: cell-rows it failed on, [7] low word: of those,
// target off the block / window, high word: band larger than a list ([3] high word: histogram mismatches, must be 0).
// TI = float, or double for float64 input whose samples are float32-representable (float32 archives promoted by a
// reader): the samples are narrowed on load, `narrow_flag` is set as soon as one does not survive the round trip, the
// kernel gives up and the float64 kernel queued behind it (which looks at the same flag) does the work instead
// (the protocol of kernels_ring2.hip).
// X64 (TI = double): genuinely float64 samples.  The rings hold the HIGH words of the 64-bit keys -- histogram, walk,
// band compaction, sort and the slow path run on them exactly as on float32 keys -- and, in a second set of tuples,
// the LOW words; once the two order statistics are known by their high words, one pass over the rings fetches their low
// words (a tie of high words -- repeated values, or distinct doubles within 2^-20 relative of each other at the target
// rank -- is settled by successive minima of the low words: kernels_ring2.hip).  `narrow_flag` is then the RUN flag:
// the kernel is queued behind the narrowing one and returns unless that one gave up.
template <int YPS, int SUBS, bool STATS, typename TI = float, bool X64 = false>
__global__ __launch_bounds__(64 * waves3(SUBS, X64 ? 8 : 4), 2) void clim_ring3_f32(
    const TI* __restrict__ ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* __restrict__ table,
    const uint32_t* __restrict__ sflags, int32_t step_min, const DevChunk* __restrict__ chunks, double q,
    int negate, int32_t ntracks, double* __restrict__ thresh, double* __restrict__ seas, int64_t ldo,
    unsigned long long* __restrict__ stats, uint32_t* __restrict__ narrow_flag) {
    static_assert(!X64 || sizeof(TI) == 8, "the 64-bit mode takes double input");
    constexpr bool kNarrow = sizeof(TI) == 8 && !X64;
    if constexpr (kNarrow) {
        if (*narrow_flag != 0) return;           // the probe (or another workgroup) already found a lossy sample
    }
    if constexpr (X64) {
        if (narrow_flag != nullptr && *narrow_flag == 0) return;     // the narrowing kernel did the work
    }
    bool lossy = false;
    constexpr int W = 5;
    constexpr int R = 2 * W + 1;
    static_assert(SUBS == 8 || SUBS == 4 || SUBS == 2, "8, 4 or 2 lanes per cell");
    // (the narrowing instantiation keeps the float32 shape: its waves never wait for each other, and single-wave
    // workgroups measured 7 % slower)
    constexpr int kWaves3 = waves3(SUBS, X64 ? 8 : 4);
    constexpr int NTP = SUBS * YPS;
    constexpr int CPWAVE = 64 / SUBS;
    constexpr int NB = Cfg3<SUBS>::NB;           // buckets per cell
    constexpr int HS = NB + 1;                   // words per cell histogram (one bank of skew per cell)
    constexpr int CAP = Cfg3<SUBS>::CAP;         // list slots per lane that are sorted
    constexpr int LW = Cfg3<SUBS>::LW;           // entries per lane list = most keys a band may hold (MCAP)
    constexpr int LWL = X64 ? 2 * LW : LW;       // words per lane list (64-bit mode: an entry is a high and a low word)
    constexpr int Q = 16 / SUBS;                 // buckets per lane of the 16 the walk looks at
    constexpr int J = 5;
    constexpr int JM = Cfg3<SUBS>::JM;
    constexpr uint32_t SLACK = JM - 2;
    constexpr uint32_t ALLC = (1u << YPS) - 1u;
    constexpr int EDGE_LO = Cfg3<SUBS>::EDGE_LO, EDGE_HI = Cfg3<SUBS>::EDGE_HI;    // rebuild when the target bucket is this close to an end
    // band population x 16 (running mean): a rebuild is asked for outside [LO_TRIG, HI_TRIG]; whenever the wave
    // rebuilds, every cell outside [LO_ADJ, HI_ADJ] changes its bucket width too (so that it does not ask for
    // a rebuild of its own a few rows later)
    // (PS: the thresholds in units of 1/16 key, 16 = as measured for 4 and 8 lanes per cell; the 2-lane layout, whose
    // lane holds half a cell's band, keeps its bands smaller)
    constexpr int32_t PS = Cfg3<SUBS>::PS;
    constexpr int32_t M16_TARGET = PS * 5, M16_HI_TRIG = PS * 11, M16_LO_TRIG = PS * 5 / 2, M16_HI_ADJ = PS * 15 / 2,
                      M16_LO_ADJ = PS * 7 / 2;

    // Row sums (every layout, float32 keys and the 64-bit mode): the cell's sum of the values in ring slot k, for every slot, as float64 in
    // LDS.  A row then adds (sum of the pushed samples) - (row sum of the slot it overwrites) to the cell's running total:
    // the evicted samples are not converted and summed again (10 x (key -> float, cvt, add) per lane and row: 45 vector
    // instructions, 20 of them float64), and the total is the cell's, not the lane's.
    constexpr bool kRowSum = true;
    constexpr int RSW = kRowSum ? 2 * R : 0;     // words per cell
    constexpr int kRsBase = kWaves3 * CPWAVE * HS + 64 * kWaves3 * LWL;
    static_assert(!kRowSum || kRsBase % 2 == 0, "row sums are 8-byte aligned");
    __shared__ __attribute__((aligned(16))) uint32_t lds[kRsBase + kWaves3 * CPWAVE * RSW];

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

    uint32_t* const hist = lds + (wave * CPWAVE + cw) * HS;
    uint32_t* const list = lds + kWaves3 * CPWAVE * HS + threadIdx.x * LWL;
    // LDS byte address of this lane's list (the low 32 bits of a generic LDS pointer are the LDS offset)
    const uint32_t list_addr =
        static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint32_t*)list));
    // lane constants of the sort (lower / upper lane of a pair) and of the scan over the lanes of a cell
    const uint32_t bnd1 = (sub & 1) ? 0xFFFFFFFFu : 0u, bnd2 = (sub & 2) ? 0xFFFFFFFFu : 0u,
                   bnd4 = (sub & 4) ? 0xFFFFFFFFu : 0u;
    const uint32_t mk1 = sub >= 1 ? 0xFFFFFFFFu : 0u, mk2 = sub >= 2 ? 0xFFFFFFFFu : 0u,
                   mk4 = sub >= 4 ? 0xFFFFFFFFu : 0u;
    const uint32_t mk12 = sub <= 12 / Q - 1 ? 0xFFFFFFFFu : 0u;       // lanes that hold buckets 0..11 of a block

    typedef uint32_t RingT __attribute__((ext_vector_type(R)));
    RingT ring[YPS];
#pragma unroll
    for (int y = 0; y < YPS; ++y) ring[y] = kInv3;
    RingT ringlo[X64 ? YPS : 1];                 // 64-bit mode: the low words of the keys, slot for slot
#pragma unroll
    for (int y = 0; y < (X64 ? YPS : 1); ++y) ringlo[y] = kInv3;
    // the value a key stands for (0 for an invalid key): slot k of track y / a key with its low word
    auto val_at = [&](int y, int k) __attribute__((always_inline)) -> double {
        if constexpr (X64) return value_of_key64_3(opaque3(ring[y][k]), opaque3(ringlo[y][k]));
        else return value_of_key3(opaque3(ring[y][k]));
    };
    double lsum = 0.0;       // (kRowSum: the CELL's running total, uniform over its lanes)
    uint32_t nval = 0;
    double* const rowsum = reinterpret_cast<double*>(lds + kRsBase) + (kRowSum ? (wave * CPWAVE + cw) * R : 0);
    if constexpr (kRowSum) {
#pragma unroll
        for (int k = sub; k < R; k += SUBS) rowsum[k] = 0.0;     // the rings start invalid: every value is 0
    }
    bool rs_stale = false;   // the row sums no longer describe the slots (a held track was rotated): re-sum

    // The address of every track's next sample.  kPtr (float32 input, up to 11 tracks per lane -- where 10 more registers do
    // not spill): a 64-bit pointer per track; a row that follows its predecessor adds the row stride, two full-rate
    // additions instead of one v_mad_u64_u32 (configs[2] 53.0 -> 52.4 ms).  Otherwise a 32-bit time index per track and one
    // v_mad_u64_u32 per load (12 tracks per lane: 31.1 -> 33.3 ms with the pointers, they spill).
    constexpr bool kPtr = !X64 && sizeof(TI) == 4 && YPS <= 11;
    const char* ap[kPtr ? YPS : 1];
    uint32_t tix[kPtr ? 1 : YPS];
    const uint32_t last_step = padded_last ? 0u : 1u;
    auto entries_of = [&](int32_t step, uint32_t (&e)[YPS]) {
        const uint32_t* p = tab + static_cast<int64_t>(step - step_min) * NTP;
#pragma unroll
        for (int y = 0; y < YPS; ++y) e[y] = p[y * SUBS];
    };
    // (the row stride in BYTES as a 32-bit number -- the launcher refuses ld >= 2^30 -- so that a sample address is ONE
    // v_mad_u64_u32 with the column pointer as its addend)
    const uint32_t ld4 = static_cast<uint32_t>(ld) * static_cast<uint32_t>(sizeof(TI));
    auto point_at = [&](int32_t step) {
        uint32_t e[YPS];
        entries_of(step, e);
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            const uint32_t t = minu3((e[y] >> 1) - 2u, tmax);
            if constexpr (kPtr) ap[y] = reinterpret_cast<const char*>(col) + static_cast<uint64_t>(t) * ld4;
            else tix[y] = t;
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            if constexpr (kPtr) ap[y] += (y == YPS - 1) ? static_cast<uint64_t>(last_step * ld4) : static_cast<uint64_t>(ld4);
            else tix[y] += (y == YPS - 1) ? last_step : 1u;
        }
    };
    auto request = [&](TI (&x)[YPS]) {
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            if constexpr (kPtr) x[y] = *reinterpret_cast<const TI*>(ap[y]);
            else x[y] = *reinterpret_cast<const TI*>(reinterpret_cast<const char*>(col) + static_cast<uint64_t>(tix[y]) * ld4);
        }
    };

    TI x_in[YPS];           // the samples as requested (one row ahead)
    point_at(ch.warm_start);
    request(x_in);

    int m = (ch.warm_start - step_min) % R;
    // carried across rows, uniform over the lanes of a cell
    uint32_t pc = 0, Fc = 0;  // pivot with its exact count #{keys <= pc}; with a valid window: pc = edge(A) - 1
    uint32_t have_c = 0;
    float kpr = 8192.0f;      // keys per rank near the target
    bool kpr_seen = false;
    bool clean = false;
    // histogram window of the cell: keys [hbase, hbase + (NB << hshift)), bucket = (key - hbase) >> hshift
    // clamped to 0 (everything below) .. NB-1 (everything above, the invalid keys included)
    uint32_t hbase = 0, hshift = 0, hvalid = 0;
    uint32_t A = 0;           // anchor bucket: pc + 1 is its lower edge
    int32_t m16 = M16_TARGET; // running mean of the band population, x 16
    uint32_t hbuilt = 0;      // the cell has had a window before (its bucket width is then adjusted, not re-derived)
    uint32_t st_count = 0, st_extract = 0, st_rows = 0, st_cold = 0, st_fast = 0, st_cell = 0;
    uint32_t st_band = 0, st_rebuild = 0, st_try = 0, st_fail = 0, st_lost = 0, st_cap = 0, st_mm = 0;
    uint32_t st_rb_inv = 0, st_rb_edge = 0, st_rb_m = 0, st_second = 0;

    // STATS builds: shader-clock ticks per section of the row loop (a tick waits for the LDS queue to drain, so a
    // section is charged with the LDS work it issued)
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

    auto tag_of = [&](uint32_t key) -> uint32_t {
        return minu3(__builtin_elementwise_sub_sat(key, hbase) >> hshift, static_cast<uint32_t>(NB - 1));
    };

    // inputs of the epilogue of the row this lane finishes (see below)
    uint32_t e_alo = 0, e_ahi = 0, e_n = 0, e_la = 0, e_lb = 0;
    double e_total = 0.0, e_g = 0.0;

    uint32_t hmask = 0;
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
        // the samples of this row as float32 (narrowed and checked for float64 input)
        float x_raw[YPS];
        uint32_t khi[X64 ? YPS : 1], kin_lo[X64 ? YPS : 1], kout_lo[X64 ? YPS : 1];      // 64-bit mode: key words
        uint32_t nanbits = 0;                                                              // ... bit y: sample y is NaN
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            x_raw[y] = X64 ? 0.0f : static_cast<float>(x_in[y]);
            if constexpr (kNarrow) lossy |= (static_cast<TI>(x_raw[y]) != x_in[y]) && (x_in[y] == x_in[y]);
            if constexpr (X64) {
                const double v = static_cast<double>(x_in[y]);
                key64_of3(v, negmask, khi[y], kin_lo[y]);
                nanbits |= (v != v ? 1u : 0u) << y;
            }
        }
        bool row_nan;
        if constexpr (X64) {
            row_nan = nanbits != 0;
        } else {
            float xs = x_raw[0];
#pragma unroll
            for (int y = 1; y < YPS; ++y) xs += x_raw[y];
            row_nan = xs != xs;
        }
        const bool fast = (sf & 1u) && clean && !__any(row_nan);
        auto key_in = [&](int y) -> uint32_t {
            if constexpr (X64) return khi[y];
            else return key_of_bits3(__float_as_uint(x_raw[y]), negmask);
        };
        auto is_nan = [&](int y) -> bool {
            if constexpr (X64) return ((nanbits >> y) & 1u) != 0;
            else return x_raw[y] != x_raw[y];
        };
        if (fast) {
            if constexpr (STATS) ++st_fast;
            if constexpr (X64) {
#pragma unroll
                for (int y = 0; y < YPS; ++y) kin[y] = khi[y];
                kin_lo[YPS - 1] |= padmask;
            } else if (negate) {
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
                const bool ok = !is_nan(y);
                kin[y] = ok ? key_in(y) : kInv3;
                if constexpr (X64) kin_lo[y] = ok ? kin_lo[y] : kInv3;
            }
            kin[YPS - 1] |= padmask;
            if constexpr (X64) kin_lo[YPS - 1] |= padmask;
        } else {
            uint32_t e_cur[YPS];
            entries_of(s, e_cur);
            cmask = 0;
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                const uint32_t code = e_cur[y] >> 1;
                cmask |= (e_cur[y] & 1u) << y;
                hmask |= (code == kCodeHold ? 1u : 0u) << y;
                const bool ok = code >= 2u && !is_nan(y);
                kin[y] = ok ? key_in(y) : kInv3;
                if constexpr (X64) kin_lo[y] = ok ? kin_lo[y] : kInv3;
            }
            wave_hold = __any(hmask != 0);
        }
        // ---- the one place where the rings are written (slot m of every track) ------------
#pragma unroll
        for (int y = 0; y < YPS; ++y) kout[y] = ring[y][m];
        if constexpr (X64) {
#pragma unroll
            for (int y = 0; y < YPS; ++y) kout_lo[y] = opaque3(ringlo[y][m]);      // (opaque: or the extract is sunk below
                                                                                      // the write and the old tuple kept)
        }
        if (wave_hold) {
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                kin[y] = ((hmask >> y) & 1u) ? kout[y] : kin[y];
                if constexpr (X64) kin_lo[y] = ((hmask >> y) & 1u) ? kout_lo[y] : kin_lo[y];
            }
        }
#pragma unroll
        for (int y = 0; y < YPS; ++y) ring[y][m] = kin[y];
        if constexpr (X64) {
#pragma unroll
            for (int y = 0; y < YPS; ++y) ringlo[y][m] = kin_lo[y];
        }
        // ---- the histogram mirrors the ring (a held track adds and removes the same key) ------
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            __hip_atomic_fetch_add(&hist[tag_of(kin[y])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&hist[tag_of(kout[y])], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // the free probe: how the count of keys <= the carried pivot changes (in ONE place: written into both branches
        // below, the compiler computed the twenty compares twice on the plain rows)
        uint32_t dF = 0;
#pragma unroll
        for (int y = 0; y < YPS; ++y) dF += (kin[y] <= pc ? 1u : 0u) - (kout[y] <= pc ? 1u : 0u);
        if constexpr (kRowSum) {
            // the sum of what this lane pushes (an invalid key counts 0; a held track pushes the key it evicts) ...
            double din;
            if (fast) {
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    double di;
                    if constexpr (X64) {
                        // (from the key words: the samples themselves are not kept; a padded track's key is invalid = 0)
                        di = value_of_key64_3(kin[y], kin_lo[y]);
                    } else {
                        uint32_t bi = __float_as_uint(x_raw[y]);
                        if (y == YPS - 1) bi &= ~padmask;
                        di = static_cast<double>(__uint_as_float(bi));
                    }
                    din = y == 0 ? di : din + di;
                }
                din = (negate && !X64) ? -din : din;
            } else {
                din = 0.0;
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    if constexpr (X64) din += value_of_key64_3(kin[y], kin_lo[y]);
                    else din += value_of_key3(kin[y]);
                    nval += (kin[y] != kInv3 ? 1u : 0u) - (kout[y] != kInv3 ? 1u : 0u);
                }
                rotate = wave_hold;
                clean = !__any(nval != full_valid);
            }
            // ... replaces the cell's row sum of slot m
            const double cell_in = csum<SUBS>(din);
            const double cell_out = rowsum[m];
            rowsum[m] = cell_in;
            lsum += cell_in - cell_out;
        } else
        if (fast) {
            // (the pushed samples are summed as they are, the sum changes sign for cold spells: one instruction
            // instead of one per sample)
            double din, dout;
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                double di, dq;
                if constexpr (X64) {
                    // (from the key words: the samples themselves are not kept; a padded track's key is invalid = 0)
                    di = value_of_key64_3(kin[y], kin_lo[y]);
                    dq = value_of_key64_3(kout[y], kout_lo[y]);
                } else {
                    uint32_t bi = __float_as_uint(x_raw[y]);
                    uint32_t bo = bits_of_key3(kout[y]);
                    if (y == YPS - 1) {
                        bi &= ~padmask;
                        bo &= ~padmask;
                    }
                    di = static_cast<double>(__uint_as_float(bi));
                    dq = static_cast<double>(__uint_as_float(bo));
                }
                din = y == 0 ? di : din + di;
                dout = y == 0 ? dq : dout + dq;
            }
            lsum += ((negate && !X64) ? -din : din) - dout;
        } else {
#pragma unroll
            for (int y = 0; y < YPS; ++y) {
                if constexpr (X64) {
                    lsum += value_of_key64_3(kin[y], kin_lo[y]);
                    lsum -= value_of_key64_3(kout[y], kout_lo[y]);
                } else {
                lsum += value_of_key3(kin[y]);
                lsum -= value_of_key3(kout[y]);
                }
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
        tick(0);

        // ---- select + output (not during warm-up) ---------------------------------
        if (s >= ch.begin) {
            const bool wallc = __all(cmask == ALLC);
            uint32_t n;
            double total;
            if (wallc) {
                n = csum<SUBS>(nval);
                if constexpr (kRowSum) total = lsum;
                else total = csum<SUBS>(lsum);
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
                if constexpr (kRowSum) {
                    // (a non-finite total: an infinity in the ring, or one that has just left it -- the cell's total and
                    // its row sums are taken from the rings again)
                    double tc = 0.0;
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        double v = 0.0;
#pragma unroll
                        for (int y = 0; y < YPS; ++y) v = opaque3d(v + val_at(y, k));
                        v = csum<SUBS>(v);
                        rowsum[k] = v;
                        tc += v;
                    }
                    lsum = tc;
                    total = wallc ? tc : csum<SUBS>(tl);
                } else {
                lsum = t;
                total = csum<SUBS>(tl);
                }
            }
            Fc += csum<SUBS>(dF);

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
            uint32_t la = 0, lb = 0;      // 64-bit mode: the low words of the two order statistics
            bool low_done = false;        // ... already taken from the band lists
            uint32_t alo = 0, ahi = 0, pe = 0, Fe = 0;
            bool band_done = false;       // this cell was settled by the band path
            bool lost = false;            // this cell's window must be rebuilt (anchor lost, target off the block)

            tick(1);
            // ================= band path =====================================================
            const bool btry = wallc && hvalid != 0 && have_c != 0 && n != 0;
            if (__any(btry)) {
                // ---- 1. walk: the 16 buckets next to the anchor -------------------------
                // one look at a block of 16 buckets starting at S_: where the two ranks are inside it, given the keys
                // below the block (csA, less the keys of its first 12 buckets / of the whole block where the block
                // lies below the edge the count belongs to)
                auto look = [&](uint32_t S_, bool ok_, uint32_t csA, bool sub_p12, bool sub_tot, uint32_t& Cb_,
                                uint32_t& mb_, uint32_t& k0_, uint32_t& k1_, uint32_t& CS_, uint32_t& tot_)
                                __attribute__((always_inline)) -> bool {
                    const uint32_t* hp = hist + (ok_ ? S_ : 0u) + sub * Q;
                    uint32_t pf[Q];
                    pf[0] = hp[0];
#pragma unroll
                    for (int i = 1; i < Q; ++i) pf[i] = pf[i - 1] + hp[i];
                    const uint32_t T = pf[Q - 1];
                    uint32_t incl = T;
                    incl += dpp3<kShr1>(incl) & mk1;
                    if constexpr (SUBS >= 4) incl += dpp3<kShr2>(incl) & mk2;
                    if constexpr (SUBS == 8) incl += dpp3<kShr4>(incl) & mk4;
                    const uint32_t excl = incl - T;
                    tot_ = cmax<SUBS>(incl);                                  // keys in the 16 buckets
                    // keys in buckets 0..11 of the block: the inclusive prefix at bucket 11 (the last bucket of a lane
                    // with 4 or 2 buckets per lane, the fourth of the second lane's eight at 2 lanes per cell)
                    uint32_t p12;
                    if constexpr (Q <= 4) p12 = cmax<SUBS>(incl & mk12);
                    else p12 = cmax<SUBS>(sub == 11 / Q ? excl + pf[11 % Q] : 0u);
                    CS_ = csA - (sub_p12 ? p12 : 0u) - (sub_tot ? tot_ : 0u);    // keys below the block
                    ok_ = ok_ && CS_ <= lo && lo + (need2 ? 1u : 0u) < CS_ + tot_;
                    const uint32_t t0 = lo - CS_, t1 = t0 + (need2 ? 1u : 0u);
                    uint32_t kk = 0, PB = 0, PU = 0xFFFFFFFFu;
#pragma unroll
                    for (int i = 0; i < Q; ++i) {
                        const uint32_t P = excl + pf[i];
                        const bool le0 = P <= t0;
                        kk += le0 ? 1u : 0u;
                        kk += (P <= t1) ? 0x10000u : 0u;
                        PB = le0 ? P : PB;
                    }
#pragma unroll
                    for (int i = Q - 1; i >= 0; --i) {
                        const uint32_t P = excl + pf[i];
                        PU = (P > t1) ? P : PU;
                    }
                    kk = csum<SUBS>(kk);
                    PB = cmax<SUBS>(PB);
                    PU = cmin<SUBS>(PU);
                    k0_ = kk & 0xFFFFu;
                    k1_ = kk >> 16;
                    Cb_ = CS_ + PB;                // keys below bucket B0
                    mb_ = PU - PB;                 // keys in buckets B0..B1
                    return ok_;
                };
                const bool up = Fc <= lo;                      // target at or above the anchor's lower edge
                bool bok = btry && (up || A >= 12u);
                uint32_t S = bok ? (up ? A : A - 12u) : 0u;              // first bucket of the block
                bok = bok && S >= 1u && S + 16u <= static_cast<uint32_t>(NB - 1);
                const bool blk = bok;                          // the block lies inside the window
                uint32_t Cb, mb, k0, k1, CS, tot;
                bok = look(S, bok, Fc, !up, false, Cb, mb, k0, k1, CS, tot);
                // (round 4) a target off its block but inside the window: ONE more look, at the next block in its direction
                // (with `up` it can only be above, otherwise only below), instead of the slow path and a new window
                // (not on 2 lanes per cell, where a look is 8 buckets per lane: 85.0 -> 85.4 ms on the 6-hourly share)
                const bool off = SUBS >= 4 && blk && !bok;
                if (__any(off)) {
                    const uint32_t S2 = up ? S + 16u : S - 16u;
                    bool ok2 = off && (up || S >= 17u) && S2 + 16u <= static_cast<uint32_t>(NB - 1);
                    uint32_t Cb2, mb2, k02, k12, CS2, tot2;
                    ok2 = look(S2, ok2, up ? CS + tot : CS, false, !up, Cb2, mb2, k02, k12, CS2, tot2);
                    Cb = ok2 ? Cb2 : Cb;
                    mb = ok2 ? mb2 : mb;
                    k0 = ok2 ? k02 : k0;
                    k1 = ok2 ? k12 : k1;
                    S = ok2 ? S2 : S;
                    bok = bok || ok2;
                    if constexpr (STATS) st_second += ok2 ? 1u : 0u;
                }
                const uint32_t B0 = S + k0;
                lost = btry && !bok;                       // off the block or at an end of the window
                const bool capf = bok && mb > static_cast<uint32_t>(LW);
                bok = bok && mb <= static_cast<uint32_t>(LW);
                const uint32_t E0 = hbase + (B0 << hshift);
                const uint32_t width = bok ? ((k1 - k0 + 1u) << hshift) : 0u;
                tick(2);
                // ---- 2. compaction: the keys of the band, appended to this lane's list ----------
#pragma unroll
                for (int i = 0; i < (X64 ? 2 * CAP : CAP); i += 4)
                    *reinterpret_cast<uint4*>(list + i) = make_uint4(kInv3, kInv3, kInv3, kInv3);
                uint32_t ptr = list_addr;
                uint32_t cntl;
                uint32_t c[CAP];
                uint32_t bh[X64 ? CAP : 1], bl[X64 ? CAP : 1];     // 64-bit mode: the lane's band entries as read back
                if constexpr (X64) {
#pragma unroll
                    for (int y = 0; y < YPS; ++y) compact11x(ring[y], ringlo[y], E0, width, ptr);
                    cntl = (ptr - list_addr) >> 3;
                    tick(3);
#pragma unroll
                    for (int i = 0; i < CAP; i += 2) {
                        const uint4 v = *reinterpret_cast<const uint4*>(list + 2 * i);
                        bh[i] = v.x; bl[i] = v.y; bh[i + 1] = v.z; bl[i + 1] = v.w;
                    }
#pragma unroll
                    for (int i = 0; i < CAP; ++i) c[i] = bh[i];
                } else {
#pragma unroll
                    for (int y = 0; y < YPS; ++y) compact11(ring[y], E0, width, ptr);
                    cntl = (ptr - list_addr) >> 2;
                    tick(3);
#pragma unroll
                    for (int i = 0; i < CAP; i += 4) {
                        const uint4 v = *reinterpret_cast<const uint4*>(list + i);
                        c[i] = v.x; c[i + 1] = v.y; c[i + 2] = v.z; c[i + 3] = v.w;
                    }
                }
                // (mm == mb always: the histogram mirrors the ring; the check costs one reduction and turns a
                // bookkeeping error into a slow row instead of a wrong one)
                const uint32_t mm = csum<SUBS>(cntl);
                const uint32_t lmax = cmax<SUBS>(cntl);
                const bool mmf = bok && mm != mb;
                bok = bok && mm == mb && lmax <= static_cast<uint32_t>(CAP);
                // ---- 3. sort the cell's slots, pick entries j and j + 1 ------------------
                // (most rows no lane holds more than four band keys: the slots 4..7 are then all empty and the
                // network over four slots per lane does)
                const uint32_t j = lo - Cb, j1 = j + 1u;
                uint32_t xa, xb;
                if (CAP == 4 || __all(lmax <= 4u)) {
                    uint32_t c4[4] = {c[0], c[1], c[2], c[3]};
                    sort_cell<SUBS, 4>(c4, bnd1, bnd2, bnd4);
                    const uint32_t va = pick_reg<4>(c4, j & 3u), vb = pick_reg<4>(c4, j1 & 3u);
                    xa = cmax<SUBS>((static_cast<uint32_t>(sub) == (j >> 2)) ? va : 0u);
                    xb = cmax<SUBS>((static_cast<uint32_t>(sub) == (j1 >> 2)) ? vb : 0u);
                } else {
                    sort_cell<SUBS, CAP>(c, bnd1, bnd2, bnd4);
                    const uint32_t va = pick_reg<CAP>(c, j & (CAP - 1)), vb = pick_reg<CAP>(c, j1 & (CAP - 1));
                    xa = cmax<SUBS>((static_cast<uint32_t>(sub) == j / CAP) ? va : 0u);
                    xb = cmax<SUBS>((static_cast<uint32_t>(sub) == j1 / CAP) ? vb : 0u);
                }
                if constexpr (X64) {
                    // ---- the low words of the two picks, from the lists: every key that shares a high word with a
                    // pick is in the band (same bucket), so its multiplicity and the r-th smallest low word of the
                    // group are settled on the <= CAP entries per lane instead of a pass over the rings
                    const uint32_t xb2 = need2 ? xb : xa;
                    uint32_t ca = 0, cb = 0, ta = 0, tb = 0, fa = 0;
#pragma unroll
                    for (int i = 0; i < CAP; ++i) {
                        const bool ea = bh[i] == xa, eb = bh[i] == xb2;
                        ca += ea ? 1u : 0u;
                        cb += eb ? 1u : 0u;
                        ta = ea ? bl[i] : ta;
                        tb = eb ? bl[i] : tb;
                        fa += bh[i] < xa ? 1u : 0u;
                    }
                    ca = csum<SUBS>(ca);
                    cb = csum<SUBS>(cb);
                    ta = cmax<SUBS>(ta);            // (exact for a group of one: the other lanes offer 0)
                    tb = cmax<SUBS>(tb);
                    const bool tie_a = bok && ca > 1u, tie_b = bok && need2 && cb > 1u;
                    if (__any(tie_a || tie_b)) {
                        const uint32_t below_a = csum<SUBS>(fa);                         // band entries below group A
                        const uint32_t ra = j - below_a;                                   // rank inside group A
                        const uint32_t rb = xb2 == xa ? ra + 1u : j1 - (below_a + ca);     // rank inside group B
                        auto nth_low = [&](uint32_t H, uint32_t r, bool want) -> uint32_t {
                            uint32_t prev = 0, rem = r, ans = 0;
                            bool have_prev = false, open = want;
                            while (__any(open)) {
                                uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
                                for (int i = 0; i < CAP; ++i) {
                                    const bool in = bh[i] == H && (!have_prev || bl[i] > prev);
                                    mn = in ? minu3(mn, bl[i]) : mn;
                                }
                                mn = cmin<SUBS>(mn);
                                uint32_t cnt = 0;
#pragma unroll
                                for (int i = 0; i < CAP; ++i) cnt += (bh[i] == H && bl[i] == mn) ? 1u : 0u;
                                cnt = csum<SUBS>(cnt);
                                if (open) {
                                    if (rem < cnt || cnt == 0u) {
                                        ans = mn;
                                        open = false;
                                    } else {
                                        rem -= cnt;
                                        prev = mn;
                                        have_prev = true;
                                    }
                                }
                            }
                            return ans;
                        };
                        const uint32_t ya = nth_low(xa, ra, tie_a);
                        const uint32_t yb = nth_low(xb2, rb, tie_b);
                        ta = tie_a ? ya : ta;
                        tb = tie_b ? yb : tb;
                    }
                    if (bok) {
                        la = ta;
                        lb = tb;
                        low_done = true;
                    }
                }
                if constexpr (STATS) {
                    st_try += btry ? 1u : 0u;
                    st_fail += (btry && !bok) ? 1u : 0u;
                    st_lost += lost ? 1u : 0u;
                    st_cap += capf ? 1u : 0u;
                    st_mm += mmf ? 1u : 0u;
                }
                if (bok) {
                    alo = xa;
                    ahi = need2 ? xb : xa;
                    pe = E0 - 1u;
                    Fe = Cb;
                    A = B0;
                    resolved = true;
                    band_done = true;
                    m16 += (static_cast<int32_t>(mb << 4) - m16) >> 2;
                } else if (capf) {
                    m16 += (static_cast<int32_t>(minu3(mb, 64u) << 4) - m16) >> 2;      // overfull buckets: narrow them
                }
                // (a cell the band path could not settle for lack of list space keeps its anchor: the pivot and
                // its count stay exact, the slow path settles this row; a cell whose target left the block is
                // given a new window below)
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
                    const bool use_c = have_c != 0 && wallc;
                    uint32_t p0 = pc, F0 = 0;
                    if (use_c) F0 = Fc;
                    if (!__all(use_c || n == 0)) {
                        uint32_t pm = key_of_bits3(__float_as_uint(static_cast<float>(total / static_cast<double>(nn))), 0u);
                        if (!use_c) p0 = have_c != 0 ? pc : pm;
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
                uint32_t s_alo = 0, s_ahi = 0, s_pe = 0, s_Fe = 0;
                bool sres = resolved;         // settled (by the band path, or n == 0)
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
                        const uint32_t j = window ? lo - Fl : 0u;
                        uint32_t d_lo, d_nx;
                        top.at2(j, d_lo, d_nx);
                        const uint32_t d_hi = need2 ? d_nx : d_lo;
                        const bool exact = d_hi <= horizon || j + (need2 ? 1u : 0u) < static_cast<uint32_t>(J);
                        if (window && !exact) slack = J - 2;
                        if (window && exact) {
                            s_alo = base + d_lo;
                            s_ahi = base + d_hi;
                            s_pe = pl; s_Fe = Fl;
                            top_span = top.m[J - 1] - top.m[0];
                            sres = true;
                        } else if (adjacent && !window) {
                            s_alo = ph;
                            s_ahi = (need2 && lo + 1u >= Fh) ? base + top.m[0] : ph;
                            s_pe = ph; s_Fe = Fh;
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
                    alo = s_alo; ahi = s_ahi; pe = s_pe; Fe = s_Fe;
                    resolved = true;
                    if (n > 0) {
                        if (rank_gap > 1 || rank_gap < -1) {
                            const float obs = (static_cast<float>(alo) - static_cast<float>(p_first)) *
                                              __builtin_amdgcn_rcpf(static_cast<float>(rank_gap));
                            if (obs >= 1.0f && obs < 1.0e8f) kpr = 0.75f * kpr + 0.25f * obs;
                        }
                        if (top_span != 0) {
                            // local spacing of the keys just above the pivot: what sizes the buckets
                            const float obs = static_cast<float>(top_span) * (1.0f / static_cast<float>(J - 1));
                            if (obs >= 1.0f && obs < 1.0e8f) {
                                kpr = kpr_seen ? 0.5f * kpr + 0.5f * obs : obs;
                                kpr_seen = true;
                            }
                        }
                    }
                }
            }

            // ---- 64-bit mode: the low words of the two order statistics (alo, ahi are HIGH words) ----------
            if constexpr (X64) {
            if (!__all(low_done || n == 0)) {
                // (only rows on which some cell went through the slow path: the band path reads the low words from
                // its lists)  One pass: how many pooled keys carry each high word, and the low word of one of them
                const uint32_t la_band = la, lb_band = lb;
                la = 0;
                lb = 0;
                uint32_t ca = 0, cb = 0;
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    const bool cnt = wallc || ((cmask >> y) & 1u);
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const uint32_t key = opaque3(ring[y][k]), l = opaque3(ringlo[y][k]);
                        const bool ea = cnt && key == alo, eb = cnt && key == ahi;
                        ca += ea ? 1u : 0u;
                        cb += eb ? 1u : 0u;
                        la = ea ? l : la;
                        lb = eb ? l : lb;
                    }
                }
                ca = csum<SUBS>(ca);
                cb = csum<SUBS>(cb);
                la = cmax<SUBS>(la);            // (exact when the cell holds ONE such key: the other lanes offer 0)
                lb = cmax<SUBS>(lb);
                // ties of high words (repeated values, or distinct doubles within 2^-20 of each other at the target
                // rank): the r-th smallest low word of the group, by successive minima with their multiplicities
                const bool tie_a = n > 0 && ca > 1u, tie_b = n > 0 && need2 && cb > 1u;
                if (__any(tie_a || tie_b)) {
                    uint32_t below_a = lo;     // keys below group A: lo itself when A is a single key, otherwise counted
                    if (__any(tie_a)) {
                        const uint32_t c = count_le(tie_a ? alo - 1u : 0u);
                        below_a = tie_a ? c : lo;
                    }
                    const uint32_t ra = lo - below_a;                                    // rank inside group A
                    const uint32_t rb = ahi == alo ? ra + 1u : lo + 1u - (below_a + ca);   // rank inside group B
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
                                    const uint32_t key = opaque3(ring[y][k]), l = opaque3(ringlo[y][k]);
                                    const bool in = cnt && key == H && (!have_prev || l > prev);
                                    mn = in ? minu3(mn, l) : mn;
                                }
                            }
                            mn = cmin<SUBS>(mn);
                            uint32_t c = 0;
#pragma unroll
                            for (int y = 0; y < YPS; ++y) {
                                const bool cnt = wallc || ((cmask >> y) & 1u);
#pragma unroll
                                for (int k = 0; k < R; ++k)
                                    c += (cnt && opaque3(ring[y][k]) == H && opaque3(ringlo[y][k]) == mn) ? 1u : 0u;
                            }
                            c = csum<SUBS>(c);
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
                la = low_done ? la_band : la;
                lb = low_done ? lb_band : lb;
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
                if constexpr (X64) {
                    e_la = la;
                    e_lb = lb;
                }
                e_n = n;
                e_total = total;
                e_g = g;
            }
            if (eph == static_cast<uint32_t>(SUBS - 1) || s + 1 == ch.end) {
                double th = make_nan(), se = make_nan();
                if (e_n > 0) {
                    double v_lo, v_hi;
                    if constexpr (X64) {
                        v_lo = double_of_key64_3(e_alo, e_la);
                        v_hi = double_of_key64_3(e_ahi, e_lb);
                    } else {
                        v_lo = static_cast<double>(__uint_as_float(bits_of_key3(e_alo)));
                        v_hi = static_cast<double>(__uint_as_float(bits_of_key3(e_ahi)));
                    }
                    th = numpy_lerp(v_lo, v_hi, e_g);
                    se = e_total / static_cast<double>(e_n);
                }
                if (static_cast<uint32_t>(sub) <= eph && cell_ok) {
                    const int64_t row = static_cast<int64_t>(s) - static_cast<int64_t>(eph) + sub;
                    thresh[row * ldo + cell] = th;
                    seas[row * ldo + cell] = se;
                }
            }
            if (wallc) {
                // (rows that do not pool every track leave the carried pivot and the window alone: both count
                // ALL keys of the ring and stay exact)
                if (n > 0) {
                    if (band_done || hvalid == 0 || lost) {
                        pc = pe;
                        Fc = Fe;
                        if (!band_done) hvalid = 0;    // the pivot is no longer a bucket edge: new window below
                    }
                    have_c = 1;
                } else {
                    have_c = 0;
                    pc = 0;
                    Fc = 0;
                    hvalid = 0;
                }
            }

            tick(6);
            // ================= window (re)build ==================================================
            const bool can = wallc && n > 0;
            const bool want = can && (hvalid == 0 || A < static_cast<uint32_t>(EDGE_LO) ||
                                      A > static_cast<uint32_t>(NB - EDGE_HI) || m16 > M16_HI_TRIG || m16 < M16_LO_TRIG);
            if (__any(want)) {
                if constexpr (STATS) {
                    ++st_rebuild;
                    st_rb_inv += (can && hvalid == 0) ? 1u : 0u;
                    st_rb_edge += (can && hvalid != 0 && (A < static_cast<uint32_t>(EDGE_LO) || A > static_cast<uint32_t>(NB - EDGE_HI))) ? 1u : 0u;
                    st_rb_m += (can && hvalid != 0 && (m16 > M16_HI_TRIG || m16 < M16_LO_TRIG)) ? 1u : 0u;
                }
                if (can) {
                    // bucket width: from the key spacing on a first build, one step at a time afterwards
                    uint32_t sh;
                    if (hbuilt == 0) {
                        const float bw = fminf(fmaxf(kpr * kBucketRanks * (static_cast<float>(PS) * 0.0625f), 1.0f), 8.0e6f);
                        sh = 31u - static_cast<uint32_t>(__builtin_clz(static_cast<uint32_t>(bw)));
                        m16 = M16_TARGET;
                    } else {
                        sh = hshift;
#pragma unroll
                        for (int it = 0; it < 3; ++it) {
                            const bool dn = m16 > M16_HI_ADJ && sh > 0u, upw = m16 < M16_LO_ADJ && sh < 23u;
                            sh = dn ? sh - 1u : upw ? sh + 1u : sh;
                            m16 = dn ? m16 >> 1 : upw ? m16 << 1 : m16;
                        }
                    }
                    sh = minu3(sh, 23u);
                    hshift = sh;
                    const uint32_t half = static_cast<uint32_t>(NB / 2) << sh;
                    uint32_t nb = alo > half ? alo - half : 0u;
                    nb = minu3(nb, 0u - (static_cast<uint32_t>(NB) << sh));     // window inside the key space
                    hbase = nb;
                    hbuilt = 1;
                }
                // every lane clears its share of the cell's histogram, then adds its keys
#pragma unroll 8
                for (int i = 0; i < NB / SUBS; ++i) hist[sub * (NB / SUBS) + i] = 0u;
#pragma unroll
                for (int y = 0; y < YPS; ++y)
#pragma unroll
                    for (int k = 0; k < R; ++k)
                        __hip_atomic_fetch_add(&hist[tag_of(ring[y][k])], 1u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                // anchor: the bucket of this row's answer; its lower edge becomes the carried pivot
                const uint32_t An = can ? tag_of(alo) : 0u;
                const uint32_t pn = hbase + (An << hshift) - 1u;
                const bool okA = can && An >= 1u && An + 16u < static_cast<uint32_t>(NB);
                // The count of keys at or below the new pivot.  A cell the band path settled on this row has it without a
                // pass over its keys (round 4): its window is centred on the answer, so the pivot is alo - 1; Fc counts
                // the keys below the band and the band's keys are still in the lanes' lists.
                const bool shortc = !X64 && band_done && okA && pn == alo - 1u;
                uint32_t Fn;
                if (__all(shortc || !okA)) {
                    uint32_t below = 0;
#pragma unroll
                    for (int i = 0; i < CAP; i += 4) {
                        const uint4 v = *reinterpret_cast<const uint4*>(list + i);
                        below += (v.x < alo ? 1u : 0u) + (v.y < alo ? 1u : 0u) + (v.z < alo ? 1u : 0u) + (v.w < alo ? 1u : 0u);
                    }
                    Fn = Fc + csum<SUBS>(below);
                } else {
                    Fn = count_le(okA ? pn : 0u);
                    if constexpr (STATS) ++st_count;
                }
                if (okA) {
                    A = An;
                    pc = pn;
                    Fc = Fn;
                    hvalid = 1;
                } else if (can) {
                    hvalid = 0;
                }
            }
        }

        tick(7);
        // The waves of a workgroup read the two halves of the same 128-byte lines (a workgroup is 32 cells wide), and
        // a line stays in L2 for about three rows: they are marched in step every 32 rows.  Measured on configs[2]
        // (profiles/r3_rendezvous.txt): no rendezvous at all is 3 % faster (55.3 against 56.9 ms) but fetches 1.65 x
        // the algorithmic bytes -- every line again for the wave that comes late; bounding the lead of a wave by 8..32
        // rows through a counter in LDS does not help (1.6 x: the partner has to arrive within about three rows).
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
    if (rotate) rs_stale = true;
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
            if constexpr (X64) {
                const uint32_t lastl = opaque3(ringlo[y][R - 1]);
                asm volatile("s_nop 1");
#pragma unroll
                for (int k = R - 1; k >= 1; --k) {
                    uint32_t e = ringlo[y][k];
                    ring_sel3(e, ringlo[y][k - 1], hy);
                    ringlo[y][k] = e;
                }
                uint32_t l0 = ringlo[y][0];
                ring_sel3(l0, lastl, hy);
                ringlo[y][0] = l0;
            }
        }
    }
    if constexpr (kRowSum) {
        if (rs_stale) {
            // the held tracks' keys have moved one slot up: the row sums (and, from them, the total) are taken from the
            // rings again
            double tc = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                double v = 0.0;
#pragma unroll
                for (int y = 0; y < YPS; ++y) v = opaque3d(v + val_at(y, k));
                v = csum<SUBS>(v);
                rowsum[k] = v;
                tc += v;
            }
            lsum = tc;
            rs_stale = false;
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
        atomicAdd(&stats[5], static_cast<unsigned long long>(st_band) | (static_cast<unsigned long long>(st_rebuild) << 32));
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&stats[8 + i], tacc[i]);
    }
    if (STATS && stats != nullptr && sub == 0 && cell_ok) {
        atomicAdd(&stats[6], static_cast<unsigned long long>(st_try) | (static_cast<unsigned long long>(st_fail) << 32));
        // (slot 7: low word: cells whose target left the block or the window, high word: bands with more keys than
        // a list holds; the rest of the failures are lanes with more than CAP band keys)
        atomicAdd(&stats[7], static_cast<unsigned long long>(st_lost) | (static_cast<unsigned long long>(st_cap) << 32));
        if (st_mm) atomicAdd(&stats[3], static_cast<unsigned long long>(st_mm) << 32);
        // (cells asking for a rebuild, by reason: no window / target near an end / band population out of range)
        atomicAdd(&stats[1], static_cast<unsigned long long>(st_rb_inv) << 32);
        atomicAdd(&stats[2], static_cast<unsigned long long>(st_rb_edge) << 32);
        atomicAdd(&stats[4], static_cast<unsigned long long>(st_rb_m) << 32);
    }
}

// ---------------------------------------------------------------------------
namespace {
typedef void (*Ring3Kernel)(const float*, int64_t, int64_t, int64_t, const uint32_t*, const uint32_t*, int32_t,
                            const DevChunk*, double, int, int32_t, double*, double*, int64_t, unsigned long long*,
                            uint32_t*);
typedef void (*Ring3KernelN)(const double*, int64_t, int64_t, int64_t, const uint32_t*, const uint32_t*, int32_t,
                             const DevChunk*, double, int, int32_t, double*, double*, int64_t, unsigned long long*,
                             uint32_t*);
struct Ring3Entry { int yps, subs; Ring3Kernel fn, fn_stats; Ring3KernelN fn_narrow, fn_x64; };
// (the counter twins -- STATS = true: pass counters and section ticks -- are built with -DXMHW_RING_STATS only: tools/,
// not the product)
#ifdef XMHW_RING_STATS
#define XMHW_R3S(Y, S) clim_ring3_f32<Y, S, true>
#else
#define XMHW_R3S(Y, S) nullptr
#endif
#define XMHW_R3(Y, S) {Y, S, clim_ring3_f32<Y, S, false>, XMHW_R3S(Y, S), nullptr, nullptr}
// (with the 64-bit mode for genuinely float64 input: 8 lanes per cell, where the two rings fit the register file)
#define XMHW_R3X(Y, S) {Y, S, clim_ring3_f32<Y, S, false>, XMHW_R3S(Y, S), nullptr, clim_ring3_f32<Y, S, false, double, true>}
// (with the narrowing instantiation for float64 input: the layouts the automatic choice uses)
#define XMHW_R3N(Y, S) {Y, S, clim_ring3_f32<Y, S, false>, XMHW_R3S(Y, S), clim_ring3_f32<Y, S, false, double>, nullptr}
// (narrowing and the 64-bit mode: short records on 4 lanes per cell, both rings of 4 / 5 tracks per lane fit)
#define XMHW_R3NX(Y, S) {Y, S, clim_ring3_f32<Y, S, false>, XMHW_R3S(Y, S), clim_ring3_f32<Y, S, false, double>, clim_ring3_f32<Y, S, false, double, true>}
const Ring3Entry kRing3[] = {
    XMHW_R3X(2, 8), XMHW_R3X(3, 8), XMHW_R3X(4, 8), XMHW_R3X(5, 8), XMHW_R3X(6, 8),
    // (long records -- reanalyses, model runs: 49..96 tracks on 8 lanes per cell.  Narrowing from 9 tracks per lane up
    // spills registers -- its samples are twice as wide -- and is kept only where a named archive needs it: 11 tracks
    // per lane = ERA5 1940-2024; the others narrow on the second-generation kernel or take the 64-bit mode)
    XMHW_R3N(7, 8), XMHW_R3N(8, 8), XMHW_R3(9, 8), XMHW_R3(10, 8), XMHW_R3N(11, 8), XMHW_R3(12, 8),
    // (2 lanes per cell, 32 cells per wave: records of 9..24 tracks; narrowing up to 8 tracks per lane and at 10 = the
    // 6-hourly 20-year records of BASELINE configs[4])
    XMHW_R3N(5, 2), XMHW_R3N(6, 2), XMHW_R3N(7, 2), XMHW_R3N(8, 2), XMHW_R3(9, 2), XMHW_R3N(10, 2), XMHW_R3(11, 2),
    XMHW_R3(12, 2),
    // (4 lanes per cell: 13..48 tracks; narrowing up to 8 tracks per lane, at 10 = 40-year daily records, BASELINE
    // configs[2] / [3], and at 11 = OISST 1982-2024)
    XMHW_R3(3, 4), XMHW_R3NX(4, 4), XMHW_R3NX(5, 4), XMHW_R3N(6, 4), XMHW_R3N(7, 4), XMHW_R3N(8, 4), XMHW_R3(9, 4),
    XMHW_R3N(10, 4), XMHW_R3N(11, 4), XMHW_R3(12, 4),
};
#undef XMHW_R3
#undef XMHW_R3S
#undef XMHW_R3N
#undef XMHW_R3X
#undef XMHW_R3NX
const Ring3Entry* find_ring3(int32_t yps, int32_t subs) {
    for (const auto& e : kRing3)
        if (e.yps == yps && e.subs == subs) return &e;
    return nullptr;
}
}  // namespace

int32_t ring3_pick_yps(int32_t w, int32_t ntracks, int32_t subs) {
    if (w != 5) return 0;
    int32_t best = 0;
    for (const auto& e : kRing3)
        if (e.subs == subs && e.yps * subs >= ntracks && (best == 0 || e.yps < best)) best = e.yps;
    if (best && (best - 1) * subs >= ntracks) return 0;      // padding may only sit in the last slot of a lane
    return best;
}

bool ring_stats_built() {
#ifdef XMHW_RING_STATS
    return true;
#else
    return false;
#endif
}

bool ring3_supported(int32_t w, int32_t yps, int32_t subs) { return w == 5 && find_ring3(yps, subs) != nullptr; }

hipError_t launch_ring3_f32(const float* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                            const uint32_t* sflags, int32_t step_min, const DevChunk* chunks, int32_t nchunks,
                            int32_t w, int32_t yps, int32_t subs, int32_t ntracks, double q, int negate,
                            double* thresh, double* seas, int64_t ldo, hipStream_t stream,
                            unsigned long long* stats) {
    const Ring3Entry* e = w == 5 ? find_ring3(yps, subs) : nullptr;
    if (!e || ld >= (int64_t(1) << 30)) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    const int kWaves3 = waves3(subs, 4);
    const int64_t cells_per_block = (64 / subs) * kWaves3;
    dim3 grid(static_cast<unsigned>((C + cells_per_block - 1) / cells_per_block), static_cast<unsigned>(nchunks));
    const bool twin = stats != nullptr && e->fn_stats != nullptr;
    hipLaunchKernelGGL(twin ? e->fn_stats : e->fn, grid, dim3(64 * kWaves3), 0, stream, ts, C, ld, Tn, table, sflags,
                       step_min, chunks, q, negate, ntracks, thresh, seas, ldo, twin ? stats : nullptr,
                       static_cast<uint32_t*>(nullptr));
    return hipGetLastError();
}

bool ring3_x64_supported(int32_t w, int32_t yps, int32_t subs) {
    const Ring3Entry* e = w == 5 ? find_ring3(yps, subs) : nullptr;
    return e != nullptr && e->fn_x64 != nullptr;
}

// genuinely float64 samples (64-bit keys as high / low words); run_flag: device flag of the narrowing launch queued
// before it (nullptr: always run; 0 at run time: nothing to do)
hipError_t launch_ring3_f64(const double* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                            const uint32_t* sflags, int32_t step_min, const DevChunk* chunks, int32_t nchunks,
                            int32_t w, int32_t yps, int32_t subs, int32_t ntracks, double q, int negate, double* thresh,
                            double* seas, int64_t ldo, hipStream_t stream, const uint32_t* run_flag) {
    const Ring3Entry* e = w == 5 ? find_ring3(yps, subs) : nullptr;
    if (!e || !e->fn_x64 || ld >= (int64_t(1) << 29)) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    const int kWaves3 = waves3(subs, 8);
    const int64_t cells_per_block = (64 / subs) * kWaves3;
    dim3 grid(static_cast<unsigned>((C + cells_per_block - 1) / cells_per_block), static_cast<unsigned>(nchunks));
    hipLaunchKernelGGL(e->fn_x64, grid, dim3(64 * kWaves3), 0, stream, ts, C, ld, Tn, table, sflags, step_min, chunks, q,
                       negate, ntracks, thresh, seas, ldo, static_cast<unsigned long long*>(nullptr),
                       const_cast<uint32_t*>(run_flag));
    return hipGetLastError();
}

bool ring3_narrowing_supported(int32_t w, int32_t yps, int32_t subs) {
    const Ring3Entry* e = w == 5 ? find_ring3(yps, subs) : nullptr;
    return e != nullptr && e->fn_narrow != nullptr;
}

// float64 input on the float32 kernel: the flag must have been cleared and the sparse probe queued by the caller
// (launch_narrow_probe, kernels_ring.hip); the kernel leaves as soon as the flag is set
hipError_t launch_ring3_f32_narrowing(const double* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                                      const uint32_t* sflags, int32_t step_min, const DevChunk* chunks,
                                      int32_t nchunks, int32_t w, int32_t yps, int32_t subs, int32_t ntracks, double q,
                                      int negate, double* thresh, double* seas, int64_t ldo, hipStream_t stream,
                                      uint32_t* narrow_flag) {
    const Ring3Entry* e = w == 5 ? find_ring3(yps, subs) : nullptr;
    if (!e || !e->fn_narrow || !narrow_flag || ld >= (int64_t(1) << 29)) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    const int kWaves3 = waves3(subs, 4);
    const int64_t cells_per_block = (64 / subs) * kWaves3;
    dim3 grid(static_cast<unsigned>((C + cells_per_block - 1) / cells_per_block), static_cast<unsigned>(nchunks));
    hipLaunchKernelGGL(e->fn_narrow, grid, dim3(64 * kWaves3), 0, stream, ts, C, ld, Tn, table, sflags, step_min, chunks,
                       q, negate, ntracks, thresh, seas, ldo, static_cast<unsigned long long*>(nullptr), narrow_flag);
    return hipGetLastError();
}

}  // namespace xmhw
