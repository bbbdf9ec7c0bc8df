// kernels_sorted.hip -- the climatology kernel of rounds 5 and 6: selection on SORTED ROW-LISTS in LDS.
//
// What the ring kernels (kernels_ring*.hip) keep in registers -- the R = 2w+1 last samples of every track -- is not
// kept at all here.  The pool of a row is the union of the R last ROW-LISTS (the samples all tracks push at one step),
// and a row-list is evicted WHOLE when its slot is overwritten, so nothing is ever needed per key after its push:
//
//   * per step the 2 lanes of a cell sort the cell's new samples (a network in registers per lane, one bitonic
//     exchange between the lanes) and write the K largest keys, descending, into the slot of the evicted list.  LDS is
//     RANK-MAJOR (round 6): the key of rank r of the list in slot g sits in row r * R + g, a row = the 32 cells of the
//     wave (cell-minor: per-lane dynamic positions never meet on a bank).  A window that reaches past the last rank of
//     its list -- or above rank 0: the address wraps around -- leaves the workgroup's LDS allocation, and an LDS read
//     outside the allocation returns 0 on gfx950 (tools/ubench_ldsoob.hip; the library checks it once per device
//     before it uses this kernel): 0 is the "no key" value, so no list needs sentinels, clamps or masks.  The records of
//     37..40 tracks keep TWO TIERS: 14 ranks in LDS (20,480 bytes per wave: 8 waves per CU) and ranks 14, 15 in registers
//     of the lane that owns the list's slot; a short correction after the select lets them join the top set (4c);
//   * per list a POINTER P_i = the number of its keys inside the TOP SET (the Cs largest keys of the pool,
//     Cs = n - 1 - lo, lo = floor((n - 1) q)): order statistic lo of numpy's linear quantile is the largest key outside
//     the top set (max over the lists of key[P_i]), lo + 1 the smallest inside (min of key[P_i - 1]);
//   * a row changes the top set by the evicted list's share and the new list's (counted against the carried boundary
//     value); the pointers then move |c_new - c_evicted| keys (4 on average) in ROUNDS of a parallel merge-select: every
//     lane reads the 4 next keys of each of its lists (+ a fifth that says how far the windows can be trusted), a tree
//     of merges of sorted runs and one exchange between the lanes give the cell's 16 largest candidates, the d-th of
//     them is the threshold every list counts its window against.  What one round cannot settle (an unsafe window,
//     more than 15 keys) takes another round, or -- at most three keys left in the wave's worst cell -- is finished key
//     by key from the 22 list heads;
//   * `seas` = (sum of the 11 lists' float64 sums, added in slot order every row) / n: the same bits for every cut of
//     the row axis;
//   * quantiles <= 0.15 run MIRRORED: keys of the negated samples, the top set = the lo + 1 smallest of the pool.
//
// Everything is exact.  What can fail is the capacity of a list: a list whose K stored keys are all inside the top set
// while it holds more valid keys than K (a steep seasonal slope puts up to ~20 of a list's 40 keys among the 44
// largest of the pool), with its last stored key above the key outside the top set.  Such a cell-row is recomputed on the
// spot, by the whole wave, from the cell's samples (pool_order_stats below: exact for any pool); the list state stays
// consistent (the hidden keys are all below the stored ones) and the cell carries on by itself once the boundary has
// moved back.
//
// The kernel runs on its OWN chunks and step-table rows (plan.cpp: sorted_plan): the row axis is cut wherever the set
// of pooled tracks changes (a held step -- doy 60 --, the ends of partial years); inside a chunk every pooled track
// pushes at every row, warms up with its R-1 last pushes before the chunk, and the other tracks push nothing.  Warm-up
// rows only build lists; the chunk's first output row grows the top set from nothing.
//
// Two things about the compiled loop that cost or bought several per cent each (profiles/r6_experiments.txt): the branches
// most rows do not take are marked XMHW_COLD so that their blocks sit behind the loop, and NO `s_waitcnt vmcnt` may appear
// in the sort / bookkeeping / select blocks -- the requests for the next row's samples are in flight there
// (tools/check_sorted_waits.py, tests/test_kernel_isa.py).
//
// Lane layout: lane = 2 * cell_in_wave + sub; a wave is 32 cells = one 128-byte line of a float32 sample row; a
// workgroup is one wave.  Track k of the plan sits in lane sub = k % 2, slot k / 2 (the y-major table of the ring
// kernels, subs = 2).
//
// Reference semantics restated: window_roll() (identify.py:184-209), calculate_thresh() / calculate_seas() without
// the Feb-29 step (identify.py:233-235, :263), coldSpells negation (xmhw.py:153-154).
#include <type_traits>

#include "device_common.h"
#include "kernels.h"
#include "packed_src.h"
#include "plan.h"
#include "sortnet_gen.h"

namespace xmhw {
namespace {

typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_ld(uint32_t a) { return *reinterpret_cast<lds_u32*>(static_cast<uintptr_t>(a)); }
__device__ __forceinline__ void lds_st(uint32_t a, uint32_t v) { *reinterpret_cast<lds_u32*>(static_cast<uintptr_t>(a)) = v; }

constexpr int kSwap1 = 0xB1;      // quad_perm [1, 0, 3, 2]: the other lane of the cell
// EVERY exchange must run with both lanes of the cell active (never under a condition that can differ between them).
__device__ __forceinline__ uint32_t swp(uint32_t v) {
    // (bound_ctrl set: every lane of a quad has its partner, and the DPP-combine pass then folds the exchange into the
    // instruction that uses it)
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), kSwap1, 0xF, 0xF, true));
}
__device__ __forceinline__ double swp(double v) {
    const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
    const uint32_t lo = swp(static_cast<uint32_t>(b)), hi = swp(static_cast<uint32_t>(b >> 32));
    return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
}
__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// some lane's condition holds: the ballot compared as a scalar (HIP's __any() goes through a select and a vector compare)
__device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
// a * b + c, a and b unsigned 24-bit: one full-rate instruction (left to the compiler `base + P * stride` is a 64-bit v_mad_u64_u32)
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
// a * b + c, a and b signed 24-bit (the compiler turns a multiplication by +-1 into a negation and a select)
__device__ __forceinline__ uint32_t mad_i24(uint32_t a, int32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// key of a non-NaN float in two instructions (v_ashrrev, v_bitop3): (sign | 0x80000000) ^ bits; NEG: the key of the
// NEGATED sample, (~sign & 0x7FFFFFFF) ^ bits (the sign mask s of the sample itself, C = 0x80000000: ~s & ~C ^ bits)
template <bool NEG>
__device__ __forceinline__ uint32_t key_fast(uint32_t b) {
    const int32_t sg = static_cast<int32_t>(b) >> 31;
    // truth tables over (A = sign mask, B = bits, C = 0x80000000), bit index = A * 4 + B * 2 + C:
    //   (A | C) ^ B = 0x36      (~A & ~C) ^ B = 0xC9
    if constexpr (NEG)
        return static_cast<uint32_t>(__builtin_amdgcn_bitop3_b32(sg, static_cast<int32_t>(b), static_cast<int32_t>(0x80000000u), 0xC9));
    else
        return static_cast<uint32_t>(__builtin_amdgcn_bitop3_b32(sg, static_cast<int32_t>(b), static_cast<int32_t>(0x80000000u), 0x36));
}

// #{a0..a3 >= th}: four compares into four SGPR pairs, then four add-with-carry -- no wait states between a compare and
// the instruction that reads its mask (left to the compiler every compare goes through VCC with an s_nop behind it)
__device__ __forceinline__ uint32_t count_ge4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t th) {
    uint32_t c;
    unsigned long long s0, s1, s2, s3, sd;
    asm("v_cmp_ge_u32_e64 %[s0], %[a0], %[t]\n\t"
        "v_cmp_ge_u32_e64 %[s1], %[a1], %[t]\n\t"
        "v_cmp_ge_u32_e64 %[s2], %[a2], %[t]\n\t"
        "v_cmp_ge_u32_e64 %[s3], %[a3], %[t]\n\t"
        "v_addc_co_u32_e64 %[c], %[sd], 0, 0, %[s0]\n\t"
        "v_addc_co_u32_e64 %[c], %[sd], %[c], 0, %[s1]\n\t"
        "v_addc_co_u32_e64 %[c], %[sd], %[c], 0, %[s2]\n\t"
        "v_addc_co_u32_e64 %[c], %[sd], %[c], 0, %[s3]"
        : [c] "=&v"(c), [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3), [sd] "=&s"(sd)
        : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [t] "v"(th));
    return c;
}

typedef uint32_t V32 __attribute__((ext_vector_type(32)));
// (branches most rows do not take: their blocks go behind the loop's hot path)
#ifdef XMHW_NO_COLD
#define XMHW_COLD(x) (x)
#else
#define XMHW_COLD(x) __builtin_expect(!!(x), 0)
#endif
#define XMHW_STR2(x) #x
#define XMHW_STR(x) XMHW_STR2(x)
#ifndef XMHW_LOOP_PAD
#define XMHW_LOOP_PAD 0
#endif

// N unsorted keys -> descending.  Up to 7 keys by insertion with three-input instructions (sortnet::Ins: a 3-sorter is
// v_max3 / v_med3 / v_min3, an insertion into a sorted run of n is n + 1 independent instructions); 8, 10 and 12 keys as two
// such halves + the merge network of two sorted runs (10 keys: 12 + 12 + 26 = 50 instructions against the 29-comparator
// network's 58); 9 and 11 keys on the comparator networks.
template <int N>
__device__ __forceinline__ void sort_desc(uint32_t (&v)[N]) {
    if constexpr (N >= 3 && N <= 7) {
        sortnet::Ins<N>::run(v);
    } else if constexpr (N == 8 || N == 10 || N == 12) {
        constexpr int H = N / 2;
        uint32_t a[H], b[H];
#pragma unroll
        for (int i = 0; i < H; ++i) { a[i] = v[i]; b[i] = v[H + i]; }
        sortnet::Ins<H>::run(a);
        sortnet::Ins<H>::run(b);
#pragma unroll
        for (int i = 0; i < H; ++i) { v[i] = a[i]; v[H + i] = b[i]; }
        sortnet::MergeTop<H, H, N>::run(v);
    } else {
        sortnet::Desc<N>::run(v);
    }
}
// LDS is handed out in pieces of 1,280 bytes on gfx950 (160 KB / 128; tools/ubench_ldsoob.hip: a read at the first byte
// behind an allocation rounded up to that returns 0, the bytes between the declared size and that do not)
constexpr int kLdsGranule = 1280;


// ---- a flagged cell-row, recomputed by the whole wave from the samples themselves (round 6) ----------------------------
// The select above can fail on one cell of the wave (a row-list too short for the row).  Instead of leaving the cell-row to
// a second kernel (rounds 5: a bitmap, a work list, 11.6 ns per flagged cell-row) the wave stops for it: the 64 lanes load
// the cell's pool -- the R last table rows of its chunk x the tracks: the samples the wave itself read over the last R
// rows, L2 / Infinity-Cache hits --, key them as the main path does and find order statistics lo and lo + 1 by stepping
// from the select's own answer (wrong, but a few ranks away), key to neighbouring key, with wave-wide counts (ballot +
// popcount: every count and every decision is scalar).  Exact for any pool; what it costs is two memory latencies and
// ~300 instructions per flagged cell-row, and only waves that have one pay it.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = umin(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(-1, static_cast<int>(v), 0xB1, 0xF, 0xF, false)));    // quad_perm [1,0,3,2]
    v = umin(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(-1, static_cast<int>(v), 0x4E, 0xF, 0xF, false)));    // quad_perm [2,3,0,1]
    v = umin(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(-1, static_cast<int>(v), 0x141, 0xF, 0xF, false)));   // row_half_mirror
    v = umin(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(-1, static_cast<int>(v), 0x140, 0xF, 0xF, false)));   // row_mirror
    const uint32_t a = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 0));
    const uint32_t b = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 16));
    const uint32_t c = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 32));
    const uint32_t d = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 48));
    return umin(umin(a, b), umin(c, d));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return ~wave_min_u32(~v); }

// keys: KPL per lane, 0 = no key.  n valid keys, lo = floor((n - 1) q) (ascending order statistic), guess = a key near it
// (0: none).  Returns the keys of order statistics lo and min(lo + 1, n - 1) -- wave-uniform.
template <int KPL>
__device__ __forceinline__ void pool_order_stats(const uint32_t (&key)[KPL], uint32_t n, uint32_t lo, uint32_t guess,
                                                 uint32_t& out_lo, uint32_t& out_hi) {
    auto count_lt = [&](uint32_t v) -> uint32_t {       // valid keys below v
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < KPL; ++i)
            cnt += static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(key[i] != 0u && key[i] < v)));
        return cnt;
    };
    auto count_eq = [&](uint32_t v) -> uint32_t {
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < KPL; ++i)
            cnt += static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(key[i] == v)));
        return cnt;
    };
    auto next_above = [&](uint32_t v) -> uint32_t {     // the smallest key above v (all ones if none)
        uint32_t m = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < KPL; ++i) m = (key[i] > v && key[i] < m) ? key[i] : m;
        return wave_min_u32(m);
    };
    auto next_below = [&](uint32_t v) -> uint32_t {     // the largest valid key below v (0 if none)
        uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < KPL; ++i) m = (key[i] < v && key[i] > m) ? key[i] : m;
        return wave_max_u32(m);
    };
    uint32_t v = guess;
    bool found = false;
    uint32_t cl = 0, ev = 0;                            // keys below v, keys equal to v
    if (v != 0u) {
        cl = count_lt(v);
        ev = count_eq(v);
        for (int it = 0; it < 32 && !found; ++it) {
            if (cl <= lo && lo < cl + ev) {
                found = true;
            } else if (lo < cl) {
                v = next_below(v);
                if (v == 0u) break;
                ev = count_eq(v);
                cl -= ev;
            } else {
                const uint32_t nv = next_above(v);
                if (nv == 0xFFFFFFFFu) break;
                cl += ev;
                v = nv;
                ev = count_eq(v);
            }
        }
    }
    if (!found) {
        // the whole key, bit by bit: the largest v with #{valid keys < v} <= lo   (key 0 = no key: (0 - 1) wraps high)
        v = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = v | (1u << bit);
            uint32_t cnt = 0;
#pragma unroll
            for (int i = 0; i < KPL; ++i)
                cnt += static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(key[i] - 1u < cand - 1u)));
            if (cnt <= lo) v = cand;
        }
        cl = count_lt(v);
        ev = count_eq(v);
    }
    // v = the key of a[lo]; a[lo + 1]: v again if it is duplicated past lo, else the smallest key above v
    uint32_t vhi = v;
    if (lo + 1 < n && lo + 1 >= cl + ev) vhi = next_above(v);
    out_lo = v;
    out_hi = vhi;
}

}  // namespace

// stats (STATS builds): [0] wave-rows, [1] walk iterations (what the wave pays), [2] flagged cell-rows, [3] walk steps
// summed over cells, [8..15] shader-clock ticks per section (push, sort, bookkeeping, walk, epilogue)
// ([13] the wait for the row's samples, [14] their conversion; [8] then is the requests of the next row alone; [15] the
// recomputation of flagged cell-rows inside the kernel)
// and [5..7] rounds after a row's first one by the keys still to move in the wave's worst cell (<= 4, <= 8, more), [4] wave-rows
// that ran the correction for the register ranks (4c)
// PACKED: the samples are int16 codes read in place (kernels.h: PackedI16, packed_src.h): a row's codes become the float32
// samples the rest of the row works on -- float(code) * sf + of in mode 1, float(code) in modes 2 and 3 -- and in mode 2 the
// epilogue decodes the two selected codes and the mean of the codes in float64.
// K = keys a cell keeps of every row-list, KL <= K of them in LDS; the K - KL (0 or 2) last ranks of a list stay in
// registers of the lane that owns its slot (EXT below)
template <int YPS, int K, int KL, bool STATS, bool PACKED>
__device__ __forceinline__ void sorted_body(
    const void* __restrict__ ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* __restrict__ table,
    const uint32_t* __restrict__ sflags, const DevSortedChunk* __restrict__ chunks, double q, int negate,
    int32_t ntracks, double* __restrict__ thresh, double* __restrict__ seas, int64_t ldo,
    unsigned long long* __restrict__ stats, const PackedI16& pk) {
    constexpr int R = 11;
    constexpr uint32_t ES = PACKED ? 2u : 4u;    // bytes per stored sample                        // w = 5
    constexpr int NL = 6;                        // list slots per lane (lane 1 owns 5 and a dummy)
    constexpr int HE = (YPS + 1) / 2 * 2;        // keys per lane, padded to an even count
    constexpr int HH = HE / 2;                   // sorted keys per lane after the exchange
    constexpr int KH = K / 2;                    // keys of the new list a lane ends with
#ifndef XMHW_SERIAL_MAX
#define XMHW_SERIAL_MAX 3
#endif
    constexpr uint32_t kSerialMax = XMHW_SERIAL_MAX;   // keys left (worst cell of the wave) up to which a row is finished key by key
    constexpr int NTP = 2 * YPS;
    // LDS, rank-major: row r * R + g = rank r of the list in slot g, a row = 32 cells.  The allocation is rounded up to the
    // 1,280-byte piece LDS is handed out in on gfx950 (K = 16: 176 rows of lists + 4 rows that stay 0 = 23,040 bytes, 7
    // waves per CU), so that the first byte behind it is the first byte the hardware answers with 0.
    constexpr uint32_t LSTRIDE = 32 * 4;         // bytes between the lists of one rank
    constexpr uint32_t RSTRIDE = R * LSTRIDE;    // bytes between consecutive ranks of a list
    constexpr int LDS_BYTES = (R * KL * static_cast<int>(LSTRIDE) + kLdsGranule - 1) / kLdsGranule * kLdsGranule;
    static_assert(K <= HE && K % 2 == 0, "a list stores an even number of keys, at most what a lane holds");
    // Two tiers (K = 16, KL = 14: the 37..40-track records).  LDS is what limits the waves: 11 lists x 16 ranks are 7 waves
    // per CU, 11 x 14 are 8 -- and the eighth wave is worth 9 % (an issue-bound kernel with two waves on every SIMD).  Ranks
    // 14 and 15 of every list are kept in two register tuples instead.  The select works on the LDS ranks alone; a list
    // whose 14 LDS keys are all inside the top set and whose 15th key lies above the boundary is then put right by a
    // short correction (4c below): its register keys join the top set and as many of the smallest keys leave it.
    constexpr int EXT = K - KL;
    static_assert(EXT == 0 || ((EXT == 2 || EXT == 4) && KL >= KH), "two or four register ranks, all from lane 1's half");
    constexpr int EXN = EXT > 0 ? EXT : 1;
    static_assert(LDS_BYTES == 8960 || LDS_BYTES == 11520 || LDS_BYTES == 14080 || LDS_BYTES == 17920 || LDS_BYTES == 20480 ||
                  LDS_BYTES == 23040 || LDS_BYTES == 25600, "sorted_lds_probe() checks these allocation sizes");
    __shared__ __attribute__((aligned(16))) uint32_t lds[LDS_BYTES / 4];

    const int lane = threadIdx.x & 63;
    const int sub = lane & 1;
    const int cw = lane >> 1;
    const int64_t cell = static_cast<int64_t>(blockIdx.x) * 32 + cw;
    const bool cell_ok = cell < C;
    const DevSortedChunk ch = chunks[blockIdx.y];
    // (the chunk's own table and flag rows: row of step s = trow0 + s - warm_start)
    const int32_t step_min = ch.warm_start - ch.trow0;
    const char* col = static_cast<const char*>(ts) + (cell_ok ? cell : C - 1) * static_cast<int64_t>(ES);
    // LOW quantiles (q <= 0.15: the launchers admit q >= 0.85 and q <= 0.15) run MIRRORED: the lists keep the K SMALLEST
    // samples of a step -- the keys are those of the negated samples, so "largest" means smallest --, the top set is the
    // lo + 1 smallest of the pool, its smallest member (mirrored order) is a[lo] and the largest key outside it a[lo + 1].
    // Position and weight (lo, g) come from the caller's q as always, so the interpolation is numpy's, bit for bit.
    const bool mirror = q < 0.5;
    const int kneg = (negate != 0) != mirror ? 1 : 0;      // the keys are those of the negated samples
    const uint32_t tmax = static_cast<uint32_t>(Tn - 1);
    const bool padded_last = (YPS - 1) * 2 + sub >= ntracks;
    const uint32_t padmask = padded_last ? 0xFFFFFFFFu : 0u;
    const uint32_t* tab = table + sub;

    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u32*)lds));
    // (the array must start the workgroup's allocation: a window above rank 0 relies on its address wrapping around)
    if (lds0 != 0u) __builtin_trap();
    // (rank 0 of list 0 of this cell)
    const uint32_t lcell = static_cast<uint32_t>(cw) * 4u;
    // ---- LDS: every list empty (key 0 = no key), the rows behind the lists 0 for good -------------------------
    for (int i = lane; i < LDS_BYTES / 4; i += 64) lds[i] = 0u;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    // own list slots: global slot g = sub * NL + j; P[j] = the list's keys INSIDE the top set = index of the first key outside
    // Per own list: the pointer P (a plain array: the select works on it), and in ONE 32-element register vector TV the float64
    // sum of its samples (two words side by side: a register pair), the register ranks and its valid keys.  Everything
    // reads them with static indices but 3. below, which replaces the values of the slot the new list goes to:
    // vector[wave-uniform index] is ONE v_mov through the index register (s_set_gpr_idx) for vectors of more than eight
    // elements -- up to eight the compiler expands it into a chain of selects, six a value.
    uint32_t P[NL];
    V32 TV = 0;
#define XMHW_RSLO(j) TV[2 * (j)]
#define XMHW_RSHI(j) TV[2 * (j) + 1]
#define XMHW_EX(e, j) TV[12 + 6 * (e) + (j)]
#define XMHW_NVL(j) TV[24 + (j)]
    uint32_t lbase[NL];                          // address of the list's rank 0
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int g = sub * NL + j;
        // (lane 1's sixth slot is a dummy: every rank of it is far outside the allocation and reads 0)
        lbase[j] = g < R ? lcell + static_cast<uint32_t>(g) * LSTRIDE : 0x40000000u + lcell;
        P[j] = 0;
    }
    static_assert(KL >= 5, "a list holds a window");
    static_assert(NL == 6 && EXT <= 2, "the register vector holds five values of six lists");
    // cell-level state, the same in both lanes
    // (B = the carried boundary: the key outside the top set.  During a chunk's warm-up rows nothing is selected -- the lists
    // are only built --, the top set stays empty and no new key counts as above the boundary; the chunk's first output row
    // then grows the top set from nothing: four or five rounds instead of R - 1 rows of selection)
#ifdef XMHW_WARM_SELECT      // (experiment: the select runs on warm-up rows too, as in round 5)
    uint32_t Ctop = 0, n = 0, B = 0u;
#else
    uint32_t Ctop = 0, n = 0, B = 0xFFFFFFFFu;
#endif
    double total = 0.0;

    // ---- sample addresses: a 64-bit pointer per track ------------------------------------------------------
    const uint32_t ld4 = static_cast<uint32_t>(ld) * ES;    // bytes between the steps of a cell
    const char* ap[YPS];
    auto entries_of = [&](int32_t step, uint32_t (&e)[YPS]) {
        const uint32_t* p = tab + static_cast<int64_t>(step - step_min) * NTP;
#pragma unroll
        for (int y = 0; y < YPS; ++y) e[y] = p[y * 2];
    };
    auto point_at = [&](int32_t step) {
        uint32_t e[YPS];
        entries_of(step, e);
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            const uint32_t t = umin((e[y] >> 1) - 2u, tmax);
            ap[y] = col + static_cast<uint64_t>(t) * ld4;
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int y = 0; y < YPS; ++y) ap[y] += (y == YPS - 1 && padded_last) ? 0u : ld4;
    };
    // (PACKED: the raw codes, sign-extended by the load; converted when their row starts)
    typename std::conditional<PACKED, int32_t, float>::type x_in[YPS];
    auto request = [&]() {
#pragma unroll
        for (int y = 0; y < YPS; ++y) {
            if constexpr (PACKED) x_in[y] = static_cast<int32_t>(*reinterpret_cast<const int16_t*>(ap[y]));
            else x_in[y] = *reinterpret_cast<const float*>(ap[y]);
        }
    };
    point_at(ch.warm_start);
    request();

    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = 0;
    if constexpr (STATS) tlast = __builtin_amdgcn_s_memtime();
    auto tick = [&](int idx) {
        if constexpr (STATS) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tacc[idx] += now - tlast;
            tlast = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    uint32_t st_rows = 0, st_iter = 0, st_flag = 0, st_steps = 0;
    uint32_t st_serial = 0, st_ext = 0;
    uint32_t st_more[4] = {0, 0, 0, 0};     // rounds after a row's first one by the keys still to move (wave maximum): <= 2, <= 4, <= 8, more

    // inputs of the epilogue of the row this lane finishes
    uint32_t e_alo = 0, e_ahi = 0, e_n = 0;
    double e_total = 0.0, e_g = 0.0;

    int32_t ep_s = -1;             // (wave-uniform) the row whose epilogue is due (-1: none), and its parity in the chunk
    uint32_t ep_eph = 0;
    auto finish_rows = [&]() {
        double th = make_nan(), se = make_nan();
        if (e_n > 0) {
            double v_lo = static_cast<double>(key_f32(e_alo));
            double v_hi = static_cast<double>(key_f32(e_ahi));
            if (mirror) {                    // (the keys were those of the negated samples)
                v_lo = -v_lo;
                v_hi = -v_hi;
            }
            se = e_total / static_cast<double>(e_n);
            if constexpr (PACKED) {
                v_lo = packed_value(pk, v_lo);
                v_hi = packed_value(pk, v_hi);
                se = packed_value(pk, se);
            }
            th = numpy_lerp(v_lo, v_hi, e_g);
        }
        if (static_cast<uint32_t>(sub) <= ep_eph && cell_ok) {
            const int64_t row = static_cast<int64_t>(ep_s) - static_cast<int64_t>(ep_eph) + sub;
            thresh[row * ldo + cell] = th;
            seas[row * ldo + cell] = se;
        }
        ep_s = -1;
    };
    bool nan_mode = false;         // (wave-uniform) the last plain row had a NaN sample
    uint32_t vi_n = 0xFFFFFFFFu, vi_lo = 0;      // the pool size the quantile position below was computed for
    double vi_g = 0.0;
    // (the slot of a step is a function of the step itself, not of the chunk: with the fixed-order total below the
    // outputs do not depend on how the row axis is cut -- a grid split over N ranks is bit-identical to the whole)
    int m = ((ch.warm_start % R) + R) % R;
    uint32_t sf_cur = __builtin_amdgcn_readfirstlane(sflags[ch.warm_start - step_min]);
    uint32_t sf_nxt = ch.warm_start + 1 < ch.end ? __builtin_amdgcn_readfirstlane(sflags[ch.warm_start + 1 - step_min]) : 0u;
#ifdef XMHW_LOOP_ALIGN
    asm volatile(".p2align " XMHW_STR(XMHW_LOOP_ALIGN) "\n\t.rept " XMHW_STR(XMHW_LOOP_PAD) "\n\ts_nop 0\n\t.endr" ::: "memory");
#endif
    for (int32_t s = ch.warm_start; s < ch.end; ++s) {
        const uint32_t sf = sf_cur;
        // (the flags of step s + 2 are asked for a row ahead: nothing waits for them)
        const uint32_t sf_nn = s + 2 < ch.end ? sflags[s + 2 - step_min] : 0u;
        // ---- 1. this row's samples -> keys (0 = invalid: NaN, outside [0, T), padding), their sum and count --------
        uint32_t k[HE];
        double din = 0.0;
        uint32_t nvin = 0;
        float xv[YPS];
        bool packed_fill = false;      // (PACKED, wave-uniform) some lane of the wave holds a fill code this row
        if constexpr (STATS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tick(5);
        }
        if constexpr (PACKED) {
            // (codes -> samples: xmhw_decode()'s arithmetic, the fill code -> NaN; big-endian codes swapped first.  Every
            // wave-uniform choice -- byte order, float32 recipe, "a lane holds a fill code" -- is ONE branch around a block
            // of YPS instructions, not a select per sample)
            if (pk.swap) {
#pragma unroll
                for (int y = 0; y < YPS; ++y)
                    x_in[y] = static_cast<int32_t>(static_cast<int16_t>(__builtin_bswap16(static_cast<uint16_t>(x_in[y]))));
            }
#pragma unroll
            for (int y = 0; y < YPS; ++y) xv[y] = static_cast<float>(x_in[y]);
            if (pk.mode == 1) {
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    float f = xv[y] * pk.sf;
                    f = f + pk.of;
                    xv[y] = f;
                }
            }
            bool anyfill = false;
#pragma unroll
            for (int y = 0; y < YPS; ++y) anyfill = anyfill || x_in[y] == pk.fill;
            packed_fill = wave_any(anyfill);
            if (packed_fill) {
#pragma unroll
                for (int y = 0; y < YPS; ++y) xv[y] = x_in[y] == pk.fill ? __uint_as_float(0x7FC00000u) : xv[y];
            }
        } else {
#pragma unroll
            for (int y = 0; y < YPS; ++y) xv[y] = x_in[y];
        }
        bool slow = !(sf & 1u) || nan_mode;
        if (!slow) {
            // a plain row: every real track pushes a sample; NaN shows in the sum (so does +inf next to -inf: those
            // rows take the general path below, which gives the same keys)
            // (cold spells: key(-x) and -sum(x), one instruction per sample less than negating the samples)
            // (keys and -- SUM -- the float64 sum of a plain row; NEG: key(-x))
            auto plain_row = [&](auto negc, auto sumc) {
                constexpr bool NEG = decltype(negc)::value, SUM = decltype(sumc)::value;
                double din1 = 0.0;
#pragma unroll
                for (int y = 0; y < YPS; ++y) {
                    uint32_t xb = __float_as_uint(xv[y]);
                    uint32_t ky = key_fast<NEG>(xb);
                    if (y == YPS - 1) {
                        xb &= ~padmask;
                        ky &= ~padmask;
                    }
                    k[y] = ky;
                    if constexpr (SUM) {
                        const double dv = static_cast<double>(__uint_as_float(xb));
                        if (y & 1) din1 = y == 1 ? dv : din1 + dv;      // (two chains: a float64 add waits for the one before it)
                        else din = y == 0 ? dv : din + dv;
                    }
                }
                if constexpr (SUM) din += din1;
            };
            bool int_sum = false;
            if constexpr (PACKED) int_sum = pk.mode != 1;
            if (int_sum) {
                // (codes that stand for themselves -- modes 2 and 3: the sum of a row is the sum of its codes, exact in 32-bit
                // integers: 20 fast-class adds instead of 20 conversions to float64 and 20 float64 adds; a fill code is
                // what sends the row to the general path, not a NaN in the sum)
                if constexpr (PACKED) {
                    int32_t isum = 0;
#pragma unroll
                    for (int y = 0; y < YPS; ++y)
                        isum += (y == YPS - 1) ? (x_in[y] & static_cast<int32_t>(~padmask)) : x_in[y];
                    if (kneg) plain_row(std::true_type{}, std::false_type{});
                    else plain_row(std::false_type{}, std::false_type{});
                    din = packed_fill ? make_nan() : static_cast<double>(isum);
                }
            } else if (kneg) {
                plain_row(std::true_type{}, std::true_type{});
            } else {
                plain_row(std::false_type{}, std::true_type{});
            }
            if (negate) din = -din;
            nvin = padded_last ? YPS - 1 : YPS;
            slow = wave_any(din != din);
            nan_mode = slow;      // (rows with NaN come in runs -- masked data: the next row goes straight to the general path)
        }
        if (XMHW_COLD(slow)) {
            din = 0.0;
            nvin = 0;
            // (the key through the two-instruction conversion, zeroed for a NaN or absent sample; the sample itself zeroed
            // before it is widened; cold spells: key(-x) and the sum negated once.  A SIMPLE row -- every real track pushes a
            // sample: masked data, configs[3] -- only has NaN to look for; the other rows also ask the step table)
#define XMHW_GENERAL_ROW(NEG_, OKEXPR)                                                                  \
    _Pragma("unroll") for (int y = 0; y < YPS; ++y) {                                                   \
        const float xs = xv[y];                                                                          \
        const bool ok = xs == xs && (OKEXPR);                                                            \
        k[y] = ok ? key_fast<NEG_>(__float_as_uint(xs)) : 0u;                                            \
        din += static_cast<double>(ok ? xs : 0.0f);                                                      \
        nvin += ok ? 1u : 0u;                                                                            \
    }
            if (sf & 1u) {
                if (kneg) { XMHW_GENERAL_ROW(true, !(y == YPS - 1 && padded_last)) }
                else { XMHW_GENERAL_ROW(false, !(y == YPS - 1 && padded_last)) }
            } else {
                uint32_t e[YPS];
                entries_of(s, e);
                if (kneg) { XMHW_GENERAL_ROW(true, (e[y] >> 1) >= 2u) }
                else { XMHW_GENERAL_ROW(false, (e[y] >> 1) >= 2u) }
            }
#undef XMHW_GENERAL_ROW
            if (negate) din = -din;
            if (sf & 1u) nan_mode = wave_any(nvin != static_cast<uint32_t>(padded_last ? YPS - 1 : YPS));
        }
#pragma unroll
        for (int y = YPS; y < HE; ++y) k[y] = 0u;
        tick(6);
        if (ep_s >= 0) finish_rows();        // the outputs of the two rows before this one (see 5. below)
        // prefetch: the samples of step s + 1, into the same registers
        if (s + 1 < ch.end) {
            if (!XMHW_COLD(!(sf_nxt & 2u))) advance();
            else point_at(s + 1);
            request();
        }
        tick(0);

        // ---- 2. sort: the lane's K largest keys (two sorted halves, merged), then one bitonic exchange between the
        // two lanes of the cell: lane 0 ends with ranks 0..K/2-1 of the cell's keys, lane 1 with ranks K/2..K-1
        uint32_t u[KH];
        {
            uint32_t ka[HE];
            {
                uint32_t h0[HH], h1[HH];
#pragma unroll
                for (int i = 0; i < HH; ++i) { h0[i] = k[i]; h1[i] = k[HH + i]; }
                sort_desc<HH>(h0);
                sort_desc<HH>(h1);
#pragma unroll
                for (int i = 0; i < HH; ++i) { ka[i] = h0[i]; ka[HH + i] = h1[i]; }
            }
            sortnet::MergeTop<HH, HH, K>::run(ka);
            uint32_t t[K];
#pragma unroll
            for (int i = 0; i < K; ++i) t[i] = umax(ka[i], swp(ka[K - 1 - i]));   // the K largest of the cell (bitonic)
            const uint32_t lb = sub ? 0u : 0xFFFFFFFFu;                         // lane 0 keeps the larger half
#pragma unroll
            for (int i = 0; i < KH; ++i) u[i] = med3u(t[i], t[i + KH], lb);
            sortnet::BitonicDesc<KH>::run(u);
        }
        tick(1);

        // ---- 3. the new list replaces the list in slot m ----------------------------------------------------------
        const int m_sub = m >= NL ? 1 : 0;
        const int mj = __builtin_amdgcn_readfirstlane(m - m_sub * NL);
        const bool own_m = sub == m_sub;
        // (this lane's first rank of list m: lane 0 holds ranks 0 .. K/2-1, lane 1 the rest)
        const uint32_t base_m = lcell + static_cast<uint32_t>(m) * LSTRIDE + static_cast<uint32_t>(sub * KH) * RSTRIDE;
#pragma unroll
        for (int i = 0; i < KH; ++i) {
            // (EXT: lane 1's last keys -- ranks KL .. K - 1 -- are not for LDS: the rows behind the lists stay 0)
            if (EXT == 0 || i < KH - EXT || sub == 0) lds_st(base_m + static_cast<uint32_t>(i) * RSTRIDE, u[i]);
        }
        // what leaves: the evicted list's share of the top set, its valid keys, its sum
        uint32_t c_old = 0, nv_old = 0;
        // (P[slot]: a chain of selects under scalar conditions)
        bool is_slot[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            int mj_ = mj;
            asm volatile("" : "+s"(mj_));
            is_slot[j] = mj_ == j;
        }
        const uint32_t nvm = TV[24 + mj];
        {
            uint32_t Pm = P[0];
#pragma unroll
            for (int j = 1; j < NL; ++j) Pm = is_slot[j] ? P[j] : Pm;
            if (own_m) {
                c_old = Pm;
                nv_old = nvm;
            }
        }
        // what comes: keys above the carried boundary join the top set
        uint32_t c_new = 0;
#pragma unroll
        for (int i = 0; i < KH; ++i) c_new += (u[i] > B) ? 1u : 0u;
        {
            uint32_t pk = (c_new << 16) | nvin;
            uint32_t po = (c_old << 16) | nv_old;
            pk += swp(pk);
            po += swp(po);
            c_new = umin(pk >> 16, static_cast<uint32_t>(KL));      // (the pointer counts LDS ranks)
            nvin = pk & 0xFFFFu;
            c_old = po >> 16;
            nv_old = po & 0xFFFFu;
        }
        din += swp(din);
        Ctop += c_new - c_old;
        n += nvin - nv_old;
#pragma unroll
        for (int j = 0; j < NL; ++j) P[j] = (own_m && is_slot[j]) ? c_new : P[j];
        // (the lanes that do not own slot m write back what they hold)
        TV[24 + mj] = own_m ? nvin : nvm;
        if constexpr (EXT > 0) {
            // (ranks KL .. K - 1 are lane 1's last keys; the lane that owns slot m keeps them)
#pragma unroll
            for (int e_ = 0; e_ < EXT; ++e_) {
                const uint32_t r_ = swp(u[KH - EXT + e_]);
                const uint32_t mine = sub ? u[KH - EXT + e_] : r_;
                const uint32_t old_ = TV[12 + 6 * e_ + mj];
                TV[12 + 6 * e_ + mj] = own_m ? mine : old_;
            }
        }
        {
            const uint64_t db = static_cast<uint64_t>(__double_as_longlong(din));
            const uint32_t ol_ = TV[2 * mj], oh_ = TV[2 * mj + 1];
            TV[2 * mj] = own_m ? static_cast<uint32_t>(db) : ol_;
            TV[2 * mj + 1] = own_m ? static_cast<uint32_t>(db >> 32) : oh_;
        }
        {
            // the pool's total: the 11 list sums added in slot order, every row (a running total would round differently
            // for every cut of the row axis, and an infinity that has left the pool would stay in it)
            double tsum = 0.0;
#pragma unroll
            for (int j = 0; j < NL; ++j)
                tsum += __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(XMHW_RSHI(j)) << 32) | XMHW_RSLO(j)));
            total = tsum + swp(tsum);
        }
        m = (m + 1 == R) ? 0 : m + 1;
        tick(2);

        // ---- 4. move the pointers until the top set holds Cs keys: a parallel merge-select ------------------------
        // (warm-up rows only build the lists: see B above)
        uint32_t a_lo = 0, a_hi = 0;
        double g = 0.0;
        bool flag = false;
#ifdef XMHW_WARM_SELECT
        if (true) {
#else
        if (s >= ch.begin) {
#endif
        // (n is the same for every cell on every row of gap-free data: the float64 position is redone only when some cell's
        // count has changed)
        if (XMHW_COLD(wave_any(n != vi_n))) {
            vi_n = n;
            const uint32_t nn_ = n ? n : 1u;
            const double vi = static_cast<double>(nn_ - 1) * q;
            const double fl = floor(vi);
            vi_g = vi - fl;
            vi_lo = static_cast<uint32_t>(fl);
        }
        const uint32_t nn = n ? n : 1u;
        g = vi_g;
        const uint32_t lo = vi_lo;
        const bool need2 = lo + 1 < nn;
        const uint32_t Cs = n ? (mirror ? lo + 1u : n - 1u - lo) : 0u;
        // Direction of the cell: GROW (keys join the top set, largest first) or SHRINK (keys leave it, smallest first).
        // Shrinking cells work on NEGATED keys (2^32 - key), so that "the key that moves next" is the largest one for
        // everybody and 0 -- no key: an invalid sample, a position outside the list -- stays the lowest for both.
        // (a cell whose top set is right already counts as growing by 0)
        const bool grow = Ctop <= Cs;
        uint32_t rem = grow ? Cs - Ctop : Ctop - Cs;
        const uint32_t steps0 = rem;
        const uint32_t cm = grow ? 0u : 0xFFFFFFFFu;
        const uint32_t cneg = grow ? 0u : 1u;
        auto cpl = [&](uint32_t v) -> uint32_t { return (v ^ cm) + cneg; };      // (its own inverse)
        const int psign = grow ? 1 : -1;
        const int32_t dstep = grow ? static_cast<int32_t>(RSTRIDE) : -static_cast<int32_t>(RSTRIDE);
        // (per list, for this row's direction: the address the window starts from when the pointer is 0)
        uint32_t wbase[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) wbase[j] = lbase[j] - (grow ? 0u : RSTRIDE);
        // (prem = the keys the cell still has to move, 0 once it has given up: what the loop conditions look at)
        uint32_t prem = rem;
        uint32_t st_iter_row = 0;
        bool st_first_done = false;
        bool first_round = true;
        while (wave_any(prem != 0u)) {
            // After a row's first round two rounds in three have at most three keys left to move in their worst cell (an
            // unsafe window, more than 15 steps): those are moved ONE BY ONE -- the largest of the 22 list heads, six reads and
            // ~70 instructions a key instead of a round's 390.
            if (XMHW_COLD(!first_round && !wave_any(prem > kSerialMax))) {
                while (wave_any(prem != 0u)) {
                    if constexpr (STATS) ++st_serial;
                    uint32_t hd[NL];
                    uint32_t hm = 0;
#pragma unroll
                    for (int j = 0; j < NL; ++j) {
                        hd[j] = cpl(lds_ld(mad_u24(P[j], RSTRIDE, wbase[j])));
                        hm = umax(hm, hd[j]);
                    }
                    const uint32_t ho = swp(hm);
                    const uint32_t cmax = umax(hm, ho);
                    const bool act = prem != 0u;                        // (the cell has keys left to move)
                    const bool dry_ = act && cmax == 0u;                // nothing left in the lists: give up (flag)
                    const bool win = act && !dry_ && hm == cmax && (sub == 0 || ho != cmax);     // lane 0 first on a tie
                    bool found = false;
#pragma unroll
                    for (int j = 0; j < NL; ++j) {
                        const bool sel = win && !found && hd[j] == cmax;
                        P[j] += sel ? static_cast<uint32_t>(psign) : 0u;
                        found = found || sel;
                    }
                    rem -= (act && !dry_) ? 1u : 0u;
                    flag = flag | dry_;
                    prem = flag ? 0u : rem;
                }
                break;
            }
            first_round = false;
            if constexpr (STATS) {
                ++st_iter;
#if defined(XMHW_HIST_W3)
                if (false) {
#elif defined(XMHW_HIST_FIRST)      // (experiment: the histogram is of the row's FIRST round instead -- thresholds XMHW_H1 / XMHW_H2)
                if (!st_first_done) {
#else
                if (st_first_done) {
#endif
                    uint32_t r_ = prem;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) r_ = umax(r_, static_cast<uint32_t>(__shfl_xor(static_cast<int>(r_), off, 64)));
                    #ifdef XMHW_HIST_FIRST
                    ++st_more[r_ <= XMHW_H1 ? 1 : r_ <= XMHW_H2 ? 2 : 3];
#else
                    ++st_more[r_ <= 2u ? 0 : r_ <= 4u ? 1 : r_ <= 8u ? 2 : 3];
#endif
                }
                st_first_done = true;
            }
            const uint32_t d = umin(prem, 15u);
            // -- the W = 4 next keys of every own list, in the order they would move.  A position past the last rank of
            //    the list, or above rank 0, is outside the allocation and reads 0 -- the LOWEST key in the cell's own order
            uint32_t a[NL][4];
            uint32_t F = 0;            // the largest FIFTH key: whatever the lists hold beyond the windows is not above it
            {
                // (addresses first, then the 30 reads back to back, then ONE wait: left to itself the compiler
                // interleaves them and waits for the LDS five times per list)
                uint32_t ad[NL][5];
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    ad[j][0] = mad_u24(P[j], RSTRIDE, wbase[j]);
#pragma unroll
                    for (int i = 1; i < 5; ++i) ad[j][i] = ad[j][i - 1] + static_cast<uint32_t>(dstep);
                }
                __builtin_amdgcn_sched_barrier(0);
                uint32_t raw[NL][5];
#pragma unroll
                for (int j = 0; j < NL; ++j)
#pragma unroll
                    for (int i = 0; i < 5; ++i) raw[j][i] = lds_ld(ad[j][i]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NL; ++j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[j][i] = cpl(raw[j][i]);
                    F = umax(F, cpl(raw[j][4]));
                }
            }
            // -- the lane's LT = 16 largest of its 24, sorted: a tree of merges of sorted runs.  (LT = 12 is exact too -- what a lane
            // holds beyond its twelfth key is not above that key, which then joins the trust bound F -- and 28 instructions a
            // round shorter; measured: -0.5 % on smooth data, +1.5 % on quantised data with sea ice, where cells that move
            // 15 keys a round take twelve of them from one lane's lists often enough.  Not used.)
#ifndef XMHW_LANE_TOP
#define XMHW_LANE_TOP 16
#endif
            constexpr int LT = XMHW_LANE_TOP;
            static_assert(LT == 12 || LT == 16, "merge networks exist for these");
            uint32_t sl[LT];
            {
                uint32_t r01[8], r23[8], r45[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    r01[i] = a[0][i]; r01[4 + i] = a[1][i];
                    r23[i] = a[2][i]; r23[4 + i] = a[3][i];
                    r45[i] = a[4][i]; r45[4 + i] = a[5][i];
                }
                sortnet::MergeTop<4, 4, 8>::run(r01);
                sortnet::MergeTop<4, 4, 8>::run(r23);
                sortnet::MergeTop<4, 4, 8>::run(r45);
                uint32_t r4[16];
#pragma unroll
                for (int i = 0; i < 8; ++i) { r4[i] = r01[i]; r4[8 + i] = r23[i]; }
                sortnet::MergeTop<8, 8, LT>::run(r4);
                uint32_t r6[LT + 8];
#pragma unroll
                for (int i = 0; i < LT; ++i) r6[i] = r4[i];
#pragma unroll
                for (int i = 0; i < 8; ++i) r6[LT + i] = r45[i];
                sortnet::MergeTop<LT, 8, LT>::run(r6);
#pragma unroll
                for (int i = 0; i < LT; ++i) sl[i] = r6[i];
            }
            if constexpr (LT < 16) F = umax(F, sl[LT - 1]);
            // -- the cell's 16 largest as a bitonic sequence, split: lane 0 takes the larger of every pair (ranks 0..7), lane 1
            // the smaller (ranks 8..15)
            uint32_t u8[8];
            {
                uint32_t t16[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    // (positions past a lane's LT keys hold nothing: 0)
                    if (i < LT && 15 - i < LT) t16[i] = umax(sl[i], swp(sl[15 - i]));
                    else if (i < LT) t16[i] = sl[i];
                    else t16[i] = swp(sl[15 - i]);
                }
                const uint32_t lb = sub ? 0u : 0xFFFFFFFFu;
#pragma unroll
                for (int i = 0; i < 8; ++i) u8[i] = med3u(t16[i], t16[i + 8], lb);
            }
            // -- the cell's key of rank d - 1 (0 if d == 0): u8 is a bitonic sequence -- the bitonic sorter PRUNED to the one
            // output asked for: a stage keeps the half the rank lies in, and "the larger or the smaller of a pair, by a bit of
            // the rank" is one v_med3 against a per-lane bound (all ones: the larger; 0: the smaller).  15 instructions
            // instead of the sorter's 24 and a 12-instruction pick
            uint32_t tl;
            {
                const uint32_t idx = (d - 1u) - (sub ? 8u : 0u);      // (rank inside this lane's half; >= 8: not here)
                const int32_t nidx = static_cast<int32_t>(~idx);
                const uint32_t b4 = static_cast<uint32_t>(__builtin_amdgcn_sbfe(nidx, 2, 1));
                const uint32_t b2 = static_cast<uint32_t>(__builtin_amdgcn_sbfe(nidx, 1, 1));
                const uint32_t b1 = static_cast<uint32_t>(__builtin_amdgcn_sbfe(nidx, 0, 1));
                const uint32_t c0 = med3u(u8[0], u8[4], b4), c1 = med3u(u8[1], u8[5], b4);
                const uint32_t c2 = med3u(u8[2], u8[6], b4), c3 = med3u(u8[3], u8[7], b4);
                const uint32_t e0 = med3u(c0, c2, b2), e1 = med3u(c1, c3, b2);
                uint32_t z = med3u(e0, e1, b1);
                z = idx < 8u ? z : 0u;
                tl = z | swp(z);
            }
            // -- how far the windows can be trusted: a key is SAFE if it is not below the largest fifth key
            F = umax(F, swp(F));
            const bool act = d != 0u;
            const bool unsafe = act && tl < F;                 // move only the safe keys this round, look again
            const bool dry = act && !unsafe && tl == 0u;       // fewer than d keys left in the lists: give up
            // (a cell that moves nothing this round -- settled, or dry -- counts against a threshold no key reaches: its counts are
            // 0 and the pointer updates below need no condition)
            const bool move = act && !dry;
            const uint32_t th = move ? (unsafe ? F : tl) : 0xFFFFFFFFu;
#ifdef XMHW_HIST_W3      // (experiment: would windows of THREE keys have been safe in this row's first round?)
            if constexpr (STATS) {
                if (st_iter_row == 0) {
                    uint32_t f3 = 0;
#pragma unroll
                    for (int j = 0; j < NL; ++j) f3 = umax(f3, a[j][3]);
                    f3 = umax(f3, swp(f3));
                    ++st_more[wave_any(act && tl < f3) ? 2 : 1];
                }
                ++st_iter_row;
            }
#endif
            uint32_t pj[NL];
            uint32_t psum = 0;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                pj[j] = count_ge4(a[j][0], a[j][1], a[j][2], a[j][3], th);
                psum += pj[j];
            }
            // (more than d keys at or above the d-th: it is tied with the keys behind it)
            uint32_t moved = psum + swp(psum);
            const bool tie = move & !unsafe & (moved > d);
            if (XMHW_COLD(wave_any(tie))) {
                asm volatile("" ::: "memory");          // (keep this a branch: plain rows never come here)
                // the d-th and the (d+1)-th key are equal: of the keys equal to tl only d - #{keys above tl} move, lists in
                // order (lane 0 first)
                uint32_t gj[NL];
                uint32_t gsum = 0;
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    // (keys above tl = keys at or above tl + 1; tl is a real key here, never all ones)
                    gj[j] = count_ge4(a[j][0], a[j][1], a[j][2], a[j][3], tl + 1u);
                    gsum += gj[j];
                }
                const uint32_t G = gsum + swp(gsum);
                const uint32_t e_mine = psum - gsum;
                const uint32_t e_other = swp(e_mine);
                int32_t room = static_cast<int32_t>(d - G) - static_cast<int32_t>(sub ? e_other : 0u);   // ties this lane may still move
                uint32_t ps2 = 0;
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    const int32_t ej = static_cast<int32_t>(pj[j] - gj[j]);
                    const int32_t take = room < 0 ? 0 : (room < ej ? room : ej);
                    room -= ej;
                    const uint32_t pt = gj[j] + static_cast<uint32_t>(take);
                    pj[j] = tie ? pt : pj[j];
                    ps2 += pt;
                }
                psum = tie ? ps2 : psum;
                moved = psum + swp(psum);
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) P[j] = mad_i24(pj[j], psign, P[j]);      // (one multiply-add: P +- the list's count)
            rem -= moved;
            // (a dry cell gives up: the row goes to the recomputation, what it could not move stays in `rem`.  An unsafe round
            // leaves keys to move, so "keys left" is all the loop has to ask)
            flag = flag | dry;
            prem = flag ? 0u : rem;
        }
        {
            // The two keys at the boundary of the top set, from the lists' pointers: a[lo] = the largest key OUTSIDE (max over
            // the lists of key[P]), a[lo + 1] = the smallest INSIDE (min of key[P - 1]).  Twelve reads and ~30 instructions
            // once a row, for every cell alike -- whichever way it moved, or not at all -- instead of tracking the last key
            // that moved through the rounds and reading one side before the select and the other after it.
            // (P == 0: nothing inside, the read leaves the allocation: 0 - 1 = the largest word; no key outside: 0)
            uint32_t ao[NL], ai[NL], ho[NL], hi_[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                ao[j] = mad_u24(P[j], RSTRIDE, lbase[j]);
                ai[j] = ao[j] - RSTRIDE;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                ho[j] = lds_ld(ao[j]);
                hi_[j] = lds_ld(ai[j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            uint32_t mx = 0, mn = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                mx = umax(mx, ho[j]);
                mn = umin(mn, hi_[j] - 1u);
            }
            a_lo = umax(mx, swp(mx));
            a_hi = umin(mn, swp(mn)) + 1u;
        }
        Ctop = mad_i24(rem, -psign, Cs);      // (Cs, or -- a cell that gave up -- what its pointers hold: Cs -+ the keys not moved)
        // (no a[lo + 1] -- lo = n - 1 --: both keys are a[lo], the key outside the top set or, mirrored, its smallest member)
        if (!need2) {
            if (mirror) a_lo = a_hi;
            else a_hi = a_lo;
        }
        if constexpr (EXT > 0) {
            // ---- 4c. the register ranks.  The select above knows the LDS ranks only: its top set is the Cs largest of THOSE
            // keys.  A register key can belong to the true top set only if its list's KL LDS keys are all inside (it is below
            // them) and it lies above the boundary.  Such keys join (pv[j] of list j, largest first) and as many keys -- the
            // smallest of the enlarged set, one by one: LDS heads or register keys -- leave.  Ctop keeps counting LDS keys
            // only, so the next row starts from the same kind of state.
            // (what every row pays: is some saturated list's first register key above the boundary? -- twelve compares and
            // scalar logic, no branch per list)
            bool anyx = false;
#pragma unroll
            for (int j = 0; j < NL; ++j) anyx = anyx | ((P[j] == static_cast<uint32_t>(KL)) & (XMHW_EX(0, j) > a_lo));
            if (XMHW_COLD(wave_any(anyx & !flag))) {
                asm volatile("" ::: "memory");
                uint32_t pv[NL];
                uint32_t need = 0;
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    const bool sat = (P[j] == static_cast<uint32_t>(KL)) & !flag;
                    uint32_t c_ = 0;
#pragma unroll
                    for (int e_ = 0; e_ < EXT; ++e_) c_ += XMHW_EX(e_, j) > a_lo ? 1u : 0u;
                    pv[j] = sat ? c_ : 0u;
                    need += pv[j];
                }
                need += swp(need);
                if constexpr (STATS) ++st_ext;
                const bool fix = need != 0u;
                auto heads = [&](uint32_t (&hd)[NL]) -> uint32_t {     // own lists' smallest inside keys - 1 (none: all ones), their minimum
                    uint32_t lv[NL];
#pragma unroll
                    for (int j = 0; j < NL; ++j) lv[j] = lds_ld(lbase[j] + (P[j] - 1u) * RSTRIDE);
                    uint32_t hm = 0xFFFFFFFFu;
#pragma unroll
                    for (int j = 0; j < NL; ++j) {
                        uint32_t ev = XMHW_EX(0, j);          // (the list's smallest register key inside: rank KL + pv - 1)
#pragma unroll
                        for (int e_ = 1; e_ < EXT; ++e_) ev = pv[j] > static_cast<uint32_t>(e_) ? XMHW_EX(e_, j) : ev;
                        hd[j] = (pv[j] != 0u ? ev : lv[j]) - 1u;
                        hm = umin(hm, hd[j]);
                    }
                    return hm;
                };
                uint32_t vlast = a_lo;
                while (wave_any(need != 0u)) {
                    uint32_t hd[NL];
                    const uint32_t hm = heads(hd);
                    const uint32_t ho = swp(hm);
                    const uint32_t cmin = umin(hm, ho);
                    const bool act = need != 0u;
                    const bool win = act && hm == cmin && (sub == 0 || ho != cmin);     // lane 0 first on a tie
                    bool found = false;
#pragma unroll
                    for (int j = 0; j < NL; ++j) {
                        const bool sel = win && !found && hd[j] == cmin;
                        const bool reg = pv[j] != 0u;
                        P[j] = (sel && !reg) ? P[j] - 1u : P[j];
                        pv[j] = (sel && reg) ? pv[j] - 1u : pv[j];
                        found = found || sel;
                    }
                    if (act) {
                        vlast = cmin + 1u;
                        need -= 1u;
                    }
                }
                uint32_t hd[NL];
                uint32_t hm = heads(hd);
                hm = umin(hm, swp(hm)) + 1u;
                uint32_t pvs = 0;
#pragma unroll
                for (int j = 0; j < NL; ++j) pvs += pv[j];
                pvs += swp(pvs);
                if (fix) {
                    a_lo = vlast;
                    a_hi = hm;
                    if (!need2) {
                        if (mirror) a_lo = a_hi;
                        else a_hi = a_lo;
                    }
                    Ctop = Cs - pvs;
                    // every register key of a list inside and more valid keys than the K the cell keeps: the list may hide keys
                    // above the boundary (its last kept key lies above it) -- the row goes to the recomputation
#pragma unroll
                    for (int j = 0; j < NL; ++j) flag = flag || (pv[j] == static_cast<uint32_t>(EXT) && XMHW_NVL(j) > static_cast<uint32_t>(K));
                }
            }
        } else {
            // a list stored to its last key, all of it inside the top set, with keys that were not stored: what was not
            // stored is not above the list's last stored key U -- the row is wrong only if U lies above the key outside
            // the top set (a tie with it is harmless: sea-ice plateaus, quantised values)
            uint32_t atb = 0;
#pragma unroll
            for (int j = 0; j < NL; ++j)
                atb |= (P[j] == static_cast<uint32_t>(K) && XMHW_NVL(j) > static_cast<uint32_t>(K)) ? (1u << j) : 0u;     // (never the dummy: no bit)
            if (XMHW_COLD(wave_any(atb != 0u))) {
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    const uint32_t Uj = lds_ld(lbase[j] + static_cast<uint32_t>(K - 1) * RSTRIDE);
                    flag = flag || (((atb >> j) & 1u) && Uj > a_lo);
                }
            }
        }
        B = a_lo;
        {
            // (NOT `flag || swp(...)`: the short-circuit would run the exchange with the flagged lanes switched off, and
            // a DPP read of a lane that is switched off returns 0 -- the partner lane would never see the flag)
            const uint32_t fl_ = flag ? 1u : 0u;
            flag = (fl_ | swp(fl_)) != 0u;
        }
        if constexpr (STATS) st_flag += (sub == 0 && cell_ok && flag) ? 1u : 0u;
        tick(3);
        if (XMHW_COLD(s >= ch.begin && wave_any(flag))) {
            // ---- 4b. the flagged cells of this row, one after the other, by the whole wave (pool_order_stats above) ---------
            constexpr int KPL = (NTP * R + 63) / 64;
            unsigned long long fm = __builtin_amdgcn_ballot_w64(flag && sub == 0 && cell_ok);
            while (fm != 0ull) {
                const int L = __builtin_ctzll(fm);
                fm &= fm - 1ull;
                const int64_t cc = static_cast<int64_t>(blockIdx.x) * 32 + (L >> 1);
                const uint32_t n_c = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(n), L));
                uint32_t lo_c = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(lo), L));
                // (mirrored keys: ascending index i of the keys is index n - 1 - i of the samples; wanted: a[lo + 1] and a[lo])
                if (mirror) lo_c = lo_c + 1u < n_c ? n_c - 2u - lo_c : 0u;
                const uint32_t guess = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(a_lo), L));
                uint32_t key[KPL];
                {
                    // (two sweeps -- table entries, samples -- so that the loads of a sweep are in flight together)
                    uint32_t ent[KPL];
#pragma unroll
                    for (int i = 0; i < KPL; ++i) {
                        const int e = lane + 64 * i;
                        const int j = e / NTP, trk = e - j * NTP;
                        ent[i] = e < NTP * R ? table[static_cast<int64_t>(s - j - step_min) * NTP + trk] : (kCodeInvalid << 1);
                    }
                    const char* colc = static_cast<const char*>(ts) + cc * static_cast<int64_t>(ES);
                    typename std::conditional<PACKED, int32_t, float>::type raw[KPL];
#pragma unroll
                    for (int i = 0; i < KPL; ++i) {
                        const uint32_t t = umin((ent[i] >> 1) - 2u, tmax);
                        const char* a_ = colc + static_cast<uint64_t>(t) * ld4;
                        if constexpr (PACKED) raw[i] = static_cast<int32_t>(*reinterpret_cast<const int16_t*>(a_));
                        else raw[i] = *reinterpret_cast<const float*>(a_);
                    }
#pragma unroll
                    for (int i = 0; i < KPL; ++i) {
                        float xs;
                        if constexpr (PACKED) {
                            int32_t c_ = raw[i];
                            if (pk.swap) c_ = static_cast<int32_t>(static_cast<int16_t>(__builtin_bswap16(static_cast<uint16_t>(c_))));
                            xs = packed_sample(pk, c_);
                        } else {
                            xs = raw[i];
                        }
                        const bool ok = xs == xs && (ent[i] >> 1) >= 2u;
                        const uint32_t kk = kneg ? key_fast<true>(__float_as_uint(xs)) : key_fast<false>(__float_as_uint(xs));
                        key[i] = ok ? kk : 0u;
                    }
                }
                uint32_t r_lo = 0, r_hi = 0;
                pool_order_stats<KPL>(key, n_c, lo_c, guess, r_lo, r_hi);
                const bool mine = (lane >> 1) == (L >> 1);
                // (mirror, no a[lo + 1] -- lo = n - 1 --: the one key asked for is a[lo])
                if (mirror && !(lo_c + 1u < n_c && n_c >= 2u)) r_hi = r_lo;
                a_lo = mine ? r_lo : a_lo;
                a_hi = mine ? r_hi : a_hi;
                flag = mine ? false : flag;
            }
            // (B stays the select's own boundary: the pointers of a flagged cell describe THAT top set, and the next row's new
            // keys must be counted against the same value -- the exact answer is for the output only)
        }
        tick(7);
        if constexpr (STATS) st_steps += (sub == 0 && cell_ok) ? steps0 : 0u;
        }
        tick(3);

        // ---- 5. output ---------------------------------------------------------------------------------------------
        if (s >= ch.begin) {
            if constexpr (STATS) ++st_rows;
            // The epilogue (key -> value, numpy's lerp, the float64 division, the stores) is the same ~60 instructions for
            // both lanes of a cell: they take turns -- lane `sub` keeps the inputs of the rows with (s - begin) % 2 == sub
            // and every second row (and at the end of the chunk) each lane finishes ITS row.
            const uint32_t eph = static_cast<uint32_t>(s - ch.begin) & 1u;
            if (static_cast<uint32_t>(sub) == eph) {
                e_alo = mirror ? a_hi : a_lo;        // a[lo]
                e_ahi = mirror ? a_lo : a_hi;        // a[lo + 1]
                e_n = n;
                e_total = total;
                e_g = g;
            }
            // (... at the START of the next row, between its key conversion and the requests for the row after: a wave waits for
            // its samples with s_waitcnt vmcnt, which counts loads and stores alike in issue order -- stores issued here, behind
            // the requests, are what the next row's wait for its last sample would have to sit out as well)
            if (eph == 1u || s + 1 == ch.end) {
                ep_s = s;
                ep_eph = eph;
            }
        }
        tick(4);
        sf_cur = sf_nxt;
        sf_nxt = __builtin_amdgcn_readfirstlane(sf_nn);
    }
    if (ep_s >= 0) finish_rows();
    if (STATS && stats != nullptr) {
        if (lane == 0) {
            atomicAdd(&stats[0], static_cast<unsigned long long>(st_rows));
            atomicAdd(&stats[1], static_cast<unsigned long long>(st_iter));
#pragma unroll
            for (int i = 1; i < 4; ++i) atomicAdd(&stats[4 + i], static_cast<unsigned long long>(st_more[i]));
            atomicAdd(&stats[4], static_cast<unsigned long long>(st_ext));      // (st_more[0] is always 0: such rows are finished key by key)
#pragma unroll
            for (int i = 0; i < 8; ++i) atomicAdd(&stats[8 + i], tacc[i]);
        }
        if (sub == 0) {
            atomicAdd(&stats[2], static_cast<unsigned long long>(st_flag));
            atomicAdd(&stats[3], static_cast<unsigned long long>(st_steps));
        }
    }
}

// (registers: LDS holds 8 waves per CU at K = 16: two waves per SIMD is what the body is asked to fit.  The shortest records --
// up to 8 tracks per lane, 8 keys per list: 11,520 bytes of LDS a wave -- are asked to fit THREE (168 registers, 20..60 bytes
// of scratch): 12- and 16-year records -4 % and -5.6 %; 9..12 tracks per lane would need 100..164 bytes of scratch and run
// 1.5 x slower that way -- profiles/r6_experiments.txt)
#ifndef XMHW_WAVES3_MAX_YPS
#define XMHW_WAVES3_MAX_YPS 8
#endif
#define XMHW_WAVES_PER_SIMD(Y) ((Y) <= XMHW_WAVES3_MAX_YPS ? 3 : 2)
template <int YPS, int K, int KL, bool STATS>
__global__ __launch_bounds__(64, XMHW_WAVES_PER_SIMD(YPS)) void clim_sorted_f32(
    const float* __restrict__ ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* __restrict__ table,
    const uint32_t* __restrict__ sflags, const DevSortedChunk* __restrict__ chunks, double q, int negate,
    int32_t ntracks, double* __restrict__ thresh, double* __restrict__ seas, int64_t ldo,
    unsigned long long* __restrict__ stats) {
    sorted_body<YPS, K, KL, STATS, false>(ts, C, ld, Tn, table, sflags, chunks, q, negate, ntracks, thresh, seas, ldo, stats,
                                      PackedI16{});
}
// the same on int16 codes (no counter twin)
template <int YPS, int K, int KL>
__global__ __launch_bounds__(64, XMHW_WAVES_PER_SIMD(YPS)) void clim_sorted_i16(
    const int16_t* __restrict__ codes, PackedI16 pk, int64_t C, int64_t ld, int64_t Tn, const uint32_t* __restrict__ table,
    const uint32_t* __restrict__ sflags, const DevSortedChunk* __restrict__ chunks, double q, int negate,
    int32_t ntracks, double* __restrict__ thresh, double* __restrict__ seas, int64_t ldo) {
    sorted_body<YPS, K, KL, false, true>(codes, C, ld, Tn, table, sflags, chunks, q, negate, ntracks, thresh, seas, ldo, nullptr, pk);
}

// ---------------------------------------------------------------------------
namespace {
typedef void (*SortedKernel)(const float*, int64_t, int64_t, int64_t, const uint32_t*, const uint32_t*,
                             const DevSortedChunk*, double, int, int32_t, double*, double*, int64_t, unsigned long long*);
typedef void (*SortedKernelI16)(const int16_t*, PackedI16, int64_t, int64_t, int64_t, const uint32_t*, const uint32_t*,
                                const DevSortedChunk*, double, int, int32_t, double*, double*, int64_t);
struct SortedEntry { int yps, k, kl; SortedKernel fn, fn_stats; SortedKernelI16 fn_i16; };
#ifdef XMHW_RING_STATS
#define XMHW_SS(Y, K, KL) clim_sorted_f32<Y, K, KL, true>
#else
#define XMHW_SS(Y, K, KL) nullptr
#endif
#define XMHW_S2(Y, K, KL) {Y, K, KL, clim_sorted_f32<Y, K, KL, false>, XMHW_SS(Y, K, KL), clim_sorted_i16<Y, K, KL>}
#define XMHW_S(Y, K) XMHW_S2(Y, K, K)
// tracks per lane -> keys stored per list: about 0.4 x the tracks of the record (a list's share of the pool's top tenth
// is a tenth of the tracks on average and reaches three to four times that on a steep seasonal slope), even, at most
// what a lane holds.  9..48 tracks.
#ifndef XMHW_K40
#define XMHW_K40 16      // (keys per list of the 37..40-track records ...
#endif
#ifndef XMHW_KL48
#define XMHW_KL48 16     // (41..48 tracks keep 18 keys per list: 16 ranks in LDS = 7 waves per CU instead of 6, two in registers;
                         //  14 + four in registers = 8 waves was measured slower: 43 tracks 14.8 against 14.55 ms, 48 tracks 17.9 / 15.6)
#endif
#ifndef XMHW_KL40
#define XMHW_KL40 14     //  ... and how many of them live in LDS: 14 = 8 waves per CU, the other two in registers)
#endif
const SortedEntry kSorted[] = {
#ifdef XMHW_SORTED_ONLY      // (tools/isa_sorted.sh: one instantiation, for a quick look at the ISA)
    XMHW_S2(20, XMHW_K40, XMHW_KL40),
#else
    XMHW_S(5, 6),   XMHW_S(6, 6),   XMHW_S(7, 8),   XMHW_S(8, 8),   XMHW_S(9, 10),  XMHW_S(10, 10), XMHW_S(11, 10),
    XMHW_S(12, 10), XMHW_S(13, 12), XMHW_S(14, 12), XMHW_S(15, 12), XMHW_S(16, 12), XMHW_S(17, 14), XMHW_S(18, 14),
    XMHW_S2(19, XMHW_K40, XMHW_KL40), XMHW_S2(20, XMHW_K40, XMHW_KL40), XMHW_S2(21, 18, XMHW_KL48), XMHW_S2(22, 18, XMHW_KL48), XMHW_S2(23, 18, XMHW_KL48), XMHW_S2(24, 18, XMHW_KL48),
#endif
};
#undef XMHW_S
#undef XMHW_S2
#undef XMHW_SS
const SortedEntry* find_sorted(int32_t yps) {
    for (const auto& e : kSorted)
        if (e.yps == yps) return &e;
    return nullptr;
}
}  // namespace

// ---------------------------------------------------------------------------
// What the rank-major lists rely on, checked on the device itself: a workgroup fills an allocation of BYTES with a
// pattern and reads where the kernel's windows can end up -- the first row behind the allocation and the rows after it
// (past the last rank of a list), the rows "above" address 0 (above rank 0: the address wraps around), the dummy list's
// base.  Every such read must return 0.  Many workgroups per CU, so that the bytes behind an allocation belong to a
// neighbour that has just written its own pattern.
namespace {
__global__ __launch_bounds__(64) void lds_outside_probe(uint32_t* __restrict__ bad, int BYTES) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];       // (BYTES of dynamic LDS: one kernel for every size)
    for (int i = threadIdx.x; i < BYTES / 4; i += 64) lds[i] = 0x80000000u | (blockIdx.x << 16) | static_cast<uint32_t>(i);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const uint32_t base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u32*)lds));
    uint32_t wrong = base != 0u ? 1u : 0u;
    const uint32_t lane4 = threadIdx.x * 4u;
    constexpr uint32_t RS = 11u * 128u;
    for (uint32_t r = 0; r < 6; ++r) {
        // (all 32 cells of a row, rows 0 .. 10 behind the allocation and r ranks further; the same above address 0)
        for (uint32_t gofs = 0; gofs + 256u <= 10u * 128u; gofs += 256u) {      // (gofs + lane4 stays inside one rank's rows)
            wrong |= lds_ld(static_cast<uint32_t>(BYTES) + r * RS + gofs + lane4);
            wrong |= lds_ld(0u - (r + 1u) * RS + gofs + lane4);
        }
        wrong |= lds_ld(0x40000000u + r * RS + lane4);
        wrong |= lds_ld(0x40000000u - (r + 1u) * RS + lane4);
    }
    // ... and a read inside the allocation returns what was stored (the probe would otherwise pass on a part that reads 0 everywhere)
    if (lds_ld(static_cast<uint32_t>(BYTES) - 256u + lane4) != (0x80000000u | (blockIdx.x << 16) | static_cast<uint32_t>(BYTES / 4 - 64 + threadIdx.x)))
        wrong = 1u;
    if (wrong != 0u) atomicAdd(bad, 1u);
}
}  // namespace

// 0 = every probe read 0 (the sorted-list kernel may run on this device), otherwise the number of lanes that did not
hipError_t sorted_lds_probe(uint32_t* d_bad, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    // (the allocation sizes of the instantiations: K = 6 .. 18 keys per list, each rounded up to the 1,280-byte piece)
    for (int bytes : {8960, 11520, 14080, 17920, 20480, 23040, 25600})
        hipLaunchKernelGGL(lds_outside_probe, dim3(2048), dim3(64), static_cast<size_t>(bytes), stream, d_bad, bytes);
    return hipGetLastError();
}

int32_t sorted_pick_yps(int32_t w, int32_t ntracks) {
    if (w != 5) return 0;
    const int32_t yps = (ntracks + 1) / 2;
    return find_sorted(yps) ? yps : 0;
}

int32_t sorted_pick_k(int32_t w, int32_t ntracks) {
    if (w != 5) return 0;
    const SortedEntry* e = find_sorted((ntracks + 1) / 2);
    return e ? e->k : 0;
}

int32_t sorted_lds_bytes(int32_t w, int32_t ntracks) {
    if (w != 5) return 0;
    const SortedEntry* e = find_sorted((ntracks + 1) / 2);
    return e ? (11 * e->kl * 128 + kLdsGranule - 1) / kLdsGranule * kLdsGranule : 0;
}

hipError_t launch_sorted_f32(const float* ts, int64_t C, int64_t ld, int64_t Tn, const uint32_t* table,
                             const uint32_t* sflags, const DevSortedChunk* chunks, int32_t nchunks,
                             int32_t w, int32_t yps, int32_t ntracks, double q, int negate, double* thresh, double* seas,
                             int64_t ldo, hipStream_t stream, unsigned long long* stats) {
    const SortedEntry* e = w == 5 ? find_sorted(yps) : nullptr;
    if (!e || ld >= (int64_t(1) << 30)) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    dim3 grid(static_cast<unsigned>((C + 31) / 32), static_cast<unsigned>(nchunks));
    const bool twin = stats != nullptr && e->fn_stats != nullptr;
    hipLaunchKernelGGL(twin ? e->fn_stats : e->fn, grid, dim3(64), 0, stream, ts, C, ld, Tn, table, sflags, chunks, q,
                       negate, ntracks, thresh, seas, ldo, twin ? stats : nullptr);
    return hipGetLastError();
}

hipError_t launch_sorted_i16(const int16_t* codes, const PackedI16& pk, int64_t C, int64_t ld, int64_t Tn,
                             const uint32_t* table, const uint32_t* sflags, const DevSortedChunk* chunks, int32_t nchunks,
                             int32_t w, int32_t yps, int32_t ntracks, double q, int negate, double* thresh, double* seas,
                             int64_t ldo, hipStream_t stream) {
    const SortedEntry* e = w == 5 ? find_sorted(yps) : nullptr;
    if (!e || !e->fn_i16 || ld >= (int64_t(1) << 30) || pk.mode < 1 || pk.mode > 3) return hipErrorInvalidValue;
    if (C <= 0 || nchunks <= 0) return hipSuccess;
    dim3 grid(static_cast<unsigned>((C + 31) / 32), static_cast<unsigned>(nchunks));
    hipLaunchKernelGGL(e->fn_i16, grid, dim3(64), 0, stream, codes, pk, C, ld, Tn, table, sflags, chunks, q, negate, ntracks,
                       thresh, seas, ldo);
    return hipGetLastError();
}

}  // namespace xmhw
