// ring3_helpers.h -- device helpers shared by the third- and fourth-generation ring kernels (kernels_ring3.hip,
// kernels_ring4.hip): DPP exchanges inside a cell, order-preserving keys, the hand-scheduled count pass, the slow
// path's extraction list and the sorting networks across the lanes of a cell.
#pragma once
#include "device_common.h"

namespace xmhw {
namespace {

// waves per workgroup: the lanes of a workgroup cover ONE 128-byte line of a sample row (32 float32 cells, 16 float64
// ones): 64 / subs cells per wave
constexpr int waves3(int subs, int itemsize) { return 128 / ((64 / subs) * itemsize); }
constexpr uint32_t kInv3 = 0xFFFFFFFFu;

template <int CTRL>
__device__ __forceinline__ uint32_t dpp3(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
// partner = lane ^ 1, ^ 2, quad mirror (^ 3), mirror inside 8 lanes (^ 7); shifts inside the 16-lane row
constexpr int kX1 = 0xB1, kX2 = 0x4E, kX3 = 0x1B, kX7 = 0x141;
constexpr int kShr1 = 0x111, kShr2 = 0x112, kShr4 = 0x114;

template <int SUBS>
__device__ __forceinline__ uint32_t csum(uint32_t v) {
    v += dpp3<kX1>(v);
    if constexpr (SUBS >= 4) v += dpp3<kX2>(v);
    if constexpr (SUBS == 8) v += dpp3<kX7>(v);
    return v;
}
template <int CTRL>
__device__ __forceinline__ double dpp3_f64(double v) {
    const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
    const uint32_t lo = dpp3<CTRL>(static_cast<uint32_t>(b));
    const uint32_t hi = dpp3<CTRL>(static_cast<uint32_t>(b >> 32));
    return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(hi) << 32) | lo));
}
template <int SUBS>
__device__ __forceinline__ double csum(double v) {
    v += dpp3_f64<kX1>(v);
    if constexpr (SUBS >= 4) v += dpp3_f64<kX2>(v);
    if constexpr (SUBS == 8) v += dpp3_f64<kX7>(v);
    return v;
}
__device__ __forceinline__ uint32_t minu3(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t maxu3(uint32_t a, uint32_t b) { return a > b ? a : b; }
template <int SUBS>
__device__ __forceinline__ uint32_t cmax(uint32_t v) {
    v = maxu3(v, dpp3<kX1>(v));
    if constexpr (SUBS >= 4) v = maxu3(v, dpp3<kX2>(v));
    if constexpr (SUBS == 8) v = maxu3(v, dpp3<kX7>(v));
    return v;
}
template <int SUBS>
__device__ __forceinline__ uint32_t cmin(uint32_t v) {
    v = minu3(v, dpp3<kX1>(v));
    if constexpr (SUBS >= 4) v = minu3(v, dpp3<kX2>(v));
    if constexpr (SUBS == 8) v = minu3(v, dpp3<kX7>(v));
    return v;
}
__device__ __forceinline__ uint32_t med3u3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t ashr31_3(uint32_t v) {
    uint32_t r;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(v));
    return r;
}
// key of a non-NaN float: negmask = 0 (heat waves) or 0xFFFFFFFF (cold spells: key(-x) = ~key(x))
__device__ __forceinline__ uint32_t key_of_bits3(uint32_t b, uint32_t negmask) {
    return b ^ (static_cast<uint32_t>(static_cast<int32_t>(b) >> 31) | 0x80000000u) ^ negmask;
}
// the same in two instructions (v_ashrrev, v_bitop3: (sign | 0x80000000) ^ b, complemented for cold spells), for the
// plain rows where every sample is a number; NEG is a compile-time choice, the caller branches on the launch's negate
template <bool NEG>
__device__ __forceinline__ uint32_t key_of_bits3_fast(uint32_t b) {
    return static_cast<uint32_t>(__builtin_amdgcn_bitop3_b32(static_cast<int32_t>(ashr31_3(b)), static_cast<int32_t>(b),
                                                             static_cast<int32_t>(0x80000000u), NEG ? 0xC9 : 0x36));
}
__device__ __forceinline__ uint32_t bits_of_key3(uint32_t k) { return k ^ (~ashr31_3(k) | 0x80000000u); }
__device__ __forceinline__ double value_of_key3(uint32_t k) {   // 0 for an invalid key
    const float f = __uint_as_float(bits_of_key3(k));
    return k == kInv3 ? 0.0 : static_cast<double>(f);
}
// 64-bit mode (genuinely float64 samples): the order-preserving 64-bit key of a double as a HIGH word -- what the rings,
// the histogram and the whole selection work on, exactly as on a float32 key -- and a LOW word kept beside it
// (kernels_ring2.hip: key64_of)
__device__ __forceinline__ void key64_of3(double v, uint32_t negmask, uint32_t& hi, uint32_t& lo) {
    const uint64_t b = static_cast<uint64_t>(__double_as_longlong(v));
    const uint32_t bh = static_cast<uint32_t>(b >> 32), bl = static_cast<uint32_t>(b);
    const uint32_t sgn = ashr31_3(bh);                       // all ones for a negative sample
    hi = bh ^ (sgn | 0x80000000u) ^ negmask;
    lo = bl ^ sgn ^ negmask;
}
// the double a VALID key pair stands for (the negated sample under coldSpells)
__device__ __forceinline__ double double_of_key64_3(uint32_t hi, uint32_t lo) {
    const uint32_t sgn = ~ashr31_3(hi);                      // all ones if the value is negative
    const uint32_t bh = hi ^ (sgn | 0x80000000u), bl = lo ^ sgn;
    return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(bh) << 32) | bl));
}
__device__ __forceinline__ double value_of_key64_3(uint32_t hi, uint32_t lo) {      // 0 for an invalid key
    return hi == kInv3 ? 0.0 : double_of_key64_3(hi, lo);
}
__device__ __forceinline__ uint32_t opaque3(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}
// (a running float64 sum made opaque after every term: the rare paths that re-sum a whole ring otherwise have all their
// keys converted ahead of the first addition -- one register pair per key)
__device__ __forceinline__ double opaque3d(double v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ void ring_sel3(uint32_t& slot, uint32_t other, unsigned long long take_other) {
    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(slot) : "v"(other), "s"(take_other));
}

// c + #{r[i] <= p}, 11 keys, hand-scheduled (see kernels_ring2.hip: count_le11)
template <class RingT>
__device__ __forceinline__ uint32_t count_le11_3(const RingT& r, uint32_t p, uint32_t& c, uint32_t d) {
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

// Band compaction, 11 keys: every key k with (k - e0) < w (unsigned: e0 <= k < e0 + w) is appended to the
// list at LDS byte address p (p += 4).  The 11 compares are issued first (their lane masks go to SGPR
// pairs), then each mask becomes EXEC for one ds_write + v_add: 3 vector instructions, one scalar and one
// LDS instruction per key.  (Branching over the ds_write and the v_add where no lane of the wave has a band key
// at that ring position -- about half of the positions -- made the pass slower: 5,290 against 3,930 cycles per
// wave-row; tools/ubench_lds.hip: a taken s_cbranch behind an EXEC write costs as much as the masked write.  Taking
// the positions in PAIRS -- one v_cndmask + ONE masked ds_write per pair unless some lane matches at both -- halves
// the LDS instructions and was slower too: 5,920 cycles, kernel 71.5 against 65.0 ms; profiles/r3_compaction_variants.txt)
// EXEC is saved and restored (the call sites are wave-uniform).
template <class RingT>
__device__ __forceinline__ void compact11(const RingT& r, uint32_t e0, uint32_t w, uint32_t& p) {
    unsigned long long m0, m1, m2, m3, m4, m5, m6, m7, m8, m9, m10, sv;
    uint32_t t0, t1;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_sub_u32 %[t0], %[k0], %[e0]\n\t"
        "v_sub_u32 %[t1], %[k1], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m0], %[t0], %[w]\n\t"
        "v_sub_u32 %[t0], %[k2], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m1], %[t1], %[w]\n\t"
        "v_sub_u32 %[t1], %[k3], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m2], %[t0], %[w]\n\t"
        "v_sub_u32 %[t0], %[k4], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m3], %[t1], %[w]\n\t"
        "v_sub_u32 %[t1], %[k5], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m4], %[t0], %[w]\n\t"
        "v_sub_u32 %[t0], %[k6], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m5], %[t1], %[w]\n\t"
        "v_sub_u32 %[t1], %[k7], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m6], %[t0], %[w]\n\t"
        "v_sub_u32 %[t0], %[k8], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m7], %[t1], %[w]\n\t"
        "v_sub_u32 %[t1], %[k9], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m8], %[t0], %[w]\n\t"
        "v_sub_u32 %[t0], %[k10], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m9], %[t1], %[w]\n\t"
        "v_cmp_lt_u32_e64 %[m10], %[t0], %[w]\n\t"
        "s_and_b64 exec, %[sv], %[m0]\n\t"
        "ds_write_b32 %[p], %[k0]\n\t"
        "v_add_u32 %[p], 4, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m1]\n\t"
        "ds_write_b32 %[p], %[k1]\n\t"
        "v_add_u32 %[p], 4, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m2]\n\t"
        "ds_write_b32 %[p], %[k2]\n\t"
        "v_add_u32 %[p], 4, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m3]\n\t"
        "ds_write_b32 %[p], %[k3]\n\t"
        "v_add_u32 %[p], 4, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m4]\n\t"
        "ds_write_b32 %[p], %[k4]\n\t"
        "v_add_u32 %[p], 4, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m5]\n\t"
        "ds_write_b32 %[p], %[k5]\n\t"
        "v_add_u32 %[p], 4, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m6]\n\t"
        "ds_write_b32 %[p], %[k6]\n\t"
        "v_add_u32 %[p], 4, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m7]\n\t"
        "ds_write_b32 %[p], %[k7]\n\t"
        "v_add_u32 %[p], 4, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m8]\n\t"
        "ds_write_b32 %[p], %[k8]\n\t"
        "v_add_u32 %[p], 4, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m9]\n\t"
        "ds_write_b32 %[p], %[k9]\n\t"
        "v_add_u32 %[p], 4, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m10]\n\t"
        "ds_write_b32 %[p], %[k10]\n\t"
        "v_add_u32 %[p], 4, %[p]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [p] "+v"(p), [t0] "=&v"(t0), [t1] "=&v"(t1), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2),
          [m3] "=&s"(m3), [m4] "=&s"(m4), [m5] "=&s"(m5), [m6] "=&s"(m6), [m7] "=&s"(m7), [m8] "=&s"(m8),
          [m9] "=&s"(m9), [m10] "=&s"(m10), [sv] "=&s"(sv)
        : [k0] "v"(r[0]), [k1] "v"(r[1]), [k2] "v"(r[2]), [k3] "v"(r[3]), [k4] "v"(r[4]), [k5] "v"(r[5]),
          [k6] "v"(r[6]), [k7] "v"(r[7]), [k8] "v"(r[8]), [k9] "v"(r[9]), [k10] "v"(r[10]), [e0] "v"(e0),
          [w] "v"(w)
        : "memory", "scc");
}

// 64-bit mode: compact11 with the LOW word of every band key written beside its high word (ds_write2_b32: list entries
// are pairs, p += 8), so that the low words of the order statistics can be read from the lists instead of being
// fetched by another pass over the rings.  In two halves (an asm statement takes 30 operands).
template <int K0, int N, class RingT>
__device__ __forceinline__ void compact_half_x(const RingT& r, const RingT& rl, uint32_t e0, uint32_t w, uint32_t& p) {
    static_assert(N == 5 || N == 6, "halves of 11");
    unsigned long long m0, m1, m2, m3, m4, m5, sv;
    uint32_t t0, t1;
    constexpr int K5 = N == 6 ? K0 + 5 : K0 + 4;      // (a sixth key of the second half: its mask is forced to 0)
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_sub_u32 %[t0], %[k0], %[e0]\n\t"
        "v_sub_u32 %[t1], %[k1], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m0], %[t0], %[w]\n\t"
        "v_sub_u32 %[t0], %[k2], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m1], %[t1], %[w]\n\t"
        "v_sub_u32 %[t1], %[k3], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m2], %[t0], %[w]\n\t"
        "v_sub_u32 %[t0], %[k4], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m3], %[t1], %[w]\n\t"
        "v_sub_u32 %[t1], %[k5], %[e0]\n\t"
        "v_cmp_lt_u32_e64 %[m4], %[t0], %[w]\n\t"
        "v_cmp_lt_u32_e64 %[m5], %[t1], %[w]\n\t"
        "s_and_b64 %[m5], %[m5], %[last]\n\t"
        "s_and_b64 exec, %[sv], %[m0]\n\t"
        "ds_write2_b32 %[p], %[k0], %[l0] offset1:1\n\t"
        "v_add_u32 %[p], 8, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m1]\n\t"
        "ds_write2_b32 %[p], %[k1], %[l1] offset1:1\n\t"
        "v_add_u32 %[p], 8, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m2]\n\t"
        "ds_write2_b32 %[p], %[k2], %[l2] offset1:1\n\t"
        "v_add_u32 %[p], 8, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m3]\n\t"
        "ds_write2_b32 %[p], %[k3], %[l3] offset1:1\n\t"
        "v_add_u32 %[p], 8, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m4]\n\t"
        "ds_write2_b32 %[p], %[k4], %[l4] offset1:1\n\t"
        "v_add_u32 %[p], 8, %[p]\n\t"
        "s_and_b64 exec, %[sv], %[m5]\n\t"
        "ds_write2_b32 %[p], %[k5], %[l5] offset1:1\n\t"
        "v_add_u32 %[p], 8, %[p]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [p] "+v"(p), [t0] "=&v"(t0), [t1] "=&v"(t1), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2),
          [m3] "=&s"(m3), [m4] "=&s"(m4), [m5] "=&s"(m5), [sv] "=&s"(sv)
        : [k0] "v"(r[K0]), [k1] "v"(r[K0 + 1]), [k2] "v"(r[K0 + 2]), [k3] "v"(r[K0 + 3]), [k4] "v"(r[K0 + 4]),
          [k5] "v"(r[K5]), [l0] "v"(rl[K0]), [l1] "v"(rl[K0 + 1]), [l2] "v"(rl[K0 + 2]), [l3] "v"(rl[K0 + 3]),
          [l4] "v"(rl[K0 + 4]), [l5] "v"(rl[K5]), [e0] "v"(e0), [w] "v"(w),
          [last] "s"(N == 6 ? ~0ull : 0ull)
        : "memory", "scc");
}
template <class RingT>
__device__ __forceinline__ void compact11x(const RingT& r, const RingT& rl, uint32_t e0, uint32_t w, uint32_t& p) {
    compact_half_x<0, 6>(r, rl, e0, w, p);
    compact_half_x<6, 5>(r, rl, e0, w, p);
}

// ---- slow path: the round-2 extraction list (kernels_ring2.hip: Top2), on the adjacent lane layout ----
template <int J, int JM>
struct Top3 {
    static_assert(JM >= J && JM <= 8, "merged width: J..8");
    uint32_t m[JM];
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int i = 0; i < JM; ++i) m[i] = 0xFFFFFFFFu;
    }
    __device__ __forceinline__ void insert(uint32_t d) {
#pragma unroll
        for (int i = J - 1; i >= 1; --i) m[i] = med3u3(m[i - 1], m[i], d);
        m[0] = minu3(m[0], d);
    }
    template <int SUBS>
    __device__ __forceinline__ uint32_t horizon() const { return cmin<SUBS>(m[J - 1]); }
    template <int CTRL, bool FIRST = false>
    __device__ __forceinline__ void merge() {
        constexpr int N = FIRST ? J : JM;
        uint32_t b[N];
#pragma unroll
        for (int i = 0; i < N; ++i) b[i] = dpp3<CTRL>(m[i]);
#pragma unroll
        for (int i = 0; i < JM; ++i) {
            const int k = JM - 1 - i;
            if (i < N && k < N) m[i] = minu3(m[i], b[k]);
            else if (k < N) m[i] = b[k];
        }
        constexpr int OFF = 8 - JM;
#pragma unroll
        for (int d = 4; d >= 1; d >>= 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if ((k & d) == 0 && k >= OFF && k + d < 8) {
                    const uint32_t lo_ = minu3(m[k - OFF], m[k - OFF + d]);
                    const uint32_t hi_ = maxu3(m[k - OFF], m[k - OFF + d]);
                    m[k - OFF] = lo_;
                    m[k - OFF + d] = hi_;
                }
            }
        }
    }
    template <int SUBS>
    __device__ __forceinline__ void merge_cell() {
        merge<kX1, true>();
        if constexpr (SUBS >= 4) merge<kX2>();
        if constexpr (SUBS == 8) merge<kX7>();
    }
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
    __device__ __forceinline__ uint32_t count_below(uint32_t d) const {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < JM; ++i) c += (m[i] < d) ? 1u : 0u;
        return c;
    }
};

// ---- fast path: sort the CAP x SUBS slots of a cell ascending in lane-major order ----------------
// in-lane: optimal networks (5 comparators for 4, 19 for 8); across lanes: bitonic merges whose first
// step pairs slot r of a lane with slot CAP-1-r of its mirror lane, so that every comparator is
// ascending; the lower lane keeps the minimum (v_med3 with a bound of 0), the upper one the maximum
// (bound 0xFFFFFFFF)
__device__ __forceinline__ void cswap(uint32_t& a, uint32_t& b) {
    const uint32_t lo = minu3(a, b), hi = maxu3(a, b);
    a = lo;
    b = hi;
}
template <int CAP>
__device__ __forceinline__ void sort_lane(uint32_t (&c)[CAP]) {
    if constexpr (CAP == 4) {
        cswap(c[0], c[1]); cswap(c[2], c[3]); cswap(c[0], c[2]); cswap(c[1], c[3]); cswap(c[1], c[2]);
    } else {
        static_assert(CAP == 8, "4 or 8 slots per lane");
        cswap(c[0], c[1]); cswap(c[2], c[3]); cswap(c[4], c[5]); cswap(c[6], c[7]);
        cswap(c[0], c[2]); cswap(c[1], c[3]); cswap(c[4], c[6]); cswap(c[5], c[7]);
        cswap(c[1], c[2]); cswap(c[5], c[6]); cswap(c[0], c[4]); cswap(c[3], c[7]);
        cswap(c[1], c[5]); cswap(c[2], c[6]);
        cswap(c[1], c[4]); cswap(c[3], c[6]);
        cswap(c[2], c[4]); cswap(c[3], c[5]);
        cswap(c[3], c[4]);
    }
}
template <int CAP>
__device__ __forceinline__ void clean_lane(uint32_t (&c)[CAP]) {      // bitonic -> sorted, in-lane
#pragma unroll
    for (int d = CAP / 2; d >= 1; d >>= 1)
#pragma unroll
        for (int r = 0; r < CAP; ++r)
            if ((r & d) == 0) cswap(c[r], c[r + d]);
}
template <int CAP, int CTRL, bool MIRROR>
__device__ __forceinline__ void cross_step(uint32_t (&c)[CAP], uint32_t bound) {
    uint32_t b[CAP];
#pragma unroll
    for (int r = 0; r < CAP; ++r) b[r] = dpp3<CTRL>(c[MIRROR ? CAP - 1 - r : r]);
#pragma unroll
    for (int r = 0; r < CAP; ++r) c[r] = med3u3(c[r], b[r], bound);
}
template <int SUBS, int CAP>
__device__ __forceinline__ void sort_cell(uint32_t (&c)[CAP], uint32_t bnd1, uint32_t bnd2, uint32_t bnd4) {
    sort_lane<CAP>(c);
    cross_step<CAP, kX1, true>(c, bnd1);                   // groups of 2 lanes
    clean_lane<CAP>(c);
    if constexpr (SUBS >= 4) {
        cross_step<CAP, kX3, true>(c, bnd2);               // groups of 4 lanes
        cross_step<CAP, kX1, false>(c, bnd1);
        clean_lane<CAP>(c);
    }
    if constexpr (SUBS == 8) {
        cross_step<CAP, kX7, true>(c, bnd4);               // groups of 8 lanes
        cross_step<CAP, kX2, false>(c, bnd2);
        cross_step<CAP, kX1, false>(c, bnd1);
        clean_lane<CAP>(c);
    }
}
template <int CAP>
__device__ __forceinline__ uint32_t pick_reg(const uint32_t (&c)[CAP], uint32_t r) {
    uint32_t v = c[0];
#pragma unroll
    for (int i = 1; i < CAP; ++i) {
        v = (r == static_cast<uint32_t>(i)) ? c[i] : v;
        asm volatile("" : "+v"(v));
    }
    return v;
}

constexpr int kBudget3 = 6;

}  // namespace
}  // namespace xmhw
