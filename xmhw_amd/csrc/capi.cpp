// capi.cpp -- the C ABI declared in include/xmhw_amd.h.
#include "../../include/xmhw_amd.h"

#include <hip/hip_runtime.h>
#include <unistd.h>

#include <cerrno>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "kernels.h"
#include "plan.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    // out of device memory is reported as such wherever it happens (callers that cache device
    // buffers release them and retry on XMHW_ERR_NOMEM)
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        return fail(XMHW_ERR_NOMEM, std::string(what) + ": out of device memory");
    }
    return fail(XMHW_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace
// shared with comm.cpp
int xmhw_set_error_(int code, const std::string& msg) { return fail(code, msg); }
namespace {
#define HIP_TRY(expr)                                        \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return hip_fail(_e, #expr);    \
    } while (0)

constexpr int kSubs = 8;
// the sorted-list kernel keeps the K largest keys of a row-list, K sized for the top tenth of the pool (four times a list's
// average share): quantiles from here up -- and, mirrored (the K smallest keys: round 6), from 1 - that down.  In between a
// list's share outgrows K and too many cell-rows would be recomputed: those calls run on the ring layout.
constexpr double kSortedMinQ = 0.85;
inline bool sorted_serves(double q) { return q >= kSortedMinQ || q <= 1.0 - kSortedMinQ; }

}  // namespace

struct xmhw_plan {
    xmhw::Plan host;
    // device state (lazy, per current device at first use)
    std::mutex mu;
    bool uploaded = false;
    int32_t yps = 0;          // ring kernel years-per-lane (0: ring not available)
    int32_t subs = 0;         // ... and its lanes per cell (8, or 16 for records of 49..96 tracks)
    int32_t nchunks = 0;
    uint32_t* d_table = nullptr;
    int32_t yps2 = 0;         // second-generation float32 ring kernel (kernels_ring2.hip), 0: not available
    int32_t subs2 = 0;        // ... and its lanes per cell (8, or 4 for variant 7)
    int32_t ring2_variant = -2;   // -2: auto (0 or 7, whichever pads fewer tracks); -1: off (round-1 kernel); 0..7: see kernels_ring2.hip
    uint32_t* d_table2 = nullptr;
    uint32_t* d_sflags = nullptr;
    int32_t yps64 = 0;        // 64-bit mode on 16 lanes per cell, short records: tracks per lane (1..3), 0: none
    uint32_t* d_table64 = nullptr;   // ... and its step table
    xmhw::DevChunk* d_chunks = nullptr;
    int32_t* d_row_ptr = nullptr;
    int32_t* d_centres = nullptr;
    unsigned long long* d_stats = nullptr;  // debug: ring kernel pass counters (xmhw_plan_debug_stats)
    uint32_t* d_tablex = nullptr;           // the 64-bit mode's own table (8 or 4 lanes) when the float32 layout is another one
    int32_t ypsx = 0, subsx = 0;
    uint32_t* d_narrow_flag = nullptr;      // float64 input: set when a sample is not float32-representable
    bool narrowing = true;                  // xmhw_plan_set_narrowing
    // sorted-list kernel (kernels_sorted.hip): its table (2 lanes per cell), the chunks of the regular rows it serves,
    // the chunks of the other rows (they stay on the ring kernel), the bitmap of cell-rows it hands to the generic kernel
    int32_t yps_s = 0;
    uint32_t* d_table_s = nullptr;
    xmhw::DevSortedChunk* d_chunks_s = nullptr;
    uint32_t* d_sflags_s = nullptr;
    xmhw::DevChunk* d_chunks_i = nullptr;
    int32_t nchunks_s = 0, nchunks_i = 0;
    int64_t sorted_pieces = -1;             // the number of pieces the sorted chunks were cut into
    int32_t sorted_built = 0;               // chunks in d_chunks_s (nchunks_s = that, or 0 while the plan is not sorted-usable)
    // optional timing of the main kernel of every raw-climatology call (xmhw_plan_set_timing): a ring of event pairs
    bool timing = false;
    hipEvent_t tev[32] = {};
    uint64_t tcalls = 0;

    ~xmhw_plan() {
        if (d_stats) (void)hipFree(d_stats);
        for (hipEvent_t e : tev) if (e) (void)hipEventDestroy(e);
        if (d_table_s) (void)hipFree(d_table_s);
        if (d_chunks_s) (void)hipFree(d_chunks_s);
        if (d_sflags_s) (void)hipFree(d_sflags_s);
        if (d_chunks_i) (void)hipFree(d_chunks_i);
        if (d_narrow_flag) (void)hipFree(d_narrow_flag);
        if (d_tablex) (void)hipFree(d_tablex);
        if (d_table) (void)hipFree(d_table);
        if (d_table64) (void)hipFree(d_table64);
        if (d_table2) (void)hipFree(d_table2);
        if (d_sflags) (void)hipFree(d_sflags);
        if (d_chunks) (void)hipFree(d_chunks);
        if (d_row_ptr) (void)hipFree(d_row_ptr);
        if (d_centres) (void)hipFree(d_centres);
    }
};

namespace {

int32_t ring2_resolved(const xmhw_plan* p);
int32_t ring2_legacy(const xmhw_plan* p);

// float64 samples on the second-generation kernel's 64-bit mode: which layout, if any.  The float32 layout of
// the plan (8 or 4 lanes per cell) as long as a lane holds at most 4 tracks (44 keys and their low words fit the
// registers); otherwise, and for short records, 16 lanes per cell (variant 12, the table of the 16-lane rings).
struct X64Choice { int32_t variant = -1, yps = 0; };
X64Choice x64_choice(const xmhw_plan* p) {
    static const bool on = [] { const char* v = std::getenv("XMHW_RING2_F64"); return !(v && v[0] == '0'); }();
    X64Choice c;
    if (!on || p->ring2_variant == -1) return c;
    // 8 lanes per cell: low words in registers up to 4 tracks per lane (9..32 tracks), in LDS at 5 and 6 (33..48) ...
    const int32_t y8 = xmhw::ring2_pick_yps(p->host.w, p->host.ntracks, 8);
    static const bool lds_on = [] { const char* v = std::getenv("XMHW_RING2_F64_LDS"); return !(v && v[0] == '0'); }();
    // (the third-generation kernel's 64-bit mode where both rings fit its registers: up to 5 tracks per lane = 9..40
    // tracks, layout 20; XMHW_RING3_F64=0 keeps the second-generation kernel)
    static const bool r3_on = [] { const char* v = std::getenv("XMHW_RING3_F64"); return !(v && v[0] == '0'); }();
    // (13..20 tracks: the 4-lane layout of the float32 path, 16 cells per wave, on the plan's own table;
    // XMHW_RING3_F64_LANES=8 keeps the 8-lane layout)
    static const bool r3_4 = [] { const char* v = std::getenv("XMHW_RING3_F64_LANES"); return !(v && v[0] == '8'); }();
    if (r3_on && r3_4) {
        const int32_t y4 = xmhw::ring3_pick_yps(p->host.w, p->host.ntracks, 4);
        if (y4 > 0 && xmhw::ring3_x64_supported(p->host.w, y4, 4)) {
            c.variant = 21;
            c.yps = y4;
            return c;
        }
    }
    if (r3_on && y8 > 0 && xmhw::ring3_pick_yps(p->host.w, p->host.ntracks, 8) == y8 &&
        xmhw::ring3_x64_supported(p->host.w, y8, 8)) {
        c.variant = 20;
        c.yps = y8;
        return c;
    }
    if (y8 > 0 && (y8 <= 4 || lds_on) && xmhw::ring2_x64_supported(p->host.w, y8, 8)) {
        c.variant = 8;
        c.yps = y8;
        return c;
    }
    // ... 16 lanes per cell for longer and for very short records (the table of the 16-lane rings)
    const int32_t y16 = xmhw::ring2_pick_yps(p->host.w, p->host.ntracks, 12);
    // (its table: the plan's own ring2 table where the float32 layout is the same 16 lanes, the 16-lane table kept for
    // the 64-bit mode otherwise)
    if (y16 > 0 && y16 <= 6 && xmhw::ring2_x64_supported(p->host.w, y16, 12)) {
        c.variant = 12;
        c.yps = y16;
    }
    return c;
}
bool x64_usable(const xmhw_plan* p) { return x64_choice(p).variant >= 0; }

int32_t resolve_kernel(const xmhw_plan* p, int elem_bytes) {
    if (p->host.kernel_choice == XMHW_KERNEL_GENERIC) return XMHW_KERNEL_GENERIC;
    if (elem_bytes == 8) {
        // float64: the second-generation kernel's 64-bit mode where it is instantiated (w = 5, up to 96 tracks),
        // the generic kernel otherwise.  The round-1 float64 ring (kernels_ring64.hip) is gone: round 2's
        // randomised cross-check found it returning wrong rows on clustered doubles and it was never repaired;
        // an explicit XMHW_KERNEL_RING request on a plan the 64-bit mode does not cover is refused
        // (XMHW_ERR_UNSUPPORTED) instead of being served by a kernel known to be wrong.
        const bool x64 = x64_usable(p);
        if (p->host.kernel_choice == XMHW_KERNEL_RING) return x64 ? XMHW_KERNEL_RING : -1;
        return x64 ? XMHW_KERNEL_RING : XMHW_KERNEL_GENERIC;
    }
    const int32_t yps = xmhw::ring_pick(p->host.w, p->host.ntracks, elem_bytes, nullptr);
    if (p->host.kernel_choice == XMHW_KERNEL_RING) return yps ? XMHW_KERNEL_RING : -1;
    return yps ? XMHW_KERNEL_RING : XMHW_KERNEL_GENERIC;
}

int32_t auto_chunks(const xmhw_plan* p, int64_t C) {
    if (p->host.nchunks_req > 0) return std::min(p->host.nchunks_req, p->host.D);
    // enough waves to fill 256 CUs x 16 waves a few times over; each chunk
    // re-reads 2w rows per track and cold-starts its bracket, so keep them long
    int64_t waves = (C + 7) / 8;
    int64_t want = (4 * 4096 + waves - 1) / std::max<int64_t>(waves, 1);
    {
        // the third-generation kernel runs two waves per SIMD (2,048 at a time) of 16 or 8 cells: twice that many
        // waves in all is enough, and every further chunk costs its warm-up rows (1 degree grid, 64,800 cells: 3.67 ms
        // with 1 or 2 chunks, 3.87 with 3, 4.16 with 6).
        // The model behind it (round 4): a workgroup is 2 waves, a CU holds 4, the chip 1,024 at a time.  With n chunks
        // a grid is W = n * ceil(C / 32) workgroups of (D / n + 2w) rows each and runs in about
        // ceil(W / 1024) * (D / n + 2w) row-times.  The 1 degree grid (2,025 workgroups per chunk, D = 366): n = 1
        // -> 2 rounds x 376 = 752; n = 2 -> 4 x 193 = 772; n = 3 -> 6 x 132 = 792; n = 6 -> 12 x 71 = 852 --
        // the measured order.  One or two chunks fill the last round to 99 %: there is no tail to remove, and what
        // keeps this grid at 11 % of the roofline against 15 % for the 40-year one is the record, not the grid:
        // 30 tracks pad to 32 (6 % idle ring slots) and the per-row costs that do not depend on the number of
        // tracks (walk, sort, epilogue, row overhead: ~40 % of a row) are spread over 120 bytes of samples per
        // cell-row instead of 160.
        const int32_t v = ring2_resolved(p);
        if (v >= 20) {
            const int64_t cpw = 64 / xmhw::ring2_subs(v);
            waves = (C + cpw - 1) / cpw;
            want = (4096 + waves - 1) / std::max<int64_t>(waves, 1);
        }
    }
    want = std::max<int64_t>(1, std::min<int64_t>(want, p->host.D / 24));
    return static_cast<int32_t>(std::max<int64_t>(want, 1));
}

// the ring2 variant float32 input runs on: the requested one, or (auto) 8 lanes per cell unless the
// 4-lane layout pads fewer tracks (20 tracks: 4 x 5 exactly against 8 x 3 = 24) and does not spill;
// both with the lanes' lists merged into a wider window (variants 8 and 10: measured 3-4 % faster than
// the plain 0 and 7)
int32_t ring2_legacy(const xmhw_plan* p) {
    const int32_t y8 = xmhw::ring2_pick_yps(p->host.w, p->host.ntracks, 8);
    const int32_t y4 = xmhw::ring2_pick_yps(p->host.w, p->host.ntracks, 10);
    if (y4 && y4 <= 8 && (!y8 || y4 * 4 < y8 * 8)) return 10;
    if (!y8 && xmhw::ring2_pick_yps(p->host.w, p->host.ntracks, 12) >= 4) return 12;
    return 8;
}

int32_t ring2_resolved(const xmhw_plan* p) {
    if (p->ring2_variant != -2 && p->ring2_variant != XMHW_LAYOUT_SORTED) return p->ring2_variant;
    // the third-generation kernel (kernels_ring3.hip) on 4 lanes per cell where a lane holds at least 4 tracks
    // (w = 5, 13..48 tracks).  1,036,800 cells, daily (tools/bench_ring2.py --years, counters on): 40 tracks 57.6 ms
    // against 80 ms for the second-generation layouts, 24 tracks 41.8 against 59.8, 20 tracks 38.3 against 44.0,
    // 16 tracks 34.5 against 35.3; 12 tracks 31.0 against 29.7 -- with so few keys per lane its per-row overheads
    // (histogram, walk, sort) outweigh the cheaper selection.  The 6-hourly share of configs[4] (20 tracks): 116 against
    // 130 ms.
    // ... on 2 lanes per cell (32 cells per wave) for records of 9..24 tracks: 518,400 cells daily, counters on: 13
    // tracks 12.4 against 17.7 ms on 4 lanes, 18 tracks 14.6 / 19.5, 22 tracks 17.5 / 21.3, 24 tracks 20.4 / 21.1; the
    // 6-hourly share of configs[4] (20 tracks, 405,000 cells): 48.1 / 58.9
    if (xmhw::ring3_pick_yps(p->host.w, p->host.ntracks, 2) >= 5) return 22;
    if (xmhw::ring3_pick_yps(p->host.w, p->host.ntracks, 4) >= 4) return 21;
    // ... and on 8 lanes per cell for longer records (49..88 tracks, 7..11 per lane) instead of the second-generation
    // kernel's 16-lane layout: 259,200 cells daily, counters on: 50 tracks 24.9 against 36.6 ms, 65 tracks 31.6 / 45.1,
    // 85 tracks 42.8 / 51.0; 96 tracks (12 per lane, 256 registers) 53.5 / 52.0 -- those stay where they were
    {
        const int32_t y8 = xmhw::ring3_pick_yps(p->host.w, p->host.ntracks, 8);
        if (y8 >= 7 && y8 <= 11) return 20;
    }
    return ring2_legacy(p);
}

// float64 input that is really float32 (decoded archives): which ring2 variant narrows it, and on which of the
// plan's tables.  The third-generation kernel narrows on the layouts the automatic choice uses (4..12 tracks per lane
// at 4 lanes per cell); any other plan whose float32 layout is variant 20 / 21 narrows on the second-generation
// kernel: its table is the plan's own when the lane layout is the same (4 lanes: variant 10), the 64-bit mode's
// 8-lane table otherwise.
struct NarrowChoice { int32_t variant = -1, yps = 0; const uint32_t* table = nullptr; };
NarrowChoice narrow_choice(const xmhw_plan* p) {
    NarrowChoice c;
    int32_t v = ring2_resolved(p);
    if (v < 0) return c;
    if (v >= 20 && p->yps2 && p->subs2 == xmhw::ring2_subs(v) &&
        xmhw::ring2_narrowing_supported(p->host.w, p->yps2, v)) {
        c.variant = v;
        c.yps = p->yps2;
        c.table = p->d_table2;
        return c;
    }
    if (v >= 20) v = ring2_legacy(p);
    const int32_t subs = xmhw::ring2_subs(v);
    const int32_t yps = xmhw::ring2_pick_yps(p->host.w, p->host.ntracks, v);
    if (!yps || !xmhw::ring2_narrowing_supported(p->host.w, yps, v)) return c;
    const uint32_t* t = nullptr;
    if (p->subs2 == subs && p->yps2 == yps) t = p->d_table2;
    else if (p->subsx == subs && p->ypsx == yps) t = p->d_tablex;
    if (!t) return c;
    c.variant = v;
    c.yps = yps;
    c.table = t;
    return c;
}

// The sorted-list kernel serves float32 input of plans with w = 5 whose record it is instantiated for, under the
// automatic layout choice or XMHW_LAYOUT_SORTED (environment XMHW_SORTED=0 turns it off); the rows it cannot serve
// (plan.h: sorted_segments) need the ring kernel.
bool sorted_usable(const xmhw_plan* p) {
    static const bool on = [] { const char* v = std::getenv("XMHW_SORTED"); return !(v && v[0] == '0'); }();
    if (!on && p->ring2_variant != XMHW_LAYOUT_SORTED) return false;
    if (p->ring2_variant != -2 && p->ring2_variant != XMHW_LAYOUT_SORTED) return false;
    if (p->host.kernel_choice == XMHW_KERNEL_GENERIC) return false;
    if (xmhw::sorted_pick_yps(p->host.w, p->host.ntracks) == 0) return false;
    return resolve_kernel(p, 4) == XMHW_KERNEL_RING;
}

// The sorted-list kernel's lists rely on LDS reads outside the workgroup's allocation returning 0 (kernels_sorted.hip).
// Checked once per device of this process before the kernel is used there; a device that answers otherwise keeps the ring
// kernels.  -1 = not probed yet, 1 = holds, 0 = does not.
int sorted_device_ok() {
    static std::mutex mu;
    static int state[64];
    static bool init = false;
    std::lock_guard<std::mutex> lock(mu);
    if (!init) { for (int& v : state) v = -1; init = true; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (state[dev] >= 0) return state[dev];
    uint32_t* d_bad = nullptr;
    uint32_t bad = 1;
    if (hipMalloc(&d_bad, sizeof(uint32_t)) != hipSuccess) return 0;
    const bool ran = xmhw::sorted_lds_probe(d_bad, nullptr) == hipSuccess &&
                     hipMemcpy(&bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d_bad);
    if (!ran) return 0;          // (not cached: a transient failure is probed again)
    state[dev] = bad == 0 ? 1 : 0;
    return state[dev];
}

// tables and chunks of the sorted-list kernel (under the plan's lock)
int upload_sorted(xmhw_plan* p, int64_t C) {
    const xmhw::Plan& h = p->host;
    if (!sorted_usable(p) || sorted_device_ok() != 1) { p->nchunks_s = 0; return XMHW_OK; }
    const int32_t yps = xmhw::sorted_pick_yps(h.w, h.ntracks);
    const int64_t waves = (C + 31) / 32;
    // the kernel's own chunks and table rows (plan.h: sorted_plan); a small grid is cut into more pieces so that it still
    // fills the chip (7 waves per CU; every piece pays R - 1 warm-up rows).  The tables depend on (tracks per lane, pieces)
    // only: the slabs of one threshold() call -- a different cell count each -- share them.
    const int64_t pieces = h.nchunks_req > 0 ? h.nchunks_req : (1536 + waves - 1) / std::max<int64_t>(waves, 1);
    if (!p->d_table_s || p->yps_s != yps || p->sorted_pieces != pieces) {
        HIP_TRY(hipDeviceSynchronize());
        for (void** q : {reinterpret_cast<void**>(&p->d_table_s), reinterpret_cast<void**>(&p->d_sflags_s),
                         reinterpret_cast<void**>(&p->d_chunks_s), reinterpret_cast<void**>(&p->d_chunks_i)})
            if (*q) { HIP_TRY(hipFree(*q)); *q = nullptr; }
        const xmhw::Plan::SortedPlan sp = h.sorted_plan(2 * yps, 24, pieces);
        p->sorted_built = 0;
        p->nchunks_i = 0;
        if (!sp.chunks.empty()) {
            // longest chunk first: workgroups start in the order of their index, blockIdx.y = the chunk, so the launch ends
            // with the short chunks (a 40-year daily plan: rows [60, 366), then [0, 59), then the Feb-29 row) instead of a
            // tail of 316-row waves
            std::vector<xmhw::DevSortedChunk> cs(sp.chunks.size());
            for (size_t i = 0; i < cs.size(); ++i) cs[i] = {sp.chunks[i].warm_start, sp.chunks[i].begin, sp.chunks[i].end, sp.chunks[i].trow0};
            std::stable_sort(cs.begin(), cs.end(), [](const xmhw::DevSortedChunk& a, const xmhw::DevSortedChunk& b) {
                return a.end - a.warm_start > b.end - b.warm_start;
            });
            HIP_TRY(hipMalloc(&p->d_table_s, sizeof(uint32_t) * sp.table.size()));
            HIP_TRY(hipMemcpy(p->d_table_s, sp.table.data(), sizeof(uint32_t) * sp.table.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc(&p->d_sflags_s, sizeof(uint32_t) * sp.flags.size()));
            HIP_TRY(hipMemcpy(p->d_sflags_s, sp.flags.data(), sizeof(uint32_t) * sp.flags.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc(&p->d_chunks_s, sizeof(xmhw::DevSortedChunk) * cs.size()));
            HIP_TRY(hipMemcpy(p->d_chunks_s, cs.data(), sizeof(xmhw::DevSortedChunk) * cs.size(), hipMemcpyHostToDevice));
            p->sorted_built = static_cast<int32_t>(cs.size());
        }
        p->yps_s = yps;
        p->sorted_pieces = pieces;
    }
    p->nchunks_s = p->sorted_built;
    return XMHW_OK;
}

int upload(xmhw_plan* p, int64_t C) {
    std::lock_guard<std::mutex> lock(p->mu);
    {
        const int rc = upload_sorted(p, C);
        if (rc != XMHW_OK) return rc;
    }
    const int32_t nchunks = auto_chunks(p, C);
    const X64Choice xc0 = x64_choice(p);
    const int32_t xsubs0 = xc0.variant == 21 ? 4 : 8;
    const bool need_x = (xc0.variant == 8 || xc0.variant == 20 || xc0.variant == 21) &&
                        !(p->subs2 == xsubs0 && p->yps2 == xc0.yps);
    if (p->uploaded && nchunks == p->nchunks &&
        p->subs2 == xmhw::ring2_subs(ring2_resolved(p)) &&
        p->yps2 == (ring2_resolved(p) >= 0 ? xmhw::ring2_pick_yps(p->host.w, p->host.ntracks, ring2_resolved(p)) : 0) &&
        (!need_x || (p->ypsx == xc0.yps && p->subsx == xsubs0)))
        return XMHW_OK;
    const xmhw::Plan& h = p->host;
    if (!p->uploaded) {
        // every table is allocated at most once: a failed upload can be retried without leaking
        auto put = [](auto** dst, const auto& v) -> hipError_t {
            using E = typename std::remove_reference<decltype(v)>::type::value_type;
            if (*dst == nullptr) {
                hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), sizeof(E) * v.size());
                if (e != hipSuccess) { *dst = nullptr; return e; }
            }
            return hipMemcpy(*dst, v.data(), sizeof(E) * v.size(), hipMemcpyHostToDevice);
        };
        HIP_TRY(put(&p->d_row_ptr, h.row_ptr));
        HIP_TRY(put(&p->d_centres, h.centres));
        p->yps = xmhw::ring_pick(h.w, h.ntracks, 4, &p->subs);
        if (p->yps) HIP_TRY(put(&p->d_table, h.ring_table(p->subs, p->yps)));
        HIP_TRY(put(&p->d_sflags, h.step_flags()));
        {
            const int32_t y16 = xmhw::ring2_pick_yps(h.w, h.ntracks, 12);
            p->yps64 = (y16 >= 1 && y16 <= 6) ? y16 : 0;      // (also for 49..96 tracks: their float32 layout may be another)
        }
        if (p->yps64) HIP_TRY(put(&p->d_table64, h.ring_table(16, p->yps64)));
    }
    // the second-generation ring kernel's table depends on the variant's lanes per cell
    {
        const int32_t v2 = ring2_resolved(p);
        const int32_t subs2 = xmhw::ring2_subs(v2);
        const int32_t yps2 = v2 >= 0 ? xmhw::ring2_pick_yps(h.w, h.ntracks, v2) : 0;
        if (yps2 != p->yps2 || subs2 != p->subs2) {
            if (p->d_table2) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(p->d_table2)); p->d_table2 = nullptr; }
            p->yps2 = yps2;
            p->subs2 = subs2;
            if (yps2) {
                const std::vector<uint32_t> t2 = h.ring_table(subs2, yps2);
                HIP_TRY(hipMalloc(&p->d_table2, sizeof(uint32_t) * t2.size()));
                HIP_TRY(hipMemcpy(p->d_table2, t2.data(), sizeof(uint32_t) * t2.size(), hipMemcpyHostToDevice));
            }
        }
    }
    {
        // the 64-bit mode's own table (8 lanes, or 4 for short records), when the float32 layout of this plan is a
        // different one
        const X64Choice xc = x64_choice(p);
        const int32_t xsubs = xc.variant == 21 ? 4 : 8;
        if ((xc.variant == 8 || xc.variant == 20 || xc.variant == 21) && !(p->subs2 == xsubs && p->yps2 == xc.yps) &&
            !(p->ypsx == xc.yps && p->subsx == xsubs)) {
            if (p->d_tablex) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(p->d_tablex)); p->d_tablex = nullptr; }
            const std::vector<uint32_t> tx = h.ring_table(xsubs, xc.yps);
            HIP_TRY(hipMalloc(&p->d_tablex, sizeof(uint32_t) * tx.size()));
            HIP_TRY(hipMemcpy(p->d_tablex, tx.data(), sizeof(uint32_t) * tx.size(), hipMemcpyHostToDevice));
            p->ypsx = xc.yps;
            p->subsx = xsubs;
        }
    }
    if (p->yps || p->yps64 || p->yps2) {
        std::vector<xmhw::Chunk> ch = h.make_chunks(nchunks);
        std::vector<xmhw::DevChunk> dch(ch.size());
        for (size_t i = 0; i < ch.size(); ++i) dch[i] = {ch[i].warm_start, ch[i].begin, ch[i].end};
        if (p->d_chunks) { HIP_TRY(hipFree(p->d_chunks)); p->d_chunks = nullptr; }
        HIP_TRY(hipMalloc(&p->d_chunks, sizeof(xmhw::DevChunk) * dch.size()));
        HIP_TRY(hipMemcpy(p->d_chunks, dch.data(), sizeof(xmhw::DevChunk) * dch.size(),
                          hipMemcpyHostToDevice));
        p->nchunks = static_cast<int32_t>(dch.size());
    } else {
        p->nchunks = nchunks;
    }
    p->uploaded = true;
    return XMHW_OK;
}

template <typename T>
int clim_raw(xmhw_plan* plan, const T* ts, int64_t C, int64_t ld, double q, int negate,
             double* thresh, double* seas, int64_t ldo, void* stream) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    if (C < 0 || ld < C || ldo < C) return fail(XMHW_ERR_INVALID, "bad C/ld/ldo");
    if (!(q >= 0.0 && q <= 1.0)) return fail(XMHW_ERR_INVALID, "quantile must be in [0, 1]");
    if (C == 0) return XMHW_OK;
    if (!ts || !thresh || !seas) return fail(XMHW_ERR_INVALID, "NULL device buffer");
    const int32_t kernel = resolve_kernel(plan, sizeof(T));
    if (kernel < 0)
        return fail(XMHW_ERR_UNSUPPORTED, "ring kernel not available for this window/track count/dtype");
    int rc = upload(plan, C);
    if (rc != XMHW_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const xmhw::Plan& h = plan->host;
    hipError_t e;
    if (kernel == XMHW_KERNEL_RING) {
        if constexpr (sizeof(T) == 4) {
            // float32: the sorted-list kernel on the regular rows (high percentiles: a list keeps its K largest keys),
            // the ring kernel on the others, the generic kernel on the cell-rows the sorted kernel flagged
            const bool sorted = plan->nchunks_s > 0 && sorted_serves(q);
            const xmhw::DevChunk* rchunks = sorted ? plan->d_chunks_i : plan->d_chunks;
            const int32_t rn = sorted ? plan->nchunks_i : plan->nchunks;
            unsigned long long* rstats = sorted ? nullptr : plan->d_stats;
            e = hipSuccess;
            hipEvent_t t0 = nullptr, t1 = nullptr;
            if (plan->timing) {
                const int slot = static_cast<int>(plan->tcalls % 16);
                for (int i = 0; i < 2; ++i)
                    if (!plan->tev[2 * slot + i]) HIP_TRY(hipEventCreate(&plan->tev[2 * slot + i]));
                t0 = plan->tev[2 * slot];
                t1 = plan->tev[2 * slot + 1];
                plan->tcalls++;
            }
            if (sorted) {
                if (t0) e = hipEventRecord(t0, st);
                if (e == hipSuccess)
                    e = xmhw::launch_sorted_f32(reinterpret_cast<const float*>(ts), C, ld, h.T, plan->d_table_s,
                                                plan->d_sflags_s, plan->d_chunks_s, plan->nchunks_s, h.w,
                                                plan->yps_s, h.ntracks, q, negate, thresh, seas, ldo, st, plan->d_stats);
                if (e == hipSuccess && t1) e = hipEventRecord(t1, st);
            } else if (t0) {
                e = hipEventRecord(t0, st);
            }
            if (e != hipSuccess) {
            } else if (rn == 0) {
            } else
            if (plan->yps2 && ring2_resolved(plan) >= 0 && xmhw::ring2_f32_supported(h.w, plan->yps2, ring2_resolved(plan)))
                e = xmhw::launch_ring2_f32(reinterpret_cast<const float*>(ts), C, ld, h.T, plan->d_table2, plan->d_sflags,
                                           h.step_min, rchunks, rn, h.w, plan->yps2, h.ntracks,
                                           ring2_resolved(plan), q, negate, thresh, seas, ldo, st, rstats);
            else
            e = xmhw::launch_ring_f32(reinterpret_cast<const float*>(ts), C, ld, plan->d_table,
                                      h.step_min, rchunks, rn, h.w, plan->yps, plan->subs, q,
                                      negate, thresh, seas, ldo, st, rstats);
            if (e == hipSuccess && !sorted && t1) e = hipEventRecord(t1, st);
        } else {
            // float64 input: if every sample is float32-representable (decoded int16 / float32
            // archives) the float32 kernel gives the same pools at 2.7x the rate.  All decisions are
            // taken on the device so that the call stays asynchronous: probe -> narrowing float32
            // kernel (stops at the first lossy sample) -> float64 kernel (runs only if flagged).
            const uint32_t* run_flag = nullptr;
            e = hipSuccess;
            const NarrowChoice nc = narrow_choice(plan);
            if (plan->narrowing && nc.variant >= 0) {
                // the second-generation kernel narrows too (the shipped layouts)
                if (!plan->d_narrow_flag) HIP_TRY(hipMalloc(&plan->d_narrow_flag, sizeof(uint32_t)));
                e = xmhw::launch_narrow_probe(reinterpret_cast<const double*>(ts), h.T, C, ld, plan->d_narrow_flag, st);
                if (e == hipSuccess)
                    e = xmhw::launch_ring2_f32_narrowing(reinterpret_cast<const double*>(ts), C, ld, h.T, nc.table,
                                                         plan->d_sflags, h.step_min, plan->d_chunks, plan->nchunks, h.w,
                                                         nc.yps, h.ntracks, nc.variant, q, negate, thresh, seas, ldo, st,
                                                         plan->d_narrow_flag);
                run_flag = plan->d_narrow_flag;
            } else if (plan->narrowing && plan->yps) {
                if (!plan->d_narrow_flag) HIP_TRY(hipMalloc(&plan->d_narrow_flag, sizeof(uint32_t)));
                e = xmhw::launch_ring_f32_narrowing(reinterpret_cast<const double*>(ts), h.T, C, ld, plan->d_table,
                                                    h.step_min, plan->d_chunks, plan->nchunks, h.w, plan->yps,
                                                    plan->subs, q, negate, thresh, seas, ldo, st, plan->d_stats,
                                                    plan->d_narrow_flag);
                run_flag = plan->d_narrow_flag;
            }
            // genuinely float64 samples: the second-generation kernel's 64-bit mode (resolve_kernel() returned
            // XMHW_KERNEL_RING only because it is instantiated for this plan)
            if (e == hipSuccess) {
                const X64Choice xc = x64_choice(plan);
                if (xc.variant >= 0)
                    e = xmhw::launch_ring2_f64(reinterpret_cast<const double*>(ts), C, ld, h.T,
                                               xc.variant == 12 ? ((plan->subs2 == 16 && plan->yps2 == xc.yps) ? plan->d_table2 : plan->d_table64)
                                               : (plan->subs2 == (xc.variant == 21 ? 4 : 8) && plan->yps2 == xc.yps) ? plan->d_table2
                                               : plan->d_tablex,
                                               plan->d_sflags,
                                               h.step_min, plan->d_chunks, plan->nchunks, h.w, xc.yps, h.ntracks, xc.variant,
                                               q, negate, thresh, seas, ldo, st, run_flag);
                else
                    return fail(XMHW_ERR_UNSUPPORTED, "no float64 ring kernel for this plan");
            }
        }
    } else {
        const uint32_t* run_flag = nullptr;
        e = hipSuccess;
        if constexpr (sizeof(T) == 8) {
            // no float64 ring for this plan (e.g. a record of more than 48 tracks), but the float32 ring
            // covers it: float32-representable data still takes the fast kernel, the generic one
            // runs only if the narrowing gave up
            const NarrowChoice nc = narrow_choice(plan);
            if (plan->narrowing && nc.variant >= 0 && plan->host.kernel_choice == XMHW_KERNEL_AUTO) {
                if (!plan->d_narrow_flag) HIP_TRY(hipMalloc(&plan->d_narrow_flag, sizeof(uint32_t)));
                e = xmhw::launch_narrow_probe(reinterpret_cast<const double*>(ts), h.T, C, ld, plan->d_narrow_flag, st);
                if (e == hipSuccess)
                    e = xmhw::launch_ring2_f32_narrowing(reinterpret_cast<const double*>(ts), C, ld, h.T, nc.table,
                                                         plan->d_sflags, h.step_min, plan->d_chunks, plan->nchunks, h.w,
                                                         nc.yps, h.ntracks, nc.variant, q, negate, thresh, seas, ldo, st,
                                                         plan->d_narrow_flag);
                run_flag = plan->d_narrow_flag;
            } else if (plan->narrowing && plan->yps && plan->host.kernel_choice == XMHW_KERNEL_AUTO) {
                if (!plan->d_narrow_flag) HIP_TRY(hipMalloc(&plan->d_narrow_flag, sizeof(uint32_t)));
                e = xmhw::launch_ring_f32_narrowing(reinterpret_cast<const double*>(ts), h.T, C, ld, plan->d_table,
                                                    h.step_min, plan->d_chunks, plan->nchunks, h.w, plan->yps,
                                                    plan->subs, q, negate, thresh, seas, ldo, st, plan->d_stats,
                                                    plan->d_narrow_flag);
                run_flag = plan->d_narrow_flag;
            }
        }
        if (e == hipSuccess)
            e = xmhw::launch_generic<T>(ts, h.T, C, ld, plan->d_row_ptr, plan->d_centres, h.D, h.w, q,
                                        negate, thresh, seas, ldo, st, run_flag);
    }
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    return XMHW_OK;
}

int row_index(const xmhw::Plan& h, int32_t label) {
    auto it = std::lower_bound(h.doys.begin(), h.doys.end(), label);
    if (it == h.doys.end() || *it != label) return -1;
    return static_cast<int>(it - h.doys.begin());
}

template <typename T>
int clim_oneshot(const T* ts, const int32_t* doy, int64_t Tn, int64_t C, int32_t D, int32_t w,
                 double q, int smooth, int smooth_w, int feb29_fix, int negate, double* thresh,
                 double* seas, void* stream) {
    if (smooth && (smooth_w <= 0 || smooth_w % 2 == 0))
        return fail(XMHW_ERR_INVALID, "smoothPercentileWidth should be odd");
    xmhw_plan* plan = nullptr;
    int rc = xmhw_plan_create(doy, Tn, w, &plan);
    if (rc != XMHW_OK) return rc;
    if (plan->host.D != D) {
        xmhw_plan_destroy(plan);
        return fail(XMHW_ERR_INVALID, "D does not match the number of distinct doy labels");
    }
    double *rt = nullptr, *rs = nullptr;
    const size_t bytes = sizeof(double) * static_cast<size_t>(D) * static_cast<size_t>(C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto cleanup = [&]() {
        if (rt) (void)hipFree(rt);
        if (rs) (void)hipFree(rs);
        xmhw_plan_destroy(plan);
    };
    if (C == 0) { cleanup(); return XMHW_OK; }
    const bool need_finish = smooth || feb29_fix;
    if (need_finish) {
        if (hipMalloc(&rt, bytes) != hipSuccess || hipMalloc(&rs, bytes) != hipSuccess) {
            cleanup();
            return fail(XMHW_ERR_NOMEM, "hipMalloc of the raw climatology failed");
        }
    }
    rc = clim_raw<T>(plan, ts, C, C, q, negate, need_finish ? rt : thresh, need_finish ? rs : seas, C,
                     stream);
    if (rc == XMHW_OK && need_finish)
        rc = xmhw_clim_finish(plan, rt, rs, C, C, feb29_fix, smooth, smooth_w, thresh, seas, stream);
    hipError_t e = hipStreamSynchronize(st);
    cleanup();
    if (rc != XMHW_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    return XMHW_OK;
}

// ---- per-call tables kept on the device between calls ------------------------------------------
// Every detect-side entry needs row_of_t (and the tiled exceedance kernel its chunk tables) on the
// device.  Round 1 allocated, uploaded, synchronised and freed them on every call, which made the
// "asynchronous" stream argument a fiction.  Now a small cache keyed by the table's content keeps them
// (uploaded once, with a blocking copy, at the first call that sees them), and scratch buffers are
// kept per stream and grow on demand: steady-state calls neither allocate nor synchronise.
struct ChunkTables {
    int32_t *tile_begin = nullptr, *t0 = nullptr, *i0 = nullptr, *n = nullptr;
    int32_t ntiles = 0;
    int64_t nchunks = 0;
};
struct RowsEntry {
    std::vector<int32_t> host;
    int device = 0;
    int32_t* d_rows = nullptr;
    std::map<std::pair<int64_t, int>, ChunkTables> chunks;      // (D, tile) -> tables
    uint64_t stamp = 0;
    ~RowsEntry() {
        if (d_rows) (void)hipFree(d_rows);
        for (auto& kv : chunks)
            for (int32_t* p : {kv.second.tile_begin, kv.second.t0, kv.second.i0, kv.second.n})
                if (p) (void)hipFree(p);
    }
};
std::mutex g_cache_mu;
std::vector<std::shared_ptr<RowsEntry>> g_rows_cache;
uint64_t g_stamp = 0;
constexpr size_t kRowsCacheSlots = 8;

// returns nullptr and sets *err on failure.  The caller HOLDS the returned pointer until its launches are queued: an
// entry another thread evicts meanwhile is destroyed (hipFree, which waits for the device) only when the last holder
// lets go of it.
using RowsRef = std::shared_ptr<RowsEntry>;
RowsRef cached_rows(const int32_t* row_of_t, int64_t Tn, hipError_t* err) {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    int dev = 0;
    *err = hipGetDevice(&dev);
    if (*err != hipSuccess) return nullptr;
    for (auto& e : g_rows_cache)
        if (e->device == dev && static_cast<int64_t>(e->host.size()) == Tn &&
            std::memcmp(e->host.data(), row_of_t, sizeof(int32_t) * static_cast<size_t>(Tn)) == 0) {
            e->stamp = ++g_stamp;
            return e;
        }
    auto e = std::make_shared<RowsEntry>();
    e->host.assign(row_of_t, row_of_t + Tn);
    e->device = dev;
    *err = hipMalloc(&e->d_rows, sizeof(int32_t) * static_cast<size_t>(Tn));
    if (*err != hipSuccess) { e->d_rows = nullptr; return nullptr; }
    *err = hipMemcpy(e->d_rows, row_of_t, sizeof(int32_t) * static_cast<size_t>(Tn), hipMemcpyHostToDevice);
    if (*err != hipSuccess) return nullptr;
    if (g_rows_cache.size() >= kRowsCacheSlots) {
        size_t oldest = 0;
        for (size_t i = 1; i < g_rows_cache.size(); ++i)
            if (g_rows_cache[i]->stamp < g_rows_cache[oldest]->stamp) oldest = i;
        (void)hipDeviceSynchronize();          // nothing in flight may still read the evicted tables
        g_rows_cache.erase(g_rows_cache.begin() + static_cast<long>(oldest));
    }
    e->stamp = ++g_stamp;
    g_rows_cache.push_back(e);
    return e;
}

// scratch memory per (device, stream): grows on demand (the only synchronising moment), never shrinks.  A caller
// keeps the ScratchRef for the length of its call: a buffer that another thread on the same stream outgrows meanwhile
// is freed only when its last holder is done.
using ScratchRef = std::shared_ptr<void>;
struct Scratch { ScratchRef ptr; size_t cap = 0; };
std::map<std::pair<int, void*>, Scratch> g_scratch;
hipError_t scratch_get(hipStream_t st, size_t bytes, void** out, ScratchRef* keep) {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    Scratch& sc = g_scratch[{dev, static_cast<void*>(st)}];
    if (sc.cap < bytes) {
        if (sc.ptr) {
            e = hipStreamSynchronize(st);
            if (e != hipSuccess) return e;
            sc.ptr.reset();
            sc.cap = 0;
        }
        void* p = nullptr;
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return e;
        sc.ptr = ScratchRef(p, [](void* q) { (void)hipFree(q); });
        sc.cap = bytes;
    }
    *out = sc.ptr.get();
    *keep = sc.ptr;
    return hipSuccess;
}
void release_cached_tables() {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    (void)hipDeviceSynchronize();
    g_rows_cache.clear();
    g_scratch.clear();
}

template <typename T>
int detect_events(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* thresh, int64_t ldt,
                  const int32_t* row_of_t, int32_t min_duration, int32_t join_gaps, int32_t max_gap,
                  int32_t negate, int32_t* events, int32_t* start, int32_t* end, uint8_t* bthresh,
                  int64_t ldo, int32_t* nevents, void* stream) {
    if (Tn <= 0 || C < 0 || ld < C || ldt < C || ldo < C) return fail(XMHW_ERR_INVALID, "bad T/C/ld/ldt/ldo");
    if (min_duration < 1 || max_gap < 0) return fail(XMHW_ERR_INVALID, "minDuration must be >= 1 and maxGap >= 0");
    if (C == 0) return XMHW_OK;
    if (!ts || !thresh || !row_of_t || !events || !start || !end)
        return fail(XMHW_ERR_INVALID, "NULL buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipSuccess;
    const RowsRef rows = cached_rows(row_of_t, Tn, &e);
    if (!rows) return hip_fail(e, "row table upload");
    e = xmhw::launch_detect<T>(ts, Tn, C, ld, thresh, ldt, rows->d_rows, min_duration, join_gaps, max_gap, negate,
                               events, start, end, bthresh, ldo, nevents, st);
    if (e != hipSuccess) return hip_fail(e, "detect_events launch");
    return XMHW_OK;
}

template <typename T>
int event_stats(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* seas, const double* thresh,
                int64_t ldc, const int32_t* row_of_t, int32_t negate, const int32_t* events, int64_t ldo,
                const int64_t* offsets, double* table, void* stream) {
    if (Tn <= 0 || C < 0 || ld < C || ldc < C || ldo < C) return fail(XMHW_ERR_INVALID, "bad T/C/ld/ldc/ldo");
    if (C == 0) return XMHW_OK;
    if (!ts || !seas || !thresh || !row_of_t || !events || !offsets)
        return fail(XMHW_ERR_INVALID, "NULL buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipSuccess;
    const RowsRef rows = cached_rows(row_of_t, Tn, &e);
    if (!rows) return hip_fail(e, "row table upload");
    e = xmhw::launch_event_stats<T>(ts, Tn, C, ld, seas, thresh, ldc, rows->d_rows, negate, events, ldo, offsets, table, st);
    if (e != hipSuccess) return hip_fail(e, "event_stats launch");
    return XMHW_OK;
}

static std::atomic<int> g_exceed_kernel{0};   // 0 auto, 1 per-step kernel, 2 tiled kernel (xmhw_set_exceed_kernel)

// Chunks for exceed_bits_tiled: maximal segments of consecutive steps with consecutive rows inside one
// tile of `tile` rows, grouped by tile (time order kept inside a tile).
struct ExceedChunks {
    std::vector<int32_t> tile_begin, t0, i0, n;
    int32_t ntiles = 0;
    ExceedChunks(const int32_t* row_of_t, int64_t Tn, int64_t D, int tile) {
        ntiles = static_cast<int32_t>((D + tile - 1) / tile);
        std::vector<std::vector<int32_t>> per(ntiles);          // chunk start steps per tile
        std::vector<int32_t> len;                                 // by start step (sparse via map below)
        std::vector<int32_t> start, count;
        for (int64_t t = 0; t < Tn;) {
            const int32_t r = row_of_t[t];
            const int32_t k = r / tile;
            int64_t e = t + 1;
            while (e < Tn && row_of_t[e] == row_of_t[e - 1] + 1 && row_of_t[e] / tile == k) ++e;
            per[k].push_back(static_cast<int32_t>(start.size()));
            start.push_back(static_cast<int32_t>(t));
            count.push_back(static_cast<int32_t>(e - t));
            t = e;
        }
        tile_begin.assign(ntiles + 1, 0);
        for (int32_t k = 0; k < ntiles; ++k) {
            tile_begin[k + 1] = tile_begin[k] + static_cast<int32_t>(per[k].size());
            for (int32_t id : per[k]) {
                t0.push_back(start[id]);
                i0.push_back(row_of_t[start[id]] - k * tile);
                n.push_back(count[id]);
            }
        }
    }
};

template <typename T>
int exceed_bits(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* thresh, int64_t ldt, int64_t D,
                const int32_t* row_of_t, int32_t negate, uint64_t* bits, int64_t ldb, void* stream) {
    if (Tn <= 0 || C < 0 || ld < C || ldt < C || ldb < C || D <= 0) return fail(XMHW_ERR_INVALID, "bad T/C/ld/ldt/ldb/D");
    if (Tn >= (int64_t{1} << 31)) return fail(XMHW_ERR_INVALID, "T too large");
    if (C == 0) return XMHW_OK;
    if (!ts || !thresh || !row_of_t || !bits) return fail(XMHW_ERR_INVALID, "NULL buffer");
    for (int64_t t = 0; t < Tn; ++t)
        if (row_of_t[t] < 0 || row_of_t[t] >= D) return fail(XMHW_ERR_INVALID, "row_of_t outside [0, D)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    constexpr int kTile = sizeof(T) == 4 ? 64 : 32;
    hipError_t e = hipSuccess;
    const RowsRef rows = cached_rows(row_of_t, Tn, &e);
    if (!rows) return hip_fail(e, "row table upload");
    // chunk tables of the tiled kernel: built and uploaded once per (row table, D, tile)
    ChunkTables* ct = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        auto key = std::make_pair(D, kTile);
        auto it = rows->chunks.find(key);
        if (it == rows->chunks.end()) {
            const ExceedChunks ch(row_of_t, Tn, D, kTile);
            ChunkTables t;
            t.ntiles = ch.ntiles;
            t.nchunks = static_cast<int64_t>(ch.t0.size());
            auto put = [&](int32_t** dst, const std::vector<int32_t>& v) -> hipError_t {
                const size_t bytes = sizeof(int32_t) * (v.empty() ? 1 : v.size());
                hipError_t err = hipMalloc(dst, bytes);
                if (err == hipSuccess && !v.empty()) err = hipMemcpy(*dst, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice);
                return err;
            };
            for (auto pr : {std::make_pair(&t.tile_begin, &ch.tile_begin), std::make_pair(&t.t0, &ch.t0),
                            std::make_pair(&t.i0, &ch.i0), std::make_pair(&t.n, &ch.n)})
                if (e == hipSuccess) e = put(pr.first, *pr.second);
            if (e != hipSuccess) return hip_fail(e, "chunk table upload");
            it = rows->chunks.emplace(key, t).first;
        }
        ct = &it->second;
    }
    // The tiled kernel pays one pass over its unrolled tile per chunk: worth it when chunks are long
    // (calendar-like labels); an arbitrary label sequence falls back to the per-step kernel.
    // One thread walks all tiles of a cell, so small grids (too few workgroups to fill 256 CUs) keep the
    // per-step kernel, which also splits the time axis over blocks.
    const int mode = g_exceed_kernel.load();
    const bool tiled = mode == 2 || (mode == 0 && ct->nchunks * kTile <= 4 * Tn && C >= 131072);
    const int64_t W = (Tn + 63) / 64;
    float* thf = nullptr;
    ScratchRef scratch_keep;
    if constexpr (sizeof(T) == 4) {
        // float32 series: compare against the float32 floor of the thresholds (same results, see
        // kernels_events.hip), 4 instead of 8 bytes per threshold.  Only the addressed (D, C) region is
        // converted: `thresh` may point into a wider array (column block k0 of a (D, ldt) climatology),
        // where D * ldt elements would overrun it.  The copy lives in this stream's scratch buffer.
        void* sp = nullptr;
        e = scratch_get(st, sizeof(float) * static_cast<size_t>(D) * static_cast<size_t>(C), &sp, &scratch_keep);
        if (e != hipSuccess) return hip_fail(e, "scratch allocation");
        thf = static_cast<float*>(sp);
        e = xmhw::launch_floor_to_f32(thresh, D, C, ldt, thf, C, st);
        if (e != hipSuccess) return hip_fail(e, "floor_to_f32 launch");
    }
    const int64_t ldtf = sizeof(T) == 4 ? C : ldt;   // leading dimension of the thresholds the kernels read
    if (tiled) {
        e = ldb == C ? hipMemsetAsync(bits, 0, sizeof(uint64_t) * static_cast<size_t>(W) * static_cast<size_t>(C), st)
                     : hipMemset2DAsync(bits, sizeof(uint64_t) * static_cast<size_t>(ldb), 0,
                                        sizeof(uint64_t) * static_cast<size_t>(C), static_cast<size_t>(W), st);
        if (e == hipSuccess) {
            if constexpr (sizeof(T) == 4)
                e = xmhw::launch_exceed_bits_tiled<float, float, 64>(ts, C, ld, thf, ldtf, D, ct->tile_begin, ct->ntiles,
                                                                     ct->t0, ct->i0, ct->n, negate, bits, ldb, st);
            else
                e = xmhw::launch_exceed_bits_tiled<double, double, 32>(ts, C, ld, thresh, ldt, D, ct->tile_begin, ct->ntiles,
                                                                       ct->t0, ct->i0, ct->n, negate, bits, ldb, st);
        }
        if (e != hipSuccess) return hip_fail(e, "exceed_bits_tiled launch");
        return XMHW_OK;
    }
    if constexpr (sizeof(T) == 4)
        e = xmhw::launch_exceed_bits<float, float>(ts, Tn, C, ld, thf, ldtf, rows->d_rows, negate, bits, ldb, st);
    else
        e = xmhw::launch_exceed_bits<double, double>(ts, Tn, C, ld, thresh, ldt, rows->d_rows, negate, bits, ldb, st);
    if (e != hipSuccess) return hip_fail(e, "exceed_bits launch");
    return XMHW_OK;
}

template <typename T>
int event_stats_sparse(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* seas, const double* thresh,
                       int64_t ldc, const int32_t* row_of_t, int32_t negate, int64_t n_events, double* table,
                       void* stream) {
    if (Tn <= 0 || C < 0 || ld < C || ldc < C || n_events < 0) return fail(XMHW_ERR_INVALID, "bad T/C/ld/ldc/n_events");
    if (n_events == 0) return XMHW_OK;
    if (!ts || !seas || !thresh || !row_of_t || !table) return fail(XMHW_ERR_INVALID, "NULL buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipSuccess;
    const RowsRef rows = cached_rows(row_of_t, Tn, &e);
    if (!rows) return hip_fail(e, "row table upload");
    e = xmhw::launch_event_stats_sparse<T>(ts, Tn, ld, seas, thresh, ldc, rows->d_rows, negate, n_events, table, st);
    if (e != hipSuccess) return hip_fail(e, "event_stats_sparse launch");
    return XMHW_OK;
}

template <typename T>
int event_intermediate(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* seas, const double* thresh,
                       int64_t ldc, const int32_t* row_of_t, int32_t negate, const int32_t* events, int64_t ldo,
                       double* out, int64_t ldv, uint8_t* dur, void* stream) {
    if (Tn <= 0 || C < 0 || ld < C || ldc < C || ldo < C || ldv < C)
        return fail(XMHW_ERR_INVALID, "bad T/C/ld/ldc/ldo/ldv");
    if (C == 0) return XMHW_OK;
    if (!ts || !seas || !thresh || !row_of_t || !events || !out || !dur) return fail(XMHW_ERR_INVALID, "NULL buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipSuccess;
    const RowsRef rows = cached_rows(row_of_t, Tn, &e);
    if (!rows) return hip_fail(e, "row table upload");
    e = xmhw::launch_event_intermediate<T>(ts, Tn, C, ld, seas, thresh, ldc, rows->d_rows, negate, events, ldo, out, ldv, dur,
                                           st);
    if (e != hipSuccess) return hip_fail(e, "event_intermediate launch");
    return XMHW_OK;
}

template <typename T>
int clim_host(const T* ts, const int32_t* doy, int64_t Tn, int64_t C, int32_t D, int32_t w, double q,
              int smooth, int smooth_w, int feb29_fix, int negate, double* thresh, double* seas) {
    if (C == 0) return XMHW_OK;
    if (!ts || !thresh || !seas) return fail(XMHW_ERR_INVALID, "NULL host buffer");
    T* d_ts = nullptr;
    double *d_th = nullptr, *d_se = nullptr;
    const size_t in_bytes = sizeof(T) * static_cast<size_t>(Tn) * static_cast<size_t>(C);
    const size_t out_bytes = sizeof(double) * static_cast<size_t>(D) * static_cast<size_t>(C);
    auto cleanup = [&]() {
        if (d_ts) (void)hipFree(d_ts);
        if (d_th) (void)hipFree(d_th);
        if (d_se) (void)hipFree(d_se);
    };
    if (hipMalloc(&d_ts, in_bytes) != hipSuccess || hipMalloc(&d_th, out_bytes) != hipSuccess ||
        hipMalloc(&d_se, out_bytes) != hipSuccess) {
        cleanup();
        return fail(XMHW_ERR_NOMEM, "hipMalloc failed");
    }
    hipError_t e = hipMemcpy(d_ts, ts, in_bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { cleanup(); return hip_fail(e, "hipMemcpy H2D"); }
    int rc = clim_oneshot<T>(d_ts, doy, Tn, C, D, w, q, smooth, smooth_w, feb29_fix, negate, d_th, d_se,
                             nullptr);
    if (rc == XMHW_OK) {
        e = hipMemcpy(thresh, d_th, out_bytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(seas, d_se, out_bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy D2H");
    }
    cleanup();
    return rc;
}

}  // namespace

template <typename T>
static int block_time(const T* ts, int64_t Tn, int64_t C, int64_t ld, const double* cats, int64_t ldcat,
                      const int32_t* bin_of_t, int32_t nbins, double* out, int64_t ldo, void* stream) {
    if (C < 0 || Tn <= 0 || nbins <= 0 || ldo < C || ld < C || (cats && ldcat < C)) return fail(XMHW_ERR_INVALID, "bad C/T/nbins/ld");
    if (C == 0) return XMHW_OK;
    if (!ts || !bin_of_t || !out) return fail(XMHW_ERR_INVALID, "NULL buffer");
    hipError_t e = xmhw::launch_block_time<T>(ts, Tn, C, ld, cats, ldcat, bin_of_t, nbins, out, ldo,
                                              static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "block_time launch");
    return XMHW_OK;
}

extern "C" {

int xmhw_version(void) { return 1000 * 0 + 1; }
const char* xmhw_arch(void) { return "gfx950"; }
const char* xmhw_last_error(void) { return g_err.c_str(); }

int xmhw_device_count(int* count) {
    if (!count) return fail(XMHW_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return hip_fail(e, "hipGetDeviceCount"); }
    *count = n;
    return XMHW_OK;
}
int xmhw_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return XMHW_OK;
}
int xmhw_get_device(int* device) {
    if (!device) return fail(XMHW_ERR_INVALID, "device is NULL");
    HIP_TRY(hipGetDevice(device));
    return XMHW_OK;
}
int xmhw_device_info(int device, char* name, int name_len, int* compute_units, uint64_t* hbm_bytes) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
        std::snprintf(name, static_cast<size_t>(name_len), "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return XMHW_OK;
}

int xmhw_malloc(void** dev_ptr, size_t bytes) {
    if (!dev_ptr) return fail(XMHW_ERR_INVALID, "dev_ptr is NULL");
    *dev_ptr = nullptr;
    if (bytes == 0) return XMHW_OK;
    hipError_t e = hipMalloc(dev_ptr, bytes);
    if (e == hipErrorOutOfMemory) return fail(XMHW_ERR_NOMEM, "hipMalloc: out of memory");
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    return XMHW_OK;
}
int xmhw_free(void* dev_ptr) {
    if (dev_ptr) HIP_TRY(hipFree(dev_ptr));
    return XMHW_OK;
}
int xmhw_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return XMHW_OK;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return XMHW_OK;
}
int xmhw_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return XMHW_OK;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return XMHW_OK;
}
int xmhw_memcpy2d_h2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                      void* stream) {
    if (width == 0 || height == 0) return XMHW_OK;
    if (!dst || !src || dpitch < width || spitch < width) return fail(XMHW_ERR_INVALID, "bad pointer/pitch");
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return XMHW_OK;
}
int xmhw_host_alloc(void** host_ptr, size_t bytes) {
    if (!host_ptr) return fail(XMHW_ERR_INVALID, "host_ptr is NULL");
    *host_ptr = nullptr;
    if (bytes == 0) return XMHW_OK;
    hipError_t e = hipHostMalloc(host_ptr, bytes, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) return fail(XMHW_ERR_NOMEM, "hipHostMalloc: out of memory");
    if (e != hipSuccess) return hip_fail(e, "hipHostMalloc");
    return XMHW_OK;
}
int xmhw_host_free(void* host_ptr) {
    if (host_ptr) HIP_TRY(hipHostFree(host_ptr));
    return XMHW_OK;
}
int xmhw_memcpy2d_h2d_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                            void* stream) {
    if (width == 0 || height == 0) return XMHW_OK;
    if (!dst || !src || dpitch < width || spitch < width) return fail(XMHW_ERR_INVALID, "bad pointer/pitch");
    HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    return XMHW_OK;
}
int xmhw_memcpy_h2d_async(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return XMHW_OK;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    return XMHW_OK;
}
int xmhw_event_sync(void* event) {
    HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(event)));
    return XMHW_OK;
}
int xmhw_memcpy_d2h_async(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return XMHW_OK;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    return XMHW_OK;
}
int xmhw_decode(const void* raw_dev, int raw_itemsize, int big_endian, int64_t rows, int64_t cols, int64_t ld_raw,
                void* out_dev, int out_itemsize, int64_t ld_out, int has_scale, double scale_factor, double add_offset,
                int has_fill, double fill_value, void* stream) {
    if (rows < 0 || cols < 0 || ld_raw < cols || ld_out < cols) return fail(XMHW_ERR_INVALID, "bad rows/cols/ld");
    if (rows == 0 || cols == 0) return XMHW_OK;
    if (!raw_dev || !out_dev) return fail(XMHW_ERR_INVALID, "NULL device buffer");
    hipError_t e = xmhw::launch_decode(raw_dev, raw_itemsize, big_endian, rows, cols, ld_raw, out_dev, out_itemsize, ld_out,
                                       scale_factor, add_offset, has_scale, has_fill, fill_value,
                                       static_cast<hipStream_t>(stream));
    if (e == hipErrorInvalidValue)
        return fail(XMHW_ERR_UNSUPPORTED, "decode: stored/decoded type pair not supported (int16->f32/f64, f32->f32, f64->f64)");
    if (e != hipSuccess) return hip_fail(e, "decode launch");
    return XMHW_OK;
}
int xmhw_encode_i16(const float* in_dev, int64_t rows, int64_t cols, int64_t ld_in, int16_t* out_dev, int64_t ld_out,
                    double scale_factor, double add_offset, int32_t fill_code, void* stream) {
    if (rows < 0 || cols < 0 || ld_in < cols || ld_out < cols) return fail(XMHW_ERR_INVALID, "bad rows/cols/ld");
    if (!(scale_factor == scale_factor) || scale_factor == 0.0 || !(add_offset == add_offset))
        return fail(XMHW_ERR_INVALID, "scale_factor must be a non-zero number, add_offset a number");
    if (fill_code < -32768 || fill_code > 32767) return fail(XMHW_ERR_INVALID, "fill_code is not an int16");
    if (rows == 0 || cols == 0) return XMHW_OK;
    if (!in_dev || !out_dev) return fail(XMHW_ERR_INVALID, "NULL device buffer");
    hipError_t e = xmhw::launch_encode_i16(in_dev, rows, cols, ld_in, out_dev, ld_out, scale_factor, add_offset, fill_code,
                                           static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "encode launch");
    return XMHW_OK;
}
int xmhw_read_rows(int fd, int64_t file_offset, int64_t row_pitch, int64_t row_bytes, int64_t rows, void* dst_host) {
    // pread() copies from the page cache (or the disk) straight into the caller's buffer -- for the ingest
    // path a page-locked staging buffer: no page of the file is ever mapped into this process, so many
    // threads calling this at once do not queue up on the address space's page-fault path the way
    // copies out of an mmap() do
    if (fd < 0 || file_offset < 0 || row_pitch < row_bytes || row_bytes < 0 || rows < 0)
        return fail(XMHW_ERR_INVALID, "bad fd/offset/pitch/row_bytes/rows");
    if (rows == 0 || row_bytes == 0) return XMHW_OK;
    if (!dst_host) return fail(XMHW_ERR_INVALID, "NULL destination");
    char* dst = static_cast<char*>(dst_host);
    for (int64_t r = 0; r < rows; ++r) {
        int64_t done = 0;
        while (done < row_bytes) {
            const ssize_t got = ::pread(fd, dst + r * row_bytes + done, static_cast<size_t>(row_bytes - done),
                                        static_cast<off_t>(file_offset + r * row_pitch + done));
            if (got < 0) {
                if (errno == EINTR) continue;
                return fail(XMHW_ERR_INVALID, std::string("pread: ") + std::strerror(errno));
            }
            if (got == 0) return fail(XMHW_ERR_INVALID, "pread: unexpected end of file");
            done += got;
        }
    }
    return XMHW_OK;
}
int xmhw_pad_gaps(void* ts_dev, int itemsize, int64_t T, int64_t C, int64_t ld, const double* x_dev, double max_gap,
                  void* stream) {
    if (T < 0 || C < 0 || ld < C) return fail(XMHW_ERR_INVALID, "bad T/C/ld");
    if (itemsize != 4 && itemsize != 8) return fail(XMHW_ERR_INVALID, "itemsize must be 4 or 8");
    if (!(max_gap == max_gap)) return fail(XMHW_ERR_INVALID, "max_gap is NaN");
    if (T == 0 || C == 0) return XMHW_OK;
    if (!ts_dev || !x_dev) return fail(XMHW_ERR_INVALID, "NULL device buffer");
    hipError_t e = xmhw::launch_pad_gaps(ts_dev, itemsize, T, C, ld, x_dev, max_gap, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "pad_gaps launch");
    return XMHW_OK;
}
int xmhw_memset(void* dst, int value, size_t bytes, void* stream) {
    if (bytes == 0) return XMHW_OK;
    HIP_TRY(hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(stream)));
    return XMHW_OK;
}
int xmhw_stream_create(void** stream) {
    if (!stream) return fail(XMHW_ERR_INVALID, "stream is NULL");
    hipStream_t s;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return XMHW_OK;
}
int xmhw_stream_destroy(void* stream) {
    if (stream) HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return XMHW_OK;
}
int xmhw_stream_sync(void* stream) {
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return XMHW_OK;
}
int xmhw_event_create(void** event) {
    if (!event) return fail(XMHW_ERR_INVALID, "event is NULL");
    hipEvent_t ev;
    HIP_TRY(hipEventCreate(&ev));
    *event = ev;
    return XMHW_OK;
}
int xmhw_event_destroy(void* event) {
    if (event) HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(event)));
    return XMHW_OK;
}
int xmhw_event_record(void* event, void* stream) {
    HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
    return XMHW_OK;
}
int xmhw_stream_wait_event(void* stream, void* event) {
    HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
    return XMHW_OK;
}
int xmhw_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (!ms) return fail(XMHW_ERR_INVALID, "ms is NULL");
    HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
    HIP_TRY(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
    return XMHW_OK;
}

namespace {
// the layouts this build instantiates (include/xmhw_amd.h: XMHW_LAYOUT_*)
bool layout_compiled(int32_t layout) {
    switch (layout) {
        case -2: case -1: case 8: case 10: case 12: case 20: case 21: case 22: case XMHW_LAYOUT_SORTED: return true;
#ifdef XMHW_RING4
        case 30: case 31: case 32: return true;
#endif
        default: return false;
    }
}
}  // namespace

int xmhw_plan_create(const int32_t* doy_host, int64_t T, int32_t window_half_width, xmhw_plan** plan) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    *plan = nullptr;
    if (!doy_host) return fail(XMHW_ERR_INVALID, "doy is NULL");
    xmhw_plan* p = new (std::nothrow) xmhw_plan();
    if (!p) return fail(XMHW_ERR_NOMEM, "out of host memory");
    if (!p->host.build(doy_host, T, window_half_width)) {
        std::string msg = p->host.error;
        delete p;
        return fail(XMHW_ERR_INVALID, msg);
    }
    // (the environment sets the default layout of new plans; a number this build does not have is ignored)
    if (const char* v = std::getenv("XMHW_RING2")) {
        const int32_t lay = std::atoi(v);
        if (layout_compiled(lay) && (lay != XMHW_LAYOUT_SORTED || xmhw::sorted_pick_yps(p->host.w, p->host.ntracks) != 0))
            p->ring2_variant = lay;
    }
    *plan = p;
    return XMHW_OK;
}
int xmhw_plan_set_layout(xmhw_plan* plan, int32_t layout) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    if (!layout_compiled(layout))
        return fail(XMHW_ERR_UNSUPPORTED,
                    "layout must be one of the XMHW_LAYOUT_* constants this library was built with: -2 (auto), -1, 8, 10, 12, "
                    "20, 21, 22, 40 (the plain second-generation layouts 0..7, 9, 11 left the build in round 4; 30..32 need "
                    "make RING4=1)");
    if (layout == XMHW_LAYOUT_SORTED && xmhw::sorted_pick_yps(plan->host.w, plan->host.ntracks) == 0)
        return fail(XMHW_ERR_UNSUPPORTED, "the sorted-list kernel is not instantiated for this window / record length");
    plan->ring2_variant = layout;
    return XMHW_OK;
}
int xmhw_sorted_device_ok(int32_t* holds) {
    if (!holds) return fail(XMHW_ERR_INVALID, "NULL argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(XMHW_ERR_HIP, "no HIP device");
    *holds = sorted_device_ok() == 1 ? 1 : 0;
    return XMHW_OK;
}
int xmhw_plan_layout_in_use(const xmhw_plan* plan, int32_t* layout) {
    if (!plan || !layout) return fail(XMHW_ERR_INVALID, "NULL argument");
    if (sorted_usable(plan)) { *layout = XMHW_LAYOUT_SORTED; return XMHW_OK; }
    const int32_t v2 = ring2_resolved(plan);
    const bool ring = resolve_kernel(plan, 4) == XMHW_KERNEL_RING;
    const int32_t y2 = v2 >= 0 ? xmhw::ring2_pick_yps(plan->host.w, plan->host.ntracks, v2) : 0;
    *layout = (ring && y2 > 0 && xmhw::ring2_f32_supported(plan->host.w, y2, v2)) ? v2 : -1;
    return XMHW_OK;
}
// deprecated aliases (rounds 2 and 3)
int xmhw_plan_set_ring2(xmhw_plan* plan, int32_t variant) { return xmhw_plan_set_layout(plan, variant); }
int xmhw_plan_ring2_in_use(const xmhw_plan* plan, int32_t* variant) { return xmhw_plan_layout_in_use(plan, variant); }
int xmhw_plan_f64_mode(const xmhw_plan* plan, int32_t* variant) {
    if (!plan || !variant) return fail(XMHW_ERR_INVALID, "NULL argument");
    *variant = resolve_kernel(plan, 8) == XMHW_KERNEL_RING ? x64_choice(plan).variant : -1;
    return XMHW_OK;
}
int xmhw_plan_destroy(xmhw_plan* plan) {
    delete plan;
    return XMHW_OK;
}
int xmhw_plan_info(const xmhw_plan* plan, int32_t* D, int32_t* ntracks, int32_t* kernel, int32_t* nsteps,
                   int32_t* step_min) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    if (D) *D = plan->host.D;
    if (ntracks) *ntracks = plan->host.ntracks;
    if (kernel) *kernel = resolve_kernel(plan, 4);
    if (nsteps) *nsteps = plan->host.nsteps;
    if (step_min) *step_min = plan->host.step_min;
    return XMHW_OK;
}
int xmhw_plan_doys(const xmhw_plan* plan, int32_t* doys_out) {
    if (!plan || !doys_out) return fail(XMHW_ERR_INVALID, "NULL argument");
    std::memcpy(doys_out, plan->host.doys.data(), sizeof(int32_t) * plan->host.doys.size());
    return XMHW_OK;
}
int xmhw_plan_set_kernel(xmhw_plan* plan, int32_t kernel) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    if (kernel < XMHW_KERNEL_AUTO || kernel > XMHW_KERNEL_GENERIC)
        return fail(XMHW_ERR_INVALID, "unknown kernel selector");
    plan->host.kernel_choice = kernel;
    return XMHW_OK;
}
int xmhw_plan_set_timing(xmhw_plan* plan, int32_t enable) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    plan->timing = enable != 0;
    return XMHW_OK;
}
int xmhw_plan_kernel_ms(xmhw_plan* plan, int32_t calls_back, float* ms) {
    if (!plan || !ms) return fail(XMHW_ERR_INVALID, "NULL argument");
    if (calls_back < 0 || calls_back >= 16 || static_cast<uint64_t>(calls_back) >= plan->tcalls)
        return fail(XMHW_ERR_INVALID, "no such timed call");
    const int slot = static_cast<int>((plan->tcalls - 1 - static_cast<uint64_t>(calls_back)) % 16);
    if (!plan->tev[2 * slot] || !plan->tev[2 * slot + 1]) return fail(XMHW_ERR_INVALID, "no such timed call");
    HIP_TRY(hipEventSynchronize(plan->tev[2 * slot + 1]));
    HIP_TRY(hipEventElapsedTime(ms, plan->tev[2 * slot], plan->tev[2 * slot + 1]));
    return XMHW_OK;
}
int xmhw_plan_set_narrowing(xmhw_plan* plan, int32_t enable) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    plan->narrowing = enable != 0;
    return XMHW_OK;
}
int xmhw_plan_narrowed(xmhw_plan* plan, int32_t* narrowed_out) {
    if (!plan || !narrowed_out) return fail(XMHW_ERR_INVALID, "NULL argument");
    *narrowed_out = 0;
    if (!plan->d_narrow_flag) return XMHW_OK;
    uint32_t flag = 1;
    HIP_TRY(hipMemcpy(&flag, plan->d_narrow_flag, sizeof(uint32_t), hipMemcpyDeviceToHost));
    *narrowed_out = flag == 0 ? 1 : 0;
    return XMHW_OK;
}
int xmhw_plan_set_chunks(xmhw_plan* plan, int32_t nchunks) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    if (nchunks < 0) return fail(XMHW_ERR_INVALID, "nchunks must be >= 0");
    plan->host.nchunks_req = nchunks;
    return XMHW_OK;
}
int xmhw_plan_chunks_in_use(const xmhw_plan* plan, int64_t C, int32_t* nchunks) {
    if (!plan || !nchunks) return fail(XMHW_ERR_INVALID, "NULL argument");
    if (C < 0) return fail(XMHW_ERR_INVALID, "bad C");
    *nchunks = auto_chunks(plan, std::max<int64_t>(C, 1));
    return XMHW_OK;
}
int xmhw_debug_stats_available(void) { return xmhw::ring_stats_built() ? 1 : 0; }
int xmhw_plan_debug_stats_n(xmhw_plan* plan, int enable, uint64_t* out, int32_t n) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    if (out && n <= 0) return fail(XMHW_ERR_INVALID, "n must be > 0");
    if (enable && !plan->d_stats) {
        HIP_TRY(hipMalloc(&plan->d_stats, 16 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(plan->d_stats, 0, 16 * sizeof(unsigned long long)));
    }
    if (out) {
        if (!plan->d_stats) return fail(XMHW_ERR_INVALID, "stats not enabled");
        unsigned long long all[16];
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(all, plan->d_stats, sizeof(all), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(plan->d_stats, 0, sizeof(all)));
        for (int32_t i = 0; i < std::min<int32_t>(n, 16); ++i) out[i] = all[i];
    }
    return XMHW_OK;
}
// (the round-1 contract: 8 values)
int xmhw_plan_debug_stats(xmhw_plan* plan, int enable, uint64_t* out8) {
    return xmhw_plan_debug_stats_n(plan, enable, out8, 8);
}
int xmhw_plan_table(const xmhw_plan* plan, int32_t years_per_lane, uint32_t* table_out,
                    int32_t* ntracks_padded) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    if (years_per_lane <= 0) return fail(XMHW_ERR_INVALID, "years_per_lane must be > 0");
    if (plan->host.ntracks > kSubs * years_per_lane)
        return fail(XMHW_ERR_INVALID, "years_per_lane too small for the number of tracks");
    if (ntracks_padded) *ntracks_padded = kSubs * years_per_lane;
    if (table_out) {
        std::vector<uint32_t> tab = plan->host.ring_table(kSubs, years_per_lane);
        std::memcpy(table_out, tab.data(), sizeof(uint32_t) * tab.size());
    }
    return XMHW_OK;
}

int xmhw_plan_sorted_info(const xmhw_plan* plan, int64_t C, int32_t* keys_per_list, int32_t* lds_bytes_per_wave,
                          int32_t* pieces) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    const int32_t k = xmhw::sorted_pick_k(plan->host.w, plan->host.ntracks);
    if (k == 0) return fail(XMHW_ERR_UNSUPPORTED, "the sorted-list kernel is not instantiated for this window / record length");
    if (keys_per_list) *keys_per_list = k;
    if (lds_bytes_per_wave) *lds_bytes_per_wave = xmhw::sorted_lds_bytes(plan->host.w, plan->host.ntracks);
    if (pieces) {
        const int64_t waves = (std::max<int64_t>(C, 1) + 31) / 32;
        *pieces = static_cast<int32_t>(plan->host.nchunks_req > 0 ? plan->host.nchunks_req : (1536 + waves - 1) / waves);
    }
    return XMHW_OK;
}

int xmhw_plan_sorted_table(const xmhw_plan* plan, int32_t pieces, int32_t* nchunks, int32_t* nrows, int32_t* ntp,
                           int32_t* chunks_out, uint32_t* table_out, uint32_t* flags_out) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    const int32_t yps = xmhw::sorted_pick_yps(plan->host.w, plan->host.ntracks);
    if (yps == 0) return fail(XMHW_ERR_UNSUPPORTED, "the sorted-list kernel is not instantiated for this window / record length");
    const xmhw::Plan::SortedPlan sp = plan->host.sorted_plan(2 * yps, 24, std::max(pieces, 1));
    if (nchunks) *nchunks = static_cast<int32_t>(sp.chunks.size());
    if (nrows) *nrows = static_cast<int32_t>(sp.flags.size());
    if (ntp) *ntp = 2 * yps;
    if (chunks_out)
        for (size_t i = 0; i < sp.chunks.size(); ++i) {
            chunks_out[4 * i + 0] = sp.chunks[i].warm_start;
            chunks_out[4 * i + 1] = sp.chunks[i].begin;
            chunks_out[4 * i + 2] = sp.chunks[i].end;
            chunks_out[4 * i + 3] = sp.chunks[i].trow0;
        }
    if (table_out) std::memcpy(table_out, sp.table.data(), sizeof(uint32_t) * sp.table.size());
    if (flags_out) std::memcpy(flags_out, sp.flags.data(), sizeof(uint32_t) * sp.flags.size());
    return XMHW_OK;
}

int xmhw_clim_raw_f32(xmhw_plan* plan, const float* ts, int64_t C, int64_t ld, double q, int negate,
                      double* thresh, double* seas, int64_t ldo, void* stream) {
    return clim_raw<float>(plan, ts, C, ld, q, negate, thresh, seas, ldo, stream);
}
int xmhw_clim_raw_f64(xmhw_plan* plan, const double* ts, int64_t C, int64_t ld, double q, int negate,
                      double* thresh, double* seas, int64_t ldo, void* stream) {
    return clim_raw<double>(plan, ts, C, ld, q, negate, thresh, seas, ldo, stream);
}

int xmhw_clim_raw_i16(xmhw_plan* plan, const int16_t* codes, int64_t C, int64_t ld, int big_endian, int has_scale,
                      double scale_factor, double add_offset, int has_fill, int32_t fill_code, int decoded_itemsize, double q,
                      int negate, double* thresh, double* seas, int64_t ldo, void* stream) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    if (C < 0 || ld < C || ldo < C) return fail(XMHW_ERR_INVALID, "bad C/ld/ldo");
    if (!(q >= 0.0 && q <= 1.0)) return fail(XMHW_ERR_INVALID, "quantile must be in [0, 1]");
    if (decoded_itemsize != 4 && decoded_itemsize != 8) return fail(XMHW_ERR_INVALID, "decoded_itemsize must be 4 or 8");
    if (has_scale && !(scale_factor == scale_factor && add_offset == add_offset && scale_factor != 0.0))
        return fail(XMHW_ERR_INVALID, "scale_factor must be a non-zero number, add_offset a number");
    if (C == 0) return XMHW_OK;
    if (!codes || !thresh || !seas) return fail(XMHW_ERR_INVALID, "NULL device buffer");
    // the codes are read in place by the sorted-list kernel and its recomputation only: the plans and quantiles those
    // serve (w = 5, 9..48 tracks, quantile >= 0.85).  Anything else: xmhw_decode() + xmhw_clim_raw_f32 / _f64.
    int rc = upload(plan, C);
    if (rc != XMHW_OK) return rc;
    if (!(plan->nchunks_s > 0 && sorted_serves(q)))
        return fail(XMHW_ERR_UNSUPPORTED, "packed input runs on the sorted-list kernel only (w = 5, 9..48 tracks, quantile >= 0.85 or <= 0.15): "
                                         "decode the series (xmhw_decode) and call xmhw_clim_raw_f32 / _f64");
    xmhw::PackedI16 pk;
    pk.swap = big_endian ? 1 : 0;
    pk.fill = has_fill ? fill_code : 0x7FFFFFFF;
    int kneg = negate ? 1 : 0;
    if (!has_scale) {
        pk.mode = 3;
    } else if (decoded_itemsize == 4) {
        // float32 decode: the kernels key and sum float(code) * sf + of, exactly the series xmhw_decode() would write
        pk.mode = 1;
        pk.sf = static_cast<float>(scale_factor);
        pk.of = static_cast<float>(add_offset);
    } else {
        // float64 decode: code -> value is monotone (decreasing for a negative scale_factor: the kernels then key the
        // negated codes); the two selected codes and the mean of the codes are decoded in float64
        pk.mode = 2;
        pk.s = scale_factor;
        pk.o = add_offset;
        pk.val_neg = negate ? 1 : 0;
        kneg = (negate ? 1 : 0) ^ (scale_factor < 0.0 ? 1 : 0);
    }
    pk.key_neg = kneg;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const xmhw::Plan& h = plan->host;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (plan->timing) {
        const int slot = static_cast<int>(plan->tcalls % 16);
        for (int i = 0; i < 2; ++i)
            if (!plan->tev[2 * slot + i]) HIP_TRY(hipEventCreate(&plan->tev[2 * slot + i]));
        t0 = plan->tev[2 * slot];
        t1 = plan->tev[2 * slot + 1];
        plan->tcalls++;
    }
    hipError_t e = hipSuccess;
    if (t0) e = hipEventRecord(t0, st);
    if (e == hipSuccess)
        e = xmhw::launch_sorted_i16(codes, pk, C, ld, h.T, plan->d_table_s, plan->d_sflags_s, plan->d_chunks_s, plan->nchunks_s,
                                    h.w, plan->yps_s, h.ntracks, q, kneg, thresh, seas, ldo, st);
    if (e == hipSuccess && t1) e = hipEventRecord(t1, st);
    if (e != hipSuccess) return hip_fail(e, "packed climatology launch");
    return XMHW_OK;
}

int xmhw_clim_finish(const xmhw_plan* plan, const double* thresh_in, const double* seas_in, int64_t C,
                     int64_t ldo, int feb29_fix, int smooth, int smooth_width, double* thresh_out,
                     double* seas_out, void* stream) {
    if (!plan) return fail(XMHW_ERR_INVALID, "plan is NULL");
    if (smooth && (smooth_width <= 0 || smooth_width % 2 == 0))
        return fail(XMHW_ERR_INVALID, "Running average window should be odd");
    if (C < 0 || ldo < C) return fail(XMHW_ERR_INVALID, "bad C/ldo");
    if (C == 0) return XMHW_OK;
    if (!thresh_in || !seas_in || !thresh_out || !seas_out)
        return fail(XMHW_ERR_INVALID, "NULL device buffer");
    if (thresh_in == thresh_out || seas_in == seas_out)
        return fail(XMHW_ERR_INVALID, "in and out may not alias");
    const xmhw::Plan& h = plan->host;
    void* flags = nullptr;      // per-column "has an absent group" flags of the one-pass finish kernel
    ScratchRef scratch_keep;
    if (smooth && smooth_width == 31) {
        hipError_t se = scratch_get(static_cast<hipStream_t>(stream), static_cast<size_t>(C), &flags, &scratch_keep);
        if (se != hipSuccess) return hip_fail(se, "scratch allocation");
    }
    hipError_t e = xmhw::launch_finish(thresh_in, seas_in, C, ldo, h.D, row_index(h, 59), row_index(h, 60),
                                       row_index(h, 61), feb29_fix, smooth, smooth ? smooth_width : 1,
                                       thresh_out, seas_out, static_cast<hipStream_t>(stream),
                                       static_cast<uint8_t*>(flags));
    if (e != hipSuccess) return hip_fail(e, "clim_finish launch");
    return XMHW_OK;
}

int xmhw_clim_f32(const float* ts, const int32_t* doy, int64_t T, int64_t C, int32_t D, int32_t w,
                  double q, int smooth, int smooth_w, int feb29_fix, int negate, double* thresh,
                  double* seas, void* stream) {
    return clim_oneshot<float>(ts, doy, T, C, D, w, q, smooth, smooth_w, feb29_fix, negate, thresh, seas,
                               stream);
}
int xmhw_clim_f64(const double* ts, const int32_t* doy, int64_t T, int64_t C, int32_t D, int32_t w,
                  double q, int smooth, int smooth_w, int feb29_fix, int negate, double* thresh,
                  double* seas, void* stream) {
    return clim_oneshot<double>(ts, doy, T, C, D, w, q, smooth, smooth_w, feb29_fix, negate, thresh, seas,
                                stream);
}
int xmhw_clim_host_f32(const float* ts, const int32_t* doy, int64_t T, int64_t C, int32_t D, int32_t w,
                       double q, int smooth, int smooth_w, int feb29_fix, int negate, double* thresh,
                       double* seas) {
    return clim_host<float>(ts, doy, T, C, D, w, q, smooth, smooth_w, feb29_fix, negate, thresh, seas);
}
int xmhw_clim_host_f64(const double* ts, const int32_t* doy, int64_t T, int64_t C, int32_t D, int32_t w,
                       double q, int smooth, int smooth_w, int feb29_fix, int negate, double* thresh,
                       double* seas) {
    return clim_host<double>(ts, doy, T, C, D, w, q, smooth, smooth_w, feb29_fix, negate, thresh, seas);
}

int xmhw_land_mask_f32(const float* ts, int64_t T, int64_t C, int64_t ld, int anynans, uint8_t* keep,
                       void* stream) {
    if (C < 0 || ld < C || T <= 0) return fail(XMHW_ERR_INVALID, "bad T/C/ld");
    hipError_t e = xmhw::launch_land_mask<float>(ts, T, C, ld, anynans, keep, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "land_mask launch");
    return XMHW_OK;
}
int xmhw_land_mask_f64(const double* ts, int64_t T, int64_t C, int64_t ld, int anynans, uint8_t* keep,
                       void* stream) {
    if (C < 0 || ld < C || T <= 0) return fail(XMHW_ERR_INVALID, "bad T/C/ld");
    hipError_t e = xmhw::launch_land_mask<double>(ts, T, C, ld, anynans, keep, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "land_mask launch");
    return XMHW_OK;
}

int xmhw_land_mask_i16(const int16_t* codes, int64_t T, int64_t C, int64_t ld, int big_endian, int has_fill,
                       int32_t fill_code, int anynans, uint8_t* keep, void* stream) {
    if (C < 0 || ld < C || T <= 0) return fail(XMHW_ERR_INVALID, "bad T/C/ld");
    if (has_fill && (fill_code < -32768 || fill_code > 32767)) return fail(XMHW_ERR_INVALID, "fill_code is not an int16");
    if (C == 0) return XMHW_OK;
    if (!codes || !keep) return fail(XMHW_ERR_INVALID, "NULL device buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!has_fill) {           // no code means "missing": every cell stays
        hipError_t e = hipMemsetAsync(keep, 1, static_cast<size_t>(C), st);
        if (e != hipSuccess) return hip_fail(e, "land_mask memset");
        return XMHW_OK;
    }
    uint16_t raw = static_cast<uint16_t>(static_cast<int16_t>(fill_code));
    if (big_endian) raw = static_cast<uint16_t>((raw >> 8) | (raw << 8));
    hipError_t e = xmhw::launch_land_mask_i16(codes, T, C, ld, static_cast<int16_t>(raw), anynans, keep, st);
    if (e != hipSuccess) return hip_fail(e, "land_mask launch");
    return XMHW_OK;
}
int xmhw_gather_cells_i16(const int16_t* in, int64_t rows, int64_t ld_in, const int64_t* index, int64_t n,
                          int16_t* out, int64_t ld_out, void* stream) {
    if (rows < 0 || n < 0 || ld_out < n) return fail(XMHW_ERR_INVALID, "bad rows/n/ld_out");
    hipError_t e = xmhw::launch_gather_cells<int16_t>(in, rows, ld_in, index, n, out, ld_out,
                                                      static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "gather_cells launch");
    return XMHW_OK;
}
int xmhw_gather_cells_f32(const float* in, int64_t rows, int64_t ld_in, const int64_t* index, int64_t n,
                          float* out, int64_t ld_out, void* stream) {
    if (rows < 0 || n < 0 || ld_out < n) return fail(XMHW_ERR_INVALID, "bad rows/n/ld_out");
    hipError_t e = xmhw::launch_gather_cells<float>(in, rows, ld_in, index, n, out, ld_out,
                                                    static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "gather_cells launch");
    return XMHW_OK;
}
int xmhw_gather_cells_f64(const double* in, int64_t rows, int64_t ld_in, const int64_t* index, int64_t n,
                          double* out, int64_t ld_out, void* stream) {
    if (rows < 0 || n < 0 || ld_out < n) return fail(XMHW_ERR_INVALID, "bad rows/n/ld_out");
    hipError_t e = xmhw::launch_gather_cells<double>(in, rows, ld_in, index, n, out, ld_out,
                                                     static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "gather_cells launch");
    return XMHW_OK;
}
int xmhw_scatter_cells_f64(const double* in, int64_t rows, int64_t ld_in, const int64_t* index, int64_t n,
                           double* out, int64_t ld_out, void* stream) {
    if (rows < 0 || n < 0 || ld_in < n) return fail(XMHW_ERR_INVALID, "bad rows/n/ld_in");
    hipError_t e = xmhw::launch_scatter_cells(in, rows, ld_in, index, n, out, ld_out, ld_out,
                                              static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "scatter_cells launch");
    return XMHW_OK;
}

int xmhw_detect_events_f32(const float* ts, int64_t T, int64_t C, int64_t ld, const double* thresh, int64_t ldt,
                           const int32_t* row_of_t, int32_t min_duration, int32_t join_gaps, int32_t max_gap,
                           int32_t negate, int32_t* events, int32_t* start, int32_t* end, uint8_t* bthresh,
                           int64_t ldo, int32_t* nevents, void* stream) {
    return detect_events<float>(ts, T, C, ld, thresh, ldt, row_of_t, min_duration, join_gaps, max_gap, negate,
                                events, start, end, bthresh, ldo, nevents, stream);
}
int xmhw_detect_events_f64(const double* ts, int64_t T, int64_t C, int64_t ld, const double* thresh, int64_t ldt,
                           const int32_t* row_of_t, int32_t min_duration, int32_t join_gaps, int32_t max_gap,
                           int32_t negate, int32_t* events, int32_t* start, int32_t* end, uint8_t* bthresh,
                           int64_t ldo, int32_t* nevents, void* stream) {
    return detect_events<double>(ts, T, C, ld, thresh, ldt, row_of_t, min_duration, join_gaps, max_gap, negate,
                                 events, start, end, bthresh, ldo, nevents, stream);
}

int xmhw_count_events(const int32_t* start, int64_t T, int64_t C, int64_t ldo, int32_t* nevents, void* stream) {
    if (T <= 0 || C < 0 || ldo < C) return fail(XMHW_ERR_INVALID, "bad T/C/ldo");
    if (C == 0) return XMHW_OK;
    if (!start || !nevents) return fail(XMHW_ERR_INVALID, "NULL buffer");
    hipError_t e = xmhw::launch_count_events(start, T, C, ldo, nevents, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "count_events launch");
    return XMHW_OK;
}
int xmhw_event_stats_f32(const float* ts, int64_t T, int64_t C, int64_t ld, const double* seas,
                         const double* thresh, int64_t ldc, const int32_t* row_of_t, int32_t negate,
                         const int32_t* events, int64_t ldo, const int64_t* offsets, double* table, void* stream) {
    return event_stats<float>(ts, T, C, ld, seas, thresh, ldc, row_of_t, negate, events, ldo, offsets, table, stream);
}
int xmhw_release_cached_tables(void) {
    release_cached_tables();
    return XMHW_OK;
}
int xmhw_offsets_from_counts(const int32_t* counts_dev, int64_t n, int64_t* offsets_dev, void* stream) {
    if (n < 0) return fail(XMHW_ERR_INVALID, "n must be >= 0");
    if (!offsets_dev || (n > 0 && !counts_dev)) return fail(XMHW_ERR_INVALID, "NULL buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    void* sp = nullptr;
    ScratchRef scratch_keep;
    const int64_t nblocks = (n + 1023) / 1024;
    hipError_t e = scratch_get(st, sizeof(int64_t) * static_cast<size_t>(nblocks + 1), &sp, &scratch_keep);
    if (e != hipSuccess) return hip_fail(e, "scratch allocation");
    e = xmhw::launch_offsets_from_counts(counts_dev, n, offsets_dev, static_cast<int64_t*>(sp), st);
    if (e != hipSuccess) return hip_fail(e, "offsets_from_counts launch");
    return XMHW_OK;
}
int xmhw_set_exceed_kernel(int32_t mode) {
    if (mode < 0 || mode > 2) return fail(XMHW_ERR_INVALID, "mode must be 0 (auto), 1 (per-step) or 2 (tiled)");
    g_exceed_kernel = mode;
    return XMHW_OK;
}
int xmhw_exceed_bits_f32(const float* ts, int64_t T, int64_t C, int64_t ld, const double* thresh, int64_t ldt,
                         int64_t D, const int32_t* row_of_t, int32_t negate, uint64_t* bits, int64_t ldb,
                         void* stream) {
    return exceed_bits<float>(ts, T, C, ld, thresh, ldt, D, row_of_t, negate, bits, ldb, stream);
}
int xmhw_exceed_bits_f64(const double* ts, int64_t T, int64_t C, int64_t ld, const double* thresh, int64_t ldt,
                         int64_t D, const int32_t* row_of_t, int32_t negate, uint64_t* bits, int64_t ldb,
                         void* stream) {
    return exceed_bits<double>(ts, T, C, ld, thresh, ldt, D, row_of_t, negate, bits, ldb, stream);
}
int xmhw_events_from_bits(const uint64_t* bits, int64_t T, int64_t C, int64_t ldb, int32_t min_duration,
                          int32_t join_gaps, int32_t max_gap, const int64_t* offsets, int32_t* nevents,
                          double* table, void* stream) {
    if (T <= 0 || C < 0 || ldb < C) return fail(XMHW_ERR_INVALID, "bad T/C/ldb");
    if (C == 0) return XMHW_OK;
    if (!bits || (!offsets && !nevents) || (offsets && !table)) return fail(XMHW_ERR_INVALID, "NULL buffer");
    if (min_duration < 1 || max_gap < 0) return fail(XMHW_ERR_INVALID, "bad minDuration/maxGap");
    hipError_t e = xmhw::launch_events_from_bits(bits, T, C, ldb, min_duration, join_gaps, max_gap, offsets,
                                                 nevents, table, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "events_from_bits launch");
    return XMHW_OK;
}
int xmhw_event_stats_sparse_f32(const float* ts, int64_t T, int64_t C, int64_t ld, const double* seas,
                                const double* thresh, int64_t ldc, const int32_t* row_of_t, int32_t negate,
                                int64_t n_events, double* table, void* stream) {
    return event_stats_sparse<float>(ts, T, C, ld, seas, thresh, ldc, row_of_t, negate, n_events, table, stream);
}
int xmhw_event_stats_sparse_f64(const double* ts, int64_t T, int64_t C, int64_t ld, const double* seas,
                                const double* thresh, int64_t ldc, const int32_t* row_of_t, int32_t negate,
                                int64_t n_events, double* table, void* stream) {
    return event_stats_sparse<double>(ts, T, C, ld, seas, thresh, ldc, row_of_t, negate, n_events, table, stream);
}
int xmhw_event_intermediate_f32(const float* ts, int64_t T, int64_t C, int64_t ld, const double* seas,
                                const double* thresh, int64_t ldc, const int32_t* row_of_t, int32_t negate,
                                const int32_t* events, int64_t ldo, double* out, int64_t ldv, uint8_t* dur,
                                void* stream) {
    return event_intermediate<float>(ts, T, C, ld, seas, thresh, ldc, row_of_t, negate, events, ldo, out, ldv, dur,
                                     stream);
}
int xmhw_event_intermediate_f64(const double* ts, int64_t T, int64_t C, int64_t ld, const double* seas,
                                const double* thresh, int64_t ldc, const int32_t* row_of_t, int32_t negate,
                                const int32_t* events, int64_t ldo, double* out, int64_t ldv, uint8_t* dur,
                                void* stream) {
    return event_intermediate<double>(ts, T, C, ld, seas, thresh, ldc, row_of_t, negate, events, ldo, out, ldv, dur,
                                      stream);
}
int xmhw_event_stats_f64(const double* ts, int64_t T, int64_t C, int64_t ld, const double* seas,
                         const double* thresh, int64_t ldc, const int32_t* row_of_t, int32_t negate,
                         const int32_t* events, int64_t ldo, const int64_t* offsets, double* table, void* stream) {
    return event_stats<double>(ts, T, C, ld, seas, thresh, ldc, row_of_t, negate, events, ldo, offsets, table, stream);
}

int xmhw_block_events(const double* table, const int64_t* offsets, int64_t C, const int32_t* bin_of_t, int64_t T,
                      int32_t nbins, int32_t mtime_column, double* out, int64_t ldo, void* stream) {
    if (C < 0 || T <= 0 || nbins <= 0 || ldo < C) return fail(XMHW_ERR_INVALID, "bad C/T/nbins/ldo");
    if (mtime_column < 0 || mtime_column >= xmhw::kEventColumns) return fail(XMHW_ERR_INVALID, "bad mtime column");
    if (C == 0) return XMHW_OK;
    if (!table || !offsets || !bin_of_t || !out) return fail(XMHW_ERR_INVALID, "NULL buffer");
    hipError_t e = xmhw::launch_block_events(table, offsets, C, bin_of_t, T, nbins, mtime_column, out, ldo,
                                             static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "block_events launch");
    return XMHW_OK;
}
int xmhw_block_time_f32(const float* ts, int64_t T, int64_t C, int64_t ld, const double* cats, int64_t ldcat,
                        const int32_t* bin_of_t, int32_t nbins, double* out, int64_t ldo, void* stream) {
    return block_time<float>(ts, T, C, ld, cats, ldcat, bin_of_t, nbins, out, ldo, stream);
}
int xmhw_block_time_f64(const double* ts, int64_t T, int64_t C, int64_t ld, const double* cats, int64_t ldcat,
                        const int32_t* bin_of_t, int32_t nbins, double* out, int64_t ldo, void* stream) {
    return block_time<double>(ts, T, C, ld, cats, ldcat, bin_of_t, nbins, out, ldo, stream);
}

int xmhw_synth_sst_f32(float* ts, int64_t T, int64_t C, int64_t ld, int64_t cell0, uint64_t seed,
                       double nan_frac, void* stream) {
    if (C < 0 || ld < C || T <= 0) return fail(XMHW_ERR_INVALID, "bad T/C/ld");
    hipError_t e = xmhw::launch_synth<float>(ts, T, C, ld, cell0, seed, nan_frac, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "synth launch");
    return XMHW_OK;
}
int xmhw_synth_sst_ex_f32(float* ts, int64_t T, int64_t C, int64_t ld, int64_t cell0, uint64_t seed, double nan_frac,
                          double quant, double ice_frac, double rho, int64_t ice_patch, void* stream) {
    if (!ts || T < 0 || C < 0 || ld < C) return fail(XMHW_ERR_INVALID, "bad argument");
    if (!(rho >= 0.0 && rho < 1.0) || quant < 0.0 || ice_frac < 0.0 || ice_frac > 1.0)
        return fail(XMHW_ERR_INVALID, "rho must be in [0, 1), quant >= 0, ice_frac in [0, 1]");
    hipError_t e = xmhw::launch_synth_ex<float>(ts, T, C, ld, cell0, seed, nan_frac, quant, ice_frac, rho, ice_patch,
                                                static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "synth launch");
    return XMHW_OK;
}
int xmhw_synth_sst_f64(double* ts, int64_t T, int64_t C, int64_t ld, int64_t cell0, uint64_t seed,
                       double nan_frac, void* stream) {
    if (C < 0 || ld < C || T <= 0) return fail(XMHW_ERR_INVALID, "bad T/C/ld");
    hipError_t e = xmhw::launch_synth<double>(ts, T, C, ld, cell0, seed, nan_frac, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "synth launch");
    return XMHW_OK;
}

}  // extern "C"
