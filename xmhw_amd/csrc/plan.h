// plan.h -- host-side plan: everything the kernels need that depends only on
// the doy labels (the reference's window_roll()/groupby("doy") bookkeeping,
// xmhw/identify.py:184-209, :233, :263).  Pure C++, no HIP calls here.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace xmhw {

// Ring-kernel step table -------------------------------------------------
// The time axis is cut into TRACKS: maximal runs of strictly increasing doy
// label (one calendar year each on a daily axis; one cycle on a tstep axis).
// Inside a track consecutive time steps have increasing labels, so when the
// kernel walks the rows (distinct labels, ascending) a track's pool window
// [t-w, t+w] slides by exactly one sample per row at which the track has a
// centre.  Per (step, track) one 32-bit entry:
//   bit 0      counted: the track has a centre at this row -> its window is
//              part of this row's pool
//   bits 1..31 code: 0 = HOLD (track has no centre at this row but has one
//              before and after, e.g. doy 60 in a non-leap year: keep the
//              window, do not advance);
//              1 = PUSH an invalid sample (outside [0,T) or padding);
//              >=2 = PUSH sample t = code - 2.
// Steps run from step_min = -(2w) (window warm-up before row 0) to D-1.
constexpr uint32_t kCodeHold = 0;
constexpr uint32_t kCodeInvalid = 1;
constexpr uint32_t make_entry(uint32_t code, bool counted) { return (code << 1) | (counted ? 1u : 0u); }

struct Chunk {
    int32_t warm_start;  // first step executed (ring warm-up, no output)
    int32_t begin;       // first row with output
    int32_t end;         // one past the last row with output
};

struct Plan {
    int64_t T = 0;
    int32_t w = 0;
    int32_t R = 1;  // 2w+1
    int32_t D = 0;
    std::vector<int32_t> doys;       // [D] distinct labels ascending
    std::vector<int32_t> row_of_t;   // [T]
    int32_t ntracks = 0;
    std::vector<int64_t> track_begin, track_end;  // [ntracks] time ranges
    // generic kernel: CSR of centres per row
    std::vector<int32_t> row_ptr;    // [D+1]
    std::vector<int32_t> centres;    // [T]
    int32_t max_centres = 0;         // max centres in a row
    int32_t step_min = 0;            // -(R-1)
    int32_t nsteps = 0;              // D + R - 1
    int32_t kernel_choice = 0;       // XMHW_KERNEL_* requested (0 auto)
    int32_t nchunks_req = 0;         // 0 auto

    std::string error;

    bool build(const int32_t* doy, int64_t T, int32_t w);
    // table[nsteps][ntp], ntp = subs * yps >= ntracks; track k -> lane group
    // (sub = k / yps, slot = k % yps)
    // (the table is [step][track]; kernels_ring2.hip reads it y-major instead: track k -> sub k % subs,
    // slot k / subs, so that padding only ever sits in the last slot of a lane)
    std::vector<uint32_t> ring_table(int32_t subs, int32_t yps) const;
    // per step: bit 0 = SIMPLE (every real track pushes a valid sample and is part of the pool);
    // bit 1 = CONSEC (every real track pushes the sample following the one it pushed at the previous step)
    std::vector<uint32_t> step_flags() const;
    std::vector<Chunk> make_chunks(int32_t nchunks) const;
    // first step a ring kernel must execute so that every track has pushed R-1 samples before row `begin`
    int32_t warm_start_for(int32_t begin) const;
    // Rows the sorted-list kernel (kernels_sorted.hip) can serve.  Its pool is the union of the R last ROW-LISTS
    // (what all tracks pushed at one step), so row s is REGULAR iff for every real track no step in [s-R+1, s] is a
    // HOLD and a track that pushed a valid sample in that span is part of the row's pool.  Returns the maximal runs
    // of rows, alternating, in ascending order.
    struct Segment { int32_t begin, end; bool regular; };
    std::vector<Segment> sorted_segments() const;
    // The sorted-list kernel's own chunks and step table.  The row axis is cut wherever the set of tracks that are part
    // of the pool changes (a held step -- doy 60 in the non-leap years --, the first / last centre of a partial year);
    // inside a chunk that set S is constant and every track of S pushes at every row.  A chunk gets its OWN table rows
    // (warm-up + output rows): a track of S warms up with its R-1 last pushes before the chunk (held steps skipped, so
    // that its window is what the reference pools), a track outside S pushes nothing.  Every row of the plan is served.
    // `pieces` > 1 cuts long chunks further (small grids).  table: [rows][ntp] entries (track k at index k), flags: [rows]
    // (bit 0 SIMPLE, bit 1 CONSEC as step_flags()), chunk.trow0 = the chunk's first table row (that of its warm_start).
    struct SortedChunk { int32_t warm_start, begin, end, trow0; };
    struct SortedPlan {
        std::vector<SortedChunk> chunks;
        std::vector<uint32_t> table, flags;
    };
    SortedPlan sorted_plan(int32_t ntp, int32_t min_rows_per_piece, int64_t pieces_wanted) const;
};

}  // namespace xmhw
