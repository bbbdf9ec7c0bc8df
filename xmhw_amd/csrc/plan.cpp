// plan.cpp -- host-side plan builder (see plan.h).
#include "plan.h"

#include <algorithm>

namespace xmhw {

bool Plan::build(const int32_t* doy, int64_t T_, int32_t w_) {
    if (T_ <= 0) { error = "time axis is empty"; return false; }
    if (w_ < 0) { error = "windowHalfWidth must be >= 0"; return false; }
    if (T_ > (int64_t(1) << 30) - 4) { error = "time axis too long"; return false; }
    T = T_;
    w = w_;
    R = 2 * w + 1;
    doys.assign(doy, doy + T);
    std::sort(doys.begin(), doys.end());
    doys.erase(std::unique(doys.begin(), doys.end()), doys.end());
    D = static_cast<int32_t>(doys.size());
    row_of_t.resize(T);
    for (int64_t t = 0; t < T; ++t)
        row_of_t[t] = static_cast<int32_t>(std::lower_bound(doys.begin(), doys.end(), doy[t]) - doys.begin());
    // tracks: maximal runs of strictly increasing label
    track_begin.clear();
    track_end.clear();
    track_begin.push_back(0);
    for (int64_t t = 1; t < T; ++t) {
        if (doy[t] <= doy[t - 1]) {
            track_end.push_back(t);
            track_begin.push_back(t);
        }
    }
    track_end.push_back(T);
    ntracks = static_cast<int32_t>(track_begin.size());
    // CSR of centres per row (ascending t inside a row)
    row_ptr.assign(D + 1, 0);
    for (int64_t t = 0; t < T; ++t) row_ptr[row_of_t[t] + 1]++;
    max_centres = 0;
    for (int32_t r = 0; r < D; ++r) {
        max_centres = std::max(max_centres, row_ptr[r + 1]);
        row_ptr[r + 1] += row_ptr[r];
    }
    centres.resize(T);
    std::vector<int32_t> fill(row_ptr.begin(), row_ptr.end() - 1);
    for (int64_t t = 0; t < T; ++t) centres[fill[row_of_t[t]]++] = static_cast<int32_t>(t);
    step_min = -(R - 1);
    nsteps = D + R - 1;
    return true;
}

std::vector<uint32_t> Plan::step_flags() const {
    const std::vector<uint32_t> tab = ring_table(1, ntracks);      // [step][track]
    std::vector<uint32_t> f(static_cast<size_t>(nsteps), 0u);
    for (int32_t i = 0; i < nsteps; ++i) {
        bool simple = true, consec = i > 0;
        for (int32_t k = 0; k < ntracks; ++k) {
            const uint32_t e = tab[static_cast<size_t>(i) * ntracks + k];
            simple = simple && (e & 1u) && (e >> 1) >= 2u;
            if (i > 0) {
                const uint32_t p = tab[static_cast<size_t>(i - 1) * ntracks + k];
                consec = consec && (e >> 1) >= 2u && (p >> 1) >= 2u && (e >> 1) == (p >> 1) + 1u;
            }
        }
        f[i] = (simple ? 1u : 0u) | (consec ? 2u : 0u);
    }
    return f;
}

std::vector<uint32_t> Plan::ring_table(int32_t subs, int32_t yps) const {
    const int32_t ntp = subs * yps;
    std::vector<uint32_t> tab(static_cast<size_t>(nsteps) * ntp, make_entry(kCodeInvalid, true));
    if (ntracks > ntp) return {};
    std::vector<int64_t> centre_at_row(D);
    for (int32_t k = 0; k < ntracks; ++k) {
        const int64_t tb = track_begin[k], te = track_end[k];
        std::fill(centre_at_row.begin(), centre_at_row.end(), int64_t(-1));
        for (int64_t t = tb; t < te; ++t) centre_at_row[row_of_t[t]] = t;
        const int32_t f = row_of_t[tb], l = row_of_t[te - 1];
        for (int32_t i = 0; i < nsteps; ++i) {
            const int32_t s = step_min + i;
            uint32_t e;
            if (s < f) {
                // warm-up towards the first centre: virtual centre tb-(f-s)
                const int64_t ld = tb - (f - s) + w;
                if (f - s <= R - 1 && ld >= 0 && ld < T)
                    e = make_entry(static_cast<uint32_t>(ld + 2), false);
                else
                    e = make_entry(kCodeInvalid, false);
            } else if (s <= l) {
                const int64_t t = centre_at_row[s];
                if (t >= 0) {
                    const int64_t ld = t + w;
                    e = make_entry(ld < T ? static_cast<uint32_t>(ld + 2) : kCodeInvalid, true);
                } else {
                    e = make_entry(kCodeHold, false);
                }
            } else {
                e = make_entry(kCodeInvalid, false);
            }
            tab[static_cast<size_t>(i) * ntp + k] = e;
        }
    }
    return tab;
}

int32_t Plan::warm_start_for(int32_t begin) const {
    if (R == 1) return begin;
    int32_t ws = begin;
    for (int32_t k = 0; k < ntracks; ++k) {
        const int64_t tb = track_begin[k], te = track_end[k];
        const int32_t f = row_of_t[tb], l = row_of_t[te - 1];
        // rows of [f, l] at which the track has no centre are HOLD steps: they push nothing
        std::vector<uint8_t> present(static_cast<size_t>(std::max(l - f + 1, 0)), 0);
        for (int64_t t = tb; t < te; ++t) present[row_of_t[t] - f] = 1;
        int32_t pushes = 0, s = begin;
        while (pushes < R - 1 && s > step_min) {
            --s;
            const bool hold = s >= f && s <= l && !present[s - f];
            if (!hold) ++pushes;
        }
        ws = std::min(ws, s);
    }
    return std::max(ws, step_min);
}

std::vector<Chunk> Plan::make_chunks(int32_t nchunks) const {
    nchunks = std::max(1, std::min(nchunks, D));
    std::vector<Chunk> out;
    for (int32_t j = 0; j < nchunks; ++j) {
        Chunk c;
        c.begin = static_cast<int32_t>(int64_t(D) * j / nchunks);
        c.end = static_cast<int32_t>(int64_t(D) * (j + 1) / nchunks);
        if (c.end <= c.begin) continue;
        c.warm_start = warm_start_for(c.begin);
        out.push_back(c);
    }
    return out;
}

std::vector<Plan::Segment> Plan::sorted_segments() const {
    const std::vector<uint32_t> tab = ring_table(1, ntracks);      // [step][track]
    std::vector<uint8_t> regular(static_cast<size_t>(D), 1);
    for (int32_t s = 0; s < D; ++s) {
        bool ok = true;
        for (int32_t k = 0; k < ntracks && ok; ++k) {
            bool any_valid = false;
            for (int32_t j = s - (R - 1); j <= s; ++j) {
                const uint32_t e = tab[static_cast<size_t>(j - step_min) * ntracks + k];
                if ((e >> 1) == kCodeHold) ok = false;
                any_valid = any_valid || (e >> 1) >= 2u;
            }
            const uint32_t es = tab[static_cast<size_t>(s - step_min) * ntracks + k];
            if (any_valid && !(es & 1u)) ok = false;
        }
        regular[s] = ok ? 1 : 0;
    }
    std::vector<Segment> out;
    for (int32_t s = 0; s < D;) {
        int32_t e = s;
        while (e < D && regular[e] == regular[s]) ++e;
        out.push_back({s, e, regular[s] != 0});
        s = e;
    }
    return out;
}

Plan::SortedPlan Plan::sorted_plan(int32_t ntp, int32_t min_rows_per_piece, int64_t pieces_wanted) const {
    SortedPlan out;
    if (ntracks > ntp || D <= 0) return out;
    const std::vector<uint32_t> tab = ring_table(1, ntracks);      // [step][track]
    auto orig = [&](int32_t step, int32_t k) -> uint32_t {
        return tab[static_cast<size_t>(step - step_min) * ntracks + k];
    };
    // the tracks that are part of row s
    auto counted = [&](int32_t s, int32_t k) -> bool { return (orig(s, k) & 1u) != 0; };
    std::vector<int32_t> cuts;                                     // first rows of the chunks
    for (int32_t s = 0; s < D; ++s) {
        bool cut = s == 0;
        for (int32_t k = 0; k < ntracks && !cut; ++k) cut = counted(s, k) != counted(s - 1, k);
        if (cut) cuts.push_back(s);
    }
    cuts.push_back(D);
    const int64_t total_rows = D;
    for (size_t ci = 0; ci + 1 < cuts.size(); ++ci) {
        const int32_t cb = cuts[ci], ce = cuts[ci + 1];
        std::vector<uint8_t> inS(static_cast<size_t>(ntracks));
        for (int32_t k = 0; k < ntracks; ++k) inS[k] = counted(cb, k) ? 1 : 0;
        // (a track of S pushes at every row of the chunk: it has a centre there)
        const int32_t len = ce - cb;
        int64_t pieces = std::max<int64_t>(1, pieces_wanted * len / std::max<int64_t>(total_rows, 1));
        pieces = std::max<int64_t>(1, std::min<int64_t>(pieces, len / std::max(min_rows_per_piece, 1)));
        for (int64_t pj = 0; pj < pieces; ++pj) {
            const int32_t b = cb + static_cast<int32_t>(len * pj / pieces);
            const int32_t e = cb + static_cast<int32_t>(len * (pj + 1) / pieces);
            if (e <= b) continue;
            SortedChunk ch;
            ch.warm_start = b - (R - 1);
            ch.begin = b;
            ch.end = e;
            ch.trow0 = static_cast<int32_t>(out.flags.size());
            const int32_t nrows = e - ch.warm_start;
            const size_t base = out.table.size();
            out.table.resize(base + static_cast<size_t>(nrows) * ntp, make_entry(kCodeInvalid, true));
            for (int32_t k = 0; k < ntracks; ++k) {
                if (!inS[k]) continue;
                // output rows: the plan's own entries
                for (int32_t s = b; s < e; ++s)
                    out.table[base + static_cast<size_t>(s - ch.warm_start) * ntp + k] = orig(s, k);
                // warm-up: the R-1 last pushes before row b, held steps skipped
                int32_t src = b;
                for (int32_t i = 1; i <= R - 1; ++i) {
                    uint32_t ent = make_entry(kCodeInvalid, true);      // (nothing left before the table: pushes nothing)
                    while (src > step_min) {
                        --src;
                        const uint32_t o = orig(src, k);
                        if ((o >> 1) != kCodeHold) { ent = o | 1u; break; }
                    }
                    out.table[base + static_cast<size_t>(b - i - ch.warm_start) * ntp + k] = ent;
                }
            }
            for (int32_t v = 0; v < nrows; ++v) {
                bool simple = true, consec = v > 0;
                for (int32_t k = 0; k < ntracks; ++k) {
                    const uint32_t en = out.table[base + static_cast<size_t>(v) * ntp + k];
                    simple = simple && (en >> 1) >= 2u;
                    if (v > 0) {
                        const uint32_t pr = out.table[base + static_cast<size_t>(v - 1) * ntp + k];
                        consec = consec && (en >> 1) >= 2u && (pr >> 1) >= 2u && (en >> 1) == (pr >> 1) + 1u;
                    }
                }
                out.flags.push_back((simple ? 1u : 0u) | (consec ? 2u : 0u));
            }
            out.chunks.push_back(ch);
        }
    }
    return out;
}

}  // namespace xmhw
