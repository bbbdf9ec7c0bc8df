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

std::vector<Chunk> Plan::make_chunks(int32_t nchunks) const {
    nchunks = std::max(1, std::min(nchunks, D));
    // per (row, track) hold flags decide how far back a chunk must warm up:
    // every track needs R-1 PUSH steps before the chunk's first row.
    std::vector<std::vector<uint8_t>> hold(ntracks, std::vector<uint8_t>(D, 0));
    for (int32_t k = 0; k < ntracks; ++k) {
        const int64_t tb = track_begin[k], te = track_end[k];
        const int32_t f = row_of_t[tb], l = row_of_t[te - 1];
        std::vector<uint8_t> present(D, 0);
        for (int64_t t = tb; t < te; ++t) present[row_of_t[t]] = 1;
        for (int32_t r = f; r <= l; ++r) hold[k][r] = !present[r];
    }
    std::vector<Chunk> out;
    for (int32_t j = 0; j < nchunks; ++j) {
        Chunk c;
        c.begin = static_cast<int32_t>(int64_t(D) * j / nchunks);
        c.end = static_cast<int32_t>(int64_t(D) * (j + 1) / nchunks);
        if (c.end <= c.begin) continue;
        int32_t ws = c.begin;
        for (int32_t k = 0; k < ntracks; ++k) {
            int32_t pushes = 0, s = c.begin;
            while (pushes < R - 1 && s > step_min) {
                --s;
                if (s < 0 || !hold[k][s]) ++pushes;
            }
            ws = std::min(ws, s);
        }
        if (R == 1) ws = c.begin;
        c.warm_start = std::max(ws, step_min);
        out.push_back(c);
    }
    return out;
}

}  // namespace xmhw
