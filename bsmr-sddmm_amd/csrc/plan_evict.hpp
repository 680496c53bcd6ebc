// Eviction of outlier entries from the dense part.
//
// The default dense layout stores a block's destinations as 8-bit offsets into a window of < 255 entries of every row
// (csrc/plan_pack.hpp).  ONE (block, row) whose entries span 255 or more positions of P - a long row with many residue
// entries between two of the panel's dense columns - used to flip the whole plan to direct 16-bit offsets: twice the tile
// bytes and the slower store path for every block, because of a handful of entries.  Here those entries leave the dense
// part instead: per (new block, row) the densest window of < 255 positions stays, what lies outside becomes residue.  The
// dense / sparse assignment of these entries changes (like promotion and folding, the plan's own moves; values and
// destinations do not); the panel's column list does not, so the blocks the packers cut afterwards are the ones looked at
// here.  Only outliers: when more than 1 / 16 of the dense entries would have to go (unsorted CSR rows), the RPHM is left
// as it is and the plan takes the direct offsets as before.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "bsmr_hip.h"
#include "plan_pack.hpp"

namespace bsmr {

struct EvictedRphm {
    std::vector<uint32_t> blockValues, sparseOffsets, sparseValues, sparseRows, sparseCols;
    bsmr_rphm_desc desc{};
    uint64_t evicted = 0;   // 0: nothing done, `desc` not filled
};

// d: host arrays, one panel per group (the default layout).  Returns a bsmr_hip.h status.
inline int evictWideRows(const bsmr_rphm_desc* d, EvictedRphm& out) {
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const uint32_t P = d->num_row_panels;
    const uint64_t numBlocks = d->block_offsets[P];
    out.evicted = 0;
    if (P == 0 || numBlocks == 0) return BSMR_OK;
    struct Gone {
        uint32_t col, row, value;
        uint64_t at;   // index into block_values
    };
    std::vector<std::vector<Gone>> gone(P);
    std::vector<uint64_t> denseOfWorker(packThreads(), 0);
    parallelByWeight(P, d->block_offsets, 256, [&](size_t q0, size_t q1, size_t w) {
        std::vector<std::pair<uint32_t, uint32_t>> cols;   // (column id, slot in the panel)
        for (size_t q = q0; q < q1; ++q) {
            const uint64_t first = d->block_offsets[q], count = d->block_offsets[q + 1] - first;
            cols.clear();
            for (uint64_t t = 0; t < count * 16; ++t) {
                const uint32_t c = d->dense_cols[first * 16 + t];
                if (c < d->N) cols.push_back({c, (uint32_t)t});
            }
            std::sort(cols.begin(), cols.end());
            bool twice = false;
            for (size_t u = 1; u < cols.size(); ++u) twice |= cols[u].first == cols[u - 1].first;
            if (twice) continue;   // (a column listed twice: the packer merges its slots; not looked at here)
            for (size_t u0 = 0; u0 < cols.size(); u0 += 16) {
                const size_t u1 = std::min(cols.size(), u0 + 16);
                for (uint32_t r = 0; r < 16; ++r) {
                    uint32_t v[16], slot[16], n = 0;
                    for (size_t u = u0; u < u1; ++u) {
                        const uint32_t t = cols[u].second;
                        const uint64_t at = (first + t / 16) * 256 + r * 16 + t % 16;
                        if (d->block_values[at] == kNone) continue;
                        v[n] = d->block_values[at];
                        slot[n] = (uint32_t)u;
                        ++n;
                    }
                    denseOfWorker[w] += n;
                    if (n < 2) continue;
                    uint32_t lo = v[0], hi = v[0];
                    for (uint32_t i = 1; i < n; ++i) { lo = std::min(lo, v[i]); hi = std::max(hi, v[i]); }
                    if (hi - lo < kWindowMax) continue;
                    // the window [start, start + kWindowMax) that keeps the most entries; the earliest among equals
                    uint32_t bestStart = lo, bestKept = 0;
                    for (uint32_t i = 0; i < n; ++i) {
                        uint32_t kept = 0;
                        for (uint32_t j = 0; j < n; ++j) kept += v[j] >= v[i] && v[j] - v[i] < kWindowMax;
                        if (kept > bestKept || (kept == bestKept && v[i] < bestStart)) { bestKept = kept; bestStart = v[i]; }
                    }
                    for (uint32_t i = 0; i < n; ++i) {
                        if (v[i] >= bestStart && v[i] - bestStart < kWindowMax) continue;
                        const uint32_t t = cols[slot[i]].second;
                        gone[q].push_back({cols[slot[i]].first, r, v[i], (first + t / 16) * 256 + r * 16 + t % 16});
                    }
                }
            }
        }
    });
    uint64_t total = 0, dense = 0;
    for (const auto& g : gone) total += g.size();
    for (const uint64_t n : denseOfWorker) dense += n;
    if (total == 0 || total * 16 > dense) return BSMR_OK;

    out.blockValues.resize(numBlocks * 256);
    parallelChunks(numBlocks, 1024, [&](size_t b0, size_t b1, size_t) {
        std::memcpy(out.blockValues.data() + b0 * 256, d->block_values + b0 * 256, (b1 - b0) * 1024);
    });
    out.sparseOffsets.assign((size_t)P + 1, 0);
    for (uint32_t q = 0; q < P; ++q) {
        for (const Gone& g : gone[q]) out.blockValues[g.at] = kNone;
        out.sparseOffsets[q + 1] = out.sparseOffsets[q] + (d->sparse_value_offsets[q + 1] - d->sparse_value_offsets[q]) + (uint32_t)gone[q].size();
    }
    const uint64_t numSparse = out.sparseOffsets[P];
    if ((uint64_t)d->sparse_value_offsets[P] + total > 0xFFFFFFFFull) return BSMR_ERR_INVALID_ARG;
    out.sparseValues.resize(numSparse);
    out.sparseRows.resize(numSparse);
    out.sparseCols.resize(numSparse);
    parallelChunks(P, 64, [&](size_t q0, size_t q1, size_t) {
        struct Entry {
            uint32_t col, row, value;
        };
        std::vector<Entry> entries;
        for (size_t q = q0; q < q1; ++q) {
            const uint32_t b = d->sparse_value_offsets[q], e = d->sparse_value_offsets[q + 1];
            size_t at = out.sparseOffsets[q];
            if (gone[q].empty()) {   // as it was
                for (uint32_t i = b; i < e; ++i, ++at) {
                    out.sparseCols[at] = d->sparse_col_indices[i];
                    out.sparseRows[at] = d->sparse_relative_rows[i];
                    out.sparseValues[at] = d->sparse_values[i];
                }
                continue;
            }
            entries.clear();
            for (uint32_t i = b; i < e; ++i) entries.push_back({d->sparse_col_indices[i], d->sparse_relative_rows[i], d->sparse_values[i]});
            for (const Gone& g : gone[q]) entries.push_back({g.col, g.row, g.value});
            // (the residue's order inside a panel: by column, then row - src/BSMR.cpp of this repository, as the folding does)
            std::stable_sort(entries.begin(), entries.end(), [](const Entry& x, const Entry& y) { return x.col != y.col ? x.col < y.col : x.row < y.row; });
            for (const Entry& x : entries) {
                out.sparseCols[at] = x.col;
                out.sparseRows[at] = x.row;
                out.sparseValues[at] = x.value;
                ++at;
            }
        }
    });
    out.desc = *d;
    out.desc.block_values = out.blockValues.data();
    out.desc.sparse_value_offsets = out.sparseOffsets.data();
    out.desc.sparse_values = out.sparseValues.data();
    out.desc.sparse_relative_rows = out.sparseRows.data();
    out.desc.sparse_col_indices = out.sparseCols.data();
    out.evicted = total;
    return BSMR_OK;
}

}  // namespace bsmr
