// Device-plan step before packing: residue entries of the RPHM that are cheaper as MFMA tiles are regrouped
// into extra 16x16 blocks of their panel.
//
// The reference cuts a panel's count-ordered columns into 16-column blocks and calls a block dense when it
// holds at least ceil(delta * 256) entries (reference src/colReordering.cu:244-271, the delta of the caller).
// That cut-off balances the reference's two CUDA kernels.  On MI355X one dense block costs about as much as
// 11-27 residue entries (mycielskian15: K=32 0.12 ns / 11 ps, K=128 0.29 ns / 14 ps, K=512 0.91 ns / 40 ps -
// a block gathers 16 columns of B, an entry one), blocks at the sparse end of a panel ride in the dense
// kernel's tail almost for free, and a residue that disappears saves a launch and a kernel boundary (3-5 us).
// So a panel whose residue columns, cut into 16s, average at least `minAverage` (16) entries per block gives
// ALL of its residue to the dense path, and when the other panels would keep less than a quarter of all
// entries they follow.  Measured (alpha = delta = 0.3, tools/promote_lab.py): mycielskian15 K=32 50.2 -> 26.3 us,
// K=128 66.6 -> 47.5, K=512 184 -> 143; mycielskian14 K=128 34.2 -> 23.8; nips-like K=32 12.0 -> 8.7;
// 4096^2 Bernoulli(0.1) K=512 delta=0.1 66.4 -> 43.1; reddit-like shard K=256 (residue 9.5 M entries): nothing
// moved 0.60 ms, panels averaging >= 20 only 0.434 ms, all of it 0.403 ms.
// The RPHM, its statistics and the reference-visible split stay as they are; only the device plan computes
// those entries with the other kernel (same operands, same accuracy class as the low-precision residue).
#pragma once

#include <algorithm>
#include <cstdint>
#include <memory>
#include <vector>

#include "bsmr_hip.h"
#include "plan_pack.hpp"

namespace bsmr {

struct PromotedRphm {
    std::vector<uint32_t> denseCols, blockOffsets;
    // (1 KiB per block: allocated without a fill and written once, panel by panel, by the workers - a vector's
    // single-threaded fill of 0.5 GB was a third of this step on the reddit-like shard)
    std::unique_ptr<uint32_t[]> blockValues;
    std::vector<uint32_t> sparseOffsets, sparseValues, sparseRows, sparseCols;
    bsmr_rphm_desc desc{};
    uint64_t promotedEntries = 0, promotedBlocks = 0;
};

// Returns true and fills `out` when blocks were promoted.  `minAverage`: entries per 16-column block that a
// panel's residue needs on average; `minEntries`: promoted entries below which a plan WITHOUT a dense part
// (fewer than `smallDense` dense entries: the plan folds those into the residue) is left alone - a first dense
// block brings the conversion pass and a second launch with it.
// `minColumnDegree`: stored entries per column of S (nnz / N) the pattern needs at all - the dense kernel gathers 16
// columns of B per block and is only cheap when those columns are reused from cache by other panels (graph / ML
// patterns: mycielskian15 113, nips-like 60, Bernoulli 4096^2 410; mesh / FEM patterns: cop20k-like 11, wathen100 8,
// Trefethen 14 - their blocks cost 0.58 ns instead of 0.29 and the conversion of A is paid on top).
// `headMin`: a panel that does not qualify as a whole still gives its leading blocks with at least this many entries
// (0 = none; reddit-like shard K=256: a block costs 0.71 ns, a residue entry 44 ps).
// Malformed input is left to packPlan's validation (returns false).
inline bool promoteSparseBlocks(const bsmr_rphm_desc& in, uint32_t minAverage, uint64_t minEntries, uint64_t smallDense,
                                uint32_t minColumnDegree, uint32_t headMin, PromotedRphm& out) {
    const uint32_t P = in.num_row_panels;
    const uint64_t numSparse = in.sparse_value_offsets[P];
    if (minAverage == 0 || numSparse == 0 || numSparse > 0xFFFFFFF0ull) return false;
    if ((uint64_t)in.nnz < (uint64_t)minColumnDegree * in.N) return false;
    constexpr uint32_t kNone = 0xFFFFFFFFu;

    // pass 1, per panel: columns of the residue by descending count (ties: ascending id, the reference's
    // order), cut into 16s
    std::vector<uint32_t> cell(numSparse, kNone);      // promoted entry: 256 * (block within panel's new blocks) + 16 * row + column slot
    std::vector<uint32_t> newBlocks((size_t)P, 0);
    std::vector<std::vector<uint32_t>> newCols((size_t)P);
    const unsigned workers = packThreads();
    std::vector<uint8_t> bad(workers, 0), qualifies((size_t)P, 0);
    std::vector<uint32_t> headBlocks((size_t)P, 0), take((size_t)P, 0);
    parallelChunks(P, 16, [&](size_t q0, size_t q1, size_t w) {
        struct Column {
            uint32_t col, count, first;
        };
        std::vector<uint32_t> order;
        std::vector<Column> columns;
        std::vector<uint8_t> taken;
        for (size_t q = q0; q < q1; ++q) {
            const uint32_t s0 = in.sparse_value_offsets[q], s1 = in.sparse_value_offsets[q + 1];
            if (s1 == s0) continue;
            order.resize(s1 - s0);
            for (uint32_t i = 0; i < s1 - s0; ++i) {
                order[i] = s0 + i;
                if (in.sparse_col_indices[s0 + i] >= in.N || in.sparse_relative_rows[s0 + i] >= 16) bad[w] = 1;
            }
            if (bad[w]) return;
            bool sorted = true;  // the host pipeline's residue is already in column order
            for (uint32_t i = s0 + 1; i < s1 && sorted; ++i) sorted = in.sparse_col_indices[i - 1] <= in.sparse_col_indices[i];
            if (!sorted)
                std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
                    return in.sparse_col_indices[x] != in.sparse_col_indices[y]
                               ? in.sparse_col_indices[x] < in.sparse_col_indices[y]
                               : x < y;
                });
            columns.clear();
            for (uint32_t i = 0; i < order.size(); ++i) {
                const uint32_t c = in.sparse_col_indices[order[i]];
                if (columns.empty() || columns.back().col != c) columns.push_back({c, 0, i});
                ++columns.back().count;
            }
            std::stable_sort(columns.begin(), columns.end(),
                             [](const Column& x, const Column& y) { return x.count > y.count; });
            const uint64_t blocksNeeded = (columns.size() + 15) / 16;
            qualifies[q] = (uint64_t)(s1 - s0) >= (uint64_t)minAverage * blocksNeeded;
            bool head = headMin != 0;
            for (size_t c0 = 0; c0 < columns.size(); c0 += 16) {
                const size_t c1 = std::min(columns.size(), c0 + 16);
                const uint32_t block = newBlocks[q]++;
                uint32_t inBlock = 0;
                for (size_t c = c0; c < c1; ++c) inBlock += columns[c].count;
                head = head && inBlock >= headMin;
                if (head) headBlocks[q] = block + 1;
                taken.assign(256, 0);
                for (size_t c = c0; c < c1; ++c) {
                    newCols[q].push_back(columns[c].col);
                    for (uint32_t i = columns[c].first; i < columns[c].first + columns[c].count; ++i) {
                        const uint32_t e = order[i];
                        const uint32_t slot = in.sparse_relative_rows[e] * 16 + (uint32_t)(c - c0);
                        if (taken[slot]) continue;  // a repeated (row, column) keeps its extra copies in the residue
                        taken[slot] = 1;
                        cell[e] = block * 256 + slot;
                    }
                }
                for (size_t c = c1; c < c0 + 16; ++c) newCols[q].push_back(in.N);  // padding column
            }
        }
    });
    for (unsigned w = 0; w < workers; ++w)
        if (bad[w]) return false;
    // which panels: those that qualify; all of them when what the others would leave is a small rest of a plan
    // that has a dense part anyway (a residue of a few thousand entries costs a launch and a kernel boundary,
    // 3-5 us; 4096^2 Bernoulli(0.1) delta = 0.1, K = 512, 21 % left at 16 per block: 66.6 -> 43.5 us)
    const uint64_t oldBlocks = in.block_offsets[P];
    const uint64_t oldDense = in.nnz >= numSparse ? in.nnz - numSparse : 0;
    uint64_t qualified = 0;
    for (uint32_t q = 0; q < P; ++q)
        if (qualifies[q]) qualified += in.sparse_value_offsets[q + 1] - in.sparse_value_offsets[q];
    const bool hasDense = oldBlocks != 0 && oldDense >= smallDense;
    const bool everything = (hasDense || qualified >= minEntries) && (numSparse - qualified) * 4 < in.nnz;
    uint64_t totalMoved = 0, totalNew = 0;
    for (uint32_t q = 0; q < P; ++q) {
        take[q] = everything || qualifies[q] ? newBlocks[q] : headBlocks[q];
        totalNew += take[q];
        if (take[q])
            for (uint32_t i = in.sparse_value_offsets[q]; i < in.sparse_value_offsets[q + 1]; ++i)
                totalMoved += cell[i] != kNone && cell[i] / 256 < take[q];
    }
    if (totalNew == 0 || (!hasDense && totalMoved < minEntries)) return false;
    if (oldBlocks + totalNew > 0x00FFFFFFull) return false;

    // pass 2: the panel's blocks are its own followed by the promoted ones
    out.blockOffsets.assign((size_t)P + 1, 0);
    out.sparseOffsets.assign((size_t)P + 1, 0);
    for (uint32_t q = 0; q < P; ++q) {
        out.blockOffsets[q + 1] = out.blockOffsets[q] + (in.block_offsets[q + 1] - in.block_offsets[q]) + take[q];
        uint32_t kept = 0;
        for (uint32_t i = in.sparse_value_offsets[q]; i < in.sparse_value_offsets[q + 1]; ++i)
            kept += cell[i] == kNone || cell[i] / 256 >= take[q];
        out.sparseOffsets[q + 1] = out.sparseOffsets[q] + kept;
    }
    const uint64_t blocks = out.blockOffsets[P];
    out.denseCols.resize(blocks * 16);
    out.blockValues.reset(new uint32_t[std::max<uint64_t>(blocks * 256, 1)]);
    out.sparseValues.resize(out.sparseOffsets[P]);
    out.sparseRows.resize(out.sparseOffsets[P]);
    out.sparseCols.resize(out.sparseOffsets[P]);
    parallelChunks(P, 16, [&](size_t q0, size_t q1, size_t) {
        for (size_t q = q0; q < q1; ++q) {
            const uint64_t own = in.block_offsets[q + 1] - in.block_offsets[q];
            const uint64_t b0 = out.blockOffsets[q];
            if (own) {
                std::copy(in.dense_cols + (uint64_t)in.block_offsets[q] * 16, in.dense_cols + (uint64_t)in.block_offsets[q + 1] * 16,
                          out.denseCols.begin() + b0 * 16);
                std::copy(in.block_values + (uint64_t)in.block_offsets[q] * 256,
                          in.block_values + (uint64_t)in.block_offsets[q + 1] * 256, out.blockValues.get() + b0 * 256);
            }
            std::fill(out.blockValues.get() + (b0 + own) * 256, out.blockValues.get() + (b0 + own + take[q]) * 256, kNone);
            std::copy(newCols[q].begin(), newCols[q].begin() + (size_t)take[q] * 16, out.denseCols.begin() + (b0 + own) * 16);
            uint32_t at = out.sparseOffsets[q];
            for (uint32_t i = in.sparse_value_offsets[q]; i < in.sparse_value_offsets[q + 1]; ++i) {
                if (cell[i] == kNone || cell[i] / 256 >= take[q]) {
                    out.sparseValues[at] = in.sparse_values[i];
                    out.sparseRows[at] = in.sparse_relative_rows[i];
                    out.sparseCols[at] = in.sparse_col_indices[i];
                    ++at;
                } else {
                    out.blockValues[(b0 + own) * 256 + cell[i]] = in.sparse_values[i];
                }
            }
        }
    });
    out.desc = in;
    out.desc.dense_cols = out.denseCols.data();
    out.desc.block_offsets = out.blockOffsets.data();
    out.desc.block_values = out.blockValues.get();
    out.desc.sparse_value_offsets = out.sparseOffsets.data();
    out.desc.sparse_values = out.sparseValues.data();
    out.desc.sparse_relative_rows = out.sparseRows.data();
    out.desc.sparse_col_indices = out.sparseCols.data();
    out.promotedEntries = totalMoved;
    out.promotedBlocks = totalNew;
    return true;
}

}  // namespace bsmr
