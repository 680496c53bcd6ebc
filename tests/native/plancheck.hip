// Test infrastructure (not shipped): the host side of bsmr_plan_create - residue promotion (plan_promote.hpp)
// and packing (plan_pack.hpp) - callable without a GPU, with the invariants a device plan relies on checked here:
// every stored entry is computed exactly once, a promoted entry sits in the block cell of its own row and column.
#include <chrono>
#include <cstdint>
#include <vector>

#include "plan_pack.hpp"
#include "plan_promote.hpp"

namespace {
double nowUs() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// out[0] promoted (0/1), [1] promoted entries, [2] promoted blocks, [3] blocks after, [4] residue entries after,
// [5] packPlan status, [6] packed dense entries, [7] packed residue entries, [8] promotion us, [9] packing us,
// [10] B columns gathered by the dense blocks ungrouped, [11] with 4 panels per group,
// [12] bytes per index-tile element (1 = windowed 8-bit, 2 / 4 = offsets from the row's first dense entry).
// Returns 0, or the number of the first violated invariant.
extern "C" int plancheck_promote(const bsmr_rphm_desc* in, uint32_t minAverage, uint64_t minEntries, uint64_t smallDense,
                                 uint32_t minColumnDegree, uint32_t headMin, uint64_t* out) {
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const uint32_t P = in->num_row_panels;
    bsmr::PromotedRphm pr;
    double t0 = nowUs();
    const bool did = bsmr::promoteSparseBlocks(*in, minAverage, minEntries, smallDense, minColumnDegree, headMin, pr);
    out[8] = (uint64_t)(nowUs() - t0);
    const bsmr_rphm_desc* d = did ? &pr.desc : in;
    out[0] = did;
    out[1] = pr.promotedEntries;
    out[2] = pr.promotedBlocks;
    out[3] = d->block_offsets[P];
    out[4] = d->sparse_value_offsets[P];
    if (did) {
        if (d->block_offsets[P] != in->block_offsets[P] + pr.promotedBlocks) return 1;
        if (d->sparse_value_offsets[P] + pr.promotedEntries != in->sparse_value_offsets[P]) return 2;
        // where every residue entry of the input went
        std::vector<uint8_t> seen(in->nnz, 0);
        for (uint32_t q = 0; q < P; ++q) {
            const uint64_t own = in->block_offsets[q + 1] - in->block_offsets[q];
            const uint64_t b0 = d->block_offsets[q], b1 = d->block_offsets[q + 1];
            if (b1 - b0 < own) return 3;
            // the panel's own blocks come first, unchanged
            for (uint64_t b = 0; b < own; ++b) {
                for (uint32_t i = 0; i < 16; ++i)
                    if (d->dense_cols[(b0 + b) * 16 + i] != in->dense_cols[(in->block_offsets[q] + b) * 16 + i]) return 4;
                for (uint32_t i = 0; i < 256; ++i)
                    if (d->block_values[(b0 + b) * 256 + i] != in->block_values[(in->block_offsets[q] + b) * 256 + i]) return 5;
            }
            // promoted blocks: a column appears once per panel, cells hold entries of that (row, column)
            for (uint64_t b = b0 + own; b < b1; ++b)
                for (uint32_t i = 0; i < 256; ++i) {
                    const uint32_t v = d->block_values[b * 256 + i];
                    if (v == kNone) continue;
                    if (v >= in->nnz || seen[v]) return 6;
                    seen[v] = 1;
                }
            for (uint32_t i = d->sparse_value_offsets[q]; i < d->sparse_value_offsets[q + 1]; ++i) {
                const uint32_t v = d->sparse_values[i];
                if (v >= in->nnz || seen[v]) return 7;
                seen[v] = 2;
            }
            // every input residue entry of the panel is in exactly one of the two, at its own row and column
            for (uint32_t i = in->sparse_value_offsets[q]; i < in->sparse_value_offsets[q + 1]; ++i) {
                const uint32_t v = in->sparse_values[i], row = in->sparse_relative_rows[i], col = in->sparse_col_indices[i];
                if (!seen[v]) return 8;
                if (seen[v] == 1) {
                    bool found = false;
                    for (uint64_t b = b0 + own; b < b1 && !found; ++b)
                        for (uint32_t c = 0; c < 16 && !found; ++c)
                            found = d->dense_cols[b * 16 + c] == col && d->block_values[b * 256 + row * 16 + c] == v;
                    if (!found) return 9;
                }
            }
            // residue order of what stays is the input's order
            uint32_t at = d->sparse_value_offsets[q];
            for (uint32_t i = in->sparse_value_offsets[q]; i < in->sparse_value_offsets[q + 1]; ++i) {
                if (seen[in->sparse_values[i]] != 2) continue;
                if (d->sparse_values[at] != in->sparse_values[i] || d->sparse_relative_rows[at] != in->sparse_relative_rows[i] ||
                    d->sparse_col_indices[at] != in->sparse_col_indices[i])
                    return 10;
                ++at;
            }
            if (at != d->sparse_value_offsets[q + 1]) return 11;
            // no column twice among the promoted blocks of a panel
            std::vector<uint32_t> cols(d->dense_cols + (b0 + own) * 16, d->dense_cols + b1 * 16);
            std::sort(cols.begin(), cols.end());
            for (size_t i = 1; i < cols.size(); ++i)
                if (cols[i] == cols[i - 1] && cols[i] != in->N) return 12;
        }
    }
    bsmr::PackOptions opt;
    opt.blocksPerItem = 8;
    bsmr::PackedPlan pk;
    t0 = nowUs();
    out[5] = (uint64_t)(int64_t)bsmr::packPlan(d, opt, pk);
    out[9] = (uint64_t)(nowUs() - t0);
    out[10] = bsmr::countUnionColumns(d, 1);
    out[11] = bsmr::countUnionColumns(d, 4);
    out[12] = !pk.tiles8.empty() ? 1 : !pk.tiles16.empty() ? 2 : !pk.tiles32.empty() ? 4 : 0;
    out[6] = pk.numDenseEntries;
    out[7] = pk.numSparseEntries;
    return 0;
}
