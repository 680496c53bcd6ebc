// Per-panel column reordering and dense/sparse split.
//
// Behaviour: reference src/colReordering.cu:244-404.  For every 16-row panel the
// non-empty columns are ordered by their nnz count inside the panel (descending,
// ties by ascending column id), padded with the sentinel column id N to a
// multiple of 16 and cut into 16-column blocks; a block is dense iff it holds at
// least ceil(delta * 256) entries.
//
// The reference counts into a dense u32[N] array per panel (0.9 MB per thread on
// a reddit-sized matrix) and compacts it; here the panel's column ids are sorted
// and run-length encoded, which costs O(nnz_panel log nnz_panel) and no O(N) pass.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <numeric>
#include <vector>

#include <omp.h>
#include "util.hpp"

#include "BSMR.hpp"
#include "bsmr_hip.h"

std::pair<UIN, UIN> analysisDescendingOrderColSegment(
    const float blockDensityThreshold, const std::vector<UIN>& numOfNonZeroInEachColSegment) {
    const UIN threshold = static_cast<UIN>(std::ceil(blockDensityThreshold * BLOCK_SIZE));
    const size_t n = numOfNonZeroInEachColSegment.size();
    size_t visited = 0;
    UIN dense = 0;
    for (; visited + BLOCK_COL_SIZE <= n; visited += BLOCK_COL_SIZE) {
        UIN inBlock = 0;
        for (UIN i = 0; i < BLOCK_COL_SIZE; ++i) inBlock += numOfNonZeroInEachColSegment[visited + i];
        if (inBlock >= threshold) dense += BLOCK_COL_SIZE;
    }
    // leftover (< 16) columns of an unpadded list: only the non-empty ones count
    while (visited < n && numOfNonZeroInEachColSegment[visited] > 0) ++visited;
    return std::make_pair(dense, static_cast<UIN>(visited) - dense);
}

void colReordering_cpu(const sparseMatrix::CSR<float>& matrix, const UIN numRowPanels,
                       const std::vector<UIN>& reorderedRows, const float blockDensityThreshold,
                       std::vector<UIN>& denseCols, std::vector<UIN>& denseColOffsets,
                       std::vector<UIN>& sparseCols, std::vector<UIN>& sparseColOffsets,
                       std::vector<UIN>& sparseDataOffsets, float& time) {
    const auto t0 = std::chrono::steady_clock::now();

    std::vector<std::vector<UIN>> panelCols(numRowPanels);  // sorted + padded column list
    std::vector<UIN> numDense(numRowPanels, 0), numSparse(numRowPanels, 0),
        sparseEntries(numRowPanels, 0);

#pragma omp parallel num_threads(util::hostThreads(omp_get_max_threads()))
    {
        std::vector<UIN> ids;     // column id of every entry of the panel
        std::vector<UIN> cols;    // distinct columns, ascending
        std::vector<UIN> counts;  // their counts
        std::vector<UIN> perm;
#pragma omp for schedule(dynamic, 16)
        for (long long p = 0; p < static_cast<long long>(numRowPanels); ++p) {
            const size_t first = static_cast<size_t>(p) * ROW_PANEL_SIZE;
            const size_t last = std::min(first + ROW_PANEL_SIZE, reorderedRows.size());
            ids.clear();
            for (size_t i = first; i < last; ++i) {
                const UIN row = reorderedRows[i];
                ids.insert(ids.end(), matrix.colIndices().begin() + matrix.rowOffsets()[row],
                           matrix.colIndices().begin() + matrix.rowOffsets()[row + 1]);
            }
            std::sort(ids.begin(), ids.end());
            cols.clear();
            counts.clear();
            for (size_t i = 0; i < ids.size();) {
                size_t j = i;
                while (j < ids.size() && ids[j] == ids[i]) ++j;
                cols.push_back(ids[i]);
                counts.push_back(static_cast<UIN>(j - i));
                i = j;
            }
            // stable: equal counts stay in ascending column order
            perm.resize(cols.size());
            std::iota(perm.begin(), perm.end(), 0);
            std::stable_sort(perm.begin(), perm.end(),
                             [&](UIN a, UIN b) { return counts[a] > counts[b]; });
            const size_t padded = (cols.size() + BLOCK_COL_SIZE - 1) / BLOCK_COL_SIZE * BLOCK_COL_SIZE;
            std::vector<UIN>& sortedCols = panelCols[p];
            sortedCols.assign(padded, matrix.col());
            std::vector<UIN> sortedCounts(padded, 0);
            for (size_t i = 0; i < perm.size(); ++i) {
                sortedCols[i] = cols[perm[i]];
                sortedCounts[i] = counts[perm[i]];
            }
            const auto [dense, sparse] =
                analysisDescendingOrderColSegment(blockDensityThreshold, sortedCounts);
            numDense[p] = dense;
            numSparse[p] = sparse;
            UIN residue = 0;
            for (size_t i = dense; i < static_cast<size_t>(dense) + sparse; ++i) residue += sortedCounts[i];
            sparseEntries[p] = residue;
        }
    }

    auto exclusiveScan = [&](const std::vector<UIN>& in, std::vector<UIN>& out) {
        out.assign(static_cast<size_t>(numRowPanels) + 1, 0);
        for (size_t p = 0; p < numRowPanels; ++p) out[p + 1] = out[p] + in[p];
    };
    exclusiveScan(sparseEntries, sparseDataOffsets);
    exclusiveScan(numDense, denseColOffsets);
    exclusiveScan(numSparse, sparseColOffsets);

    denseCols.resize(denseColOffsets[numRowPanels]);
    sparseCols.resize(sparseColOffsets[numRowPanels]);
#pragma omp parallel for schedule(static) num_threads(util::hostThreads(omp_get_max_threads()))
    for (long long p = 0; p < static_cast<long long>(numRowPanels); ++p) {
        const std::vector<UIN>& c = panelCols[p];
        std::copy(c.begin(), c.begin() + numDense[p], denseCols.begin() + denseColOffsets[p]);
        std::copy(c.begin() + numDense[p], c.end(), sparseCols.begin() + sparseColOffsets[p]);
    }

    time = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// ---------------------------------------------------------------------------
// device path: bsmr_col_reorder, csrc/colreorder_capi.hpp
// ---------------------------------------------------------------------------
namespace {
int g_colReorderingDevice = -2;
}
void setColReorderingDevice(int device) { g_colReorderingDevice = device; }
int colReorderingDevice() { return g_colReorderingDevice; }

bool colReordering_device(const sparseMatrix::CSR<float>& matrix, const std::vector<UIN>& reorderedRows,
                          const float blockDensityThreshold, int device, std::vector<UIN>& denseCols,
                          std::vector<UIN>& denseColOffsets, std::vector<UIN>& sparseCols,
                          std::vector<UIN>& sparseColOffsets, std::vector<UIN>& sparseDataOffsets,
                          BSMR::DeviceRphmArrays& rphm, float& time) {
    const auto t0 = std::chrono::steady_clock::now();
    bsmr_colreorder* h = nullptr;
    const int st = bsmr_col_reorder(&h, device, matrix.row(), matrix.col(), matrix.rowOffsets().data(),
                                    matrix.colIndices().data(), reorderedRows.data(),
                                    static_cast<uint32_t>(reorderedRows.size()), blockDensityThreshold);
    if (st != BSMR_OK) return false;
    bsmr_colreorder_sizes sz{};
    bsmr_col_reorder_sizes(h, &sz);
    denseCols.resize(sz.num_dense_cols);
    sparseCols.resize(sz.num_sparse_cols);
    denseColOffsets.resize(static_cast<size_t>(sz.num_row_panels) + 1);
    sparseColOffsets.resize(static_cast<size_t>(sz.num_row_panels) + 1);
    sparseDataOffsets.resize(static_cast<size_t>(sz.num_row_panels) + 1);
    rphm.blockOffsets.resize(static_cast<size_t>(sz.num_row_panels) + 1);
    // the column lists and the offsets come down; block values and the residue's arrays stay behind the handle
    const int fs = bsmr_col_reorder_fetch(h, denseCols.data(), denseColOffsets.data(), sparseCols.data(), sparseColOffsets.data(),
                                          sparseDataOffsets.data(), rphm.blockOffsets.data(), nullptr, nullptr, nullptr, nullptr);
    rphm.handle.reset(h, [](bsmr_colreorder* p) { bsmr_col_reorder_free(p); });
    if (fs != BSMR_OK) {
        rphm.handle.reset();
        return false;
    }
    rphm.numBlocks = sz.num_blocks;
    rphm.numSparseEntries = sz.num_sparse_entries;
    rphm.valid = true;
    rphm.deviceMs = sz.elapsed_ms;
    time = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
}
