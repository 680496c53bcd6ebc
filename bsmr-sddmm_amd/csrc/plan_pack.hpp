// Host-side packing of the reference-layout RPHM arrays (bsmr_rphm_desc) into the
// device format the gfx950 kernels read.  Pure host C++ (no HIP), used by
// bsmr_plan_create.
//
// Dense part.  H consecutive row panels form a row GROUP.  For every group the
// union of its panels' dense columns is ordered by entry count (descending, ties
// by column id) and cut into 16-column blocks.  A block stores
//   blockCols [16]          column ids (padding -> column 0, never written)
//   tiles     [H][256]      per panel: row-relative CSR offsets in accumulator
//                           (lane-major) order, all-ones = no entry
//   blockMask               bit h set iff tiles[h] holds an entry
// An entry (row i of panel p, column j) appears in the tile of (p, j) only if
// column j is in panel p's OWN dense list: entries of sparse columns stay on the
// sparse path even when another panel of the group made column j dense.
//
// Sparse part: the reference's three arrays, with the relative row packed to a byte.
#pragma once

#include <algorithm>
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "bsmr_hip.h"
#include "sddmm_kernels.hpp"

namespace bsmr {

struct PackOptions {
    int group = 0;            // panels per group: 1, 2 or 4; 0 = choose by estimated traffic
    int blocksPerItem = 16;   // dense blocks per workgroup item
    int sparsePerItem = 256;  // sparse entries per workgroup item
    bool forceWideTiles = false;
};

struct PackedPlan {
    uint32_t H = 1;
    uint32_t numGroups = 0;
    std::vector<uint32_t> panelRows;      // [P*16]   (sparse kernel)
    std::vector<uint32_t> groupRows;      // [G*16H]
    std::vector<uint32_t> groupRowBase;   // [G*16H]
    std::vector<uint32_t> blockCols;      // [NB*16]
    std::vector<uint16_t> tiles16;        // [NB*H*256] or empty
    std::vector<uint32_t> tiles32;        // [NB*H*256] or empty
    std::vector<uint8_t> blockMask;       // [NB]
    std::vector<DenseItem> denseItems;
    std::vector<uint32_t> entryCol, entryDst;
    std::vector<uint8_t> entryRow;
    std::vector<SparseItem> sparseItems;
    uint64_t numBlocks = 0, numTiles = 0, numDenseEntries = 0, numSparseEntries = 0;
    uint64_t unionColumns = 0;            // sum over groups of distinct dense columns
};

namespace detail {

struct ColumnUse {
    uint32_t col;
    uint32_t count;              // entries of this column in the dense parts of the group
    int32_t slot[kMaxGroup];     // position in each panel's dense list, -1 = not dense there
};

// Distinct dense columns of the panels [p0, p0+h) with their per-panel slots.
inline void unionColumns(const bsmr_rphm_desc* d, uint32_t p0, uint32_t h, uint32_t P,
                         std::vector<ColumnUse>& out, std::unordered_map<uint32_t, uint32_t>& index) {
    out.clear();
    index.clear();
    for (uint32_t k = 0; k < h && p0 + k < P; ++k) {
        const uint32_t p = p0 + k;
        const uint64_t firstBlock = d->block_offsets[p];
        const uint32_t slots = (d->block_offsets[p + 1] - d->block_offsets[p]) * 16u;
        for (uint32_t t = 0; t < slots; ++t) {
            const uint32_t col = d->dense_cols[firstBlock * 16 + t];
            if (col >= d->N) continue;  // padding sentinel
            const uint32_t* tile = d->block_values + (firstBlock + t / 16) * 256;
            uint32_t cnt = 0;
            for (uint32_t r = 0; r < 16; ++r) cnt += tile[r * 16 + t % 16] != 0xFFFFFFFFu;
            auto it = index.find(col);
            if (it == index.end()) {
                it = index.emplace(col, (uint32_t)out.size()).first;
                ColumnUse u{col, 0, {-1, -1, -1, -1}};
                out.push_back(u);
            }
            out[it->second].count += cnt;
            out[it->second].slot[k] = (int32_t)t;
        }
    }
    std::sort(out.begin(), out.end(), [](const ColumnUse& a, const ColumnUse& b) {
        return a.count != b.count ? a.count > b.count : a.col < b.col;
    });
}

}  // namespace detail

// Estimated bytes one SDDMM moves for group size h (K = 128 proxy: 256 B per
// gathered column, 512 B per non-empty tile).
inline uint64_t estimateTraffic(const bsmr_rphm_desc* d, uint32_t h) {
    const uint32_t P = d->num_row_panels;
    std::vector<detail::ColumnUse> cols;
    std::unordered_map<uint32_t, uint32_t> index;
    uint64_t bytes = 0;
    for (uint32_t p0 = 0; p0 < P; p0 += h) {
        detail::unionColumns(d, p0, h, P, cols, index);
        const uint64_t blocks = (cols.size() + 15) / 16;
        bytes += blocks * 16 * 256;
        for (uint64_t b = 0; b < blocks; ++b)
            for (uint32_t k = 0; k < h; ++k) {
                bool any = false;
                for (uint64_t u = b * 16; u < std::min<uint64_t>(b * 16 + 16, cols.size()) && !any; ++u)
                    any = cols[u].slot[k] >= 0;
                bytes += any ? 512 : 0;
            }
    }
    return bytes;
}

// Returns a bsmr_hip.h status.
inline int packPlan(const bsmr_rphm_desc* d, const PackOptions& opt, PackedPlan& out) {
    const uint32_t P = d->num_row_panels;
    const uint64_t numRefBlocks = d->block_offsets[P];
    const uint64_t numSparse = d->sparse_value_offsets[P];

    for (uint32_t p = 0; p < P; ++p)
        if (d->block_offsets[p + 1] < d->block_offsets[p] ||
            d->sparse_value_offsets[p + 1] < d->sparse_value_offsets[p])
            return BSMR_ERR_BAD_PLAN;

    // ---- rows -------------------------------------------------------------
    out.panelRows.assign((size_t)P * 16, 0);
    for (size_t i = 0; i < out.panelRows.size(); ++i) {
        const uint32_t row = i < d->num_nonzero_rows ? d->reordered_rows[i] : d->reordered_rows[0];
        if (row >= d->M) return BSMR_ERR_BAD_PLAN;
        out.panelRows[i] = row;
    }

    // rowBase of a panel row = smallest CSR index among its dense entries
    std::vector<uint32_t> panelRowBase((size_t)P * 16, 0);
    uint32_t maxOffset = 0;
    out.numDenseEntries = 0;
    for (uint32_t p = 0; p < P; ++p) {
        uint32_t lo[16], hi[16];
        for (int r = 0; r < 16; ++r) { lo[r] = 0xFFFFFFFFu; hi[r] = 0; }
        for (uint64_t b = d->block_offsets[p]; b < d->block_offsets[p + 1]; ++b) {
            const uint32_t* tile = d->block_values + b * 256;
            for (uint32_t i = 0; i < 256; ++i) {
                const uint32_t v = tile[i];
                if (v == 0xFFFFFFFFu) continue;
                if (v >= d->nnz) return BSMR_ERR_BAD_PLAN;
                ++out.numDenseEntries;
                lo[i >> 4] = std::min(lo[i >> 4], v);
                hi[i >> 4] = std::max(hi[i >> 4], v);
            }
        }
        for (int r = 0; r < 16; ++r) {
            if (lo[r] == 0xFFFFFFFFu) continue;
            panelRowBase[(size_t)p * 16 + r] = lo[r];
            maxOffset = std::max(maxOffset, hi[r] - lo[r]);
        }
    }
    for (uint64_t i = 0; i < numRefBlocks * 16; ++i)
        if (d->dense_cols[i] > d->N) return BSMR_ERR_BAD_PLAN;

    // ---- group size ---------------------------------------------------------
    uint32_t H = opt.group == 1 || opt.group == 2 || opt.group == 4 ? (uint32_t)opt.group : 0;
    // Default 1.  Grouping halves the gathered B bytes on the nips-like matrix
    // (H=4: 345k -> 179k column gathers) but multiplies the per-tile work (tile
    // loads, masked scatter stores), which is what the kernel is bound by at
    // L2-resident sizes: measured 14.2 us (H=1) vs 18.9 us (H=4) on MI355X
    // (profiles/r01_dense_ablation.md).  Kept selectable for HBM-resident operands.
    if (H == 0) H = 1;
    out.H = H;
    const uint32_t G = (P + H - 1) / H;
    out.numGroups = G;
    out.groupRows.assign((size_t)G * 16 * H, out.panelRows.empty() ? 0 : out.panelRows[0]);
    out.groupRowBase.assign((size_t)G * 16 * H, 0);
    std::copy(out.panelRows.begin(), out.panelRows.end(), out.groupRows.begin());
    std::copy(panelRowBase.begin(), panelRowBase.end(), out.groupRowBase.begin());

    // ---- dense blocks -------------------------------------------------------
    const bool wide = opt.forceWideTiles || maxOffset >= 0xFFFFu;
    std::vector<detail::ColumnUse> cols;
    std::unordered_map<uint32_t, uint32_t> index;
    out.blockCols.clear();
    out.blockMask.clear();
    out.tiles16.clear();
    out.tiles32.clear();
    out.denseItems.clear();
    out.unionColumns = 0;
    const uint32_t perItem = (uint32_t)std::max(1, opt.blocksPerItem);
    for (uint32_t gi = 0; gi < G; ++gi) {
        const uint32_t p0 = gi * H;
        detail::unionColumns(d, p0, H, P, cols, index);
        out.unionColumns += cols.size();
        const uint32_t blocks = (uint32_t)((cols.size() + 15) / 16);
        const uint64_t firstBlock = out.blockMask.size();
        out.blockCols.resize((firstBlock + blocks) * 16, 0);
        out.blockMask.resize(firstBlock + blocks, 0);
        if (wide) out.tiles32.resize((firstBlock + blocks) * H * 256, 0xFFFFFFFFu);
        else out.tiles16.resize((firstBlock + blocks) * H * 256, 0xFFFFu);
        for (size_t u = 0; u < cols.size(); ++u) {
            const uint64_t b = firstBlock + u / 16;
            const uint32_t cc = (uint32_t)(u % 16);
            out.blockCols[b * 16 + cc] = cols[u].col;
            for (uint32_t k = 0; k < H && p0 + k < P; ++k) {
                const int32_t t = cols[u].slot[k];
                if (t < 0) continue;
                const uint32_t p = p0 + k;
                const uint32_t* tile = d->block_values + ((uint64_t)d->block_offsets[p] + t / 16) * 256;
                for (uint32_t r = 0; r < 16; ++r) {
                    const uint32_t v = tile[r * 16 + t % 16];
                    if (v == 0xFFFFFFFFu) continue;
                    const uint32_t off = v - panelRowBase[(size_t)p * 16 + r];
                    const uint32_t lane = (r >> 2) * 16 + cc, i = r & 3u;
                    const size_t at = (b * H + k) * 256 + lane * 4 + i;
                    if (wide) out.tiles32[at] = off;
                    else out.tiles16[at] = (uint16_t)off;
                    out.blockMask[b] |= (uint8_t)(1u << k);
                }
            }
        }
        for (uint32_t b = 0; b < blocks; b += perItem)
            out.denseItems.push_back(
                DenseItem{gi, (uint32_t)(firstBlock + b), std::min(perItem, blocks - b), 0});
    }
    out.numBlocks = out.blockMask.size();
    out.numTiles = 0;
    for (const uint8_t m : out.blockMask) out.numTiles += __builtin_popcount(m);

    // ---- sparse residue -------------------------------------------------------
    out.entryCol.resize(numSparse);
    out.entryDst.resize(numSparse);
    out.entryRow.resize(numSparse);
    for (uint64_t i = 0; i < numSparse; ++i) {
        if (d->sparse_col_indices[i] >= d->N || d->sparse_values[i] >= d->nnz ||
            d->sparse_relative_rows[i] >= 16)
            return BSMR_ERR_BAD_PLAN;
        out.entryCol[i] = d->sparse_col_indices[i];
        out.entryDst[i] = d->sparse_values[i];
        out.entryRow[i] = (uint8_t)d->sparse_relative_rows[i];
    }
    out.numSparseEntries = numSparse;
    if (out.numDenseEntries + numSparse != d->nnz) return BSMR_ERR_BAD_PLAN;

    const uint32_t perWG = (uint32_t)std::max(32, opt.sparsePerItem);
    out.sparseItems.clear();
    for (uint32_t p = 0; p < P; ++p)
        for (uint32_t s = d->sparse_value_offsets[p]; s < d->sparse_value_offsets[p + 1]; s += perWG)
            out.sparseItems.push_back(
                SparseItem{p, s, std::min<uint32_t>(perWG, d->sparse_value_offsets[p + 1] - s), 0});
    return BSMR_OK;
}

}  // namespace bsmr
