// Test infrastructure (not shipped): the host side of bsmr_plan_create - residue promotion (plan_promote.hpp)
// and packing (plan_pack.hpp) - callable without a GPU, with the invariants a device plan relies on checked here:
// every stored entry is computed exactly once, a promoted entry sits in the block cell of its own row and column.
#include <chrono>
#include <cstdint>
#include <vector>

#include "plan_pack.hpp"
#include "plan_promote.hpp"
#include "tile_format.hpp"

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
    out[12] = !pk.tiles8.empty() || !pk.tilesMask.empty() ? 1 : !pk.tiles16.empty() ? 2 : !pk.tiles32.empty() ? 4 : 0;
    out[6] = pk.numDenseEntries;
    out[7] = pk.numSparseEntries;
    return 0;
}

// The "tiles" dense format (csrc/tile_format.hpp) packed for H panels per group and checked the way the kernel reads
// it: every dense entry of the RPHM is listed exactly once, in a block that holds its column at the listed slot, on
// the listed row of its group, and row base + offset is its CSR index; entry lists start on multiples of 4 and are
// ordered by (row, column); a block's mask has exactly the panels that own an entry; counts fit the LDS room.
// out[0] blocks, [1] tiles, [2] union columns, [3] entries, [4] items, [5] entry cap, [6] bytes, [7..9] census blocks / tiles / columns.
extern "C" int plancheck_tiles(const bsmr_rphm_desc* d, uint32_t H, uint32_t blocksPerItem, uint64_t* out) {
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    bsmr::HostDense hd;
    int st = bsmr::collectDense(d, hd);
    if (st != BSMR_OK) return 100 + st;
    bsmr::TileFormatHost f;
    st = bsmr::packTiles(hd, H, blocksPerItem, f);
    if (st != BSMR_OK) return 200 + st;
    const bsmr::TileCensus census = bsmr::tileCensus(hd, H);
    out[0] = f.numBlocks; out[1] = f.numTiles; out[2] = f.unionColumns; out[3] = f.numEntries;
    out[4] = f.items.size(); out[5] = f.entryCap; out[6] = f.bytes();
    out[7] = census.blocks; out[8] = census.tiles; out[9] = census.unionColumns;
    const uint32_t R = 16 * H;
    // expected: CSR index -> (original row, column) from the RPHM
    std::vector<uint32_t> wantRow(d->nnz, kNone), wantCol(d->nnz, kNone);
    uint64_t denseEntries = 0;
    for (uint32_t p = 0; p < d->num_row_panels; ++p)
        for (uint64_t b = d->block_offsets[p]; b < d->block_offsets[p + 1]; ++b)
            for (uint32_t i = 0; i < 256; ++i) {
                const uint32_t v = d->block_values[b * 256 + i];
                if (v == kNone) continue;
                const size_t slot = (size_t)p * 16 + i / 16;
                wantRow[v] = slot < d->num_nonzero_rows ? d->reordered_rows[slot] : kNone;
                wantCol[v] = d->dense_cols[b * 16 + i % 16];
                ++denseEntries;
            }
    if (f.numEntries != denseEntries) return 1;
    std::vector<uint8_t> seen(d->nnz, 0);
    std::vector<uint8_t> blockSeen(f.numBlocks, 0);
    uint64_t tiles = 0, cols = 0;
    for (size_t it = 0; it < f.items.size(); ++it) {
        const bsmr::TileItem& item = f.items[it];
        if (item.count == 0 || item.count > bsmr::kTileMaxItemBlocks || item.first + item.count > f.numBlocks) return 2;
        if (it && f.blockCols[(size_t)f.items[it - 1].first * 16] > f.blockCols[(size_t)item.first * 16]) return 3;  // column order
        for (uint32_t b = item.first; b < item.first + item.count; ++b) {
            if (blockSeen[b]++) return 4;
            const uint32_t start = f.blockInfo[4 * (size_t)b], mask = f.blockInfo[4 * (size_t)b + 1] & 0xFFFFu;
            const uint32_t n = f.blockInfo[4 * (size_t)b + 1] >> 16;
            uint32_t quarter[4] = {0, 0, 0, 0};
            if (start % 4 || n == 0 || n > f.entryCap || (size_t)start + (n + 255) / 256 * 256 > f.entries.size()) return 5;
            uint32_t got = 0, last = 0;
            for (uint32_t e = 0; e < n; ++e) {
                const uint32_t w = f.entries[start + e];
                const uint32_t row = w & 255u, slot = (w >> 8) & 15u, off = w >> 12;
                if (row < R) ++quarter[row / (4 * H)];
                if (row >= R) return 6;
                const uint32_t idx = f.itemRowBase[it * R + row] + off;
                if (idx >= d->nnz || seen[idx]++) return 7;
                if (f.groupRows[(size_t)item.group * R + row] != wantRow[idx]) return 8;
                if (f.blockCols[(size_t)b * 16 + slot] != wantCol[idx]) return 9;
                const uint32_t key = row * 16 + slot;
                if (e && key < last) return 10;   // (row, column) order
                last = key;
                got |= 1u << (row / 16);
            }
            if (got != mask) return 11;
            if ((quarter[0] | (quarter[1] << 16)) != f.blockInfo[4 * (size_t)b + 2] ||
                (quarter[2] | (quarter[3] << 16)) != f.blockInfo[4 * (size_t)b + 3]) return 19;
            tiles += __builtin_popcount(mask);
            for (uint32_t c = 0; c < 16; ++c) {
                if (c && f.blockCols[(size_t)b * 16 + c] != 0 && f.blockCols[(size_t)b * 16 + c] <= f.blockCols[(size_t)b * 16 + c - 1]) return 12;
                if (f.blockCols[(size_t)b * 16 + c] >= d->N) return 13;
            }
        }
    }
    for (uint64_t b = 0; b < f.numBlocks; ++b)
        if (!blockSeen[b]) return 14;
    for (uint32_t v = 0; v < d->nnz; ++v)
        if ((wantRow[v] != kNone) != (seen[v] != 0)) return 15;
    if (tiles != f.numTiles) return 16;
    (void)cols;
    if (census.blocks > f.numBlocks || census.unionColumns != f.unionColumns) return 17;   // (blocks cut at the entry cap add to the census)
    if (census.blocks == f.numBlocks && census.tiles != f.numTiles) return 18;
    return 0;
}

#include "sweep_format.hpp"

// csrc/sweep_format.hpp read the way denseSweep reads it: every dense entry of the RPHM is listed exactly once, in
// the (item, wave, block) list of its row group / panel / column block, with the slab slot of its accumulator cell.
// out[0] items, [1] words, [2] groups, [3] strips, [4] most entries in one (wave, block) step, [5] bytes.
extern "C" int plancheck_sweep(const bsmr_rphm_desc* d, uint32_t W, uint32_t PW, uint32_t stripBlocks, uint64_t* out) {
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    bsmr::HostDense hd;
    int st = bsmr::collectDense(d, hd);
    if (st != BSMR_OK) return 100 + st;
    bsmr::SweepFormatHost f;
    st = bsmr::packSweep(hd, W, PW, stripBlocks, f);
    if (st != BSMR_OK) return 200 + st;
    out[0] = f.items.size(); out[1] = hd.entries(); out[2] = f.numGroups; out[3] = f.numStrips;
    out[4] = f.maxStepEntries; out[5] = f.bytes();
    std::vector<uint32_t> wantRow(d->nnz, kNone), wantCol(d->nnz, kNone);
    uint64_t denseEntries = 0;
    for (uint32_t p = 0; p < d->num_row_panels; ++p)
        for (uint64_t b = d->block_offsets[p]; b < d->block_offsets[p + 1]; ++b)
            for (uint32_t i = 0; i < 256; ++i) {
                const uint32_t v = d->block_values[b * 256 + i];
                if (v == kNone) continue;
                const size_t slot = (size_t)p * 16 + i / 16;
                wantRow[v] = slot < d->num_nonzero_rows ? d->reordered_rows[slot] : kNone;
                wantCol[v] = d->dense_cols[b * 16 + i % 16];
                ++denseEntries;
            }
    if (hd.entries() != denseEntries) return 1;
    const uint32_t GP = W * PW, NCB = (d->N + 15) / 16;
    if (f.items.size() != (size_t)f.numGroups * f.numStrips) return 2;
    if (f.panelRows.size() != (size_t)f.numGroups * GP * 16) return 3;
    std::vector<uint8_t> seen(d->nnz, 0);
    std::vector<uint32_t> blockCover(NCB, 0);
    uint64_t expectStart = 0;
    for (size_t it = 0; it < f.items.size(); ++it) {
        const bsmr::SweepItem& item = f.items[it];
        if (item.group >= f.numGroups || item.numBlocks == 0 || item.numBlocks > bsmr::kSweepMaxBlocks ||
            item.firstBlock + item.numBlocks > NCB)
            return 4;
        if (it / f.numGroups != item.firstBlock / stripBlocks || it % f.numGroups != item.group) return 5;   // strip-major
        if (item.group == 0)
            for (uint32_t b = 0; b < item.numBlocks; ++b) ++blockCover[item.firstBlock + b];
        for (uint32_t w = 0; w < W; ++w) {
            const uint32_t* starts = &f.starts[(size_t)item.startsBase + (size_t)w * (item.numBlocks + 1)];
            if (starts[0] != expectStart) return 6;    // the lists follow each other without gaps
            for (uint32_t b = 0; b < item.numBlocks; ++b) {
                if (starts[b + 1] < starts[b] || starts[b + 1] - starts[b] > ((f.maxStepEntries + 3u) & ~3u) || starts[b] % 4) return 7;
                for (uint32_t e = starts[b]; e < starts[b + 1]; ++e) {
                    const uint32_t word = f.words[e], slot = word & 1023u, rw = (word >> 10) & 63u, off = word >> 16;
                    if (word == bsmr::kSweepNoEntry) {   // padding to a multiple of 4 words, at the end of the list only
                        if (starts[b + 1] - e > 3) return 17;
                        continue;
                    }
                    if (e + 1 < starts[b + 1] && f.words[e + 1] != bsmr::kSweepNoEntry && e > starts[b] && f.words[e - 1] == bsmr::kSweepNoEntry) return 18;
                    if (rw >= 16 * PW) return 8;
                    const uint32_t j = rw / 16, r = rw % 16;
                    if (slot >= PW * 256 || slot / 256 != j) return 9;
                    const uint32_t reg = slot & 3u, lane = (slot >> 2) & 63u;
                    if (reg != (r & 3u) || lane / 16 != r / 4) return 10;   // accumulator layout: lane = 16 (r / 4) + c, register r % 4
                    const uint32_t c = lane & 15u;
                    const uint32_t idx = f.rowStart[(it * W + w) * (16 * PW) + rw] + off;
                    if (idx >= d->nnz || seen[idx]++) return 11;
                    if (f.panelRows[((size_t)item.group * GP + w * PW + j) * 16 + r] != wantRow[idx]) return 12;
                    if ((item.firstBlock + b) * 16 + c != wantCol[idx]) return 13;
                }
            }
            expectStart = starts[item.numBlocks];
        }
        if (expectStart - f.starts[item.startsBase] > f.maxItemWords) return 19;
    }
    if (f.words.size() < expectStart + bsmr::kSweepWordSlack) return 14;   // the loaders move an item's words in whole pseudo-images
    for (uint32_t b = 0; b < NCB; ++b)
        if (blockCover[b] != 1) return 15;
    for (uint32_t v = 0; v < d->nnz; ++v)
        if ((wantRow[v] != kNone) != (seen[v] != 0)) return 16;
    return 0;
}

#include "gemm_format.hpp"

// csrc/gemm_format.hpp read the way denseGemm reads it: every dense entry of the RPHM is listed exactly once, in the
// (item, wave, pass) list of its macro-tile / panel / column block, with the slab slot of its accumulator cell; the item
// order is gemmItemPlace's; lists are ordered by (row, column).
// out[0] items, [1] entries, [2] groups, [3] strips, [4] full grid, [5] bytes, [6] tiles, [7] longest list.
extern "C" int plancheck_gemm(const bsmr_rphm_desc* d, uint32_t PM, uint32_t NB, uint32_t balance, uint64_t* out) {
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    bsmr::HostDense hd;
    int st = bsmr::collectDense(d, hd);
    if (st != BSMR_OK) return 100 + st;
    bsmr::GemmFormatHost f;
    st = bsmr::packGemm(hd, PM, NB, f, balance != 0);
    if (st != BSMR_OK) return 200 + st;
    out[0] = f.items.size(); out[1] = hd.entries(); out[2] = f.numGroups; out[3] = f.numStrips; out[4] = f.fullGrid;
    out[5] = f.bytes(); out[6] = f.numTiles; out[7] = f.maxListWords;
    std::vector<uint32_t> wantRow(d->nnz, kNone), wantCol(d->nnz, kNone);
    uint64_t denseEntries = 0;
    for (uint32_t p = 0; p < d->num_row_panels; ++p)
        for (uint64_t b = d->block_offsets[p]; b < d->block_offsets[p + 1]; ++b)
            for (uint32_t i = 0; i < 256; ++i) {
                const uint32_t v = d->block_values[b * 256 + i];
                if (v == kNone) continue;
                const size_t slot = (size_t)p * 16 + i / 16;
                wantRow[v] = slot < d->num_nonzero_rows ? d->reordered_rows[slot] : kNone;
                wantCol[v] = d->dense_cols[b * 16 + i % 16];
                ++denseEntries;
            }
    if (hd.entries() != denseEntries) return 1;
    if (denseEntries == 0) return f.items.empty() ? 0 : 2;
    const uint32_t TM = PM * 16, m = PM / bsmr::kGemmWavesM, n = NB / bsmr::kGemmWavesN;
    const uint32_t Q = (m * n + bsmr::kGemmPassTiles - 1) / bsmr::kGemmPassTiles, L = bsmr::kGemmWaves * Q;
    if (f.passes != Q || f.panelRows.size() != (size_t)f.numGroups * TM) return 3;
    // colOf: every column that holds a dense entry in exactly one slot (unused slots name column 0); natural order without
    // balancing; with it, the strips' entry counts differ by at most two columns' worth (the columns were dealt by
    // descending count, back and forth) - counted below from the lists themselves
    if (f.colOf.size() != (size_t)f.numStrips * NB * 16) return 21;
    std::vector<uint32_t> degree(d->N, 0);
    uint32_t maxDegree = 0;
    for (uint32_t v = 0; v < d->nnz; ++v)
        if (wantCol[v] != kNone) maxDegree = std::max(maxDegree, ++degree[wantCol[v]]);
    {
        std::vector<uint32_t> slots(d->N, 0);
        for (size_t q = 0; q < f.colOf.size(); ++q) {
            const uint32_t c = f.colOf[q];
            if (c >= d->N) return 22;
            if (!balance && q < d->N && c != q) return 23;
            ++slots[c];
        }
        for (uint32_t c = 1; c < d->N; ++c)
            if (degree[c] ? slots[c] != 1 : slots[c] > 1) return 24;
        if (slots[0] == 0) return 25;
    }
    std::vector<uint64_t> perStrip(f.numStrips, 0);
    if (f.rowStart.size() != f.items.size() * TM || f.lists.size() != f.items.size() * (L + 1)) return 4;
    if (f.fullGrid != (f.items.size() == (size_t)f.numGroups * f.numStrips)) return 5;
    std::vector<uint8_t> seen(d->nnz, 0);
    uint64_t expectStart = 0, place = 0;
    for (size_t it = 0; it < f.items.size(); ++it) {
        const bsmr::GemmItem& item = f.items[it];
        if (item.group >= f.numGroups || item.firstBlock % NB || item.firstBlock / NB >= f.numStrips) return 6;
        if (item.listBase != it * (L + 1)) return 7;
        // the order of gemmItemPlace, macro-tiles without entries left out
        for (;; ++place) {
            if (place >= (uint64_t)f.numGroups * f.numStrips) return 8;
            uint32_t g, s;
            bsmr::gemmItemPlace((uint32_t)place, f.numGroups, f.numStrips, g, s);
            if (g == item.group && s == item.firstBlock / NB) break;
        }
        ++place;
        uint64_t itemEntries = 0;
        for (uint32_t w = 0; w < bsmr::kGemmWaves; ++w)
            for (uint32_t q = 0; q < Q; ++q) {
                const uint32_t b = f.lists[item.listBase + w * Q + q], e = f.lists[item.listBase + w * Q + q + 1];
                if (b != expectStart || e < b || (e - b) % 4 || e - b > f.maxListWords) return 9;   // lists follow each other, padded to 4
                expectStart = e;
                const uint32_t wm = w / bsmr::kGemmWavesN, wn = w % bsmr::kGemmWavesN;
                uint64_t lastKey = 0;
                bool padding = false;
                for (uint32_t i = b; i < e; ++i) {
                    const uint32_t word = f.words[i];
                    if (word == bsmr::kGemmNoEntry) {
                        if (e - i > 3) return 10;   // padding only at the end of a list
                        padding = true;
                        continue;
                    }
                    if (padding) return 11;
                    const uint32_t slot = word & 4095u, rw = (word >> 12) & 127u, off = word >> 19;
                    const uint32_t tp = slot >> 8, reg = (slot >> 6) & 3u, lane = slot & 63u;   // gemmSlabSlot
                    const uint32_t t = q * bsmr::kGemmPassTiles + tp;
                    if (t >= m * n || rw >= m * 16) return 12;
                    const uint32_t tm = t / n, tn = t % n, r = rw % 16;
                    if (rw / 16 != tm || reg != (r & 3u) || lane / 16 != r / 4) return 13;   // accumulator layout
                    const uint32_t c = lane & 15u;
                    const uint32_t rowInTile = wm * (TM / 2) + rw, colInTile = (wn * n + tn) * 16 + c;
                    const uint32_t idx = f.rowStart[it * TM + rowInTile] + off;
                    if (off >= bsmr::kGemmMaxOffset || idx >= d->nnz || seen[idx]++) return 14;
                    if (f.panelRows[(size_t)item.group * TM + rowInTile] != wantRow[idx]) return 15;
                    if (f.colOf[(size_t)item.firstBlock * 16 + colInTile] != wantCol[idx]) return 16;
                    const uint64_t key = ((uint64_t)rw << 32) | off;
                    if (i > b && key < lastKey) return 17;   // (row, position in P) order
                    lastKey = key;
                    ++itemEntries;
                    ++perStrip[item.firstBlock / NB];
                }
            }
        if (itemEntries == 0) return 18;   // macro-tiles without entries are not items
    }
    if (f.words.size() < expectStart + bsmr::kGemmWordSlack) return 19;
    for (uint32_t v = 0; v < d->nnz; ++v)
        if ((wantRow[v] != kNone) != (seen[v] != 0)) return 20;
    out[8] = *std::max_element(perStrip.begin(), perStrip.end());
    out[9] = *std::min_element(perStrip.begin(), perStrip.end());
    // where the natural order is lopsided (a natural strip above twice the mean), hot columns (more than twice the average
    // count) are dealt evenly and the others lie in natural order: no strip then holds more than its even share of the hot
    // entries (within two columns' worth) plus a full strip of columns at the threshold; otherwise the order is the natural one
    {
        const uint32_t S = f.numStrips, TN = NB * 16;
        uint64_t fullest = 0;
        for (uint32_t st = 0; st < S; ++st) {
            uint64_t sum = 0;
            for (uint64_t c = (uint64_t)st * TN; c < std::min<uint64_t>(d->N, (uint64_t)(st + 1) * TN); ++c) sum += degree[c];
            fullest = std::max(fullest, sum);
        }
        const bool lopsided = fullest * S > 2ull * denseEntries;
        out[10] = lopsided;
        const uint64_t threshold = 2ull * denseEntries / std::max<uint32_t>(1u, d->N) + 1ull;
        uint64_t hotEntries = 0;
        for (uint32_t c = 0; c < d->N; ++c)
            if (degree[c] > threshold) hotEntries += degree[c];
        if (balance && lopsided && out[8] > hotEntries / S + 2ull * maxDegree + (uint64_t)TN * threshold) return 26;
        if (balance && !lopsided)
            for (uint32_t c = 0; c < d->N; ++c)
                if (f.colOf[c] != c) return 27;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
#include "plan_evict.hpp"

// csrc/plan_evict.hpp on the host: packs `d` (default layout), evicts the outlier entries if a (block, row) is too wide for
// the 8-bit windows, packs again.  Checked here: every stored entry keeps its (panel, row in panel, column) and has exactly
// one place afterwards - a block cell or the residue -, the panels' column lists and block counts are unchanged, and no
// (new block, row) of the result spans kWindowMax or more.
// out[0] too wide before, [1] entries evicted, [2] too wide after, [3] tile bytes after (1 = 8-bit windows),
// [4] dense entries after, [5] residue entries after, [6] dense entries before.
extern "C" int plancheck_evict(const bsmr_rphm_desc* d, uint64_t* out) {
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const uint32_t P = d->num_row_panels;
    bsmr::PackOptions opt;
    bsmr::PackedPlan before;
    if (bsmr::packPlan(d, opt, before) != BSMR_OK) return 1;
    out[0] = before.tooWide;
    out[6] = before.numDenseEntries;
    bsmr::EvictedRphm ev;
    if (bsmr::evictWideRows(d, ev) != BSMR_OK) return 2;
    out[1] = ev.evicted;
    const bsmr_rphm_desc* e = ev.evicted ? &ev.desc : d;
    // the place of every entry, before and after
    auto places = [&](const bsmr_rphm_desc* x, std::vector<uint64_t>& where, std::vector<uint8_t>& dense) -> int {
        where.assign(x->nnz, ~0ull);
        dense.assign(x->nnz, 0);
        for (uint32_t q = 0; q < P; ++q) {
            for (uint64_t b = x->block_offsets[q]; b < x->block_offsets[q + 1]; ++b)
                for (uint32_t i = 0; i < 256; ++i) {
                    const uint32_t v = x->block_values[b * 256 + i];
                    if (v == kNone) continue;
                    if (v >= x->nnz || where[v] != ~0ull) return 3;
                    where[v] = ((uint64_t)q << 40) | ((uint64_t)(i / 16) << 32) | x->dense_cols[b * 16 + i % 16];
                    dense[v] = 1;
                }
            for (uint32_t i = x->sparse_value_offsets[q]; i < x->sparse_value_offsets[q + 1]; ++i) {
                const uint32_t v = x->sparse_values[i];
                if (v >= x->nnz || where[v] != ~0ull) return 4;
                where[v] = ((uint64_t)q << 40) | ((uint64_t)x->sparse_relative_rows[i] << 32) | x->sparse_col_indices[i];
            }
        }
        return 0;
    };
    std::vector<uint64_t> w0, w1;
    std::vector<uint8_t> d0, d1;
    if (int rc = places(d, w0, d0)) return rc;
    if (int rc = places(e, w1, d1)) return rc + 10;
    uint64_t moved = 0;
    for (uint32_t v = 0; v < d->nnz; ++v) {
        if (w0[v] != w1[v]) return 5;
        if (d1[v] && !d0[v]) return 6;   // nothing becomes dense
        moved += d0[v] && !d1[v];
    }
    if (moved != ev.evicted) return 7;
    for (uint32_t q = 0; q <= P; ++q)
        if (e->block_offsets[q] != d->block_offsets[q]) return 8;
    for (uint64_t i = 0; i < (uint64_t)d->block_offsets[P] * 16; ++i)
        if (e->dense_cols[i] != d->dense_cols[i]) return 9;
    bsmr::PackedPlan after;
    if (bsmr::packPlan(e, opt, after) != BSMR_OK) return 20;
    out[2] = after.tooWide;
    out[3] = after.staged ? 1 : (after.tiles16.empty() ? 4 : 2);
    out[4] = after.numDenseEntries;
    out[5] = after.numSparseEntries;
    if (after.numDenseEntries + after.numSparseEntries != d->nnz) return 21;
    if (ev.evicted && after.numDenseEntries + ev.evicted != before.numDenseEntries) return 22;
    if (ev.evicted && after.numBlocks != before.numBlocks) return 23;
    return 0;
}
