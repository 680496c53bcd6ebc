// Device format of the dense part, "tiles" form (round 2), and its host packer.  Pure host C++.
//
// What the reference keeps per dense block is a 1 KiB row-major tile of absolute CSR indices
// (reference src/BSMR.cpp:143-174).  Here the dense part of H consecutive row panels (a row GROUP,
// 16*H rows) is stored as
//   blockCols [16]   the group's union of dense columns, ascending column id, cut into 16s
//                    (padding -> column 0, never written)
//   blockInfo [4]    {first entry, panel mask | entry count << 16, c0 | c1 << 16, c2 | c3 << 16}: bit h of the
//                    mask = panel h has an entry in the block (tiles without entries are not multiplied);
//                    c_w = entries on the w-th quarter of the group's rows (the four waves of the shared-B
//                    kernel own a quarter each; entries are ordered by row, so a quarter is a sub-range)
//   entries   [u32]  one word per stored entry of the block, ordered by (row, column):
//                    row in group (8 bits) | column slot (4 bits) << 8 | offset (20 bits) << 12,
//                    offset = CSR index minus the work item's base for that row (itemRowBase)
// so the destination metadata is 4 bytes per ENTRY (the r01 format: 256 bytes per (panel, block) tile,
// whatever it held) and does not grow when panels are grouped.  The kernel (tile_kernels.hpp) multiplies
// a gathered 16-column B image against every panel of the group that has an entry in it, drops the
// 16x16 results into a small LDS slab and lets every lane carry ONE entry from the slab to P - the
// sparse mask is applied by construction, 64 entries per store instruction.
//
// An entry (row i of panel p, column j) is in the list only if column j is in panel p's OWN dense list:
// the dense / sparse assignment of every nnz stays the RPHM's (after promotion / folding), whatever H.
#pragma once

#include <algorithm>
#include <cstdint>
#include <memory>
#include <type_traits>
#include <utility>
#include <vector>

#include "bsmr_hip.h"
#include "plan_pack.hpp"

namespace bsmr {

constexpr uint32_t kTileMaxGroup = 16;        // panels per group the kernels are built for
constexpr uint32_t kTileMaxItemBlocks = 32;   // blocks per work item (column ids of an item sit in LDS)
constexpr uint32_t kTileOffsetBits = 20;      // entry offset from the item's row base
constexpr uint32_t kTileEntryChunk = 256;     // entries one LDS-DMA instruction moves (64 lanes x 16 B)

struct TileItem {
    uint32_t group;
    uint32_t first;   // first block (global block id)
    uint32_t count;   // blocks
    uint32_t pad;
};

// A vector whose resize() leaves new elements unwritten: the entry lists are tens to hundreds of MB that the next statement
// fills; value-initialising them first is one more pass over freshly mapped pages (20 ms for a reddit-like shard).
template <typename T>
struct DefaultInitAllocator : std::allocator<T> {
    template <typename U>
    struct rebind {
        using other = DefaultInitAllocator<U>;
    };
    DefaultInitAllocator() = default;
    template <typename U>
    DefaultInitAllocator(const DefaultInitAllocator<U>&) noexcept {}
    template <typename U>
    void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) {
        ::new (static_cast<void*>(p)) U;
    }
    template <typename U, typename... Args>
    void construct(U* p, Args&&... args) {
        ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
    }
};
template <typename T>
using RawVector = std::vector<T, DefaultInitAllocator<T>>;

// Dense entries of every panel (after the plan's own moves), ordered by (panel, column, row in panel).
// Kept by the plan on the host: formats for other group sizes are packed from it on demand.
struct HostDense {
    uint32_t M = 0, N = 0, nnz = 0, numPanels = 0;
    std::vector<uint32_t> panelRows;   // [P*16] original row ids (padding rows repeat row 0 of the list)
    std::vector<uint64_t> offsets;     // [P+1]
    RawVector<uint32_t> col, idx;
    RawVector<uint8_t> row;
    uint64_t entries() const { return offsets.empty() ? 0 : offsets.back(); }
};

struct TileFormatHost {
    uint32_t H = 1, numGroups = 0;
    uint32_t entryCap = 64;                    // LDS room for a block's entry words: max entries per block, rounded up to 64
    std::vector<uint32_t> groupRows;           // [G*16H]
    std::vector<uint32_t> blockCols;           // [NB*16]
    std::vector<uint32_t> blockInfo;           // [NB*4]
    std::vector<uint32_t> entries;             // every block's list padded to a multiple of 4, + one chunk at the end
    std::vector<TileItem> items;
    std::vector<uint32_t> itemRowBase;         // [items*16H]
    uint64_t numBlocks = 0, numTiles = 0, unionColumns = 0, numEntries = 0;
    size_t bytes() const {
        return 4 * (groupRows.size() + blockCols.size() + blockInfo.size() + entries.size() + itemRowBase.size()) +
               sizeof(TileItem) * items.size();
    }
};

// Dense part of `d` as per-panel (column, row, CSR index) lists.  Returns a bsmr_hip.h status.
inline int collectDense(const bsmr_rphm_desc* d, HostDense& out) {
    constexpr uint32_t kNone = 0xFFFFFFFFu;
    const uint32_t P = d->num_row_panels;
    out.M = d->M; out.N = d->N; out.nnz = d->nnz; out.numPanels = P;
    out.panelRows.assign((size_t)P * 16, 0);
    for (size_t i = 0; i < out.panelRows.size(); ++i) {
        const uint32_t row = i < d->num_nonzero_rows ? d->reordered_rows[i] : d->reordered_rows[0];
        if (row >= d->M) return BSMR_ERR_BAD_PLAN;
        out.panelRows[i] = row;
    }
    out.offsets.assign((size_t)P + 1, 0);
    for (uint32_t p = 0; p < P; ++p)
        if (d->block_offsets[p + 1] < d->block_offsets[p]) return BSMR_ERR_BAD_PLAN;
    // (a sweep over all block values, 1 KiB per block: by all host threads - 22 MB for the nips-like plan, 5 of the 10 ms of this function)
    parallelByWeight(P, d->block_offsets, 512, [&](size_t p0, size_t p1, size_t) {
        for (size_t p = p0; p < p1; ++p) {
            uint64_t n = 0;
            for (uint64_t b = d->block_offsets[p]; b < d->block_offsets[p + 1]; ++b)
                for (uint32_t i = 0; i < 256; ++i) n += d->block_values[b * 256 + i] != kNone;
            out.offsets[p + 1] = n;
        }
    });
    for (uint32_t p = 0; p < P; ++p) out.offsets[p + 1] += out.offsets[p];
    const uint64_t total = out.offsets[P];
    out.col.resize(total);
    out.idx.resize(total);
    out.row.resize(total);
    std::vector<uint8_t> bad(packThreads(), 0);
    parallelByWeight(P, d->block_offsets, 512, [&](size_t p0, size_t p1, size_t w) {
        std::vector<uint64_t> keys;   // (column, row) | position: one plain sort
        std::vector<uint32_t> vals;
        for (size_t p = p0; p < p1; ++p) {
            keys.clear();
            vals.clear();
            for (uint64_t b = d->block_offsets[p]; b < d->block_offsets[p + 1]; ++b)
                for (uint32_t c = 0; c < 16; ++c) {
                    const uint32_t col = d->dense_cols[b * 16 + c];
                    if (col >= d->N) {
                        if (col > d->N) bad[w] = 1;
                        continue;   // padding sentinel
                    }
                    for (uint32_t r = 0; r < 16; ++r) {
                        const uint32_t v = d->block_values[b * 256 + r * 16 + c];
                        if (v == kNone) continue;
                        if (v >= d->nnz) { bad[w] = 1; continue; }
                        keys.push_back(((uint64_t)col << 36) | ((uint64_t)r << 32) | vals.size());
                        vals.push_back(v);
                    }
                }
            if (keys.size() != out.offsets[p + 1] - out.offsets[p]) { bad[w] = 1; continue; }
            if (!std::is_sorted(keys.begin(), keys.end())) std::sort(keys.begin(), keys.end());   // (blocks in column order arrive sorted)
            uint64_t at = out.offsets[p];
            for (const uint64_t k : keys) {
                out.col[at] = (uint32_t)(k >> 36);
                out.row[at] = (uint8_t)((k >> 32) & 15u);
                out.idx[at] = vals[(size_t)(k & 0xFFFFFFFFu)];
                ++at;
            }
        }
    });
    for (const uint8_t b : bad)
        if (b) return BSMR_ERR_BAD_PLAN;
    return BSMR_OK;
}

// Blocks, tiles and union columns a format with H panels per group would have (no packing).
struct TileCensus {
    uint64_t blocks = 0, tiles = 0, unionColumns = 0, groups = 0;
};
inline TileCensus tileCensus(const HostDense& hd, uint32_t H) {
    const uint32_t P = hd.numPanels, G = (P + H - 1) / H;
    std::vector<TileCensus> part(packThreads());
    parallelChunks(G, 8, [&](size_t g0, size_t g1, size_t w) {
        std::vector<uint64_t> keys;   // column << 8 | panel in group
        TileCensus c;
        for (size_t g = g0; g < g1; ++g) {
            keys.clear();
            for (uint32_t h = 0; h < H && g * H + h < P; ++h) {
                const uint32_t p = (uint32_t)(g * H + h);
                uint32_t last = 0xFFFFFFFFu;
                for (uint64_t i = hd.offsets[p]; i < hd.offsets[p + 1]; ++i)
                    if (hd.col[i] != last) {
                        last = hd.col[i];
                        keys.push_back(((uint64_t)last << 8) | h);
                    }
            }
            if (keys.empty()) continue;
            std::sort(keys.begin(), keys.end());
            uint64_t cols = 0, tiles = 0;
            uint32_t mask = 0, last = 0xFFFFFFFFu;
            for (const uint64_t k : keys) {
                const uint32_t col = (uint32_t)(k >> 8);
                if (col != last) {
                    if (cols % 16 == 0) {   // a new block starts
                        tiles += __builtin_popcount(mask);
                        mask = 0;
                    }
                    ++cols;
                    last = col;
                }
                mask |= 1u << (k & 0xFFu);
            }
            tiles += __builtin_popcount(mask);
            c.unionColumns += cols;
            c.blocks += (cols + 15) / 16;
            c.tiles += tiles;
            ++c.groups;
        }
        part[w].blocks += c.blocks;
        part[w].tiles += c.tiles;
        part[w].unionColumns += c.unionColumns;
        part[w].groups += c.groups;
    });
    TileCensus out;
    for (const TileCensus& c : part) {
        out.blocks += c.blocks;
        out.tiles += c.tiles;
        out.unionColumns += c.unionColumns;
        out.groups += c.groups;
    }
    return out;
}

// Packs the tiles format for H panels per group.  blocksPerItem <= kTileMaxItemBlocks.
inline int packTiles(const HostDense& hd, uint32_t H, uint32_t blocksPerItem, TileFormatHost& out) {
    if (H == 0 || H > kTileMaxGroup || (H & (H - 1))) return BSMR_ERR_INVALID_ARG;
    blocksPerItem = std::max(1u, std::min(blocksPerItem, kTileMaxItemBlocks));
    const uint32_t P = hd.numPanels, G = (P + H - 1) / H, R = 16 * H;
    out = TileFormatHost{};
    out.H = H;
    out.numGroups = G;
    out.groupRows.assign((size_t)G * R, hd.panelRows.empty() ? 0 : hd.panelRows[0]);
    std::copy(hd.panelRows.begin(), hd.panelRows.end(), out.groupRows.begin());

    struct GroupOut {
        std::vector<uint32_t> cols, info, entries, rowBase;
        std::vector<TileItem> items;   // first = block within the group
        uint64_t tiles = 0, unionColumns = 0, numEntries = 0;
        uint32_t maxEntries = 0;
        bool overflow = false;
    };
    std::vector<GroupOut> groups(G);
    parallelChunks(G, 4, [&](size_t g0, size_t g1, size_t) {
        struct Key {
            uint32_t col, row, idx;   // row in group
        };
        std::vector<Key> keys;
        std::vector<uint32_t> blockStart, blockNcols;   // entry range / distinct columns of every block of the group
        std::vector<Key> sorted;
        for (size_t g = g0; g < g1; ++g) {
            GroupOut& go = groups[g];
            keys.clear();
            for (uint32_t h = 0; h < H && g * H + h < P; ++h) {
                const uint32_t p = (uint32_t)(g * H + h);
                for (uint64_t i = hd.offsets[p]; i < hd.offsets[p + 1]; ++i)
                    keys.push_back(Key{hd.col[i], h * 16u + hd.row[i], hd.idx[i]});
            }
            if (keys.empty()) continue;
            // by column (union columns ascending), then row, then index (repeated entries of one cell keep CSR order)
            std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) {
                return a.col != b.col ? a.col < b.col : a.row != b.row ? a.row < b.row : a.idx < b.idx;
            });
            // blocks of 16 distinct columns; a block whose entries exceed the cap the kernel has LDS room for
            // (128 per panel: tiles half full) is cut at a column boundary into blocks with fewer columns
            const uint32_t cap = 128u * H;
            sorted.clear();
            blockStart.clear();
            blockNcols.clear();
            size_t i = 0;
            while (i < keys.size()) {
                const size_t b0 = i;
                uint32_t cols[16], ncols = 0, mask = 0;
                while (i < keys.size()) {
                    if (ncols == 0 || keys[i].col != cols[ncols - 1]) {
                        if (ncols == 16) break;
                        // entries of this column
                        size_t j = i;
                        while (j < keys.size() && keys[j].col == keys[i].col) ++j;
                        if (ncols > 0 && (j - b0) > cap) break;   // would overflow: close the block before this column
                        cols[ncols++] = keys[i].col;
                    }
                    mask |= 1u << (keys[i].row >> 4);
                    ++i;
                }
                const uint32_t count = (uint32_t)(i - b0);   // <= cap + 16H: fits the 16-bit count field
                for (uint32_t c = 0; c < 16; ++c) go.cols.push_back(c < ncols ? cols[c] : 0u);
                uint32_t quarter[4] = {0, 0, 0, 0};
                for (size_t e = b0; e < i; ++e) ++quarter[keys[e].row / (4u * H)];
                go.info.push_back((uint32_t)b0);   // patched to the global entry index below
                go.info.push_back(mask | (count << 16));
                go.info.push_back(quarter[0] | (quarter[1] << 16));
                go.info.push_back(quarter[2] | (quarter[3] << 16));
                go.tiles += __builtin_popcount(mask);
                go.unionColumns += ncols;
                go.maxEntries = std::max(go.maxEntries, count);
                blockStart.push_back((uint32_t)b0);
                blockNcols.push_back(ncols);
            }
            blockStart.push_back((uint32_t)keys.size());
            const uint32_t nb = (uint32_t)blockStart.size() - 1;
            // work items: runs of <= blocksPerItem blocks; row bases; entries in (row, column) order per block
            for (uint32_t first = 0; first < nb; first += blocksPerItem) {
                const uint32_t count = std::min(blocksPerItem, nb - first);
                go.items.push_back(TileItem{(uint32_t)g, first, count, 0});
                const size_t rb = go.rowBase.size();
                go.rowBase.resize(rb + R, 0xFFFFFFFFu);
                for (uint32_t e = blockStart[first]; e < blockStart[first + count]; ++e)
                    go.rowBase[rb + keys[e].row] = std::min(go.rowBase[rb + keys[e].row], keys[e].idx);
                for (uint32_t b = first; b < first + count; ++b) {
                    const uint32_t e0 = blockStart[b], e1 = blockStart[b + 1];
                    go.info[4 * (size_t)b] = (uint32_t)go.entries.size();   // group-local, multiple of 4
                    const uint32_t* cols = &go.cols[(size_t)b * 16];
                    const uint32_t used = blockNcols[b];   // columns ascending, padding (0) only behind them
                    sorted.assign(keys.begin() + e0, keys.begin() + e1);
                    std::sort(sorted.begin(), sorted.end(), [](const Key& a, const Key& b2) {
                        return a.row != b2.row ? a.row < b2.row : a.col != b2.col ? a.col < b2.col : a.idx < b2.idx;
                    });
                    for (const Key& k : sorted) {
                        const uint32_t slot = (uint32_t)(std::lower_bound(cols, cols + used, k.col) - cols);
                        const uint32_t off = k.idx - go.rowBase[rb + k.row];
                        if (off >> kTileOffsetBits) go.overflow = true;
                        go.entries.push_back(k.row | (slot << 8) | (off << 12));
                    }
                    while (go.entries.size() & 3u) go.entries.push_back(0u);
                }
                for (uint32_t r = 0; r < R; ++r)
                    if (go.rowBase[rb + r] == 0xFFFFFFFFu) go.rowBase[rb + r] = 0;
            }
            go.numEntries = keys.size();
        }
    });
    // concatenate
    uint64_t nb = 0, ne = 0, ni = 0;
    for (const GroupOut& go : groups) {
        if (go.overflow) return BSMR_ERR_BAD_PLAN;   // offsets beyond 2^21 of one row inside an item: not representable
        nb += go.info.size() / 4;
        ne += go.entries.size();
        ni += go.items.size();
    }
    if (ne + kTileEntryChunk > 0xFFFFFFF0ull || nb > 0xFFFFFFF0ull) return BSMR_ERR_BAD_PLAN;
    out.blockCols.reserve(nb * 16);
    out.blockInfo.reserve(nb * 4);
    out.entries.reserve(ne + kTileEntryChunk);
    out.items.reserve(ni);
    out.itemRowBase.reserve(ni * R);
    uint32_t maxEntries = 0;
    for (GroupOut& go : groups) {
        const uint32_t blockBase = (uint32_t)(out.blockInfo.size() / 4), entryBase = (uint32_t)out.entries.size();
        out.blockCols.insert(out.blockCols.end(), go.cols.begin(), go.cols.end());
        for (size_t b = 0; b < go.info.size() / 4; ++b) {
            out.blockInfo.push_back(go.info[4 * b] + entryBase);
            out.blockInfo.push_back(go.info[4 * b + 1]);
            out.blockInfo.push_back(go.info[4 * b + 2]);
            out.blockInfo.push_back(go.info[4 * b + 3]);
        }
        out.entries.insert(out.entries.end(), go.entries.begin(), go.entries.end());
        for (TileItem it : go.items) {
            it.first += blockBase;
            out.items.push_back(it);
        }
        out.itemRowBase.insert(out.itemRowBase.end(), go.rowBase.begin(), go.rowBase.end());
        out.numTiles += go.tiles;
        out.unionColumns += go.unionColumns;
        out.numEntries += go.numEntries;
        maxEntries = std::max(maxEntries, go.maxEntries);
        GroupOut().cols.swap(go.cols);
        GroupOut().entries.swap(go.entries);
    }
    out.entries.resize(out.entries.size() + kTileEntryChunk, 0u);   // the last block's DMA reads whole chunks
    out.numBlocks = out.blockInfo.size() / 4;
    out.entryCap = std::max(64u, (maxEntries + 63u) / 64u * 64u);   // LDS room per block: whole 256-byte lines
    // items in the order of their first column: XCD x executes the x-th eighth of the list and keeps one column
    // range of B in its L2 (as in the r01 format); the per-item row bases are permuted along
    std::vector<uint32_t> order(out.items.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return out.blockCols[(size_t)out.items[a].first * 16] < out.blockCols[(size_t)out.items[b].first * 16];
    });
    std::vector<TileItem> items(order.size());
    std::vector<uint32_t> rb(out.itemRowBase.size());
    for (size_t i = 0; i < order.size(); ++i) {
        items[i] = out.items[order[i]];
        std::copy_n(out.itemRowBase.begin() + (size_t)order[i] * R, R, rb.begin() + i * R);
    }
    out.items.swap(items);
    out.itemRowBase.swap(rb);
    return BSMR_OK;
}

}  // namespace bsmr
