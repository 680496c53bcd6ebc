// Host-side packing of the reference-layout RPHM arrays (bsmr_rphm_desc) into the
// device format the gfx950 kernels read.  Pure host C++ (no HIP), used by
// bsmr_plan_create.
//
// Dense part.  H consecutive row panels form a row GROUP.  For every group the
// union of its panels' dense columns is ordered (by column id by default: an
// XCD then works on one column range of B, and a workgroup's outputs for one
// row are neighbours in P) and cut into 16-column blocks.  A block stores
//   blockCols [16]      column ids (padding -> column 0, never written)
//   blockMask           bit h set iff panel h of the group has an entry in the block
//   tiles     [H][256]  per panel the destination of every accumulator element, in
//                       accumulator (lane-major) order: element [4*lane + i] belongs
//                       to tile row 4*(lane>>4)+i, tile column lane&15.
// An entry (row i of panel p, column j) appears in the tile of (p, j) only if
// column j is in panel p's OWN dense list: entries of sparse columns stay on the
// sparse path even when another panel of the group made column j dense.
//
// Two destination encodings:
//   STAGED (default): a work item (run of blocks of one group) writes, for every
//     row, into a window of < 255 consecutive entries of P.  Tiles hold 8-bit
//     offsets into that window (0xFF = none); the kernel assembles the windows in
//     LDS and writes them out with coalesced stores, guided by a per-row ownership
//     bitmap (window positions owned by the sparse path or by other items are
//     skipped).  Items are cut so that every window fits.
//   DIRECT (fallback when a single block already spans >= 255 entries of some row,
//     e.g. CSR rows whose columns are not sorted): 16-bit (or 32-bit) offsets from
//     the row's first dense entry, scattered straight to P.
//
// Sparse part: the reference's three arrays, with the relative row packed to a byte.
#pragma once

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include <exception>

#include "bsmr_hip.h"
#include "sddmm_kernels.hpp"

namespace bsmr {

// Host threads for plan packing: hardware threads capped by the control group's CPU quota and
// BSMR_HOST_THREADS (same rule as util::hostThreads of the host library).
inline unsigned packThreads() {
    static unsigned cached = 0;
    if (cached) return cached;
    long limit = (long)std::max(1u, std::thread::hardware_concurrency());
    if (const char* env = std::getenv("BSMR_HOST_THREADS")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v > 0) limit = std::min(limit, v);
    } else if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        long long period = 0;
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm' && period > 0) {
            const long long quota = std::strtoll(q, nullptr, 10);
            if (quota > 0) limit = std::min<long>(limit, (long)((quota + period - 1) / period));
        }
        std::fclose(f);
    }
    cached = (unsigned)std::max(1L, std::min(limit, 32L));
    return cached;
}

// fn(begin, end, worker) over [0, n) in contiguous chunks, one per worker
template <typename F>
inline void parallelChunks(size_t n, size_t minPerWorker, F fn) {
    const size_t workers = std::max<size_t>(1, std::min<size_t>(packThreads(), n / std::max<size_t>(1, minPerWorker)));
    if (workers <= 1) {
        fn((size_t)0, n, (size_t)0);
        return;
    }
    // what a worker throws (std::bad_alloc from a vector, a length_error) is carried to the calling thread, which
    // has the C ABI's catch around it; left in the worker it would end the process (std::terminate)
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> failed(workers);
    const size_t per = (n + workers - 1) / workers;
    for (size_t w = 0; w < workers; ++w) {
        const size_t b = std::min(n, w * per), e = std::min(n, b + per);
        if (b < e)
            pool.emplace_back([=, &fn, &failed]() {
                try {
                    fn(b, e, w);
                } catch (...) {
                    failed[w] = std::current_exception();
                }
            });
    }
    for (std::thread& t : pool) t.join();
    for (const std::exception_ptr& f : failed)
        if (f) std::rethrow_exception(f);
}

// The same with chunks of about equal WEIGHT: prefix[i] = weight of [0, i) (n + 1 entries, ascending).  Panels of a clustered
// matrix differ by two orders of magnitude in their block counts; equal panel counts would leave most workers idle.
template <typename W, typename F>
inline void parallelByWeight(size_t n, const W* prefix, uint64_t minPerWorker, F fn) {
    const uint64_t total = n ? (uint64_t)prefix[n] - (uint64_t)prefix[0] : 0;
    const size_t workers = std::max<size_t>(1, std::min<size_t>(packThreads(), (size_t)(total / std::max<uint64_t>(1, minPerWorker))));
    if (workers <= 1) {
        fn((size_t)0, n, (size_t)0);
        return;
    }
    std::vector<size_t> cut(workers + 1, n);
    cut[0] = 0;
    for (size_t w = 1; w < workers; ++w) {
        const uint64_t want = (uint64_t)prefix[0] + total * w / workers;
        cut[w] = std::max<size_t>(cut[w - 1], (size_t)(std::lower_bound(prefix, prefix + n + 1, want, [](const W& a, uint64_t b) { return (uint64_t)a < b; }) - prefix));
        cut[w] = std::min(cut[w], n);
    }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> failed(workers);
    for (size_t w = 0; w < workers; ++w) {
        const size_t b = cut[w], e = cut[w + 1];
        if (b < e)
            pool.emplace_back([=, &fn, &failed]() {
                try {
                    fn(b, e, w);
                } catch (...) {
                    failed[w] = std::current_exception();
                }
            });
    }
    for (std::thread& t : pool) t.join();
    for (const std::exception_ptr& f : failed)
        if (f) std::rethrow_exception(f);
}

struct PackOptions {
    int group = 1;            // panels per group: 1, 2 or 4
    int blocksPerItem = 32;   // dense blocks per workgroup item (upper bound)
    int sparsePerItem = 256;  // sparse entries per workgroup item
    bool forceWideTiles = false;
    bool columnOrder = true;  // blocks in column-id order, items sorted by first column
    bool staged = true;       // try the staged (LDS window) destination encoding
    bool maskTiles = true;    // ... and from it the mask form (48 bytes per tile) when every tile row's entries are consecutive
    int freeResidue = 0;      // 1: residue in global column order instead of per panel (cross-check only)
    // launch order of the dense items (with columnOrder): 0 = by first column; 1 = row chunk (1/8 of the items: the XCD's
    // rows of A stay in its L2), then first column; 2 = column window of orderWindow columns, then group, then column
    int itemOrder = 0;
    uint32_t orderWindow = 8192;
    uint32_t itemSpan = 0;    // > 0: an item's blocks start inside itemSpan columns of its first block's first column
};

struct PackedPlan {
    uint32_t H = 1;
    uint32_t numGroups = 0;
    bool staged = false;
    bool tooWide = false;                 // the staged encoding was asked for and one (block, row) spans >= kWindowMax entries
    std::vector<uint32_t> panelRows;      // [P*16]   (sparse kernel)
    std::vector<uint32_t> groupRows;      // [G*16H]
    std::vector<uint32_t> rowBase;        // DIRECT: [G*16H] per group row; STAGED: [items*16H] window base
    std::vector<uint16_t> winLen;         // STAGED: [items*16H] window length (0 = row unused by the item)
    std::vector<uint32_t> winMask;        // STAGED: [items*16H*8] ownership bitmap of the window
    std::vector<uint32_t> blockCols;      // [NB*16]
    std::vector<uint8_t> tiles8;          // STAGED: [NB*H*256]
    std::vector<uint32_t> tilesMask;      // mask form of the STAGED tiles: [NB*H*12] (then tiles8 is empty)
    std::vector<uint16_t> tiles16;        // DIRECT
    std::vector<uint32_t> tiles32;        // DIRECT, rows with >= 65535 entries
    std::vector<uint8_t> blockMask;       // [NB]
    std::vector<DenseItem> denseItems;
    std::vector<uint32_t> entryCol, entryDst;
    std::vector<uint8_t> entryRow;        // row inside the panel (panel form)
    std::vector<uint32_t> entryRowId;     // row id (free form: entries in global column order, no panels)
    bool freeResidue = false;
    std::vector<SparseItem> sparseItems;
    uint64_t numBlocks = 0, numTiles = 0, numDenseEntries = 0, numSparseEntries = 0;
    uint64_t unionColumns = 0;            // sum over groups of distinct dense columns
};

constexpr uint32_t kWindow = 256;         // LDS floats per staged row
constexpr uint32_t kWindowMax = 255;      // usable window length (0xFF is the null offset)

namespace detail {

struct ColumnUse {
    uint32_t col;
    uint32_t count;              // entries of this column in the dense parts of the group
    int32_t slot[kMaxGroup];     // position in each panel's dense list, -1 = not dense there
};

// Per-column scratch of unionColumns: position of a column in the current group's list (valid where the
// stamp equals the group's).  Sized to N + 1 once per packPlan call.
struct ColumnIndex {
    std::vector<uint32_t> stamp, position;
    uint32_t epoch = 0;
    explicit ColumnIndex(size_t columns) : stamp(columns, 0), position(columns, 0) {}
};

// Distinct dense columns of the panels [p0, p0+h) with their per-panel slots.
inline void unionColumns(const bsmr_rphm_desc* d, uint32_t p0, uint32_t h, uint32_t P,
                         std::vector<ColumnUse>& out, ColumnIndex& index, bool byColumnId) {
    out.clear();
    if (++index.epoch == 0) {
        std::fill(index.stamp.begin(), index.stamp.end(), 0u);
        index.epoch = 1;
    }
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
            if (index.stamp[col] != index.epoch) {
                index.stamp[col] = index.epoch;
                index.position[col] = (uint32_t)out.size();
                ColumnUse u{col, 0, {-1, -1, -1, -1}};
                out.push_back(u);
            }
            ColumnUse& use = out[index.position[col]];
            use.count += cnt;
            use.slot[k] = (int32_t)t;
        }
    }
    if (!byColumnId) {
        std::sort(out.begin(), out.end(), [](const ColumnUse& a, const ColumnUse& b) {
            return a.count != b.count ? a.count > b.count : a.col < b.col;
        });
        return;
    }
    std::sort(out.begin(), out.end(), [](const ColumnUse& a, const ColumnUse& b) { return a.col < b.col; });
    if (h == 1) return;
    // Grouped plans: a block of 16 union columns costs one tile per panel that has an
    // entry in it.  Mixing columns shared by several panels with columns only one
    // panel uses would make almost every (block, panel) tile non-empty, so the columns
    // are partitioned: first the columns at least two panels share (column-id order),
    // then the single-panel columns, dealt 16 at a time round-robin over the panels so
    // that the four waves (one per panel) of a batch of blocks all have a tile to do.
    std::vector<ColumnUse> shared, single[kMaxGroup];
    for (const ColumnUse& c : out) {
        int panels = 0, only = 0;
        for (uint32_t k = 0; k < h; ++k)
            if (c.slot[k] >= 0) { ++panels; only = (int)k; }
        if (panels >= 2) shared.push_back(c);
        else single[only].push_back(c);
    }
    out.swap(shared);
    size_t cursor[kMaxGroup] = {0, 0, 0, 0};
    for (bool any = true; any;) {
        any = false;
        for (uint32_t k = 0; k < h; ++k) {
            const size_t n = std::min<size_t>(16, single[k].size() - cursor[k]);
            if (!n) continue;
            out.insert(out.end(), single[k].begin() + cursor[k], single[k].begin() + cursor[k] + n);
            cursor[k] += n;
            any = true;
        }
    }
}

}  // namespace detail

// Distinct dense columns summed over groups of `h` panels, without building anything: what a grouped
// format would gather.
inline uint64_t countUnionColumns(const bsmr_rphm_desc* d, uint32_t h) {
    const uint32_t P = d->num_row_panels;
    std::vector<uint32_t> stamp((size_t)d->N + 1, 0);
    uint64_t total = 0;
    for (uint32_t p0 = 0, group = 1; p0 < P; p0 += h, ++group)
        for (uint32_t p = p0; p < std::min(P, p0 + h); ++p)
            for (uint64_t i = (uint64_t)d->block_offsets[p] * 16; i < (uint64_t)d->block_offsets[p + 1] * 16; ++i) {
                const uint32_t col = d->dense_cols[i];
                if (col >= d->N || stamp[col] == group) continue;
                stamp[col] = group;
                ++total;
            }
    return total;
}

inline int packResidue(const bsmr_rphm_desc* d, const PackOptions& opt, PackedPlan& out);

// panelRows: the reordered rows, padded to whole panels
inline int packRows(const bsmr_rphm_desc* d, PackedPlan& out) {
    const uint32_t P = d->num_row_panels;
    out.panelRows.assign((size_t)P * 16, 0);
    for (size_t i = 0; i < out.panelRows.size(); ++i) {
        const uint32_t row = i < d->num_nonzero_rows ? d->reordered_rows[i] : d->reordered_rows[0];
        if (row >= d->M) return BSMR_ERR_BAD_PLAN;
        out.panelRows[i] = row;
    }
    return BSMR_OK;
}

// Returns a bsmr_hip.h status.
inline int packPlan(const bsmr_rphm_desc* d, const PackOptions& opt, PackedPlan& out) {
    // BSMR_PACK_TIMING=1: phase times on stderr
    const bool timing = std::getenv("BSMR_PACK_TIMING") != nullptr;
    auto clock0 = std::chrono::steady_clock::now();
    auto phase = [&](const char* name) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[packPlan] %-28s %8.1f ms\n", name, std::chrono::duration<double, std::milli>(now - clock0).count());
        clock0 = now;
    };
    const uint32_t P = d->num_row_panels;
    const uint64_t numRefBlocks = d->block_offsets[P];
    constexpr uint32_t kNone = 0xFFFFFFFFu;

    for (uint32_t p = 0; p < P; ++p)
        if (d->block_offsets[p + 1] < d->block_offsets[p] ||
            d->sparse_value_offsets[p + 1] < d->sparse_value_offsets[p])
            return BSMR_ERR_BAD_PLAN;

    // ---- rows -------------------------------------------------------------
    if (const int st = packRows(d, out)) return st;

    // first dense entry of every panel row (base of the DIRECT offsets)
    std::vector<uint32_t> panelRowBase((size_t)P * 16, 0);
    uint32_t maxOffset = 0;
    out.numDenseEntries = 0;
    {
        const unsigned workers = packThreads();
        std::vector<uint64_t> entries(workers, 0);
        std::vector<uint32_t> widest(workers, 0);
        std::vector<uint8_t> bad(workers, 0);
        parallelChunks(P, 64, [&](size_t q0, size_t q1, size_t w) {
            for (size_t p = q0; p < q1; ++p) {
                uint32_t lo[16], hi[16];
                for (int r = 0; r < 16; ++r) { lo[r] = kNone; hi[r] = 0; }
                for (uint64_t b = d->block_offsets[p]; b < d->block_offsets[p + 1]; ++b) {
                    const uint32_t* tile = d->block_values + b * 256;
                    for (uint32_t i = 0; i < 256; ++i) {
                        const uint32_t v = tile[i];
                        if (v == kNone) continue;
                        if (v >= d->nnz) { bad[w] = 1; continue; }
                        ++entries[w];
                        lo[i >> 4] = std::min(lo[i >> 4], v);
                        hi[i >> 4] = std::max(hi[i >> 4], v);
                    }
                }
                for (int r = 0; r < 16; ++r) {
                    if (lo[r] == kNone) continue;
                    panelRowBase[p * 16 + r] = lo[r];
                    widest[w] = std::max(widest[w], hi[r] - lo[r]);
                }
            }
        });
        for (unsigned w = 0; w < workers; ++w) {
            if (bad[w]) return BSMR_ERR_BAD_PLAN;
            out.numDenseEntries += entries[w];
            maxOffset = std::max(maxOffset, widest[w]);
        }
    }
    for (uint64_t i = 0; i < numRefBlocks * 16; ++i)
        if (d->dense_cols[i] > d->N) return BSMR_ERR_BAD_PLAN;

    phase("rows + row bases");
    // ---- groups -------------------------------------------------------------
    const uint32_t H = opt.group == 2 || opt.group == 4 ? (uint32_t)opt.group : 1u;
    out.H = H;
    const uint32_t G = (P + H - 1) / H;
    const uint32_t R = 16 * H;  // rows per group
    out.numGroups = G;
    out.groupRows.assign((size_t)G * R, out.panelRows.empty() ? 0 : out.panelRows[0]);
    std::copy(out.panelRows.begin(), out.panelRows.end(), out.groupRows.begin());
    std::vector<uint32_t> groupRowBase((size_t)G * R, 0);
    std::copy(panelRowBase.begin(), panelRowBase.end(), groupRowBase.begin());

    // ---- dense blocks: columns, masks and absolute destinations -----------------
    // phase 1 (parallel over groups): each group's column list
    std::vector<std::vector<detail::ColumnUse>> groupCols(G);
    parallelChunks(G, 8, [&](size_t g0, size_t g1, size_t) {
        detail::ColumnIndex index((size_t)d->N + 1);
        for (size_t gi = g0; gi < g1; ++gi)
            detail::unionColumns(d, (uint32_t)gi * H, H, P, groupCols[gi], index, opt.columnOrder);
    });
    std::vector<uint32_t> absTiles;               // [NB*H*256] CSR index or kNone, lane-major
    std::vector<uint32_t> groupFirstBlock(G + 1, 0);
    out.unionColumns = 0;
    for (uint32_t gi = 0; gi < G; ++gi) {
        out.unionColumns += groupCols[gi].size();
        groupFirstBlock[gi + 1] = groupFirstBlock[gi] + (uint32_t)((groupCols[gi].size() + 15) / 16);
    }
    const size_t totalBlocks = groupFirstBlock[G];
    out.blockCols.assign(totalBlocks * 16, 0);
    out.blockMask.assign(totalBlocks, 0);
    absTiles.assign(totalBlocks * H * 256, kNone);
    // phase 2 (parallel over groups): columns, masks and absolute destinations at their final places
    parallelChunks(G, 8, [&](size_t g0, size_t g1, size_t) {
        for (size_t gi = g0; gi < g1; ++gi) {
            const std::vector<detail::ColumnUse>& cols = groupCols[gi];
            const uint32_t p0 = (uint32_t)gi * H;
            const uint64_t firstBlock = groupFirstBlock[gi];
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
                        if (v == kNone) continue;
                        const uint32_t lane = (r >> 2) * 16 + cc, i = r & 3u;
                        absTiles[(b * H + k) * 256 + lane * 4 + i] = v;
                        out.blockMask[b] |= (uint8_t)(1u << k);
                    }
                }
            }
        }
    });
    { std::vector<std::vector<detail::ColumnUse>>().swap(groupCols); }
    phase("blocks (columns, tiles)");
    out.numBlocks = out.blockMask.size();
    out.numTiles = 0;
    for (const uint8_t m : out.blockMask) out.numTiles += __builtin_popcount(m);

    // tile element -> group row: lane-major index e = 4*lane + i of panel k
    auto rowOfElement = [](uint32_t k, uint32_t e) { return k * 16u + 4u * ((e >> 2) >> 4) + (e & 3u); };

    // ---- work items + destination encoding ----------------------------------------
    const uint32_t perItem = (uint32_t)std::max(1, opt.blocksPerItem);
    out.denseItems.clear();
    out.staged = opt.staged && !opt.forceWideTiles;
    std::vector<uint32_t> itemLo;  // STAGED: [items*R] window base per item row (kNone = unused)
    if (out.staged) {
      // groups are independent: every worker cuts the items of a contiguous range of groups, the ranges are
      // concatenated in group order afterwards
      const size_t ranges = std::max<size_t>(1, std::min<size_t>(packThreads(), G / 8));
      std::vector<std::vector<DenseItem>> rangeItems(ranges);
      std::vector<std::vector<uint32_t>> rangeLo(ranges);
      std::vector<uint8_t> rangeTooWide(ranges, 0);
      const size_t perRange = (G + ranges - 1) / ranges;
      parallelChunks(ranges, 1, [&](size_t r0, size_t r1, size_t) {
       for (size_t range = r0; range < r1; ++range) {
        std::vector<DenseItem>& myItems = rangeItems[range];
        std::vector<uint32_t>& myLo = rangeLo[range];
        bool staged = true;
        std::vector<uint32_t> lo(R), hi(R), blo(R), bhi(R);
        const uint32_t gEnd = (uint32_t)std::min<size_t>(G, (range + 1) * perRange);
        for (uint32_t gi = (uint32_t)(range * perRange); gi < gEnd && staged; ++gi) {
            uint32_t itemFirst = groupFirstBlock[gi], count = 0;
            std::fill(lo.begin(), lo.end(), kNone);
            std::fill(hi.begin(), hi.end(), 0u);
            auto closeItem = [&]() {
                if (!count) return;
                myItems.push_back(DenseItem{gi, itemFirst, count, 0});
                myLo.insert(myLo.end(), lo.begin(), lo.end());
                std::fill(lo.begin(), lo.end(), kNone);
                std::fill(hi.begin(), hi.end(), 0u);
                count = 0;
            };
            for (uint32_t b = groupFirstBlock[gi]; b < groupFirstBlock[gi + 1]; ++b) {
                std::fill(blo.begin(), blo.end(), kNone);
                std::fill(bhi.begin(), bhi.end(), 0u);
                for (uint32_t k = 0; k < H; ++k)
                    for (uint32_t e = 0; e < 256; ++e) {
                        const uint32_t v = absTiles[((size_t)b * H + k) * 256 + e];
                        if (v == kNone) continue;
                        const uint32_t row = rowOfElement(k, e);
                        blo[row] = std::min(blo[row], v);
                        bhi[row] = std::max(bhi[row], v);
                    }
                bool fits = count < perItem;
                if (opt.itemSpan && count && out.blockCols[(size_t)b * 16] / opt.itemSpan != out.blockCols[(size_t)itemFirst * 16] / opt.itemSpan) fits = false;
                for (uint32_t r = 0; r < R; ++r) {
                    if (blo[r] == kNone) continue;
                    if (bhi[r] - blo[r] >= kWindowMax) staged = false;  // one block alone is too wide
                    if (lo[r] != kNone && std::max(hi[r], bhi[r]) - std::min(lo[r], blo[r]) >= kWindowMax) fits = false;
                }
                if (!staged) break;
                if (!fits) closeItem();
                if (count == 0) itemFirst = b;
                for (uint32_t r = 0; r < R; ++r) {
                    if (blo[r] == kNone) continue;
                    lo[r] = std::min(lo[r], blo[r]);
                    hi[r] = std::max(hi[r], bhi[r]);
                }
                ++count;
            }
            closeItem();
        }
        rangeTooWide[range] = !staged;
       }
      });
      for (size_t range = 0; range < ranges; ++range) {
          if (rangeTooWide[range]) out.staged = false, out.tooWide = true;
          out.denseItems.insert(out.denseItems.end(), rangeItems[range].begin(), rangeItems[range].end());
          itemLo.insert(itemLo.end(), rangeLo[range].begin(), rangeLo[range].end());
      }
    }
    phase("items (window scan)");
    if (out.staged) {
        const size_t numItems = out.denseItems.size();
        out.rowBase.assign(numItems * R, 0);
        out.winLen.assign(numItems * R, 0);
        out.winMask.assign(numItems * R * (kWindow / 32), 0);
        out.tiles8.assign(out.numBlocks * H * 256, 0xFF);
        out.tiles16.clear();
        out.tiles32.clear();
        parallelChunks(numItems, 64, [&](size_t i0, size_t i1, size_t) {
            for (size_t it = i0; it < i1; ++it) {
                const DenseItem& item = out.denseItems[it];
                for (uint32_t r = 0; r < R; ++r)
                    if (itemLo[it * R + r] != kNone) out.rowBase[it * R + r] = itemLo[it * R + r];
                for (uint32_t b = item.first; b < item.first + item.count; ++b)
                    for (uint32_t k = 0; k < H; ++k)
                        for (uint32_t e = 0; e < 256; ++e) {
                            const size_t at = ((size_t)b * H + k) * 256 + e;
                            const uint32_t v = absTiles[at];
                            if (v == kNone) continue;
                            const uint32_t row = rowOfElement(k, e);
                            const uint32_t off = v - itemLo[it * R + row];
                            out.tiles8[at] = (uint8_t)off;
                            out.winMask[(it * R + row) * (kWindow / 32) + off / 32] |= 1u << (off % 32);
                            out.winLen[it * R + row] = std::max<uint16_t>(out.winLen[it * R + row], (uint16_t)(off + 1));
                        }
            }
        });
        // Mask form: a tile row's window offsets are consecutive (blocks in column order, sorted CSR rows, no entry of
        // another path in between) in every tile -> 16-bit column mask + first offset per row, 48 bytes per tile.
        if (opt.maskTiles) {
            const size_t numTiles = out.numBlocks * H;
            std::vector<uint32_t> words(numTiles * 12, 0);
            std::vector<uint8_t> bad(packThreads(), 0);
            parallelChunks(numTiles, 256, [&](size_t t0, size_t t1, size_t w) {
                for (size_t t = t0; t < t1 && !bad[w]; ++t) {
                    const uint8_t* tile = &out.tiles8[t * 256];
                    for (uint32_t row = 0; row < 16; ++row) {
                        uint32_t mask = 0, first = 0, expect = 0;
                        for (uint32_t c = 0; c < 16; ++c) {
                            const uint32_t off = tile[((row >> 2) * 16 + c) * 4 + (row & 3)];   // lane-major: lane = (row / 4) * 16 + c
                            if (off == 0xFF) continue;
                            if (!mask) first = off;
                            else if (off != expect) { bad[w] = 1; break; }
                            expect = off + 1;
                            mask |= 1u << c;
                        }
                        uint32_t* group = &words[t * 12 + (row >> 2) * 3];
                        group[(row & 3) >> 1] |= mask << (16 * (row & 1));
                        group[2] |= first << (8 * (row & 3));
                    }
                }
            });
            bool ok = true;
            for (const uint8_t b : bad) ok = ok && !b;
            if (ok) {
                out.tilesMask.swap(words);
                std::vector<uint8_t>().swap(out.tiles8);
            }
        }
    } else {
        out.denseItems.clear();
        for (uint32_t gi = 0; gi < G; ++gi)
            for (uint32_t b = groupFirstBlock[gi]; b < groupFirstBlock[gi + 1]; b += perItem)
                out.denseItems.push_back(DenseItem{gi, b, std::min(perItem, groupFirstBlock[gi + 1] - b), 0});
        out.rowBase = groupRowBase;
        out.winLen.clear();
        out.winMask.clear();
        out.tiles8.clear();
        const bool wide = opt.forceWideTiles || maxOffset >= 0xFFFFu;
        out.tiles16.assign(wide ? 0 : absTiles.size(), 0xFFFFu);
        out.tiles32.assign(wide ? absTiles.size() : 0, kNone);
        for (uint32_t gi = 0; gi < G; ++gi)
            for (uint32_t b = groupFirstBlock[gi]; b < groupFirstBlock[gi + 1]; ++b)
                for (uint32_t k = 0; k < H; ++k)
                    for (uint32_t e = 0; e < 256; ++e) {
                        const size_t at = ((size_t)b * H + k) * 256 + e;
                        const uint32_t v = absTiles[at];
                        if (v == kNone) continue;
                        const uint32_t off = v - groupRowBase[(size_t)gi * R + rowOfElement(k, e)];
                        if (wide) out.tiles32[at] = off;
                        else out.tiles16[at] = (uint16_t)off;
                    }
    }
    phase("destination encoding");
    if (opt.columnOrder) {
        // workgroup i of XCD x takes item x*(n/8)+i: an XCD then sees one column range of B.
        // The per-item arrays of the STAGED form are permuted along.
        std::vector<uint32_t> order(out.denseItems.size());
        for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
        std::vector<uint64_t> key(order.size());
        {
            // items are in group order here: the row chunk of an item = the eighth of the item list it falls in
            const uint64_t n = order.size();
            for (uint64_t i = 0; i < n; ++i) {
                const uint64_t col = out.blockCols[(size_t)out.denseItems[i].first * 16];
                uint64_t major = 0;
                if (opt.itemOrder == 1) {
                    // (whole groups: the chunk of the group's first item)
                    uint64_t f = i;
                    while (f > 0 && out.denseItems[f - 1].group == out.denseItems[i].group) --f;
                    major = f * 8 / n;
                } else if (opt.itemOrder == 2) {
                    major = (col / std::max(1u, opt.orderWindow)) << 24 | out.denseItems[i].group;
                }
                key[i] = major << 24 | col;
            }
        }
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        std::vector<DenseItem> items(order.size());
        for (size_t i = 0; i < order.size(); ++i) items[i] = out.denseItems[order[i]];
        out.denseItems.swap(items);
        if (out.staged) {
            std::vector<uint32_t> rb(out.rowBase.size()), wm(out.winMask.size());
            std::vector<uint16_t> wl(out.winLen.size());
            for (size_t i = 0; i < order.size(); ++i) {
                std::copy_n(out.rowBase.begin() + (size_t)order[i] * R, R, rb.begin() + i * R);
                std::copy_n(out.winLen.begin() + (size_t)order[i] * R, R, wl.begin() + i * R);
                std::copy_n(out.winMask.begin() + (size_t)order[i] * R * (kWindow / 32), R * (kWindow / 32),
                            wm.begin() + i * R * (kWindow / 32));
            }
            out.rowBase.swap(rb);
            out.winLen.swap(wl);
            out.winMask.swap(wm);
        }
    }

    phase("item order");
    const int st = packResidue(d, opt, out);
    phase("residue entries");
    return st;
}

// The sparse residue of the plan (out.panelRows and out.numDenseEntries are set): entries, work items, launch order.
inline int packResidue(const bsmr_rphm_desc* d, const PackOptions& opt, PackedPlan& out) {
    const uint32_t P = d->num_row_panels;
    const uint64_t numSparse = d->sparse_value_offsets[P];
    // ---- sparse residue -------------------------------------------------------
    // Entries are independent, so inside a panel they may be visited in any order:
    // with columnOrder they are sorted by column id and the work items by their first
    // column, which gives every XCD one column range of B (as for the dense blocks).
    out.entryCol.resize(numSparse);
    out.entryDst.resize(numSparse);
    out.entryRow.resize(numSparse);
    for (uint64_t i = 0; i < numSparse; ++i)
        if (d->sparse_col_indices[i] >= d->N || d->sparse_values[i] >= d->nnz ||
            d->sparse_relative_rows[i] >= 16)
            return BSMR_ERR_BAD_PLAN;
    parallelChunks(P, 256, [&](size_t q0, size_t q1, size_t) {
        std::vector<uint64_t> keys;  // (column, position in the panel): one plain sort = stable by column
        for (size_t p = q0; p < q1; ++p) {
            const uint32_t s0 = d->sparse_value_offsets[p], s1 = d->sparse_value_offsets[p + 1];
            keys.resize(s1 - s0);
            for (uint32_t i = 0; i < s1 - s0; ++i) keys[i] = ((uint64_t)d->sparse_col_indices[s0 + i] << 32) | i;
            if (opt.columnOrder) std::sort(keys.begin(), keys.end());
            for (uint32_t i = 0; i < s1 - s0; ++i) {
                const uint32_t from = s0 + (uint32_t)(keys[i] & 0xFFFFFFFFu);
                out.entryCol[s0 + i] = d->sparse_col_indices[from];
                out.entryDst[s0 + i] = d->sparse_values[from];
                out.entryRow[s0 + i] = (uint8_t)d->sparse_relative_rows[from];
            }
        }
    });
    out.numSparseEntries = numSparse;
    if (out.numDenseEntries + numSparse != d->nnz) return BSMR_ERR_BAD_PLAN;

    const uint32_t perWG = (uint32_t)std::max(32, opt.sparsePerItem);
    out.sparseItems.clear();
    for (uint32_t p = 0; p < P; ++p)
        for (uint32_t s = d->sparse_value_offsets[p]; s < d->sparse_value_offsets[p + 1]; s += perWG)
            out.sparseItems.push_back(
                SparseItem{p, s, std::min<uint32_t>(perWG, d->sparse_value_offsets[p + 1] - s), 0});
    // Free form (opt-in, BSMR_FREE_RESIDUE=1): no panels - entries in global (column, row) order, both
    // operands gathered per entry.  Measured against the panel form with items in median-column order it
    // loses everywhere (cop20k-like 54 vs 50 us, mycielskian15 88 vs 69, 4096^2 Bernoulli all-sparse
    // K=512 117 vs 81), so it is never chosen automatically; it stays as a cross-check of the panel form.
    if (numSparse && opt.freeResidue > 0) {
        std::vector<uint32_t> order(numSparse), rowId(numSparse);
        for (uint32_t q = 0; q < P; ++q)
            for (uint32_t i = d->sparse_value_offsets[q]; i < d->sparse_value_offsets[q + 1]; ++i)
                rowId[i] = out.panelRows[(size_t)q * 16 + out.entryRow[i]];
        for (uint64_t i = 0; i < numSparse; ++i) order[i] = (uint32_t)i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            return out.entryCol[a] != out.entryCol[b] ? out.entryCol[a] < out.entryCol[b] : rowId[a] < rowId[b];
        });
        const bool useFree = opt.freeResidue > 0;
        if (useFree) {
            std::vector<uint32_t> col(numSparse), dst(numSparse);
            out.entryRowId.resize(numSparse);
            for (uint64_t i = 0; i < numSparse; ++i) {
                col[i] = out.entryCol[order[i]];
                dst[i] = out.entryDst[order[i]];
                out.entryRowId[i] = rowId[order[i]];
            }
            out.entryCol.swap(col);
            out.entryDst.swap(dst);
            out.entryRow.clear();
            out.freeResidue = true;
            out.sparseItems.clear();
            for (uint64_t s0 = 0; s0 < numSparse; s0 += perWG)
                out.sparseItems.push_back(
                    SparseItem{0xFFFFFFFFu, (uint32_t)s0, (uint32_t)std::min<uint64_t>(perWG, numSparse - s0), 0});
            return BSMR_OK;
        }
    }
    // launch order = order of the items' median column: consecutive items, which share an XCD and run at
    // about the same time, then read neighbouring columns of B (banded / mesh matrices: the XCD's L2 works
    // as a sliding window).  The median, not the first column: a few far-away entries per panel would
    // otherwise decide the order (cop20k-like K=128: L2 hit rate 22 % with the first column as key).
    if (opt.columnOrder)
        std::stable_sort(out.sparseItems.begin(), out.sparseItems.end(), [&](const SparseItem& a, const SparseItem& b) {
            return out.entryCol[a.start + a.count / 2] < out.entryCol[b.start + b.count / 2];
        });
    return BSMR_OK;
}

}  // namespace bsmr
