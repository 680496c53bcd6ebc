// Packing of the dense part of a plan on the device (second half): from the reference-layout RPHM
// arrays (dense_cols, block_offsets, block_values - what RPHM::RPHM builds, reference src/BSMR.cpp:83-265) to the
// device format of csrc/plan_pack.hpp, array for array and byte for byte what packPlan() produces on the host for
// the default layout: one panel per group, blocks in column-id order, 8-bit window offsets (or their mask form),
// work items in the order of their first column.  Anything else (grouped formats, a block that spans >= 255 entries
// of a row, a column listed twice in a panel) is left to the host packer.
//
//   packKeys        every slot of every panel's dense list -> key (panel, column), value = slot in the panel
//   (radix sort)    -> the panel's columns in column-id order; padding slots (column N) sort to the end
//   packBlocks      one wave per new block: its 16 columns, the destinations in accumulator order, per-row ranges
//   packItems       one wave per panel walks its blocks and cuts the work items (windows < 255 entries, <= perItem blocks)
//   (scan)          item numbers
//   packItemRows    window base / length per item row
//   packEncode      one wave per block: 8-bit offsets, ownership bitmaps, the mask form of the tile
//   (radix sort)    items by their first column; packPermute moves the per-item arrays along
// hipCUB (rocPRIM underneath) does the sorts and scans.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

#include "plan_pack.hpp"
#include "sddmm_kernels.hpp"

namespace bsmr {

constexpr uint32_t kPackNone = 0xFFFFFFFFu;

// flags[0] |= 1: index out of range (BSMR_ERR_BAD_PLAN)   flags[0] |= 2: layout the device packer does not do
__global__ void __launch_bounds__(256)
packKeys(const uint32_t* __restrict__ denseCols, const uint32_t* __restrict__ blockOffsets, uint32_t numPanels, uint64_t numSlots,
         uint32_t N, uint64_t* __restrict__ keys, uint32_t* __restrict__ slots, uint32_t* __restrict__ flags) {
    const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (j >= numSlots) return;
    const uint32_t block = (uint32_t)(j >> 4);
    uint32_t lo = 0, hi = numPanels;   // the panel whose block range holds `block`: last p with blockOffsets[p] <= block
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (blockOffsets[mid] <= block) lo = mid;
        else hi = mid;
    }
    const uint32_t col = denseCols[j];
    if (col > N) atomicOr(flags, 1u);
    keys[j] = ((uint64_t)lo << 32) | col;
    slots[j] = (uint32_t)(j - (uint64_t)blockOffsets[lo] * 16u);
}

// columns of every panel (the sorted list up to the first padding slot) and the blocks they fill
__global__ void __launch_bounds__(256)
packPanelCols(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ blockOffsets, uint32_t numPanels, uint32_t N,
              uint32_t* __restrict__ panelCols, uint32_t* __restrict__ panelBlocks) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p > numPanels) return;
    if (p == numPanels) {
        panelBlocks[p] = 0;
        return;
    }
    const uint64_t first = (uint64_t)blockOffsets[p] * 16u, want = ((uint64_t)p << 32) | N;
    uint64_t lo = first, hi = (uint64_t)blockOffsets[p + 1] * 16u;
    while (lo < hi) {   // first slot whose key is not below (p, N)
        const uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < want) lo = mid + 1;
        else hi = mid;
    }
    panelCols[p] = (uint32_t)(lo - first);
    panelBlocks[p] = (uint32_t)((lo - first + 15u) / 16u);
}

// a column listed twice in one panel: the host packer merges the two slots; leave such input to it
__global__ void __launch_bounds__(256)
packCheckSorted(const uint64_t* __restrict__ keys, uint64_t n, uint32_t N, uint32_t* __restrict__ flags) {
    const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (j == 0 || j >= n) return;
    if (keys[j] == keys[j - 1] && (uint32_t)keys[j] < N) atomicOr(flags, 2u);
}

// One wave per new block b of panel p.  Lane l = 16 * (r >> 2) + c holds rows 4 (l >> 4) .. + 3 of column c: the
// accumulator layout of v_mfma_f32_16x16x32, so the four destinations of a lane are contiguous in absTiles.
__global__ void __launch_bounds__(256)
packBlocks(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ slots, const uint32_t* __restrict__ blockOffsets,
           const uint32_t* __restrict__ blockValues, const uint32_t* __restrict__ panelCols,
           const uint32_t* __restrict__ firstBlock, const uint32_t* __restrict__ panelOfBlock, uint32_t numBlocks, uint32_t nnz,
           uint32_t* __restrict__ blockCols, uint32_t* __restrict__ absTiles, uint32_t* __restrict__ rowLo,
           uint32_t* __restrict__ rowHi, uint8_t* __restrict__ blockMask, unsigned long long* __restrict__ counters,
           uint32_t* __restrict__ flags, uint32_t* __restrict__ blockCount) {
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (b >= numBlocks) return;
    const uint32_t lane = threadIdx.x & 63u, cc = lane & 15u, rg = lane >> 4;
    const uint32_t p = panelOfBlock[b];
    const uint32_t u = (b - firstBlock[p]) * 16u + cc;          // position in the panel's sorted column list
    const bool valid = u < panelCols[p];
    const uint64_t j = (uint64_t)blockOffsets[p] * 16u + u;
    uint32_t v[4] = {kPackNone, kPackNone, kPackNone, kPackNone};
    if (valid) {
        const uint32_t t = slots[j];
        const uint32_t* tile = blockValues + ((uint64_t)blockOffsets[p] + t / 16u) * 256u + t % 16u;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) v[i] = tile[(4u * rg + i) * 16u];
    }
    if (rg == 0) blockCols[(uint64_t)b * 16u + cc] = valid ? (uint32_t)keys[j] : 0u;
    uint32_t count = 0;
    bool bad = false;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        if (v[i] != kPackNone && v[i] >= nnz) {
            bad = true;
            v[i] = kPackNone;
        }
        count += v[i] != kPackNone;
    }
    if (bad) atomicOr(flags, 1u);
    *reinterpret_cast<uint4*>(absTiles + (uint64_t)b * 256u + lane * 4u) = make_uint4(v[0], v[1], v[2], v[3]);
    // per row of the block: smallest and largest destination (over the 16 lanes of the row group)
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        uint32_t lo = v[i], hi = v[i] == kPackNone ? 0u : v[i];
#pragma unroll
        for (uint32_t m = 1; m < 16; m <<= 1) {
            lo = min(lo, (uint32_t)__shfl_xor((int)lo, (int)m));
            hi = max(hi, (uint32_t)__shfl_xor((int)hi, (int)m));
        }
        if (cc == 0) {
            rowLo[(uint64_t)b * 16u + 4u * rg + i] = lo;
            rowHi[(uint64_t)b * 16u + 4u * rg + i] = hi;
        }
    }
#pragma unroll
    for (uint32_t m = 1; m < 64; m <<= 1) count += (uint32_t)__shfl_xor((int)count, (int)m);
    if (lane == 0) {
        blockMask[b] = count ? 1 : 0;
        if (blockCount) blockCount[b] = count;   // (for collectEmit: plans that keep the entry lists)
        if (count) {
            atomicAdd(&counters[0], (unsigned long long)count);   // dense entries
            atomicAdd(&counters[1], 1ull);                        // non-empty tiles
        }
    }
}

// The dense entries as per-panel lists ordered by (column, row in panel) - HostDense of csrc/tile_format.hpp, what the
// formats of the other dense engines are packed from - written from the new blocks, which already lie in column-id order:
// one wave per block, entry (column cc, row r) of block b goes to blockStart[b] + (entries of the block in front of it in
// (cc, r) order).  blockStart = exclusive scan of packBlocks' counts.
__global__ void __launch_bounds__(256)
collectEmit(const uint32_t* __restrict__ absTiles, const uint32_t* __restrict__ blockCols, const uint32_t* __restrict__ blockStart,
            uint32_t numBlocks, uint32_t* __restrict__ outCol, uint8_t* __restrict__ outRow, uint32_t* __restrict__ outIdx) {
    __shared__ uint32_t counts[4][64];
    const uint32_t w = threadIdx.x >> 6, b = blockIdx.x * 4u + w;
    if (b >= numBlocks) return;   // (whole waves leave: nothing below synchronises a workgroup)
    const uint32_t lane = threadIdx.x & 63u, cc = lane & 15u, rg = lane >> 4;
    const uint4 v4 = *reinterpret_cast<const uint4*>(absTiles + (uint64_t)b * 256u + lane * 4u);
    const uint32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) mine += v[i] != kPackNone;
    // lanes in list order: column major, then the row group
    const uint32_t order = cc * 4u + rg;
    counts[w][order] = mine;
    __builtin_amdgcn_wave_barrier();
    uint32_t scan = counts[w][lane];
#pragma unroll
    for (uint32_t m = 1; m < 64; m <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)scan, (int)m);
        if (lane >= m) scan += up;
    }
    __builtin_amdgcn_wave_barrier();
    counts[w][lane] = scan;   // inclusive, by list order
    __builtin_amdgcn_wave_barrier();
    uint32_t at = blockStart[b] + counts[w][order] - mine;
    const uint32_t col = blockCols[(uint64_t)b * 16u + cc];
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i)
        if (v[i] != kPackNone) {
            outCol[at] = col;
            outRow[at] = (uint8_t)(4u * rg + i);
            outIdx[at] = v[i];
            ++at;
        }
}

// panelOfBlock for the new blocks
__global__ void __launch_bounds__(256)
packPanelOfBlock(const uint32_t* __restrict__ firstBlock, uint32_t numPanels, uint32_t* __restrict__ panelOfBlock) {
    const uint32_t p = blockIdx.x;
    for (uint32_t b = firstBlock[p] + threadIdx.x; b < firstBlock[p + 1]; b += 256u) panelOfBlock[b] = p;
}

// One wave per panel (lanes 0-15 = the panel's rows) walks the blocks in order and cuts the items exactly as the
// host loop does: a block joins the open item while the item has fewer than perItem blocks and every row's window
// (smallest to largest destination) stays below kWindowMax entries.
__global__ void __launch_bounds__(64)
packItems(const uint32_t* __restrict__ firstBlock, const uint32_t* __restrict__ rowLo, const uint32_t* __restrict__ rowHi,
          uint32_t perItem, uint32_t* __restrict__ isFirst, uint32_t* __restrict__ countAtFirst,
          uint32_t* __restrict__ loAtFirst, uint32_t* __restrict__ hiAtFirst, uint32_t* __restrict__ flags,
          uint32_t* __restrict__ maxItemBlocks) {
    const uint32_t p = blockIdx.x, r = threadIdx.x & 15u;
    const bool active = threadIdx.x < 16u;
    uint32_t lo = kPackNone, hi = 0, count = 0, itemFirst = firstBlock[p], longest = 0;
    auto closeItem = [&]() {
        if (!count) return;
        if (active) {
            loAtFirst[(uint64_t)itemFirst * 16u + r] = lo;
            hiAtFirst[(uint64_t)itemFirst * 16u + r] = hi;
        }
        if (threadIdx.x == 0) {
            isFirst[itemFirst] = 1u;
            countAtFirst[itemFirst] = count;
        }
        longest = max(longest, count);
        lo = kPackNone;
        hi = 0;
        count = 0;
    };
    for (uint32_t b = firstBlock[p]; b < firstBlock[p + 1]; ++b) {
        const uint32_t blo = active ? rowLo[(uint64_t)b * 16u + r] : kPackNone;
        const uint32_t bhi = active ? rowHi[(uint64_t)b * 16u + r] : 0u;
        const bool has = blo != kPackNone;
        const bool wide = has && bhi - blo >= kWindowMax;
        const bool rowFits = !(has && lo != kPackNone && max(hi, bhi) - min(lo, blo) >= kWindowMax);
        if (__any(wide)) {
            if (threadIdx.x == 0) atomicOr(flags, 2u);   // one block alone is too wide: the direct encoding, on the host
            return;
        }
        const bool fits = count < perItem && __all(rowFits);
        if (!fits) closeItem();
        if (count == 0) itemFirst = b;
        if (has) {
            lo = min(lo, blo);
            hi = max(hi, bhi);
        }
        ++count;
    }
    closeItem();
    if (threadIdx.x == 0 && longest) atomicMax(maxItemBlocks, longest);
}

// item id of a first block = exclusive scan of isFirst; the item record and its rows' windows
__global__ void __launch_bounds__(256)
packItemRows(const uint32_t* __restrict__ isFirst, const uint32_t* __restrict__ itemBefore, const uint32_t* __restrict__ countAtFirst,
             const uint32_t* __restrict__ loAtFirst, const uint32_t* __restrict__ hiAtFirst,
             const uint32_t* __restrict__ panelOfBlock, const uint32_t* __restrict__ blockCols, uint32_t numBlocks,
             DenseItem* __restrict__ items, uint32_t* __restrict__ itemLo, uint32_t* __restrict__ rowBase,
             uint16_t* __restrict__ winLen, uint32_t* __restrict__ firstCol, uint32_t* __restrict__ itemIndex) {
    const uint32_t b = blockIdx.x * 16u + (threadIdx.x >> 4), r = threadIdx.x & 15u;
    if (b >= numBlocks || !isFirst[b]) return;
    const uint32_t id = itemBefore[b];
    const uint32_t lo = loAtFirst[(uint64_t)b * 16u + r], hi = hiAtFirst[(uint64_t)b * 16u + r];
    itemLo[(uint64_t)id * 16u + r] = lo;
    rowBase[(uint64_t)id * 16u + r] = lo == kPackNone ? 0u : lo;
    winLen[(uint64_t)id * 16u + r] = lo == kPackNone ? (uint16_t)0 : (uint16_t)(hi - lo + 1u);
    if (r == 0) {
        items[id] = DenseItem{panelOfBlock[b], b, countAtFirst[b], 0u};
        firstCol[id] = blockCols[(uint64_t)b * 16u];
        itemIndex[id] = id;
    }
}

// One wave per block: window offsets of its destinations (8 bits, 0xFF = none), the ownership bitmap of the item's
// windows, and the mask form of the tile (TileMask: per 4-row lane group two words of column masks and one of first
// offsets).  flags |= 4 when a tile row's offsets are not consecutive (then the mask form is not used).
__global__ void __launch_bounds__(256)
packEncode(const uint32_t* __restrict__ absTiles, const uint32_t* __restrict__ itemBefore, const uint32_t* __restrict__ isFirst,
           const uint32_t* __restrict__ itemLo, uint32_t numBlocks, uint8_t* __restrict__ tiles8, uint32_t* __restrict__ winMask, uint32_t* __restrict__ maskWords,
           uint32_t* __restrict__ flags) {
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (b >= numBlocks) return;
    const uint32_t lane = threadIdx.x & 63u, cc = lane & 15u, rg = lane >> 4;
    const uint32_t id = itemBefore[b] + isFirst[b] - 1u;   // items that start at or before this block, minus one
    const uint4 v4 = *reinterpret_cast<const uint4*>(absTiles + (uint64_t)b * 256u + lane * 4u);
    const uint32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
    uint32_t packed = 0, word01[2] = {0, 0}, firsts = 0;
    bool broken = false;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        const uint32_t row = 4u * rg + i;
        const bool has = v[i] != kPackNone;
        const uint32_t off = has ? v[i] - itemLo[(uint64_t)id * 16u + row] : 0xFFu;
        packed |= off << (8u * i);
        if (has) atomicOr(&winMask[((uint64_t)id * 16u + row) * (kWindow / 32u) + off / 32u], 1u << (off % 32u));
        // the row's 16 lanes: which columns have an entry, and the first one's offset
        const uint64_t ballot = __ballot(has);
        const uint32_t mask = (uint32_t)(ballot >> (16u * rg)) & 0xFFFFu;
        uint32_t first = has ? off : 0xFFFFFFFFu;
#pragma unroll
        for (uint32_t m = 1; m < 16; m <<= 1) first = min(first, (uint32_t)__shfl_xor((int)first, (int)m));
        if (mask == 0) first = 0;
        if (has && off != first + __popc(mask & ((1u << cc) - 1u))) broken = true;
        word01[i >> 1] |= mask << (16u * (i & 1u));
        firsts |= (first & 0xFFu) << (8u * i);
    }
    *reinterpret_cast<uint32_t*>(tiles8 + (uint64_t)b * 256u + lane * 4u) = packed;
    if (cc == 0) {
        uint32_t* group = maskWords + (uint64_t)b * 12u + rg * 3u;
        group[0] = word01[0];
        group[1] = word01[1];
        group[2] = firsts;
    }
    if (__any(broken) && lane == 0) atomicOr(flags, 4u);
}

// per-item arrays in launch order
__global__ void __launch_bounds__(256)
packPermute(const uint32_t* __restrict__ order, uint32_t numItems, const DenseItem* __restrict__ itemsIn,
            const uint32_t* __restrict__ rowBaseIn, const uint16_t* __restrict__ winLenIn, const uint32_t* __restrict__ winMaskIn,
            DenseItem* __restrict__ items, uint32_t* __restrict__ rowBase, uint16_t* __restrict__ winLen,
            uint32_t* __restrict__ winMask) {
    const uint32_t i = blockIdx.x * 2u + (threadIdx.x >> 7), t = threadIdx.x & 127u;   // 128 mask words per item
    if (i >= numItems) return;
    const uint32_t from = order[i];
    winMask[(uint64_t)i * 128u + t] = winMaskIn[(uint64_t)from * 128u + t];
    if (t < 16u) {
        rowBase[(uint64_t)i * 16u + t] = rowBaseIn[(uint64_t)from * 16u + t];
        winLen[(uint64_t)i * 16u + t] = winLenIn[(uint64_t)from * 16u + t];
    }
    if (t == 0) items[i] = itemsIn[from];
}

}  // namespace bsmr
