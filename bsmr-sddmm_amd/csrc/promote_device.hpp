// The plan rule of csrc/plan_promote.hpp on the device (first half): which residue entries of the RPHM
// become extra 16x16 blocks of their panel.  Same decisions and the same arrays, byte for byte, as promoteSparseBlocks()
// - the host code stays the definition and the fallback (a panel whose residue is not in column order, a (row, column)
// listed more than 16 times, the `headMin` rule) - but the 1 KiB-per-block blockValues array is built where the device
// packer (csrc/pack_device.hpp) reads it and never crosses PCIe.
//
//   promoteKeys     every residue entry -> key (panel, column), value = its index; a stable radix sort puts a panel's
//                   entries in column order (the RPHM lists them in the order of its sparse columns)
//   promotePanels   one workgroup per panel: the residue's columns as runs, the runs ranked by descending count (ties:
//                   ascending column id - a counting sort, a count is at most 16), every entry's cell in the new blocks,
//                   repeated (row, column) entries left in the residue
//   (host)          which panels give their residue (the panel rule, the "everything" rule: a few integers per panel)
//   promoteScatter  one workgroup per panel: its own blocks copied, the new blocks filled, the kept residue compacted
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace bsmr {

constexpr uint32_t kPromoteNone = 0xFFFFFFFFu;

// exclusive prefix sum of `value` over the 256 threads of the workgroup; returns the total through `total`
__device__ __forceinline__ uint32_t promoteBlockScan(uint32_t value, uint32_t* waveSums /*[4]*/, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = value;
    for (uint32_t w = 1; w < 64u; w <<= 1) {
        const uint32_t other = __shfl_up(incl, w, 64);
        if (lane >= w) incl += other;
    }
    if (lane == 63u) waveSums[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += waveSums[w];
    total = waveSums[0] + waveSums[1] + waveSums[2] + waveSums[3];
    __syncthreads();  // (the sums are reused by the next call)
    return before + incl - value;
}

__global__ void __launch_bounds__(256)
promoteKeys(const uint32_t* __restrict__ sparseOffsets, const uint32_t* __restrict__ sparseCols, uint32_t numPanels, uint32_t numSparse,
            uint64_t* __restrict__ keys, uint32_t* __restrict__ order) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= numSparse) return;
    uint32_t lo = 0, hi = numPanels;   // the panel whose residue holds e: last q with sparseOffsets[q] <= e
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sparseOffsets[mid] <= e) lo = mid;
        else hi = mid;
    }
    keys[e] = ((uint64_t)lo << 32) | sparseCols[e];
    order[e] = e;
}

// flags[0] |= 1: an index out of range (the host path reports it)   |= 2: an input the device rule does not do
// sortedKeys / order: the residue in (panel, column) order, ties in the RPHM's order; cell[] is indexed like the RPHM
__global__ void __launch_bounds__(256)
promotePanels(const uint32_t* __restrict__ sparseOffsets, const uint64_t* __restrict__ sortedKeys, const uint32_t* __restrict__ order,
              const uint32_t* __restrict__ sparseRows, uint32_t N, uint32_t minAverage, uint32_t* __restrict__ runOf, uint32_t* __restrict__ runCol,
              uint32_t* __restrict__ runFirst, uint32_t* __restrict__ runRank, uint32_t* __restrict__ colByRank,
              uint32_t* __restrict__ cell, uint32_t* __restrict__ panelRuns, uint32_t* __restrict__ panelQualifies,
              uint32_t* __restrict__ panelMovable, uint32_t* __restrict__ flags) {
    __shared__ uint32_t waveSums[4], hist[18], start[18], running[18], waveCount[4][18], movable;
    const uint32_t q = blockIdx.x, s0 = sparseOffsets[q], n = sparseOffsets[q + 1] - s0;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x < 18) hist[threadIdx.x] = running[threadIdx.x] = 0;
    if (threadIdx.x == 0) movable = 0;
    __syncthreads();
    if (n == 0) {
        if (threadIdx.x == 0) panelRuns[q] = panelQualifies[q] = panelMovable[q] = 0;
        return;
    }
    const uint64_t* keys = sortedKeys + s0;
    const uint32_t* entry = order + s0;
    // 1. runs of equal columns
    uint32_t numRuns = 0;
    for (uint32_t base = 0; base < n; base += 256u) {  // uniform
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < n;
        const uint32_t c = valid ? (uint32_t)keys[i] : 0u, prev = valid && i > 0 ? (uint32_t)keys[i - 1] : 0u;
        if (valid && (c >= N || sparseRows[entry[i]] >= 16u)) atomicOr(flags, 1u);
        const uint32_t head = valid && (i == 0 || c != prev) ? 1u : 0u;
        uint32_t heads;
        const uint32_t before = promoteBlockScan(head, waveSums, heads);
        if (valid) {
            const uint32_t r = numRuns + before + head - 1u;
            runOf[s0 + i] = r;
            if (head) {
                runCol[s0 + r] = c;
                runFirst[s0 + r] = i;
            }
        }
        numRuns += heads;
    }
    __syncthreads();
    // 2. how many runs of every length
    for (uint32_t r = threadIdx.x; r < numRuns; r += 256u) {
        const uint32_t count = (r + 1 < numRuns ? runFirst[s0 + r + 1] : n) - runFirst[s0 + r];
        if (count > 16u) atomicOr(flags, 2u);
        atomicAdd(&hist[count > 16u ? 17u : count], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // descending count: the runs of length c start behind all longer ones
        uint32_t at = 0;
        for (int c = 17; c >= 0; --c) {
            start[c] = at;
            at += hist[c];
        }
    }
    __syncthreads();
    // 3. rank of every run: stable inside a length (ascending column id = ascending run number)
    for (uint32_t base = 0; base < numRuns; base += 256u) {  // uniform
        const uint32_t r = base + threadIdx.x;
        const bool valid = r < numRuns;
        uint32_t count = 0;
        if (valid) {
            count = (r + 1 < numRuns ? runFirst[s0 + r + 1] : n) - runFirst[s0 + r];
            if (count > 16u) count = 17u;
        }
        uint32_t within = 0;
        for (uint32_t b = 1; b <= 17u; ++b) {
            const unsigned long long mask = __ballot(valid && count == b);
            if (count == b) within = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            if (lane == 0) waveCount[wave][b] = (uint32_t)__popcll(mask);
        }
        __syncthreads();
        if (valid) {
            uint32_t rank = start[count] + running[count] + within;
            for (uint32_t w = 0; w < wave; ++w) rank += waveCount[w][count];
            runRank[s0 + r] = rank;
            colByRank[s0 + rank] = runCol[s0 + r];
        }
        __syncthreads();
        if (threadIdx.x >= 1 && threadIdx.x <= 17u)
            running[threadIdx.x] += waveCount[0][threadIdx.x] + waveCount[1][threadIdx.x] + waveCount[2][threadIdx.x] + waveCount[3][threadIdx.x];
        __syncthreads();
    }
    // 4. every entry's cell: 256 * (block among the panel's new blocks) + 16 * row + column slot; a repeated (row, column)
    //    keeps its extra copies in the residue
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 256u) {
        const uint32_t r = runOf[s0 + i], rank = runRank[s0 + r], first = runFirst[s0 + r], row = sparseRows[entry[i]];
        bool repeated = false;
        for (uint32_t j = first; j < i; ++j) repeated = repeated || sparseRows[entry[j]] == row;
        cell[entry[i]] = repeated ? kPromoteNone : (rank >> 4) * 256u + row * 16u + (rank & 15u);
        mine += repeated ? 0u : 1u;
    }
    atomicAdd(&movable, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t blocks = (numRuns + 15u) >> 4;
        panelRuns[q] = numRuns;
        panelQualifies[q] = (unsigned long long)n >= (unsigned long long)minAverage * blocks ? 1u : 0u;
        panelMovable[q] = movable;
    }
}

// the promoted RPHM arrays of one panel: its own blocks, then `take` new ones; the kept residue in its old order
__global__ void __launch_bounds__(256)
promoteScatter(const uint32_t* __restrict__ oldBlockOffsets, const uint32_t* __restrict__ oldCols, const uint32_t* __restrict__ oldValues,
               const uint32_t* __restrict__ sparseOffsets, const uint32_t* __restrict__ sparseValues, const uint32_t* __restrict__ sparseRows,
               const uint32_t* __restrict__ sparseCols, const uint32_t* __restrict__ cell, const uint32_t* __restrict__ colByRank,
               const uint32_t* __restrict__ panelRuns, const uint32_t* __restrict__ take, const uint32_t* __restrict__ newBlockOffsets,
               const uint32_t* __restrict__ newSparseOffsets, uint32_t N, uint32_t* __restrict__ newCols, uint32_t* __restrict__ newValues,
               uint32_t* __restrict__ keptValues, uint32_t* __restrict__ keptRows, uint32_t* __restrict__ keptCols) {
    __shared__ uint32_t waveSums[4];
    const uint32_t q = blockIdx.x;
    const uint32_t ob = oldBlockOffsets[q], own = oldBlockOffsets[q + 1] - ob, t = take[q];
    const size_t b0 = newBlockOffsets[q];
    for (uint32_t i = threadIdx.x; i < own * 16u; i += 256u) newCols[b0 * 16u + i] = oldCols[(size_t)ob * 16u + i];
    for (uint32_t i = threadIdx.x; i < own * 256u; i += 256u) newValues[b0 * 256u + i] = oldValues[(size_t)ob * 256u + i];
    const uint32_t s0 = sparseOffsets[q], n = sparseOffsets[q + 1] - s0, runs = panelRuns[q];
    for (uint32_t i = threadIdx.x; i < t * 256u; i += 256u) newValues[(b0 + own) * 256u + i] = kPromoteNone;
    for (uint32_t i = threadIdx.x; i < t * 16u; i += 256u) newCols[(b0 + own) * 16u + i] = i < runs ? colByRank[s0 + i] : N;
    __syncthreads();  // (the fill lies behind the entries' values)
    uint32_t at = newSparseOffsets[q];
    for (uint32_t base = 0; base < n; base += 256u) {  // uniform
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < n;
        const uint32_t c = valid ? cell[s0 + i] : kPromoteNone;
        const bool moved = valid && c != kPromoteNone && (c >> 8) < t;
        const uint32_t keeps = valid && !moved ? 1u : 0u;
        uint32_t kept;
        const uint32_t before = promoteBlockScan(keeps, waveSums, kept);
        if (moved) newValues[(b0 + own) * 256u + c] = sparseValues[s0 + i];
        if (keeps) {
            keptValues[at + before] = sparseValues[s0 + i];
            keptRows[at + before] = sparseRows[s0 + i];
            keptCols[at + before] = sparseCols[s0 + i];
        }
        at += kept;
    }
}

}  // namespace bsmr
