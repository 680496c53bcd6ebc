// Column reordering, dense / sparse split and the RPHM index arrays on the device.
//
// What it produces is what the host pipeline produces, array for array and byte for byte: colReordering_cpu
// (reference src/colReordering.cu:274-404; the reference's own GPU attempt, :146-241, is unfinished) and RPHM::RPHM
// (reference src/BSMR.cpp:83-265) - denseCols / denseColOffsets / sparseCols / sparseColOffsets / sparseValueOffsets,
// blockOffsets / blockValues, sparseValues / sparseRelativeRows / sparseColIndices.
//
// How (no per-panel loops, no O(N) arrays): every stored entry becomes a 64-bit key (panel, column, row in panel);
//   1. one radix sort of the keys                         -> the panel's entries in (column, row) order
//   2. heads of equal (panel, column) runs + a scan       -> the columns of every panel with their counts
//   3. one radix sort of the runs by (panel, 16 - count, column) -> count descending, ties by ascending column id
//   4. scans over the sorted runs + one thread per panel  -> blocks of 16 columns, the dense prefix for delta
//   5. scatter kernels                                     -> the arrays above
// hipCUB (rocPRIM underneath) does the sorts and scans.  Included at the end of bsmr_capi.hip.
#pragma once

#include <hipcub/hipcub.hpp>

namespace bsmr {

// entry e of reordered row q (rows listed in reordered order): key = panel << 36 | column << 4 | row in panel
__global__ void __launch_bounds__(256)
colreorderKeys(const uint32_t* __restrict__ rowOffsets, const uint32_t* __restrict__ colIndices,
               const uint32_t* __restrict__ reorderedRows, const uint32_t* __restrict__ outStart, uint32_t numRows,
               uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6);   // one wave per row
    if (q >= numRows) return;
    const uint32_t row = reorderedRows[q], b = rowOffsets[row], e = rowOffsets[row + 1];
    const uint64_t hi = ((uint64_t)(q >> 4) << 36) | (q & 15u);
    for (uint32_t i = b + (threadIdx.x & 63u); i < e; i += 64u) {
        keys[outStart[q] + (i - b)] = hi | ((uint64_t)colIndices[i] << 4);
        vals[outStart[q] + (i - b)] = i;
    }
}

// head[i] = 1 where a new (panel, column) run starts
__global__ void __launch_bounds__(256)
colreorderHeads(const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ head) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    head[i] = i == 0 || (keys[i] >> 4) != (keys[i - 1] >> 4) ? 1u : 0u;
}

// run r (its first entry at position i): start, and the key of the second sort
__global__ void __launch_bounds__(256)
colreorderRuns(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ head, const uint32_t* __restrict__ runId,
               uint32_t n, uint32_t* __restrict__ runStart) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || !head[i]) return;
    runStart[runId[i] - 1u] = i;   // inclusive scan of heads: run ids from 1
}
__global__ void __launch_bounds__(256)
colreorderRunKeys(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ runStart, uint32_t numRuns, uint32_t n,
                  uint64_t* __restrict__ runKeys, uint32_t* __restrict__ runIndex) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= numRuns) return;
    const uint32_t start = runStart[r], end = r + 1 < numRuns ? runStart[r + 1] : n;
    const uint32_t cnt = min(end - start, 0xFFFu);   // (a cell stored twice counts twice, as on the host)
    const uint64_t k = keys[start];
    const uint64_t panel = k >> 36, col = (k >> 4) & 0xFFFFFFFFull;
    runKeys[r] = (panel << 44) | ((uint64_t)(0xFFFu - cnt) << 32) | col;   // count descending, column ascending
    runIndex[r] = r;
}

// per sorted run: panel id and count (for the scans)
__global__ void __launch_bounds__(256)
colreorderSortedRuns(const uint64_t* __restrict__ runKeys, uint32_t numRuns, uint32_t* __restrict__ runCount,
                     uint32_t* __restrict__ panelRuns) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= numRuns) return;
    const uint64_t k = runKeys[r];
    runCount[r] = 0xFFFu - (uint32_t)((k >> 32) & 0xFFFu);
    atomicAdd(&panelRuns[(uint32_t)(k >> 44)], 1u);
}

// one thread per panel: dense prefix of its blocks, sizes of its parts
__global__ void __launch_bounds__(256)
colreorderSplit(const uint32_t* __restrict__ panelRunStart /* [P+1] */, const uint32_t* __restrict__ countPrefix /* [runs+1] */,
                uint32_t numPanels, uint32_t threshold, uint32_t* __restrict__ numDense, uint32_t* __restrict__ numSparse,
                uint32_t* __restrict__ sparseEntries, uint32_t* __restrict__ numBlocks) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= numPanels) return;
    const uint32_t r0 = panelRunStart[p], r1 = panelRunStart[p + 1];
    const uint32_t blocks = (r1 - r0 + 15u) / 16u;
    // block sums are non-increasing (counts are): the dense blocks are a prefix - binary search for its end
    uint32_t lo = 0, hi = blocks;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2u;
        const uint32_t a = r0 + 16u * mid, b = min(a + 16u, r1);
        if (countPrefix[b] - countPrefix[a] >= threshold) lo = mid + 1u;
        else hi = mid;
    }
    const uint32_t denseRuns = min(16u * lo, r1 - r0);
    numBlocks[p] = lo;
    numDense[p] = 16u * lo;
    numSparse[p] = 16u * blocks - 16u * lo;
    sparseEntries[p] = countPrefix[r1] - countPrefix[r0 + denseRuns];
}

// one thread per sorted run: its column into denseCols / sparseCols, its entries into blockValues / the sparse triples
__global__ void __launch_bounds__(256)
colreorderScatter(const uint64_t* __restrict__ runKeys, const uint32_t* __restrict__ runIndex,
                  const uint32_t* __restrict__ runStart, uint32_t numRuns, uint32_t n,
                  const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                  const uint32_t* __restrict__ panelRunStart, const uint32_t* __restrict__ countPrefix,
                  const uint32_t* __restrict__ numDense, const uint32_t* __restrict__ denseColOffsets,
                  const uint32_t* __restrict__ sparseColOffsets, const uint32_t* __restrict__ sparseValueOffsets,
                  const uint32_t* __restrict__ blockOffsets, uint32_t* __restrict__ denseCols,
                  uint32_t* __restrict__ sparseCols, uint32_t* __restrict__ blockValues,
                  uint32_t* __restrict__ sparseValues, uint32_t* __restrict__ sparseRows, uint32_t* __restrict__ sparseColIdx) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= numRuns) return;
    const uint64_t k = runKeys[r];
    const uint32_t p = (uint32_t)(k >> 44), col = (uint32_t)(k & 0xFFFFFFFFull);
    const uint32_t t = r - panelRunStart[p];   // position in the panel's ordered column list
    const uint32_t src = runIndex[r];
    const uint32_t start = runStart[src], end = src + 1 < numRuns ? runStart[src + 1] : n;
    if (t < numDense[p]) {
        denseCols[denseColOffsets[p] + t] = col;
        uint32_t* tile = blockValues + ((size_t)blockOffsets[p] + t / 16u) * 256u + (t % 16u);
        for (uint32_t i = start; i < end; ++i) tile[(uint32_t)(keys[i] & 15u) * 16u] = vals[i];   // same cell twice: the later wins, as on the host
    } else {
        sparseCols[sparseColOffsets[p] + (t - numDense[p])] = col;
        const uint32_t at = sparseValueOffsets[p] + (countPrefix[r] - countPrefix[panelRunStart[p] + numDense[p]]);
        for (uint32_t i = start; i < end; ++i) {
            sparseValues[at + (i - start)] = vals[i];
            sparseRows[at + (i - start)] = (uint32_t)(keys[i] & 15u);
            sparseColIdx[at + (i - start)] = col;
        }
    }
}

__global__ void __launch_bounds__(256) fillU32(uint32_t* __restrict__ p, size_t n, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) p[i] = v;
}

}  // namespace bsmr

struct bsmr_colreorder {
    int device = 0;
    uint32_t numPanels = 0;
    uint64_t numDenseCols = 0, numSparseCols = 0, numBlocks = 0, numSparseEntries = 0;
    float elapsedMs = 0.0f;
    // device results, fetched by bsmr_col_reorder_fetch
    uint32_t *denseCols = nullptr, *sparseCols = nullptr, *blockValues = nullptr, *sparseValues = nullptr,
             *sparseRows = nullptr, *sparseColIdx = nullptr;
    std::vector<uint32_t> denseColOffsets, sparseColOffsets, sparseValueOffsets, blockOffsets;   // host (small)
    // (the result owns its device arrays: an early return of bsmr_col_reorder drops a half-built one through unique_ptr)
    ~bsmr_colreorder() {
        if (hipSetDevice(device) != hipSuccess) (void)hipGetLastError();
        void* ptrs[] = {denseCols, sparseCols, blockValues, sparseValues, sparseRows, sparseColIdx};
        for (void* q : ptrs)
            if (q) (void)hipFree(q);
    }
};

extern "C" {

int bsmr_col_reorder_free(bsmr_colreorder* h) {
    delete h;
    return BSMR_OK;
}

int bsmr_col_reorder(bsmr_colreorder** out, int device, uint32_t rows, uint32_t cols, const uint32_t* row_offsets,
                     const uint32_t* col_indices, const uint32_t* reordered_rows, uint32_t num_reordered, float delta) {
    if (!out || !row_offsets || (!reordered_rows && num_reordered)) return BSMR_ERR_INVALID_ARG;
    *out = nullptr;
    const uint32_t nnz = rows ? row_offsets[rows] : 0;
    if (nnz && !col_indices) return BSMR_ERR_INVALID_ARG;
    if (cols >= (1u << 31) || num_reordered / 16u >= (1u << 20)) return BSMR_ERR_INVALID_ARG;   // key fields: 20-bit panel, 32-bit column
    int st = useDevice(device);
    if (st != BSMR_OK) return st;
    try {
        const uint32_t P = (num_reordered + 15u) / 16u;
        // entries of the reordered rows (rows not listed - the empty ones - contribute nothing)
        std::vector<uint32_t> outStart((size_t)num_reordered + 1, 0);
        for (uint32_t q = 0; q < num_reordered; ++q) {
            if (reordered_rows[q] >= rows) return BSMR_ERR_BAD_PLAN;
            outStart[q + 1] = outStart[q] + (row_offsets[reordered_rows[q] + 1] - row_offsets[reordered_rows[q]]);
        }
        const uint32_t n = outStart[num_reordered];
        if (n > 0x7FFFFFFFu) return BSMR_ERR_INVALID_ARG;   // (hipCUB counts in int: the caller keeps the host implementation)
        std::unique_ptr<bsmr_colreorder> h(new bsmr_colreorder);
        h->device = device;
        h->numPanels = P;
        h->denseColOffsets.assign((size_t)P + 1, 0);
        h->sparseColOffsets.assign((size_t)P + 1, 0);
        h->sparseValueOffsets.assign((size_t)P + 1, 0);
        h->blockOffsets.assign((size_t)P + 1, 0);
        if (n == 0 || P == 0) {
            *out = h.release();
            return BSMR_OK;
        }
        struct EventGuard {
            hipEvent_t a = nullptr, b = nullptr;
            ~EventGuard() {
                if (a) (void)hipEventDestroy(a);
                if (b) (void)hipEventDestroy(b);
            }
        } guard;
        BSMR_HIP(hipEventCreate(&guard.a));
        BSMR_HIP(hipEventCreate(&guard.b));
        const hipEvent_t ev0 = guard.a, ev1 = guard.b;
        hipStream_t s = nullptr;
        BSMR_HIP(hipEventRecord(ev0, s));
        DeviceBuffers dev;   // scratch, freed on return
        uint32_t *dRowOffsets, *dCols, *dRows, *dStart, *dVals, *dValsAlt, *dHead, *dRunId, *dRunStart, *dRunIndex, *dRunIndexAlt,
            *dRunCount, *dCountPrefix, *dPanelRuns, *dPanelRunStart, *dNumDense, *dNumSparse, *dSparseEntries, *dNumBlocks, *dOffsets;
        uint64_t *dKeys, *dKeysAlt, *dRunKeys, *dRunKeysAlt;
        if (!dev.alloc(&dRowOffsets, (size_t)rows + 1, "hipMalloc") || !dev.alloc(&dCols, nnz, "hipMalloc") ||
            !dev.alloc(&dRows, num_reordered, "hipMalloc") || !dev.alloc(&dStart, (size_t)num_reordered + 1, "hipMalloc") ||
            !dev.alloc(&dKeys, n, "hipMalloc") || !dev.alloc(&dKeysAlt, n, "hipMalloc") || !dev.alloc(&dVals, n, "hipMalloc") ||
            !dev.alloc(&dValsAlt, n, "hipMalloc") || !dev.alloc(&dHead, n, "hipMalloc") || !dev.alloc(&dRunId, n, "hipMalloc"))
            return BSMR_ERR_OOM;
        BSMR_HIP(hipMemcpyAsync(dRowOffsets, row_offsets, ((size_t)rows + 1) * 4, hipMemcpyHostToDevice, s));
        BSMR_HIP(hipMemcpyAsync(dCols, col_indices, (size_t)nnz * 4, hipMemcpyHostToDevice, s));
        BSMR_HIP(hipMemcpyAsync(dRows, reordered_rows, (size_t)num_reordered * 4, hipMemcpyHostToDevice, s));
        BSMR_HIP(hipMemcpyAsync(dStart, outStart.data(), ((size_t)num_reordered + 1) * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(bsmr::colreorderKeys, dim3((num_reordered + 3) / 4), dim3(256), 0, s, dRowOffsets, dCols, dRows, dStart,
                           num_reordered, dKeys, dVals);
        BSMR_HIP(hipGetLastError());
        // 1. sort the entries by (panel, column, row)
        size_t tempBytes = 0, need = 0;
        hipcub::DoubleBuffer<uint64_t> kb(dKeys, dKeysAlt);
        hipcub::DoubleBuffer<uint32_t> vb(dVals, dValsAlt);
        BSMR_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, kb, vb, (int)n, 0, 56, s));
        tempBytes = need;
        BSMR_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, need, dHead, dRunId, (int)n, s));
        tempBytes = std::max(tempBytes, need);
        void* dTemp = nullptr;
        {
            uint8_t* t8 = nullptr;
            if (!dev.alloc(&t8, tempBytes + 256, "hipMalloc(sort scratch)")) return BSMR_ERR_OOM;
            dTemp = t8;
        }
        need = tempBytes;
        BSMR_HIP(hipcub::DeviceRadixSort::SortPairs(dTemp, need, kb, vb, (int)n, 0, 56, s));
        const uint64_t* sKeys = kb.Current();
        const uint32_t* sVals = vb.Current();
        // 2. runs of equal (panel, column)
        hipLaunchKernelGGL(bsmr::colreorderHeads, dim3((n + 255) / 256), dim3(256), 0, s, sKeys, n, dHead);
        need = tempBytes;
        BSMR_HIP(hipcub::DeviceScan::InclusiveSum(dTemp, need, dHead, dRunId, (int)n, s));
        uint32_t numRuns = 0;
        BSMR_HIP(hipMemcpyAsync(&numRuns, dRunId + (n - 1), 4, hipMemcpyDeviceToHost, s));
        BSMR_HIP(hipStreamSynchronize(s));
        if (!dev.alloc(&dRunStart, (size_t)numRuns + 1, "hipMalloc") || !dev.alloc(&dRunKeys, numRuns, "hipMalloc") ||
            !dev.alloc(&dRunKeysAlt, numRuns, "hipMalloc") || !dev.alloc(&dRunIndex, numRuns, "hipMalloc") ||
            !dev.alloc(&dRunIndexAlt, numRuns, "hipMalloc") || !dev.alloc(&dRunCount, (size_t)numRuns + 1, "hipMalloc") ||
            !dev.alloc(&dCountPrefix, (size_t)numRuns + 1, "hipMalloc") || !dev.alloc(&dPanelRuns, (size_t)P + 1, "hipMalloc") ||
            !dev.alloc(&dPanelRunStart, (size_t)P + 1, "hipMalloc") || !dev.alloc(&dNumDense, (size_t)P + 1, "hipMalloc") ||
            !dev.alloc(&dNumSparse, (size_t)P + 1, "hipMalloc") || !dev.alloc(&dSparseEntries, (size_t)P + 1, "hipMalloc") ||
            !dev.alloc(&dNumBlocks, (size_t)P + 1, "hipMalloc") || !dev.alloc(&dOffsets, 4 * ((size_t)P + 1), "hipMalloc"))
            return BSMR_ERR_OOM;
        hipLaunchKernelGGL(bsmr::colreorderRuns, dim3((n + 255) / 256), dim3(256), 0, s, sKeys, dHead, dRunId, n, dRunStart);
        hipLaunchKernelGGL(bsmr::colreorderRunKeys, dim3((numRuns + 255) / 256), dim3(256), 0, s, sKeys, dRunStart, numRuns, n,
                           dRunKeys, dRunIndex);
        // 3. columns of a panel by count descending, ties by ascending column id
        hipcub::DoubleBuffer<uint64_t> rkb(dRunKeys, dRunKeysAlt);
        hipcub::DoubleBuffer<uint32_t> rvb(dRunIndex, dRunIndexAlt);
        need = 0;
        BSMR_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, rkb, rvb, (int)numRuns, 0, 64, s));
        void* dTemp2 = dTemp;
        if (need > tempBytes) {
            uint8_t* t8 = nullptr;
            if (!dev.alloc(&t8, need + 256, "hipMalloc(sort scratch)")) return BSMR_ERR_OOM;
            dTemp2 = t8;
        } else {
            need = tempBytes;
        }
        BSMR_HIP(hipcub::DeviceRadixSort::SortPairs(dTemp2, need, rkb, rvb, (int)numRuns, 0, 64, s));
        const uint64_t* sRunKeys = rkb.Current();
        const uint32_t* sRunIndex = rvb.Current();
        // 4. per panel: its runs, the prefix of counts, the dense prefix of its blocks
        BSMR_HIP(hipMemsetAsync(dPanelRuns, 0, ((size_t)P + 1) * 4, s));
        BSMR_HIP(hipMemsetAsync(dRunCount + numRuns, 0, 4, s));
        hipLaunchKernelGGL(bsmr::colreorderSortedRuns, dim3((numRuns + 255) / 256), dim3(256), 0, s, sRunKeys, numRuns, dRunCount,
                           dPanelRuns);
        size_t scanNeed = 0;
        BSMR_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scanNeed, dRunCount, dCountPrefix, (int)numRuns + 1, s));
        void* dTemp3 = dTemp;
        if (scanNeed > tempBytes) {
            uint8_t* t8 = nullptr;
            if (!dev.alloc(&t8, scanNeed + 256, "hipMalloc(scan scratch)")) return BSMR_ERR_OOM;
            dTemp3 = t8;
        }
        size_t sn = std::max(scanNeed, tempBytes);
        BSMR_HIP(hipcub::DeviceScan::ExclusiveSum(dTemp3, sn, dRunCount, dCountPrefix, (int)numRuns + 1, s));
        sn = std::max(scanNeed, tempBytes);
        BSMR_HIP(hipcub::DeviceScan::ExclusiveSum(dTemp3, sn, dPanelRuns, dPanelRunStart, (int)P + 1, s));
        const uint32_t threshold = (uint32_t)std::ceil(delta * 256.0f);
        hipLaunchKernelGGL(bsmr::colreorderSplit, dim3((P + 255) / 256), dim3(256), 0, s, dPanelRunStart, dCountPrefix, P, threshold,
                           dNumDense, dNumSparse, dSparseEntries, dNumBlocks);
        BSMR_HIP(hipGetLastError());
        // offsets: small, scanned on the host
        std::vector<uint32_t> nd(P), ns(P), se(P), nb(P);
        BSMR_HIP(hipMemcpyAsync(nd.data(), dNumDense, (size_t)P * 4, hipMemcpyDeviceToHost, s));
        BSMR_HIP(hipMemcpyAsync(ns.data(), dNumSparse, (size_t)P * 4, hipMemcpyDeviceToHost, s));
        BSMR_HIP(hipMemcpyAsync(se.data(), dSparseEntries, (size_t)P * 4, hipMemcpyDeviceToHost, s));
        BSMR_HIP(hipMemcpyAsync(nb.data(), dNumBlocks, (size_t)P * 4, hipMemcpyDeviceToHost, s));
        BSMR_HIP(hipStreamSynchronize(s));
        uint64_t td = 0, ts = 0, te = 0, tb = 0;
        for (uint32_t p = 0; p < P; ++p) {
            td += nd[p]; ts += ns[p]; te += se[p]; tb += nb[p];
            if (td > 0xFFFFFFF0ull || ts > 0xFFFFFFF0ull || tb * 256ull > 0xFFFFFFFF0ull) return BSMR_ERR_OOM;
            h->denseColOffsets[p + 1] = (uint32_t)td;
            h->sparseColOffsets[p + 1] = (uint32_t)ts;
            h->sparseValueOffsets[p + 1] = (uint32_t)te;
            h->blockOffsets[p + 1] = (uint32_t)tb;
        }
        h->numDenseCols = td;
        h->numSparseCols = ts;
        h->numSparseEntries = te;
        h->numBlocks = tb;
        uint32_t* dDenseOff = dOffsets;
        uint32_t* dSparseOff = dOffsets + (P + 1);
        uint32_t* dValueOff = dOffsets + 2 * ((size_t)P + 1);
        uint32_t* dBlockOff = dOffsets + 3 * ((size_t)P + 1);
        BSMR_HIP(hipMemcpyAsync(dDenseOff, h->denseColOffsets.data(), ((size_t)P + 1) * 4, hipMemcpyHostToDevice, s));
        BSMR_HIP(hipMemcpyAsync(dSparseOff, h->sparseColOffsets.data(), ((size_t)P + 1) * 4, hipMemcpyHostToDevice, s));
        BSMR_HIP(hipMemcpyAsync(dValueOff, h->sparseValueOffsets.data(), ((size_t)P + 1) * 4, hipMemcpyHostToDevice, s));
        BSMR_HIP(hipMemcpyAsync(dBlockOff, h->blockOffsets.data(), ((size_t)P + 1) * 4, hipMemcpyHostToDevice, s));
        // 5. the arrays
        auto allocOut = [&](uint32_t** ptr, uint64_t count) {
            return hipOk(hipMalloc(reinterpret_cast<void**>(ptr), std::max<size_t>((size_t)count * 4, 16)), "hipMalloc(col reorder output)");
        };
        if (!allocOut(&h->denseCols, td) || !allocOut(&h->sparseCols, ts) || !allocOut(&h->blockValues, tb * 256) ||
            !allocOut(&h->sparseValues, te) || !allocOut(&h->sparseRows, te) || !allocOut(&h->sparseColIdx, te)) {
            bsmr_col_reorder_free(h.release());
            return BSMR_ERR_OOM;
        }
        auto fill = [&](uint32_t* ptr, uint64_t count, uint32_t v) {
            if (count) hipLaunchKernelGGL(bsmr::fillU32, dim3((unsigned)std::min<uint64_t>((count + 255) / 256, 65535)), dim3(256), 0, s, ptr, (size_t)count, v);
        };
        fill(h->denseCols, td, cols);             // padding sentinel = N
        fill(h->sparseCols, ts, cols);
        fill(h->blockValues, tb * 256, 0xFFFFFFFFu);
        hipLaunchKernelGGL(bsmr::colreorderScatter, dim3((numRuns + 255) / 256), dim3(256), 0, s, sRunKeys, sRunIndex, dRunStart, numRuns, n,
                           sKeys, sVals, dPanelRunStart, dCountPrefix, dNumDense, dDenseOff, dSparseOff, dValueOff, dBlockOff,
                           h->denseCols, h->sparseCols, h->blockValues, h->sparseValues, h->sparseRows, h->sparseColIdx);
        BSMR_HIP(hipGetLastError());
        BSMR_HIP(hipEventRecord(ev1, s));
        BSMR_HIP(hipEventSynchronize(ev1));
        (void)hipEventElapsedTime(&h->elapsedMs, ev0, ev1);
        *out = h.release();
        return BSMR_OK;
    } catch (const std::bad_alloc&) {
        return BSMR_ERR_OOM;
    } catch (...) {
        return BSMR_ERR_INVALID_ARG;
    }
}

int bsmr_col_reorder_sizes(const bsmr_colreorder* h, bsmr_colreorder_sizes* out) {
    if (!h || !out) return BSMR_ERR_INVALID_ARG;
    out->num_row_panels = h->numPanels;
    out->num_dense_cols = h->numDenseCols;
    out->num_sparse_cols = h->numSparseCols;
    out->num_blocks = h->numBlocks;
    out->num_sparse_entries = h->numSparseEntries;
    out->elapsed_ms = h->elapsedMs;
    return BSMR_OK;
}

int bsmr_col_reorder_fetch(const bsmr_colreorder* h, uint32_t* dense_cols, uint32_t* dense_col_offsets, uint32_t* sparse_cols,
                           uint32_t* sparse_col_offsets, uint32_t* sparse_value_offsets, uint32_t* block_offsets,
                           uint32_t* block_values, uint32_t* sparse_values, uint32_t* sparse_relative_rows,
                           uint32_t* sparse_col_indices) {
    if (!h) return BSMR_ERR_INVALID_ARG;
    DeviceRestore restore;   // (the calling thread's current device is the caller's business)
    BSMR_HIP(hipSetDevice(h->device));
    const size_t P1 = (size_t)h->numPanels + 1;
    auto host = [&](uint32_t* dst, const std::vector<uint32_t>& src) {
        if (dst) memcpy(dst, src.data(), P1 * 4);
    };
    host(dense_col_offsets, h->denseColOffsets);
    host(sparse_col_offsets, h->sparseColOffsets);
    host(sparse_value_offsets, h->sparseValueOffsets);
    host(block_offsets, h->blockOffsets);
    auto fetch = [&](uint32_t* dst, const uint32_t* src, uint64_t count) -> int {
        if (!dst || !count) return BSMR_OK;
        BSMR_HIP(hipMemcpy(dst, src, (size_t)count * 4, hipMemcpyDeviceToHost));
        return BSMR_OK;
    };
    int st = fetch(dense_cols, h->denseCols, h->numDenseCols);
    if (st == BSMR_OK) st = fetch(sparse_cols, h->sparseCols, h->numSparseCols);
    if (st == BSMR_OK) st = fetch(block_values, h->blockValues, h->numBlocks * 256);
    if (st == BSMR_OK) st = fetch(sparse_values, h->sparseValues, h->numSparseEntries);
    if (st == BSMR_OK) st = fetch(sparse_relative_rows, h->sparseRows, h->numSparseEntries);
    if (st == BSMR_OK) st = fetch(sparse_col_indices, h->sparseColIdx, h->numSparseEntries);
    return st;
}

int bsmr_col_reorder_device(const bsmr_colreorder* h, int* device) {
    if (!h || !device) return BSMR_ERR_INVALID_ARG;
    *device = h->device;
    return BSMR_OK;
}

int bsmr_plan_create_from_colreorder(bsmr_plan** out, const bsmr_colreorder* h, uint32_t M, uint32_t N, uint32_t nnz,
                                     const uint32_t* reordered_rows, uint32_t num_nonzero_rows, const bsmr_plan_options* options) {
    if (!out || !h) return BSMR_ERR_INVALID_ARG;
    *out = nullptr;
    if (h->numBlocks * 16 != h->numDenseCols) return BSMR_ERR_BAD_PLAN;
    bsmr_plan_options o;
    if (options) {
        if (options->struct_size < 2 * sizeof(uint32_t) || options->struct_size > sizeof(bsmr_plan_options)) return BSMR_ERR_INVALID_ARG;
        bsmr_plan_options_default(&o);
        memcpy(&o, options, options->struct_size);
        o.struct_size = (uint32_t)sizeof(bsmr_plan_options);
    } else {
        bsmr_plan_options_from_env(&o);
    }
    bsmr_rphm_desc d{};
    d.M = M;
    d.N = N;
    d.nnz = nnz;
    d.num_row_panels = h->numPanels;
    d.num_nonzero_rows = num_nonzero_rows;
    d.reordered_rows = reordered_rows;
    d.block_offsets = h->blockOffsets.data();
    d.sparse_value_offsets = h->sparseValueOffsets.data();
    const ResidentRphm res{h->denseCols, h->blockValues, h->sparseValues, h->sparseRows, h->sparseColIdx};
    int st = createPlan(out, h->device, &d, &o, &res);
    if (st != kPackOnHost) return st;
    // this plan needs the arrays on the host (an engine that keeps the dense entries there, a layout of the host packer, ...)
    try {
        std::vector<uint32_t> denseCols(h->numDenseCols), blockValues(h->numBlocks * 256), sparseValues(h->numSparseEntries),
            sparseRows(h->numSparseEntries), sparseCols(h->numSparseEntries);
        st = bsmr_col_reorder_fetch(h, denseCols.data(), nullptr, nullptr, nullptr, nullptr, nullptr, blockValues.data(), sparseValues.data(),
                                    sparseRows.data(), sparseCols.data());
        if (st != BSMR_OK) return st;
        d.dense_cols = denseCols.data();
        d.block_values = blockValues.data();
        d.sparse_values = sparseValues.data();
        d.sparse_relative_rows = sparseRows.data();
        d.sparse_col_indices = sparseCols.data();
        return bsmr_plan_create_ex(out, h->device, &d, &o);
    } catch (const std::bad_alloc&) {
        return BSMR_ERR_OOM;
    }
}

}  // extern "C"
