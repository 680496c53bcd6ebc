// Row clustering of the BSMR pipeline.
//
// Behaviour follows the reference's live path: calculateBlockSize
// (src/rowReordering.cu:1009-1025), the dispersion score + column-bin histogram
// (:49-93), the greedy BSA clustering in ascending-dispersion order with the
// normalised weighted Jaccard similarity (:235-293, :325-432, :893-1007) and the
// final stable sort by cluster id / removal of empty rows (:988-995, :1082-1090).
//
// The reference runs one single-block CUDA kernel per cluster, chained by
// device-side launches behind per-row mutexes; that pipeline is equivalent to
// finishing cluster c before cluster c+1 starts, which is what this host
// implementation does.  It never materialises the dense rows x bins table
// (5.7 GB for a reddit-sized matrix): every row keeps a sorted (bin, count)
// list, and an inverted bin -> rows index restricts each cluster's scan to the
// rows that share a bin with the running representative.  A row that shares no
// bin has similarity 0 and is rejected for every alpha >= 0, so skipping it does
// not change the outcome.
//
// Numerics: sums of squares are integers (64-bit here; the reference's 32-bit
// sums overflow on very large clusters); norms, quotients and the min/max sums
// are fp32, accumulated over the row's bins in ascending order plus one exact
// integer remainder term (see similarity()).  The reference sums all bins with a
// block-wide tree, so rows whose similarity sits within rounding of alpha may
// fall on the other side; any permutation is valid for SDDMM parity.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <functional>
#include <numeric>
#include <queue>
#include <vector>

#include "BSMR.hpp"
#include "bsmr_hip.h"

namespace {

using Clock = std::chrono::steady_clock;
inline float msSince(const Clock::time_point& t0) {
    return std::chrono::duration<float, std::milli>(Clock::now() - t0).count();
}

struct BinCount {
    UIN bin;
    UIN count;
};

// Per-row sparse histogram in CSR-like storage.
struct RowEncodings {
    std::vector<size_t> offsets;  // rows + 1
    std::vector<BinCount> items;  // sorted by bin inside a row
    std::vector<UIN> dispersion;  // 0 for empty rows
    std::vector<uint64_t> squares; // sum of count^2 per row
    const BinCount* begin(UIN row) const { return items.data() + offsets[row]; }
    const BinCount* end(UIN row) const { return items.data() + offsets[row + 1]; }
};

RowEncodings buildEncodings(const sparseMatrix::CSR<float>& m, const UIN binWidth) {
    const UIN rows = m.row();
    RowEncodings enc;
    enc.offsets.assign(static_cast<size_t>(rows) + 1, 0);
    enc.dispersion.assign(rows, 0);
    enc.squares.assign(rows, 0);
    std::vector<UIN> numBins(rows, 0);
    // pass 1: number of touched bins per row
#pragma omp parallel
    {
        std::vector<UIN> bins;
#pragma omp for schedule(dynamic, 512)
        for (long long r = 0; r < static_cast<long long>(rows); ++r) {
            const UIN b = m.rowOffsets()[r], e = m.rowOffsets()[r + 1];
            if (b == e) continue;
            bins.clear();
            for (UIN i = b; i < e; ++i) bins.push_back(m.colIndices()[i] / binWidth);
            std::sort(bins.begin(), bins.end());
            numBins[r] = static_cast<UIN>(std::unique(bins.begin(), bins.end()) - bins.begin());
        }
    }
    for (size_t r = 0; r < rows; ++r) enc.offsets[r + 1] = enc.offsets[r] + numBins[r];
    enc.items.resize(enc.offsets[rows]);
    // pass 2: fill (bin, count) and the dispersion score
#pragma omp parallel
    {
        std::vector<UIN> bins;
#pragma omp for schedule(dynamic, 512)
        for (long long r = 0; r < static_cast<long long>(rows); ++r) {
            const UIN b = m.rowOffsets()[r], e = m.rowOffsets()[r + 1];
            if (b == e) continue;
            bins.clear();
            for (UIN i = b; i < e; ++i) bins.push_back(m.colIndices()[i] / binWidth);
            std::sort(bins.begin(), bins.end());
            BinCount* out = enc.items.data() + enc.offsets[r];
            uint64_t slack = 0, sq = 0;
            size_t n = 0;
            for (size_t i = 0; i < bins.size();) {
                size_t j = i;
                while (j < bins.size() && bins[j] == bins[i]) ++j;
                out[n++] = BinCount{bins[i], static_cast<UIN>(j - i)};
                slack += binWidth - static_cast<UIN>(j - i);
                sq += static_cast<uint64_t>(j - i) * (j - i);
                i = j;
            }
            // sum over touched bins of (width - count)  +  nnz * #touched bins
            enc.dispersion[r] = static_cast<UIN>(slack + static_cast<uint64_t>(e - b) * n);
            enc.squares[r] = sq;
        }
    }
    return enc;
}

// Running representative of one cluster: dense counts + list of touched bins.
struct Representative {
    std::vector<UIN> count;     // numBins, zero outside `bins`
    std::vector<UIN> bins;      // touched bins, in order of first touch
    uint64_t sumSquares = 0;    // sum of count^2
    uint64_t total = 0;         // sum of count

    explicit Representative(size_t numBins) : count(numBins, 0) {}

    void clear() {
        for (const UIN b : bins) count[b] = 0;
        bins.clear();
        sumSquares = 0;
        total = 0;
    }

    // rep += row; returns (via newBins) the bins that were empty before.
    void add(const BinCount* rb, const BinCount* re, std::vector<UIN>& newBins) {
        newBins.clear();
        for (const BinCount* it = rb; it != re; ++it) {
            const uint64_t old = count[it->bin];
            if (old == 0) newBins.push_back(it->bin);
            sumSquares += 2 * old * it->count + static_cast<uint64_t>(it->count) * it->count;
            total += it->count;
            count[it->bin] += it->count;
        }
        bins.insert(bins.end(), newBins.begin(), newBins.end());
    }
};

// Normalised weighted Jaccard similarity of the representative x and one row y:
//   sum_b min(x_b/|x|, y_b/|y|) / sum_b max(x_b/|x|, y_b/|y|).
// Only the row's bins are visited (ascending): bins touched by the representative
// alone contribute x_b/|x| to the max-sum, and their total is (T - sum_{b in row} x_b)/|x|
// with T = sum_b x_b an exact integer - O(|row|) instead of O(|rep| + |row|).
inline float similarity(const Representative& rep, const BinCount* rb, const BinCount* re,
                        uint64_t rowSquares) {
    if (rep.sumSquares == 0 && rowSquares == 0) return 1.0f;
    if (rep.sumSquares == 0 || rowSquares == 0) return 0.0f;
    const float normRep = std::sqrt(static_cast<float>(rep.sumSquares));
    const float normRow = std::sqrt(static_cast<float>(rowSquares));
    float minSum = 0.0f, maxShared = 0.0f;
    uint64_t repInRow = 0;
    for (const BinCount* it = rb; it != re; ++it) {
        const UIN c = rep.count[it->bin];
        const float x = static_cast<float>(c) / normRep;
        const float y = static_cast<float>(it->count) / normRow;
        minSum += std::fmin(x, y);
        maxShared += std::fmax(x, y);
        repInRow += c;
    }
    const float maxSum = maxShared + static_cast<float>(rep.total - repInRow) / normRep;
    return minSum / maxSum;
}

}  // namespace

void noReorderRow(const sparseMatrix::CSR<float>& matrix, std::vector<UIN>& reorderedRows,
                  float& time) {
    const auto t0 = Clock::now();
    reorderedRows.clear();
    for (UIN r = 0; r < matrix.row(); ++r)
        if (matrix.rowOffsets()[r + 1] > matrix.rowOffsets()[r]) reorderedRows.push_back(r);
    time = msSince(t0);
}

UIN calculateBlockSize(const sparseMatrix::CSR<float>& matrix, size_t freeDeviceBytes) {
    if (freeDeviceBytes < 2) freeDeviceBytes = 2;
    const size_t rows = matrix.row();
    const UIN dueToMemory = static_cast<UIN>(std::ceil(
        static_cast<float>(rows * rows * sizeof(UIN)) / static_cast<float>(freeDeviceBytes / 2)));
    const UIN dueToLds = static_cast<UIN>(
        std::ceil(static_cast<float>(static_cast<size_t>(matrix.col()) * sizeof(UIN)) /
                  static_cast<float>(maxSharedMemoryPerBlock / 2)));
    return std::max<UIN>(16, std::max(dueToMemory, dueToLds));
}

UIN calculateBlockSize(const sparseMatrix::CSR<float>& matrix) {
    size_t freeBytes = 0, totalBytes = 0;
    if (bsmr_mem_info(0, &freeBytes, &totalBytes) != BSMR_OK || freeBytes == 0)
        freeBytes = static_cast<size_t>(288) << 30;  // MI355X HBM3E capacity
    return calculateBlockSize(matrix, freeBytes);
}

std::vector<UIN> bsa_rowReordering_host(const sparseMatrix::CSR<float>& matrix, const float alpha,
                                        const UIN block_size, int& num_clusters,
                                        float& reordering_time) {
    const auto t0 = Clock::now();
    const UIN rows = matrix.row();
    const UIN binWidth = block_size == 0 ? 16 : block_size;
    const size_t numBins = static_cast<size_t>(
        std::ceil(static_cast<float>(matrix.col()) / static_cast<float>(binWidth)));

    const RowEncodings enc = buildEncodings(matrix, binWidth);

    // rows in ascending dispersion, ties in ascending row id (stable)
    std::vector<UIN> order(rows);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](UIN a, UIN b) {
        return enc.dispersion[a] < enc.dispersion[b];
    });

    // cluster id per position of `order`; empty rows (dispersion 0) form cluster 0
    std::vector<UIN> cluster(rows, NULL_VALUE);
    UIN firstNonEmpty = 0;
    while (firstNonEmpty < rows && enc.dispersion[order[firstNonEmpty]] == 0)
        cluster[firstNonEmpty++] = 0;

    // inverted index: bin -> positions (ascending) of the non-empty rows touching it
    std::vector<size_t> invOffsets(numBins + 1, 0);
    for (const BinCount& bc : enc.items) ++invOffsets[bc.bin + 1];
    for (size_t b = 0; b < numBins; ++b) invOffsets[b + 1] += invOffsets[b];
    std::vector<UIN> invItems(enc.items.size());
    std::vector<size_t> invLen(numBins, 0);  // live length (lists are compacted lazily)
    for (UIN pos = firstNonEmpty; pos < rows; ++pos) {
        const UIN row = order[pos];
        for (const BinCount* it = enc.begin(row); it != enc.end(row); ++it)
            invItems[invOffsets[it->bin] + invLen[it->bin]++] = pos;
    }

    Representative rep(numBins);
    std::vector<UIN> seenBy(rows, 0);  // last cluster id that queued this position
    std::vector<UIN> newBins, pending;
    std::priority_queue<UIN, std::vector<UIN>, std::greater<UIN>> candidates;
    const bool scanEverything = !(alpha >= 0.0f);  // negative alpha accepts disjoint rows too

    // queue the unassigned positions > after of one bin; drops assigned ones for good
    auto enqueueBin = [&](UIN bin, UIN after, UIN clusterId) {
        UIN* list = invItems.data() + invOffsets[bin];
        size_t keep = 0;
        for (size_t i = 0; i < invLen[bin]; ++i) {
            const UIN pos = list[i];
            if (cluster[pos] != NULL_VALUE) continue;
            list[keep++] = pos;
            if (pos > after && seenBy[pos] != clusterId) {
                seenBy[pos] = clusterId;
                candidates.push(pos);
            }
        }
        invLen[bin] = keep;
    };

    UIN clusterId = 0;
    UIN start = firstNonEmpty;
    while (start < rows) {
        ++clusterId;
        cluster[start] = clusterId;
        rep.clear();
        rep.add(enc.begin(order[start]), enc.end(order[start]), newBins);
        // When the representative's bins already reach most of the remaining rows the
        // inverted index only adds work: scan every unassigned row instead, evaluating
        // the similarities of a chunk in parallel against the current representative
        // (rows before the first accepted one of a chunk were judged with the right
        // representative; the rest of the chunk is re-judged after the merge).
        bool scanAll = scanEverything;
        if (!scanAll) {
            size_t reach = 0;
            for (const UIN b : newBins) reach += invLen[b];
            scanAll = reach >= static_cast<size_t>(rows - start);
        }
        if (scanAll) {
            pending.clear();
            for (UIN pos = start + 1; pos < rows; ++pos)
                if (cluster[pos] == NULL_VALUE) pending.push_back(pos);
            constexpr size_t kChunk = 512;
            size_t i = 0;
            while (i < pending.size()) {
                const size_t n = std::min(kChunk, pending.size() - i);
                long long firstHit = static_cast<long long>(n);
#pragma omp parallel for schedule(static) reduction(min : firstHit) if (n >= 64)
                for (long long j = 0; j < static_cast<long long>(n); ++j) {
                    const UIN row = order[pending[i + j]];
                    if (similarity(rep, enc.begin(row), enc.end(row), enc.squares[row]) > alpha)
                        firstHit = std::min(firstHit, j);
                }
                if (firstHit == static_cast<long long>(n)) {
                    i += n;
                    continue;
                }
                const UIN pos = pending[i + firstHit];
                cluster[pos] = clusterId;
                rep.add(enc.begin(order[pos]), enc.end(order[pos]), newBins);
                i += static_cast<size_t>(firstHit) + 1;
            }
        } else {
            for (const UIN b : newBins) enqueueBin(b, start, clusterId);
            while (!candidates.empty()) {
                const UIN pos = candidates.top();
                candidates.pop();
                const UIN row = order[pos];
                if (similarity(rep, enc.begin(row), enc.end(row), enc.squares[row]) > alpha) {
                    cluster[pos] = clusterId;
                    rep.add(enc.begin(row), enc.end(row), newBins);
                    for (const UIN b : newBins) enqueueBin(b, pos, clusterId);
                }
            }
        }
        // the first row the cluster left behind seeds the next one
        while (start < rows && cluster[start] != NULL_VALUE) ++start;
    }

    // positions stably sorted by cluster id, mapped back to rows
    std::vector<UIN> positions(rows);
    std::iota(positions.begin(), positions.end(), 0);
    std::stable_sort(positions.begin(), positions.end(),
                     [&](UIN a, UIN b) { return cluster[a] < cluster[b]; });
    num_clusters = rows == 0 ? 0
                             : static_cast<int>(cluster[positions[rows - 1]]) +
                                   (firstNonEmpty != 0 ? 1 : 0);

    std::vector<UIN> permutation;
    permutation.reserve(rows - firstNonEmpty);
    bool leading = true;
    for (UIN i = 0; i < rows; ++i) {
        const UIN row = order[positions[i]];
        if (leading && matrix.rowOffsets()[row + 1] == matrix.rowOffsets()[row]) continue;
        leading = false;
        permutation.push_back(row);
    }
    reordering_time = msSince(t0);
    return permutation;
}
