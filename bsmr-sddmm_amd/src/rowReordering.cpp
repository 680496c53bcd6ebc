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
// rows that share a bin with the running representative.
//
// Numerics - the similarity is evaluated exactly as the reference's block
// executes it, which is what makes cluster counts and dense-block statistics equal
// the reference's published logs (tests/test_reference_logs.py):
//   * a block of T = clusterThreads(numBins) threads (:912-922); thread t owns bins
//     t, t+T, ... and accumulates them in that order in fp32;
//   * the block-wide sum (include/cudaUtil.cuh:14-45) adds the 32 lanes of a warp in
//     a balanced tree and then folds the warps with `for (s = T/64; s; s >>= 1)
//     v[w] += v[w+s]`: when T/32 is not a power of two some warps are never read, so
//     their bins take no part in the sums of squares, the min-sum or the max-sum
//     (BlockSum::live marks the bins that count);
//   * sums of squares are 32-bit unsigned with wrap-around, norms are sqrtf of their
//     float conversion, quotients are IEEE fp32 divisions.
// The fixed shape of that sum is kept as an explicit binary tree: the values of the
// representative alone are cached per node, and one (representative, row) pair only
// recomputes the leaves the row touches and their ancestors - O(|row| log T) per pair
// instead of O(numBins), with bit-identical results.  Since a similarity is only ever
// compared with alpha, the same quotient is first formed in double arithmetic in
// O(|row|); it differs from the fp32 block sum by a few 1e-6 at most, so only values
// within 1e-4 of alpha take the exact path (cop20k-like, 121 k rows: 4.7 -> 1.05 s;
// Trefethen_20000 alpha=0.3: 61 ms, wathen100: 7 ms - the reference's GPU kernels took
// 560 and 417 ms on an RTX 4090).

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <functional>
#include <numeric>
#include <queue>
#include <vector>

#include <omp.h>
#include "util.hpp"

#include "BSMR.hpp"
#include "bsmr_hip.h"

constexpr UIN WARP_SIZE_NV = 32;  // the reference's block reduction is defined on 32-lane warps

namespace {

using Clock = std::chrono::steady_clock;
inline float msSince(const Clock::time_point& t0) {
    return std::chrono::duration<float, std::milli>(Clock::now() - t0).count();
}

// threads of the reference's clustering block (src/rowReordering.cu:912-922)
UIN clusterThreads(const size_t numBins) {
    if (numBins < 32) return 32;
    const UIN candidate =
        WARP_SIZE_NV * static_cast<UIN>(std::ceil(static_cast<float>(numBins / 4) / static_cast<float>(WARP_SIZE_NV)));
    return std::min<UIN>(1024, std::max<UIN>(32, candidate));
}

// The reference's block-wide sum as a binary tree over the T per-thread partials.
// Nodes 0..T-1 are the leaves; children always have smaller ids than their parent.
struct BlockSum {
    UIN threads = 0;
    std::vector<int> left, right, parent;
    std::vector<uint8_t> live;  // node contributes to the root
    int root = 0;

    explicit BlockSum(UIN T) : threads(T) {
        const UIN warps = T / WARP_SIZE_NV;
        left.assign(T, -1);
        right.assign(T, -1);
        auto join = [&](int a, int b) {
            left.push_back(a);
            right.push_back(b);
            return static_cast<int>(left.size()) - 1;
        };
        std::vector<int> slot(warps);
        for (UIN w = 0; w < warps; ++w) {
            int lane[WARP_SIZE_NV];
            for (UIN l = 0; l < WARP_SIZE_NV; ++l) lane[l] = static_cast<int>(WARP_SIZE_NV * w + l);
            for (UIN step = 1; step < WARP_SIZE_NV; step <<= 1)  // shuffle-xor butterfly, lane 0
                for (UIN l = 0; l < WARP_SIZE_NV; l += 2 * step) lane[l] = join(lane[l], lane[l + step]);
            slot[w] = lane[0];
        }
        for (UIN s = T / (2 * WARP_SIZE_NV); s >= 1; s >>= 1)  // include/cudaUtil.cuh:37-43
            for (UIN w = 0; w < s; ++w) slot[w] = join(slot[w], slot[w + s]);
        root = slot[0];
        parent.assign(left.size(), -1);
        for (size_t n = T; n < left.size(); ++n) parent[left[n]] = parent[right[n]] = static_cast<int>(n);
        live.assign(left.size(), 0);
        live[root] = 1;
        for (int n = root; n >= static_cast<int>(T); --n)
            if (live[n]) live[left[n]] = live[right[n]] = 1;
    }
    size_t nodes() const { return left.size(); }
    bool binCounts(UIN bin) const { return live[bin % threads] != 0; }
};

struct BinCount {
    UIN bin;
    UIN count;
};

// Per-row sparse histogram in CSR-like storage.
struct RowEncodings {
    std::vector<size_t> offsets;   // rows + 1
    std::vector<BinCount> items;   // sorted by bin inside a row
    std::vector<UIN> dispersion;   // 0 for empty rows
    std::vector<uint32_t> squares; // sum of count^2 over the bins that count, modulo 2^32
    // the bins that count once more, as what the O(|row|) similarity estimate reads: bin ids and
    // y = count / |row| in double arithmetic, and their sum per row
    // (one record per row and one per bin, so that judging a row touches two places in memory)
    struct Counted {
        double y;
        UIN bin;
    };
    struct CountedRow {
        size_t first;      // into `counted`
        double sumY;
        uint32_t n;
        uint32_t squares;  // == squares[row]
    };
    std::vector<Counted> counted;
    std::vector<CountedRow> countedRow;  // rows
    const BinCount* begin(UIN row) const { return items.data() + offsets[row]; }
    const BinCount* end(UIN row) const { return items.data() + offsets[row + 1]; }
};

RowEncodings buildEncodings(const sparseMatrix::CSR<float>& m, const UIN binWidth, const BlockSum& sum) {
    const UIN rows = m.row();
    RowEncodings enc;
    enc.offsets.assign(static_cast<size_t>(rows) + 1, 0);
    enc.dispersion.assign(rows, 0);
    enc.squares.assign(rows, 0);
    std::vector<UIN> numBins(rows, 0);
    // pass 1: number of touched bins per row
#pragma omp parallel num_threads(util::hostThreads(omp_get_max_threads()))
    {
        std::vector<UIN> bins;
#pragma omp for schedule(dynamic, 512)
        for (long long r = 0; r < static_cast<long long>(rows); ++r) {
            const UIN b = m.rowOffsets()[r], e = m.rowOffsets()[r + 1];
            if (b == e) continue;
            bins.clear();
            for (UIN i = b; i < e; ++i) bins.push_back(m.colIndices()[i] / binWidth);
            if (!std::is_sorted(bins.begin(), bins.end())) std::sort(bins.begin(), bins.end());
            numBins[r] = static_cast<UIN>(std::unique(bins.begin(), bins.end()) - bins.begin());
        }
    }
    for (size_t r = 0; r < rows; ++r) enc.offsets[r + 1] = enc.offsets[r] + numBins[r];
    enc.items.resize(enc.offsets[rows]);
    // pass 2: fill (bin, count) and the dispersion score
#pragma omp parallel num_threads(util::hostThreads(omp_get_max_threads()))
    {
        std::vector<UIN> bins;
#pragma omp for schedule(dynamic, 512)
        for (long long r = 0; r < static_cast<long long>(rows); ++r) {
            const UIN b = m.rowOffsets()[r], e = m.rowOffsets()[r + 1];
            if (b == e) continue;
            bins.clear();
            for (UIN i = b; i < e; ++i) bins.push_back(m.colIndices()[i] / binWidth);
            if (!std::is_sorted(bins.begin(), bins.end())) std::sort(bins.begin(), bins.end());
            BinCount* out = enc.items.data() + enc.offsets[r];
            uint64_t slack = 0;
            uint32_t sq = 0;
            size_t n = 0;
            for (size_t i = 0; i < bins.size();) {
                size_t j = i;
                while (j < bins.size() && bins[j] == bins[i]) ++j;
                const UIN c = static_cast<UIN>(j - i);
                out[n++] = BinCount{bins[i], c};
                slack += binWidth - c;
                if (sum.binCounts(bins[i])) sq += c * c;
                i = j;
            }
            // sum over touched bins of (width - count)  +  nnz * #touched bins
            enc.dispersion[r] = static_cast<UIN>(slack + static_cast<uint64_t>(e - b) * n);
            enc.squares[r] = sq;
        }
    }
    enc.countedRow.assign(rows, RowEncodings::CountedRow{0, 0.0, 0, 0});
    size_t total = 0;
    for (size_t r = 0; r < rows; ++r) {
        uint32_t n = 0;
        if (enc.squares[r] != 0)
            for (const BinCount* it = enc.begin(static_cast<UIN>(r)); it != enc.end(static_cast<UIN>(r)); ++it)
                n += sum.binCounts(it->bin);
        enc.countedRow[r] = RowEncodings::CountedRow{total, 0.0, n, enc.squares[r]};
        total += n;
    }
    enc.counted.resize(total);
#pragma omp parallel for schedule(dynamic, 512) num_threads(util::hostThreads(omp_get_max_threads()))
    for (long long r = 0; r < static_cast<long long>(rows); ++r) {
        if (enc.squares[r] == 0) continue;
        const double normRow = std::sqrt(static_cast<float>(enc.squares[r]));
        RowEncodings::Counted* out = enc.counted.data() + enc.countedRow[r].first;
        double sumY = 0.0;
        for (const BinCount* it = enc.begin(static_cast<UIN>(r)); it != enc.end(static_cast<UIN>(r)); ++it) {
            if (!sum.binCounts(it->bin)) continue;
            const double y = static_cast<double>(it->count) / normRow;
            *out++ = RowEncodings::Counted{y, it->bin};
            sumY += y;
        }
        enc.countedRow[r].sumY = sumY;
    }
    return enc;
}

// Running representative of one cluster: dense counts, the touched bins, and the
// block-sum tree of its own normalised counts (what every pair starts from).
struct Representative {
    const BlockSum& sum;
    size_t numBins;
    std::vector<UIN> count;       // numBins, zero outside `bins`
    std::vector<UIN> bins;        // touched bins, in order of first touch
    uint32_t sumSquares = 0;      // over the bins that count, modulo 2^32 (UIN accumulator of the reference)
    uint64_t totalCounted = 0;    // sum of the counts over the bins that count
    float norm = 0.0f;            // sqrtf(float(sumSquares))
    std::vector<double> x;        // numBins: count / norm over the bins that count (the estimate's operand), else 0
    double sumX = 0.0;            // totalCounted / norm
    // per node: block sum of count/norm over the representative alone.  Only the exact evaluation reads it
    // (a handful of pairs per matrix), so it is rebuilt on demand: tree() from any thread, once per change.
    mutable std::vector<float> maxTree;
    mutable std::atomic<bool> treeValid{false};

    mutable std::vector<int> touched;     // nodes of maxTree that are not zero
    mutable std::vector<uint32_t> stamp;  // per node, == epoch: already collected by this rebuild
    mutable uint32_t epoch = 0;

    Representative(const BlockSum& s, size_t nb)
        : sum(s), numBins(nb), count(nb, 0), x(nb, 0.0), maxTree(s.nodes(), 0.0f), stamp(s.nodes(), 0) {}

    void clear() {
        for (const UIN b : bins) {
            count[b] = 0;
            x[b] = 0.0;
        }
        bins.clear();
        sumSquares = 0;
        totalCounted = 0;
    }

    // rep += row; returns (via newBins) the bins that count and were empty before.
    void add(const BinCount* rb, const BinCount* re, std::vector<UIN>& newBins) {
        newBins.clear();
        for (const BinCount* it = rb; it != re; ++it) {
            const UIN old = count[it->bin];
            if (old == 0) bins.push_back(it->bin);
            if (sum.binCounts(it->bin)) {
                if (old == 0) newBins.push_back(it->bin);
                sumSquares += 2u * old * it->count + it->count * it->count;
                totalCounted += it->count;
            }
            count[it->bin] += it->count;
        }
        norm = std::sqrt(static_cast<float>(sumSquares));
        if (sumSquares != 0) {
            const double n = norm;
            for (const UIN b : bins)
                if (sum.binCounts(b)) x[b] = static_cast<double>(count[b]) / n;
            sumX = static_cast<double>(totalCounted) / n;
        }
        treeValid.store(false, std::memory_order_release);
    }

    const std::vector<float>& tree() const {
        if (!treeValid.load(std::memory_order_acquire)) {
#pragma omp critical(bsmr_representative_tree)
            if (!treeValid.load(std::memory_order_acquire)) {
                rebuild();
                treeValid.store(true, std::memory_order_release);
            }
        }
        return maxTree;
    }

    // The tree of the representative's own normalised counts.  Every value changes with the norm, but
    // only the leaves that own one of its bins, and their ancestors, are not zero: a narrow
    // representative (banded matrices: ~30 of 6 000 bins) costs its own size, not the tree's.
    void rebuild() const {
        const UIN T = sum.threads;
        for (const int n : touched) maxTree[n] = 0.0f;
        touched.clear();
        if (sumSquares == 0) return;  // similarity never reaches the sums
        if (bins.size() * 4 >= T) {   // wide representative: plain sweep
            for (UIN t = 0; t < T; ++t) {
                float acc = 0.0f;
                if (sum.live[t])
                    for (size_t b = t; b < numBins; b += T) acc = acc + static_cast<float>(count[b]) / norm;
                maxTree[t] = acc;
            }
            for (size_t n = T; n < sum.nodes(); ++n) maxTree[n] = maxTree[sum.left[n]] + maxTree[sum.right[n]];
            touched.resize(sum.nodes());
            std::iota(touched.begin(), touched.end(), 0);
            return;
        }
        if (++epoch == 0) {
            std::fill(stamp.begin(), stamp.end(), 0u);
            epoch = 1;
        }
        size_t leaves = 0;
        for (const UIN b : bins) {
            const UIN t = b % T;
            if (!sum.live[t] || stamp[t] == epoch) continue;
            stamp[t] = epoch;
            float acc = 0.0f;
            for (size_t c = t; c < numBins; c += T) acc = acc + static_cast<float>(count[c]) / norm;
            maxTree[t] = acc;
            touched.push_back(static_cast<int>(t));
            ++leaves;
        }
        for (size_t i = 0; i < leaves; ++i)
            for (int p = sum.parent[touched[i]]; p >= 0 && stamp[p] != epoch; p = sum.parent[p]) {
                stamp[p] = epoch;
                touched.push_back(p);
            }
        std::sort(touched.begin() + leaves, touched.end());  // children before parents
        for (size_t i = leaves; i < touched.size(); ++i) {
            const int n = touched[i];
            maxTree[n] = maxTree[sum.left[n]] + maxTree[sum.right[n]];
        }
    }
};

// Set of positions with pop-min: a 64-ary tree of bit words (a priority queue of millions of cheap pushes
// and pops cost more than the similarities it fed).
class PositionSet {
public:
    explicit PositionSet(size_t n) {
        size_t words = (n + 63) / 64;
        for (;;) {
            levels_.emplace_back(std::max<size_t>(words, 1), 0ull);
            if (words <= 1) break;
            words = (words + 63) / 64;
        }
    }
    bool empty() const { return levels_.back()[0] == 0; }
    void insert(UIN pos) {
        size_t i = pos;
        for (auto& level : levels_) {
            uint64_t& w = level[i >> 6];
            const bool wasEmpty = w == 0;
            w |= 1ull << (i & 63);
            if (!wasEmpty) break;
            i >>= 6;
        }
    }
    UIN popMin() {
        size_t i = 0;
        for (size_t l = levels_.size(); l-- > 0;) i = (i << 6) | static_cast<size_t>(__builtin_ctzll(levels_[l][i]));
        size_t j = i;
        for (auto& level : levels_) {
            uint64_t& w = level[j >> 6];
            w &= ~(1ull << (j & 63));
            if (w != 0) break;
            j >>= 6;
        }
        return static_cast<UIN>(i);
    }

private:
    std::vector<std::vector<uint64_t>> levels_;  // [0] = one bit per position
};

// Per-thread scratch of the pair evaluation.
struct PairScratch {
    std::vector<UIN> rowCount;                 // numBins, zero between calls
    std::vector<float> maxValue, minValue;     // per node, valid where stamp == epoch
    std::vector<uint32_t> stamp;
    std::vector<int> dirty;
    uint32_t epoch = 0;
    PairScratch(size_t numBins, size_t nodes)
        : rowCount(numBins, 0), maxValue(nodes), minValue(nodes), stamp(nodes, 0) {}
};

// Normalised weighted Jaccard similarity of the representative x and one row y,
//   sum_b min(x_b/|x|, y_b/|y|) / sum_b max(x_b/|x|, y_b/|y|),
// as executed by the reference (src/rowReordering.cu:235-293): only the leaves (threads)
// that own one of the row's bins differ from the representative's cached tree.
float similarity(const Representative& rep, const RowEncodings& enc, const UIN row, PairScratch& sc, const float alpha) {
    const RowEncodings::CountedRow& cr = enc.countedRow[row];
    const uint32_t rowSquares = cr.squares;
    if (rep.sumSquares == 0 && rowSquares == 0) return 1.0f;
    if (rep.sumSquares == 0 || rowSquares == 0) return 0.0f;
    // The value is only compared with alpha.  The same quotient in double arithmetic, O(|row|): with x, y the
    // normalised counts, min + max = x + y in every bin, so  sum max = |x|_1 + |y|_1 - sum min  and only the
    // min-sum over the row's bins has to be formed.  The fp32 block sum differs from it by a few 1e-6 at most
    // (positive terms, < 20 roundings each), so anything further than 1e-4 from alpha is decided here; the rest
    // goes through the exact order of operations below.
    {
        const double sumX = rep.sumX, sumY = cr.sumY;
        const double a = static_cast<double>(alpha);
        // sum min <= min(|x|_1, |y|_1) and sum max >= max(|x|_1, |y|_1): rows of a different mass are rejected unseen
        const double bound = sumX < sumY ? sumX / sumY : sumY / sumX;
        if (bound < a - 1e-4) return static_cast<float>(bound);
        const RowEncodings::Counted* c = enc.counted.data() + cr.first;
        const size_t n = cr.n;
        const double* x = rep.x.data();
        double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
        size_t i = 0;
        for (; i + 4 <= n; i += 4) {
            m0 += std::min(x[c[i].bin], c[i].y);
            m1 += std::min(x[c[i + 1].bin], c[i + 1].y);
            m2 += std::min(x[c[i + 2].bin], c[i + 2].y);
            m3 += std::min(x[c[i + 3].bin], c[i + 3].y);
        }
        for (; i < n; ++i) m0 += std::min(x[c[i].bin], c[i].y);
        const double minSum = (m0 + m1) + (m2 + m3);
        const double approx = minSum / (sumX + sumY - minSum);
        if (std::fabs(approx - a) > 1e-4) return static_cast<float>(approx);
    }
    const BinCount* rb = enc.begin(row);
    const BinCount* re = enc.end(row);
    const BlockSum& sum = rep.sum;
    const UIN T = sum.threads;
    const float normRow = std::sqrt(static_cast<float>(rowSquares));
    const std::vector<float>& repTree = rep.tree();
    if (++sc.epoch == 0) {
        std::fill(sc.stamp.begin(), sc.stamp.end(), 0u);
        sc.epoch = 1;
    }
    for (const BinCount* it = rb; it != re; ++it) sc.rowCount[it->bin] = it->count;
    sc.dirty.clear();
    for (const BinCount* it = rb; it != re; ++it) {
        const UIN t = it->bin % T;
        if (!sum.live[t] || sc.stamp[t] == sc.epoch) continue;
        sc.stamp[t] = sc.epoch;
        float mx = 0.0f, mn = 0.0f;
        for (size_t b = t; b < rep.numBins; b += T) {
            const float x = static_cast<float>(rep.count[b]) / rep.norm;
            const float y = static_cast<float>(sc.rowCount[b]) / normRow;
            mn = mn + std::fmin(x, y);
            mx = mx + std::fmax(x, y);
        }
        sc.maxValue[t] = mx;
        sc.minValue[t] = mn;
        for (int p = sum.parent[t]; p >= 0 && sc.stamp[p] != sc.epoch; p = sum.parent[p]) {
            sc.stamp[p] = sc.epoch;
            sc.dirty.push_back(p);
        }
    }
    for (const BinCount* it = rb; it != re; ++it) sc.rowCount[it->bin] = 0;
    std::sort(sc.dirty.begin(), sc.dirty.end());
    for (const int n : sc.dirty) {
        const int a = sum.left[n], b = sum.right[n];
        const bool da = sc.stamp[a] == sc.epoch, db = sc.stamp[b] == sc.epoch;
        sc.maxValue[n] = (da ? sc.maxValue[a] : repTree[a]) + (db ? sc.maxValue[b] : repTree[b]);
        sc.minValue[n] = (da ? sc.minValue[a] : 0.0f) + (db ? sc.minValue[b] : 0.0f);
    }
    const bool touched = sc.stamp[sum.root] == sc.epoch;
    const float minSum = touched ? sc.minValue[sum.root] : 0.0f;
    const float maxSum = touched ? sc.maxValue[sum.root] : repTree[sum.root];
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
    if (bsmr_mem_info(pipelineDevice(), &freeBytes, &totalBytes) != BSMR_OK || freeBytes == 0)
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
    const BlockSum blockSum(clusterThreads(numBins));

    const RowEncodings enc = buildEncodings(matrix, binWidth, blockSum);

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

    // inverted index over the bins that count: bin -> positions (ascending) of the rows
    // touching it.  A row that shares no such bin with the representative has min-sum 0,
    // similarity 0, and is rejected for every alpha >= 0.
    std::vector<size_t> invOffsets(numBins + 1, 0);
    for (const BinCount& bc : enc.items)
        if (blockSum.binCounts(bc.bin)) ++invOffsets[bc.bin + 1];
    for (size_t b = 0; b < numBins; ++b) invOffsets[b + 1] += invOffsets[b];
    std::vector<UIN> invItems(invOffsets[numBins]);
    std::vector<size_t> invLen(numBins, 0);  // live length (lists are compacted lazily)
    // rows whose counted sum of squares is zero: similarity 1 with a representative in the
    // same state, 0 otherwise (src/rowReordering.cu:263-268)
    std::vector<UIN> zeroSquarePositions;
    for (UIN pos = firstNonEmpty; pos < rows; ++pos) {
        const UIN row = order[pos];
        if (enc.squares[row] == 0) {
            zeroSquarePositions.push_back(pos);
            continue;
        }
        for (const BinCount* it = enc.begin(row); it != enc.end(row); ++it)
            if (blockSum.binCounts(it->bin)) invItems[invOffsets[it->bin] + invLen[it->bin]++] = pos;
    }

    Representative rep(blockSum, numBins);
    // one short parallel region per chunk: beyond ~16 threads the barriers cost more than the
    // similarities (measured: a 128-thread box ran this 2-3x slower than an 8-thread one)
    const int maxThreads = std::min(util::hostThreads(omp_get_max_threads()), 16);
    std::vector<PairScratch> scratch;
    scratch.reserve(maxThreads);
    for (int t = 0; t < maxThreads; ++t) scratch.emplace_back(numBins, blockSum.nodes());
    std::vector<UIN> newBins, pending;
    PositionSet candidates(rows);
    const bool scanEverything = !(alpha >= 0.0f);  // negative alpha accepts disjoint rows too
    const size_t kMinChunk = static_cast<size_t>(2 * maxThreads), kMaxChunk = 1024;
    size_t chunk = 256, sinceHit = 0;  // rows judged per parallel region of the scan path, rows judged since an acceptance

    // queue the unassigned positions > after of one bin; drops assigned ones for good
    auto enqueueBin = [&](UIN bin, UIN after) {
        UIN* list = invItems.data() + invOffsets[bin];
        size_t keep = 0;
        for (size_t i = 0; i < invLen[bin]; ++i) {
            const UIN pos = list[i];
            if (cluster[pos] != NULL_VALUE) continue;
            list[keep++] = pos;
            if (pos > after) candidates.insert(pos);   // (everything at or before `after` has been judged)
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
        if (!scanAll && rep.sumSquares != 0) {
            size_t reach = 0;
            for (const UIN b : newBins) reach += invLen[b];
            scanAll = reach >= static_cast<size_t>(rows - start);
        }
        if (scanAll) {
            pending.clear();
            for (UIN pos = start + 1; pos < rows; ++pos)
                if (cluster[pos] == NULL_VALUE) pending.push_back(pos);
            // Rows behind the first accepted one of a chunk were judged against a representative that has
            // changed since: wasted work.  The chunk follows the distance between acceptances (doubling while
            // nothing is accepted), which keeps the waste near the useful work whatever the cluster sizes are.
            size_t i = 0;
            while (i < pending.size()) {
                const size_t n = std::min(chunk, pending.size() - i);
                long long firstHit = static_cast<long long>(n);
#pragma omp parallel for schedule(static) reduction(min : firstHit) num_threads(maxThreads) if (n >= 32)
                for (long long j = 0; j < static_cast<long long>(n); ++j) {
                    const UIN row = order[pending[i + j]];
                    if (similarity(rep, enc, row, scratch[omp_get_thread_num()], alpha) > alpha)
                        firstHit = std::min(firstHit, j);
                }
                if (firstHit == static_cast<long long>(n)) {
                    i += n;
                    sinceHit += n;
                    chunk = std::min<size_t>(2 * chunk, kMaxChunk);
                    continue;
                }
                sinceHit += static_cast<size_t>(firstHit) + 1;
                chunk = std::min(std::max(sinceHit, kMinChunk), kMaxChunk);
                sinceHit = 0;
                const UIN pos = pending[i + firstHit];
                cluster[pos] = clusterId;
                rep.add(enc.begin(order[pos]), enc.end(order[pos]), newBins);
                i += static_cast<size_t>(firstHit) + 1;
            }
        } else if (rep.sumSquares == 0) {
            // nothing of the seed row counts: it matches exactly the rows in the same state,
            // and merging them leaves the representative in that state
            if (1.0f > alpha) {
                size_t keep = 0;
                for (const UIN pos : zeroSquarePositions) {
                    if (cluster[pos] != NULL_VALUE) continue;
                    if (pos > start) cluster[pos] = clusterId;
                    else zeroSquarePositions[keep++] = pos;
                }
                zeroSquarePositions.resize(keep);
            }
        } else {
            for (const UIN b : newBins) enqueueBin(b, start);
            while (!candidates.empty()) {
                const UIN pos = candidates.popMin();
                const UIN row = order[pos];
                if (similarity(rep, enc, row, scratch[0], alpha) > alpha) {
                    cluster[pos] = clusterId;
                    rep.add(enc.begin(row), enc.end(row), newBins);
                    for (const UIN b : newBins) enqueueBin(b, pos);
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
    // The reference reads the *sorted* id array at the unsorted position of its last element
    // (src/rowReordering.cu:985-992: sort_by_key sorts cluster_ids in place, then
    // cluster_ids[indices[rows - 1]]); the logged count follows that, quirk included.
    num_clusters = rows == 0 ? 0
                             : static_cast<int>(cluster[positions[positions[rows - 1]]]) +
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

// ---------------------------------------------------------------------------
// device path
// ---------------------------------------------------------------------------
namespace {
int g_clusteringDevice = -2;
thread_local int g_pipelineDevice = 0;
}
void setClusteringDevice(int device) { g_clusteringDevice = device; }
int clusteringDevice() { return g_clusteringDevice; }
void setPipelineDevice(int device) { g_pipelineDevice = device < 0 ? 0 : device; }
int pipelineDevice() { return g_pipelineDevice; }

bool bsa_rowReordering_device(const sparseMatrix::CSR<float>& matrix, const float alpha, const UIN block_size,
                              int device, std::vector<UIN>& reorderedRows, int& num_clusters,
                              float& reordering_time) {
    const auto t0 = Clock::now();
    std::vector<UIN> rows(matrix.row());
    uint32_t count = 0;
    int32_t clusters = 0;
    const int st = bsmr_cluster_rows(device, matrix.row(), matrix.col(), matrix.rowOffsets().data(),
                                     matrix.colIndices().data(), block_size == 0 ? 16 : block_size, alpha,
                                     rows.data(), &count, &clusters, nullptr);
    if (st != BSMR_OK) return false;
    rows.resize(count);
    reorderedRows.swap(rows);
    num_clusters = clusters;
    reordering_time = msSince(t0);
    return true;
}
