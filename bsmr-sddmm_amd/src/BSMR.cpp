// BSMR driver object, the RPHM block format, its structural validators and the
// reordering statistics.
//
// Behaviour: reference src/BSMR.cpp -- BSMR::BSMR/rowReordering/colReordering
// (:16-81), RPHM::RPHM (:83-265), check_* (:444-824, :932-953),
// evaluationReordering (:826-930) and the original-matrix block census (:955-994).
// The implementation replaces the reference's per-row / per-panel hash maps by
// one sort of (column, local row, CSR index) triples per panel and binary
// searches, and reformulates both statistics passes from
// O(panels x blocks x nnz_panel) to O(nnz log nnz).

#include <cstdlib>
#include <string>

#include <omp.h>
#include "util.hpp"

#include "BSMR.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <map>
#include <mutex>
#include <numeric>

#include "bsmr_hip.h"

namespace {
using Clock = std::chrono::steady_clock;

inline UIN ceilDiv(UIN a, UIN b) { return (a + b - 1) / b; }

struct PanelEntry {
    UIN col;
    UIN localRow;
    UIN csrIndex;
};

// All stored entries of one panel, ordered by (col, localRow).
void gatherPanel(const sparseMatrix::CSR<float>& m, const std::vector<UIN>& reorderedRows,
                 size_t panel, std::vector<PanelEntry>& out) {
    out.clear();
    const size_t first = panel * ROW_PANEL_SIZE;
    const size_t last = std::min(first + ROW_PANEL_SIZE, reorderedRows.size());
    for (size_t i = first; i < last; ++i) {
        const UIN row = reorderedRows[i];
        for (UIN e = m.rowOffsets()[row]; e < m.rowOffsets()[row + 1]; ++e)
            out.push_back(PanelEntry{m.colIndices()[e], static_cast<UIN>(i - first), e});
    }
    std::stable_sort(out.begin(), out.end(),
                     [](const PanelEntry& a, const PanelEntry& b) { return a.col < b.col; });
}

inline std::pair<const PanelEntry*, const PanelEntry*> entriesOfColumn(
    const std::vector<PanelEntry>& v, UIN col) {
    const auto lo = std::lower_bound(v.begin(), v.end(), col,
                                     [](const PanelEntry& a, UIN c) { return a.col < c; });
    auto hi = lo;
    while (hi != v.end() && hi->col == col) ++hi;
    return {v.data() + (lo - v.begin()), v.data() + (hi - v.begin())};
}
}  // namespace

// ---------------------------------------------------------------------------
// BSMR
// ---------------------------------------------------------------------------
BSMR::BSMR(const float similarityThreshold, const float blockDensityThreshold,
           const sparseMatrix::CSR<float>& matrix, const int numIterations) {
    rowReordering(similarityThreshold, matrix, numIterations);
    colReordering(blockDensityThreshold, matrix, reorderedRows_, numIterations);
}

void BSMR::rowReordering(const float similarityThreshold, const sparseMatrix::CSR<float>& matrix,
                         const int numIterations) {
    const UIN blockSize = calculateBlockSize(matrix);
    const int iters = std::max(1, numIterations);
    float total = 0.0f;
    for (int i = 0; i < iters; ++i) {
        float once = 0.0f;
        int device = clusteringDevice();
        if (device == -2) {  // automatic
            const char* env = std::getenv("BSMR_CLUSTER");
            const std::string choice = env ? env : "";
            int count = 0;
            const bool present = choice != "host" && bsmr_device_count(&count) == BSMR_OK && count > 0;
            UIN nonEmpty = 0;
            for (UIN r = 0; r < matrix.row(); ++r) nonEmpty += matrix.rowOffsets()[r + 1] > matrix.rowOffsets()[r];
            // long rows (dense device scan wins) and enough of them to repay ~8 us per speculative pass: nips-like
            // (1 500 rows of ~500 entries) clusters in 96 ms on the host against 138-185 ms on the device
            const bool longRows = nonEmpty >= 4096 && matrix.nnz() / nonEmpty >= 32;
            device = present && (choice == "device" || longRows) ? std::min(pipelineDevice(), count - 1) : -1;
        }
        if (device < 0 ||
            !bsa_rowReordering_device(matrix, similarityThreshold, blockSize, device, reorderedRows_, numClusters_, once))
            reorderedRows_ = bsa_rowReordering_host(matrix, similarityThreshold, blockSize, numClusters_, once);
        total += once;
    }
    rowReorderingTime_ = total / iters;
    numRowPanels_ = static_cast<int>(ceilDiv(static_cast<UIN>(reorderedRows_.size()), ROW_PANEL_SIZE));
}

void BSMR::colReordering(const float blockDensityThreshold, const sparseMatrix::CSR<float>& matrix,
                         const std::vector<UIN>& reorderedRows, const int numIterations) {
    if (!reorderedRows.empty()) {
        reorderedRows_ = reorderedRows;
        numRowPanels_ = static_cast<int>(ceilDiv(static_cast<UIN>(reorderedRows_.size()), ROW_PANEL_SIZE));
    }
    const int iters = std::max(1, numIterations);
    float total = 0.0f;
    int device = colReorderingDevice();
    if (device == -2) {   // automatic: large matrices on the pipeline's device
        const char* env = std::getenv("BSMR_COLREORDER");
        const std::string choice = env ? env : "";
        int count = 0;
        const bool present = choice != "host" && bsmr_device_count(&count) == BSMR_OK && count > 0;
        device = present && (choice == "device" || matrix.nnz() >= (4u << 20)) ? std::min(pipelineDevice(), count - 1) : -1;
    }
    for (int i = 0; i < iters; ++i) {
        float once = 0.0f;
        deviceRphm_ = DeviceRphmArrays{};
        if (device < 0 || !colReordering_device(matrix, reorderedRows_, blockDensityThreshold, device, denseCols_,
                                                denseColOffsets_, sparseCols_, sparseColOffsets_, sparseValueOffsets_,
                                                deviceRphm_, once)) {
            deviceRphm_ = DeviceRphmArrays{};
            colReordering_cpu(matrix, static_cast<UIN>(numRowPanels_), reorderedRows_, blockDensityThreshold,
                              denseCols_, denseColOffsets_, sparseCols_, sparseColOffsets_,
                              sparseValueOffsets_, once);
        }
        total += once;
    }
    colReorderingTime_ = total / iters;
}

// ---------------------------------------------------------------------------
// RPHM
// ---------------------------------------------------------------------------
RPHM::RPHM(const sparseMatrix::CSR<float>& matrix, const BSMR& bsmr, int device) {
    const auto t0 = Clock::now();
    numRowPanels_ = static_cast<UIN>(bsmr.numRowPanels());
    numCols_ = matrix.col();
    reorderedRows_ = bsmr.reorderedRows();
    denseCols_ = bsmr.denseCols();
    sparseValueOffsets_ = bsmr.sparseValueOffsets();
    const auto& dco = bsmr.denseColOffsets();
    const auto& sco = bsmr.sparseColOffsets();

    // block offsets + the reference's per-thread-block work lists
    blockOffsets_.assign(static_cast<size_t>(numRowPanels_) + 1, 0);
    for (UIN p = 0; p < numRowPanels_; ++p) {
        const UIN blocks = ceilDiv(dco[p + 1] - dco[p], BLOCK_COL_SIZE);
        blockOffsets_[p + 1] = blockOffsets_[p] + blocks;
        maxNumDenseColBlocksInRowPanel_ = std::max(maxNumDenseColBlocksInRowPanel_, blocks);
        const UIN tbs = ceilDiv(blocks, each_thread_block_counts_the_number_Of_dense_blocks);
        for (UIN i = 0; i < tbs; ++i) {
            denseRowPanelIds_.push_back(p);
            denseColBlockIters_.push_back(dco[p] / BLOCK_COL_SIZE +
                                          i * each_thread_block_counts_the_number_Of_dense_blocks);
        }
        numDenseThreadBlocks_ += tbs;

        const UIN residue = sparseValueOffsets_[p + 1] - sparseValueOffsets_[p];
        const UIN stbs = ceilDiv(residue, sddmm_sparse_block_each_thread_block_counts_the_number_Of_data);
        maxNumSparseColBlocksInRowPanel_ = std::max(maxNumSparseColBlocksInRowPanel_, stbs);
        for (UIN i = 0; i < stbs; ++i) {
            sparseRowPanelIds_.push_back(p);
            sparseColBlockIters_.push_back(i * sddmm_sparse_block_each_thread_block_counts_the_number_Of_data);
        }
        numSparseThreadBlocks_ += stbs;
    }

    const size_t numBlocks = blockOffsets_.back();
    const size_t numSparse = sparseValueOffsets_.empty() ? 0 : sparseValueOffsets_.back();
    const BSMR::DeviceRphmArrays& fromDevice = bsmr.deviceRphm();
    const bool useDeviceArrays = fromDevice.valid && fromDevice.handle && fromDevice.blockOffsets == blockOffsets_ &&
                                 fromDevice.numBlocks == numBlocks && fromDevice.numSparseEntries == numSparse;
    if (useDeviceArrays) {   // the column reordering ran on the device and produced the index arrays in the same pass
        onDevice_ = fromDevice.handle;
        if (device < 0) fetchBigArrays();
    } else {
    blockValues_.assign(numBlocks * BLOCK_SIZE, NULL_VALUE);
    sparseValues_.resize(numSparse);
    sparseRelativeRows_.resize(numSparse);
    sparseColIndices_.resize(numSparse);

#pragma omp parallel num_threads(util::hostThreads(omp_get_max_threads()))
    {
        std::vector<PanelEntry> entries;
#pragma omp for schedule(dynamic, 8)
        for (long long p = 0; p < static_cast<long long>(numRowPanels_); ++p) {
            gatherPanel(matrix, reorderedRows_, static_cast<size_t>(p), entries);
            // dense part: slot t of the panel's dense column list -> block t/16, column t%16
            const size_t tileBase = static_cast<size_t>(blockOffsets_[p]) * BLOCK_SIZE;
            for (UIN t = 0; t < dco[p + 1] - dco[p]; ++t) {
                const UIN col = denseCols_[dco[p] + t];
                const auto range = entriesOfColumn(entries, col);
                for (const PanelEntry* it = range.first; it != range.second; ++it)
                    blockValues_[tileBase + static_cast<size_t>(t / BLOCK_COL_SIZE) * BLOCK_SIZE +
                                 it->localRow * BLOCK_COL_SIZE + t % BLOCK_COL_SIZE] = it->csrIndex;
            }
            // sparse part: columns in list order, rows ascending inside a column
            size_t k = sparseValueOffsets_[p];
            for (UIN s = sco[p]; s < sco[p + 1]; ++s) {
                const UIN col = bsmr.sparseCols()[s];
                const auto range = entriesOfColumn(entries, col);
                for (const PanelEntry* it = range.first; it != range.second; ++it, ++k) {
                    sparseRelativeRows_[k] = it->localRow;
                    sparseValues_[k] = it->csrIndex;
                    sparseColIndices_[k] = col;
                }
            }
        }
    }
    }
    reorderingTime_ = std::chrono::duration<float, std::milli>(Clock::now() - t0).count();

    device_ = device < 0 ? 0 : device;
    if (device >= 0) {
        bsmr_rphm_desc d{};
        d.M = matrix.row();
        d.N = matrix.col();
        d.nnz = matrix.nnz();
        d.num_row_panels = numRowPanels_;
        d.num_nonzero_rows = static_cast<uint32_t>(reorderedRows_.size());
        d.reordered_rows = reorderedRows_.data();
        d.dense_cols = denseCols_.data();
        d.block_offsets = blockOffsets_.data();
        d.block_values = blockValues_.data();
        d.sparse_value_offsets = sparseValueOffsets_.data();
        d.sparse_values = sparseValues_.data();
        d.sparse_relative_rows = sparseRelativeRows_.data();
        d.sparse_col_indices = sparseColIndices_.data();
        int where = -1;
        if (onDevice_ && bsmr_col_reorder_device(onDevice_.get(), &where) == BSMR_OK && where == device) {
            // the plan from the arrays where the column reordering left them
            planStatus_ = bsmr_plan_create_from_colreorder(&plan_, onDevice_.get(), d.M, d.N, d.nnz, d.reordered_rows, d.num_nonzero_rows, nullptr);
            // (the device arrays stay behind onDevice_ until somebody asks for the host copies - check_rphm, the density
            // statistics - or this object goes; the BSMR object that produced them drops its own reference when it is re-split)
        } else {
            planStatus_ = fetchBigArrays();
            d.block_values = blockValues_.data();
            d.sparse_values = sparseValues_.data();
            d.sparse_relative_rows = sparseRelativeRows_.data();
            d.sparse_col_indices = sparseColIndices_.data();
            if (planStatus_ == BSMR_OK) planStatus_ = bsmr_plan_create(&plan_, device, &d);
        }
        if (planStatus_ != BSMR_OK) {
            fprintf(stderr, "RPHM: device plan creation failed: %s (%s)\n", bsmr_strerror(planStatus_),
                    bsmr_last_hip_error());
            plan_ = nullptr;
        }
    }
}

int RPHM::fetchBigArrays() const {
    // (called from const getters, possibly from several threads: one fetch, its status kept)
    std::lock_guard<std::mutex> guard(fetchLock_);
    if (!onDevice_) return fetchStatus_;
    bsmr_colreorder_sizes sz{};
    int st = bsmr_col_reorder_sizes(onDevice_.get(), &sz);
    if (st == BSMR_OK) {
        try {
            blockValues_.resize(sz.num_blocks * BLOCK_SIZE);
            sparseValues_.resize(sz.num_sparse_entries);
            sparseRelativeRows_.resize(sz.num_sparse_entries);
            sparseColIndices_.resize(sz.num_sparse_entries);
        } catch (const std::bad_alloc&) {
            st = BSMR_ERR_OOM;
        }
    }
    if (st == BSMR_OK)
        st = bsmr_col_reorder_fetch(onDevice_.get(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, blockValues_.data(), sparseValues_.data(),
                                    sparseRelativeRows_.data(), sparseColIndices_.data());
    if (st != BSMR_OK) {
        // nothing half-filled stays behind (zero-filled arrays would read as "every dense cell is CSR entry 0"), the handle
        // is kept: a later call may try again
        fprintf(stderr, "RPHM: fetching the device arrays failed: %s (%s)\n", bsmr_strerror(st), bsmr_last_hip_error());
        blockValues_.clear();
        sparseValues_.clear();
        sparseRelativeRows_.clear();
        sparseColIndices_.clear();
        fetchStatus_ = st;
        return st;
    }
    onDevice_.reset();   // the device copy is not read again: the plan was built from it or will be built from the host arrays
    fetchStatus_ = BSMR_OK;
    return BSMR_OK;
}

void RPHM::release() {
    if (plan_) bsmr_plan_destroy(plan_);
    plan_ = nullptr;
}

RPHM::~RPHM() { release(); }

RPHM::RPHM(RPHM&& o) noexcept { *this = std::move(o); }

RPHM& RPHM::operator=(RPHM&& o) noexcept {
    if (this == &o) return *this;
    release();
    numRowPanels_ = o.numRowPanels_;
    maxNumDenseColBlocksInRowPanel_ = o.maxNumDenseColBlocksInRowPanel_;
    maxNumSparseColBlocksInRowPanel_ = o.maxNumSparseColBlocksInRowPanel_;
    numDenseThreadBlocks_ = o.numDenseThreadBlocks_;
    numSparseThreadBlocks_ = o.numSparseThreadBlocks_;
    numCols_ = o.numCols_;
    reorderedRows_ = std::move(o.reorderedRows_);
    denseCols_ = std::move(o.denseCols_);
    blockOffsets_ = std::move(o.blockOffsets_);
    blockValues_ = std::move(o.blockValues_);
    onDevice_ = std::move(o.onDevice_);
    sparseValueOffsets_ = std::move(o.sparseValueOffsets_);
    sparseValues_ = std::move(o.sparseValues_);
    sparseRelativeRows_ = std::move(o.sparseRelativeRows_);
    sparseColIndices_ = std::move(o.sparseColIndices_);
    denseRowPanelIds_ = std::move(o.denseRowPanelIds_);
    denseColBlockIters_ = std::move(o.denseColBlockIters_);
    sparseRowPanelIds_ = std::move(o.sparseRowPanelIds_);
    sparseColBlockIters_ = std::move(o.sparseColBlockIters_);
    reorderingTime_ = o.reorderingTime_;
    plan_ = o.plan_;
    planStatus_ = o.planStatus_;
    device_ = o.device_;
    o.plan_ = nullptr;
    return *this;
}

UIN RPHM::getNumSparseBlocks() const {
    if (sparseValueOffsets_.empty()) return 0;
    return static_cast<UIN>(sparseValueOffsets_.back() /
                            static_cast<float>(sddmm_sparse_block_each_thread_block_counts_the_number_Of_data));
}

UIN RPHM::calculateRowPanelIdByBlockValuesIndex(UIN blockValueIndex) const {
    const UIN block = blockValueIndex / BLOCK_SIZE;
    const auto it = std::upper_bound(blockOffsets_.begin(), blockOffsets_.end(), block);
    return static_cast<UIN>(it - blockOffsets_.begin()) - 1;
}

std::pair<UIN, UIN> RPHM::calculateLocalRowColByBlockValueIndex(UIN blockValueIndex) const {
    const UIN inBlock = blockValueIndex % BLOCK_SIZE;
    return {inBlock / BLOCK_COL_SIZE, inBlock % BLOCK_COL_SIZE};
}

UIN RPHM::calculateColBlockIdByBlockValueIndex(UIN blockValueIndex) const {
    const UIN panel = calculateRowPanelIdByBlockValuesIndex(blockValueIndex);
    return blockValueIndex / BLOCK_SIZE - blockOffsets_[panel];
}

std::pair<UIN, UIN> RPHM::calculateRowColByBlockValueIndex(UIN blockValueIndex) const {
    const UIN panel = calculateRowPanelIdByBlockValuesIndex(blockValueIndex);
    const auto [localRow, localCol] = calculateLocalRowColByBlockValueIndex(blockValueIndex);
    const size_t rowSlot = static_cast<size_t>(panel) * ROW_PANEL_SIZE + localRow;
    const UIN row = rowSlot < reorderedRows_.size() ? reorderedRows_[rowSlot] : NULL_VALUE;
    const UIN col = denseCols_[static_cast<size_t>(blockValueIndex / BLOCK_SIZE) * BLOCK_COL_SIZE + localCol];
    return {row, col};
}

namespace {
template <typename F>
void forEachBlockDensity(const std::vector<UIN>& blockValues, F f) {
    for (size_t b = 0; b * BLOCK_SIZE < blockValues.size(); ++b) {
        UIN n = 0;
        for (UIN i = 0; i < BLOCK_SIZE; ++i) n += blockValues[b * BLOCK_SIZE + i] != NULL_VALUE;
        f(static_cast<float>(n) / BLOCK_SIZE);
    }
}
}  // namespace

float RPHM::calculateDenseBlockAverageDensity() const {
    float total = 0.0f;
    size_t blocks = 0;
    forEachBlockDensity(blockValues(), [&](float d) { total += d; ++blocks; });
    return blocks ? total / blocks : 0.0f;
}

std::pair<float, float> RPHM::calculateMaxMinDensity() const {
    float mx = 0.0f, mn = 1.0f;
    bool any = false;
    forEachBlockDensity(blockValues(), [&](float d) { mx = std::max(mx, d); mn = std::min(mn, d); any = true; });
    return any ? std::make_pair(mx, mn) : std::make_pair(0.0f, 0.0f);
}

std::pair<float, UIN> RPHM::calculateDensityMode() const {
    std::map<UIN, UIN> histogram;  // nnz in block -> number of blocks
    forEachBlockDensity(blockValues(), [&](float d) { ++histogram[static_cast<UIN>(std::lround(d * BLOCK_SIZE))]; });
    UIN bestNnz = 0, bestFreq = 0;
    for (const auto& kv : histogram)
        if (kv.second > bestFreq) { bestNnz = kv.first; bestFreq = kv.second; }
    return {static_cast<float>(bestNnz) / BLOCK_SIZE, bestFreq};
}

// ---------------------------------------------------------------------------
// validators
// ---------------------------------------------------------------------------
namespace {

// reorderedRows must list every non-empty row exactly once and nothing else.
bool check_rowReordering(const sparseMatrix::CSR<float>& m, const RPHM& rphm) {
    std::vector<uint8_t> seen(m.row(), 0);
    for (const UIN r : rphm.reorderedRows()) {
        if (r >= m.row() || seen[r]) {
            fprintf(stderr, "Error! row %u is out of range or repeated in reorderedRows\n", r);
            return false;
        }
        if (m.rowOffsets()[r + 1] == m.rowOffsets()[r]) {
            fprintf(stderr, "Error! empty row %u appears in reorderedRows\n", r);
            return false;
        }
        seen[r] = 1;
    }
    for (UIN r = 0; r < m.row(); ++r)
        if (!seen[r] && m.rowOffsets()[r + 1] > m.rowOffsets()[r]) {
            fprintf(stderr, "Error! non-empty row %u is missing from reorderedRows\n", r);
            return false;
        }
    return true;
}

// Per panel: dense and sparse column lists are disjoint, together they are
// exactly the panel's non-empty columns (plus sentinels), counts are
// non-increasing along the concatenated list, dense blocks meet the threshold
// and no sparse block does.
bool check_colReordering(const sparseMatrix::CSR<float>& m, const BSMR& bsmr, const float delta) {
    const UIN threshold = static_cast<UIN>(std::ceil(delta * BLOCK_SIZE));
    bool ok = true;
#pragma omp parallel for schedule(dynamic, 16) num_threads(util::hostThreads(omp_get_max_threads()))
    for (long long p = 0; p < bsmr.numRowPanels(); ++p) {
        if (!ok) continue;
        std::vector<PanelEntry> entries;
        gatherPanel(m, bsmr.reorderedRows(), static_cast<size_t>(p), entries);
        std::vector<UIN> list(bsmr.denseCols().begin() + bsmr.denseColOffsets()[p],
                              bsmr.denseCols().begin() + bsmr.denseColOffsets()[p + 1]);
        const size_t numDense = list.size();
        list.insert(list.end(), bsmr.sparseCols().begin() + bsmr.sparseColOffsets()[p],
                    bsmr.sparseCols().begin() + bsmr.sparseColOffsets()[p + 1]);
        bool good = list.size() % BLOCK_COL_SIZE == 0 && numDense % BLOCK_COL_SIZE == 0;
        std::vector<UIN> counts(list.size(), 0);
        size_t covered = 0;
        std::vector<UIN> real;
        for (size_t i = 0; i < list.size() && good; ++i) {
            if (list[i] == m.col()) continue;  // padding sentinel
            const auto range = entriesOfColumn(entries, list[i]);
            counts[i] = static_cast<UIN>(range.second - range.first);
            if (counts[i] == 0) good = false;  // a listed column must be non-empty
            covered += counts[i];
            real.push_back(list[i]);
        }
        std::sort(real.begin(), real.end());
        if (std::adjacent_find(real.begin(), real.end()) != real.end()) good = false;
        if (covered != entries.size()) good = false;
        for (size_t i = 1; i < counts.size() && good; ++i)
            if (counts[i] > counts[i - 1]) good = false;
        for (size_t b = 0; b < counts.size() && good; b += BLOCK_COL_SIZE) {
            const UIN inBlock = std::accumulate(counts.begin() + b, counts.begin() + b + BLOCK_COL_SIZE, 0u);
            if ((b < numDense) != (inBlock >= threshold)) good = false;
        }
        UIN residue = 0;
        for (size_t i = numDense; i < counts.size(); ++i) residue += counts[i];
        if (residue != bsmr.sparseValueOffsets()[p + 1] - bsmr.sparseValueOffsets()[p]) good = false;
        if (!good) {
#pragma omp critical
            {
                fprintf(stderr, "Error! column reordering of row panel %lld is inconsistent\n", p);
                ok = false;
            }
        }
    }
    return ok;
}

// Every CSR index appears exactly once across blockValues and sparseValues, at a
// position whose (row, column) is the entry's own.
bool check_rphmCoverage(const sparseMatrix::CSR<float>& m, const RPHM& rphm) {
    std::vector<uint8_t> hit(m.nnz(), 0);
    std::vector<UIN> rowOf(m.nnz());
    for (UIN r = 0; r < m.row(); ++r)
        for (UIN e = m.rowOffsets()[r]; e < m.rowOffsets()[r + 1]; ++e) rowOf[e] = r;
    const auto& bv = rphm.blockValues();
    for (size_t i = 0; i < bv.size(); ++i) {
        const UIN e = bv[i];
        if (e == NULL_VALUE) continue;
        if (e >= m.nnz() || hit[e]) {
            fprintf(stderr, "Error! blockValues[%zu] = %u is out of range or repeated\n", i, e);
            return false;
        }
        const auto [row, col] = rphm.calculateRowColByBlockValueIndex(static_cast<UIN>(i));
        if (rowOf[e] != row || m.colIndices()[e] != col) {
            fprintf(stderr, "Error! blockValues[%zu] = %u sits at (%u,%u) but is entry (%u,%u)\n", i, e,
                    row, col, rowOf[e], m.colIndices()[e]);
            return false;
        }
        hit[e] = 1;
    }
    for (UIN p = 0; p < rphm.numRowPanels(); ++p) {
        for (UIN k = rphm.sparseValueOffsets()[p]; k < rphm.sparseValueOffsets()[p + 1]; ++k) {
            const UIN e = rphm.sparseValues()[k];
            if (e >= m.nnz() || hit[e]) {
                fprintf(stderr, "Error! sparseValues[%u] = %u is out of range or repeated\n", k, e);
                return false;
            }
            const size_t slot = static_cast<size_t>(p) * ROW_PANEL_SIZE + rphm.sparseRelativeRows()[k];
            if (slot >= rphm.reorderedRows().size() || rphm.reorderedRows()[slot] != rowOf[e] ||
                rphm.sparseColIndices()[k] != m.colIndices()[e]) {
                fprintf(stderr, "Error! sparse entry %u does not match CSR entry %u\n", k, e);
                return false;
            }
            hit[e] = 1;
        }
    }
    for (UIN e = 0; e < m.nnz(); ++e)
        if (!hit[e]) {
            fprintf(stderr, "Error! CSR entry %u is covered by neither part\n", e);
            return false;
        }
    return true;
}
}  // namespace

bool check_rphm(const sparseMatrix::CSR<float>& matrix, const BSMR& bsmr, const RPHM& rphm,
                const float denseColSegmentThreshold) {
    bool ok = true;
    if (!check_rowReordering(matrix, rphm)) {
        std::cerr << "Error! The row reordering is incorrect!" << std::endl;
        ok = false;
    }
    if (!check_colReordering(matrix, bsmr, denseColSegmentThreshold)) {
        std::cerr << "Error! The col reordering is incorrect!" << std::endl;
        ok = false;
    }
    if (!check_rphmCoverage(matrix, rphm)) {
        std::cerr << "Error! The rphm is incorrect!" << std::endl;
        ok = false;
    }
    return ok;
}

// ---------------------------------------------------------------------------
// statistics
// ---------------------------------------------------------------------------
std::pair<UIN, float> calculateNumDenseBlocksAndAverageDensityInOriginalMatrix(
    const float densityThreshold, const sparseMatrix::CSR<float>& matrix) {
    const UIN numPanels = ceilDiv(matrix.row(), ROW_PANEL_SIZE);
    // per panel: (column block id, nnz) for the non-empty blocks, ascending
    std::vector<std::vector<std::pair<UIN, UIN>>> census(numPanels);
#pragma omp parallel num_threads(util::hostThreads(omp_get_max_threads()))
    {
        std::vector<UIN> ids;
#pragma omp for schedule(dynamic, 64)
        for (long long p = 0; p < static_cast<long long>(numPanels); ++p) {
            const UIN r0 = static_cast<UIN>(p) * ROW_PANEL_SIZE;
            const UIN r1 = std::min(r0 + ROW_PANEL_SIZE, matrix.row());
            ids.clear();
            for (UIN e = matrix.rowOffsets()[r0]; e < matrix.rowOffsets()[r1]; ++e)
                ids.push_back(matrix.colIndices()[e] / BLOCK_COL_SIZE);
            std::sort(ids.begin(), ids.end());
            for (size_t i = 0; i < ids.size();) {
                size_t j = i;
                while (j < ids.size() && ids[j] == ids[i]) ++j;
                census[p].emplace_back(ids[i], static_cast<UIN>(j - i));
                i = j;
            }
        }
    }
    UIN numDenseBlocks = 0;
    float totalDensity = 0.0f;
    for (UIN p = 0; p < numPanels; ++p) {
        const UIN rowsHere = std::min(p * ROW_PANEL_SIZE + ROW_PANEL_SIZE, matrix.row()) - p * ROW_PANEL_SIZE;
        for (const auto& [block, nnz] : census[p]) {
            const UIN colsHere = std::min(block * BLOCK_COL_SIZE + BLOCK_COL_SIZE, matrix.col()) -
                                 block * BLOCK_COL_SIZE;
            const float density = static_cast<float>(nnz) / static_cast<float>(rowsHere * colsHere);
            if (density >= densityThreshold) {
                totalDensity += density;
                ++numDenseBlocks;
            }
        }
    }
    return {numDenseBlocks, numDenseBlocks ? totalDensity / numDenseBlocks : 0.0f};
}

void evaluationReordering(const sparseMatrix::CSR<float>& matrix, const BSMR& bsmr, Logger& logger) {
    const int numPanels = bsmr.numRowPanels();
    std::vector<std::vector<UIN>> nnzPerDenseBlock(numPanels);
    std::vector<UIN> residue(numPanels, 0);
#pragma omp parallel num_threads(util::hostThreads(omp_get_max_threads()))
    {
        std::vector<PanelEntry> entries;
#pragma omp for schedule(dynamic, 16)
        for (int p = 0; p < numPanels; ++p) {
            gatherPanel(matrix, bsmr.reorderedRows(), static_cast<size_t>(p), entries);
            const UIN d0 = bsmr.denseColOffsets()[p], d1 = bsmr.denseColOffsets()[p + 1];
            nnzPerDenseBlock[p].assign(ceilDiv(d1 - d0, BLOCK_COL_SIZE), 0);
            for (UIN t = 0; t < d1 - d0; ++t) {
                const auto range = entriesOfColumn(entries, bsmr.denseCols()[d0 + t]);
                nnzPerDenseBlock[p][t / BLOCK_COL_SIZE] += static_cast<UIN>(range.second - range.first);
            }
            for (UIN s = bsmr.sparseColOffsets()[p]; s < bsmr.sparseColOffsets()[p + 1]; ++s) {
                const auto range = entriesOfColumn(entries, bsmr.sparseCols()[s]);
                residue[p] += static_cast<UIN>(range.second - range.first);
            }
        }
    }
    int numDenseBlocks = 0, numDenseThreadBlocks = 0, numSparseThreadBlocks = 0, numSparseData = 0;
    float totalDensity = 0.0f;
    for (int p = 0; p < numPanels; ++p) {
        numDenseThreadBlocks += ceilDiv(static_cast<UIN>(nnzPerDenseBlock[p].size()),
                                        each_thread_block_counts_the_number_Of_dense_blocks);
        numSparseThreadBlocks += ceilDiv(bsmr.sparseValueOffsets()[p + 1] - bsmr.sparseValueOffsets()[p],
                                         sddmm_sparse_block_each_thread_block_counts_the_number_Of_data);
        numSparseData += residue[p];
        for (const UIN n : nnzPerDenseBlock[p]) {
            if (n == 0) continue;
            const float density = static_cast<float>(n) / static_cast<float>(BLOCK_SIZE);
            totalDensity += density;
            if (density >= logger.delta_) ++numDenseBlocks;
        }
    }
    const auto [origBlocks, origDensity] =
        calculateNumDenseBlocksAndAverageDensityInOriginalMatrix(logger.delta_, matrix);
    logger.numDenseBlock_ = numDenseBlocks;
    const float avg = totalDensity / static_cast<float>(numDenseBlocks);  // inf/nan when no block
    logger.averageDensity_ = avg > 0 ? avg : 0.0f;
    logger.numDenseThreadBlocks_ = numDenseThreadBlocks;
    logger.numSparseThreadBlocks_ = numSparseThreadBlocks;
    logger.originalNumDenseBlock_ = static_cast<int>(origBlocks);
    logger.originalAverageDensity_ = origDensity;
    logger.numSparseData_ = numSparseData;
    logger.numDenseData_ = static_cast<int>(matrix.nnz()) - numSparseData;
}
