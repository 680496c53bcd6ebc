#pragma once
// Host side of the BSMR pipeline: row clustering at threshold alpha, per-panel
// column reordering, dense/sparse block split at threshold delta, and the RPHM
// block format.  Public interface of the reference's include/BSMR.hpp:21-245;
// the implementation (src/BSMR.cpp, src/rowReordering.cpp, src/colReordering.cpp)
// is C++17/OpenMP written for this engine.
//
// RPHM differs from the reference in ownership only: the reference keeps twelve
// dev::vector<UIN> (device copies); here the index arrays stay host-visible
// std::vectors and the device-resident form lives in an opaque bsmr_plan made
// through the C ABI (include/bsmr_hip.h).

#include <string>
#include <utility>
#include <memory>
#include <mutex>
#include <vector>

struct bsmr_colreorder;   // include/bsmr_hip.h: a column-reordering result on the device

#include "Logger.hpp"
#include "Matrix.hpp"

struct bsmr_plan;

constexpr UIN ROW_PANEL_SIZE = MFMA_M;
constexpr UIN BLOCK_COL_SIZE = MFMA_N;
constexpr UIN BLOCK_SIZE = ROW_PANEL_SIZE * BLOCK_COL_SIZE;

// Launch-geometry constants of the reference's kernels (include/sddmmKernel.cuh:11-17).
// They only feed the reference-compatible work lists / log fields of RPHM and
// evaluationReordering(); the HIP kernels use their own geometry.
constexpr int each_thread_block_counts_the_number_Of_dense_blocks = 4;
constexpr int sddmm_sparse_block_number_of_thread_per_thread_block = 256;
constexpr int sddmm_sparse_block_each_thread_block_counts_the_number_Of_data =
    sddmm_sparse_block_number_of_thread_per_thread_block / 2;

class BSMR {
public:
    BSMR() = default;

    BSMR(const float similarityThreshold, const float blockDensityThreshold,
         const sparseMatrix::CSR<float>& matrix, const int numIterations = 1);

    void rowReordering(const float similarityThreshold, const sparseMatrix::CSR<float>& matrix,
                       const int numIterations = 1);

    void colReordering(const float blockDensityThreshold, const sparseMatrix::CSR<float>& matrix,
                       const std::vector<UIN>& reorderedRows = std::vector<UIN>(),
                       const int numIterations = 1);

    int numRowPanels() const { return numRowPanels_; }
    const std::vector<UIN>& reorderedRows() const { return reorderedRows_; }
    const std::vector<UIN>& denseCols() const { return denseCols_; }
    const std::vector<UIN>& denseColOffsets() const { return denseColOffsets_; }
    const std::vector<UIN>& sparseCols() const { return sparseCols_; }
    const std::vector<UIN>& sparseColOffsets() const { return sparseColOffsets_; }
    const std::vector<UIN>& sparseValueOffsets() const { return sparseValueOffsets_; }
    int numClusters() const { return numClusters_; }
    float rowReorderingTime() const { return rowReorderingTime_; }
    float colReorderingTime() const { return colReorderingTime_; }
    float reorderingTime() const { return rowReorderingTime_ + colReorderingTime_; }

    // Index arrays of the RPHM when the column reordering ran on the device (bsmr_col_reorder computes them in the
    // same pass; RPHM::RPHM then takes them instead of rebuilding them on the host).  Empty otherwise.
    // (round 3: the four big arrays - block values and the residue's three - stay on the device behind `handle`; RPHM
    // builds its plan from them there, bsmr_plan_create_from_colreorder, and fetches them when somebody asks)
    struct DeviceRphmArrays {
        bool valid = false;
        std::shared_ptr<bsmr_colreorder> handle;
        std::vector<UIN> blockOffsets;
        uint64_t numBlocks = 0, numSparseEntries = 0;
        float deviceMs = 0.0f;
    };
    const DeviceRphmArrays& deviceRphm() const { return deviceRphm_; }

private:
    DeviceRphmArrays deviceRphm_;
    int numRowPanels_ = 0;
    std::vector<UIN> reorderedRows_;
    std::vector<UIN> denseCols_;
    std::vector<UIN> denseColOffsets_;
    std::vector<UIN> sparseCols_;
    std::vector<UIN> sparseColOffsets_;
    std::vector<UIN> sparseValueOffsets_;
    int numClusters_ = 1;
    float rowReorderingTime_ = 0.0f;
    float colReorderingTime_ = 0.0f;
};

// Row-Panel Hybrid Matrix: dense 16x16 blocks as index tiles (BELL-like) plus
// the sparse residue as COO relative to the row panel.
class RPHM {
public:
    RPHM() = default;
    // device < 0: build the host arrays only (no GPU touched).
    RPHM(const sparseMatrix::CSR<float>& matrix, const BSMR& bsmr, int device = 0);
    ~RPHM();
    RPHM(const RPHM&) = delete;
    RPHM& operator=(const RPHM&) = delete;
    RPHM(RPHM&& o) noexcept;
    RPHM& operator=(RPHM&& o) noexcept;

    UIN numRowPanels() const { return numRowPanels_; }
    UIN maxNumDenseColBlocksInRowPanel() const { return maxNumDenseColBlocksInRowPanel_; }
    UIN maxNumSparseColBlocksInRowPanel() const { return maxNumSparseColBlocksInRowPanel_; }
    UIN numDenseThreadBlocks() const { return numDenseThreadBlocks_; }
    UIN numSparseThreadBlocks() const { return numSparseThreadBlocks_; }
    const std::vector<UIN>& reorderedRows() const { return reorderedRows_; }
    const std::vector<UIN>& denseCols() const { return denseCols_; }
    const std::vector<UIN>& blockValues() const { fetchBigArrays(); return blockValues_; }
    const std::vector<UIN>& blockOffsets() const { return blockOffsets_; }
    const std::vector<UIN>& sparseValueOffsets() const { return sparseValueOffsets_; }
    const std::vector<UIN>& sparseValues() const { fetchBigArrays(); return sparseValues_; }
    const std::vector<UIN>& sparseRelativeRows() const { fetchBigArrays(); return sparseRelativeRows_; }
    const std::vector<UIN>& sparseColIndices() const { fetchBigArrays(); return sparseColIndices_; }
    const std::vector<UIN>& denseRowPanelIds() const { return denseRowPanelIds_; }
    const std::vector<UIN>& denseColBlockIters() const { return denseColBlockIters_; }
    const std::vector<UIN>& sparseRowPanelIds() const { return sparseRowPanelIds_; }
    const std::vector<UIN>& sparseColBlockIters() const { return sparseColBlockIters_; }

    float time() const { return reorderingTime_; }

    // Device-resident form (nullptr when built with device < 0 or when plan
    // creation failed; planStatus() then holds the bsmr_hip.h status code).
    bsmr_plan* plan() const { return plan_; }
    int device() const { return device_; }
    int planStatus() const { return planStatus_; }

    UIN calculateRowPanelIdByBlockValuesIndex(UIN blockValueIndex) const;
    std::pair<UIN, UIN> calculateLocalRowColByBlockValueIndex(UIN blockValueIndex) const;
    std::pair<UIN, UIN> calculateRowColByBlockValueIndex(UIN blockValueIndex) const;
    UIN calculateColBlockIdByBlockValueIndex(UIN blockValueIndex) const;

    UIN getNumDenseBlocks() const { return blockOffsets_.empty() ? 0 : blockOffsets_.back(); }
    UIN getNumSparseBlocks() const;
    float calculateDenseBlockAverageDensity() const;
    std::pair<float, float> calculateMaxMinDensity() const;
    std::pair<float, UIN> calculateDensityMode() const;

private:
    void release();
    // the column reordering ran on the device: block values and the residue arrays are copied to the host on first use
    int fetchBigArrays() const;   // bsmr_hip.h status; the arrays are empty when it failed
    mutable std::shared_ptr<bsmr_colreorder> onDevice_;
    mutable std::mutex fetchLock_;
    mutable int fetchStatus_ = 0;

    UIN numRowPanels_ = 0;
    UIN maxNumDenseColBlocksInRowPanel_ = 0;
    UIN maxNumSparseColBlocksInRowPanel_ = 0;
    UIN numDenseThreadBlocks_ = 0;
    UIN numSparseThreadBlocks_ = 0;
    UIN numCols_ = 0;

    std::vector<UIN> reorderedRows_;
    std::vector<UIN> denseCols_;
    std::vector<UIN> blockOffsets_;
    mutable std::vector<UIN> blockValues_;
    std::vector<UIN> sparseValueOffsets_;
    mutable std::vector<UIN> sparseValues_;
    mutable std::vector<UIN> sparseRelativeRows_;
    mutable std::vector<UIN> sparseColIndices_;
    std::vector<UIN> denseRowPanelIds_;
    std::vector<UIN> denseColBlockIters_;
    std::vector<UIN> sparseRowPanelIds_;
    std::vector<UIN> sparseColBlockIters_;

    float reorderingTime_ = 0.0f;
    bsmr_plan* plan_ = nullptr;
    int planStatus_ = 0;
    int device_ = 0;
};

// Identity order over the non-empty rows (reference src/rowReordering.cu:15-46).
void noReorderRow(const sparseMatrix::CSR<float>& matrix, std::vector<UIN>& reorderedRows, float& time);

// Histogram bin width of the row clustering (reference src/rowReordering.cu:1009-1025):
// max(16, ceil(rows^2*4 / (freeDeviceBytes/2)), ceil(cols*4 / 24576)).  The free
// memory figure comes from the device when one is present, else 288 GB is assumed.
UIN calculateBlockSize(const sparseMatrix::CSR<float>& matrix);
UIN calculateBlockSize(const sparseMatrix::CSR<float>& matrix, size_t freeDeviceBytes);

// BSA row clustering (reference bsa_rowReordering_gpu, src/rowReordering.cu:1027-1095,
// whose mutex pipeline is equivalent to finishing cluster c before c+1 starts).
// Host implementation over sparse histograms with an inverted bin index.
// Where BSMR::rowReordering runs the clustering.  The reference always uses the GPU
// (bsa_rowReordering_gpu); both implementations here give the same row order and cluster
// count, so this is a matter of speed only.
//   -2 (default): the pipeline's device when one is present, the rows are long enough for the dense
//                 device scan to win (>= 32 stored entries per non-empty row on average) and there
//                 are at least 4096 of them (
//                 measured on MI355X: mycielskian15 0.33 s on the device against 5-12 s on
//                 the host, wathen100 0.79 s against 0.24 s); environment BSMR_CLUSTER =
//                 host | device overrides the rule
//   -1          : host
//   >= 0        : that device (falls back to the host when the table does not fit)
void setClusteringDevice(int device);
int clusteringDevice();
// The device the automatic rule above, calculateBlockSize's free-memory query and the launchers' synchronisation use
// (per thread, default 0): a process that drives several GPUs - one pipeline per rank or per shard - sets it to the
// pipeline's device, so that no rank clusters on, allocates on or leaves the process on GPU 0.
void setPipelineDevice(int device);
int pipelineDevice();
// Where BSMR::colReordering runs: -2 (default) = on the pipeline's device when one is present and S stores at least
// 4 M entries (BSMR_COLREORDER = host | device overrides), -1 = host, >= 0 = that device.  Both give the same arrays.
void setColReorderingDevice(int device);
int colReorderingDevice();
bool colReordering_device(const sparseMatrix::CSR<float>& matrix, const std::vector<UIN>& reorderedRows,
                          const float blockDensityThreshold, int device, std::vector<UIN>& denseCols,
                          std::vector<UIN>& denseColOffsets, std::vector<UIN>& sparseCols,
                          std::vector<UIN>& sparseColOffsets, std::vector<UIN>& sparseDataOffsets,
                          BSMR::DeviceRphmArrays& rphmArrays, float& time);

// bsa_rowReordering_gpu of the reference (src/rowReordering.cu:1027-1095) on the MI355X:
// csrc/cluster_kernels.hpp through bsmr_cluster_rows.  Returns false when the device path is
// unavailable (no device, table too large); the outputs are then untouched.
bool bsa_rowReordering_device(const sparseMatrix::CSR<float>& matrix, const float alpha, const UIN block_size,
                              int device, std::vector<UIN>& reorderedRows, int& num_clusters,
                              float& reordering_time);

std::vector<UIN> bsa_rowReordering_host(const sparseMatrix::CSR<float>& matrix, const float alpha,
                                        const UIN block_size, int& num_clusters,
                                        float& reordering_time);

// Source-compatible name of the reference entry point; runs the host clustering.
inline std::vector<UIN> bsa_rowReordering_gpu(const sparseMatrix::CSR<float>& matrix,
                                              const float alpha, const UIN block_size,
                                              int& num_clusters, float& reordering_time) {
    return bsa_rowReordering_host(matrix, alpha, block_size, num_clusters, reordering_time);
}

// Per-panel column reordering and dense/sparse split (reference
// src/colReordering.cu:274-404).
void colReordering_cpu(const sparseMatrix::CSR<float>& matrix, const UIN numRowPanels,
                       const std::vector<UIN>& reorderedRows, const float blockDensityThreshold,
                       std::vector<UIN>& denseCols, std::vector<UIN>& denseColOffsets,
                       std::vector<UIN>& sparseCols, std::vector<UIN>& sparseColOffsets,
                       std::vector<UIN>& sparseDataOffsets, float& time);

// (#dense column slots, #sparse column slots) of one panel whose per-column
// counts are sorted descending and padded to a multiple of 16 (reference
// src/colReordering.cu:244-271).
std::pair<UIN, UIN> analysisDescendingOrderColSegment(
    const float blockDensityThreshold, const std::vector<UIN>& numOfNonZeroInEachColSegment);

// Structural validation (reference src/BSMR.cpp:444-824, 932-953).
bool check_rphm(const sparseMatrix::CSR<float>& matrix, const BSMR& bsmr, const RPHM& rphm,
                const float denseColSegmentThreshold);

std::pair<UIN, float> calculateNumDenseBlocksAndAverageDensityInOriginalMatrix(
    const float densityThreshold, const sparseMatrix::CSR<float>& matrix);

void evaluationReordering(const sparseMatrix::CSR<float>& matrix, const BSMR& bsmr, Logger& logger);
