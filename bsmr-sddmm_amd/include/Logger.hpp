#pragma once
// Result record and `[key : value]` log printer with the keys and order of the
// reference's include/Logger.hpp:13-187, so that its scripts/analyze_results.cpp
// can parse logs produced on MI355X.  gflops = 2*NNZ*K / (ms * 1e6) (:178-180).

#include <cmath>
#include <cstddef>
#include <iomanip>
#include <iostream>
#include <string>
#include <typeinfo>

#include "Matrix.hpp"
#include "Options.hpp"

struct Dim3 {
    unsigned x = 1, y = 1, z = 1;
};

// Name of HIP device 0, or "no device" (defined in src/sddmmKernel.cpp).
std::string bsmrDeviceName();

struct Logger {
    Logger() {
#ifdef NDEBUG
        buildType_ = "Release";
#else
        buildType_ = "Debug";
#endif
        gpu_ = bsmrDeviceName();
        matrixA_type_ = typeid(MATRIX_A_TYPE).name();
        matrixB_type_ = typeid(MATRIX_B_TYPE).name();
        matrixC_type_ = typeid(MATRIX_C_TYPE).name();
        wmma_m_ = WMMA_M;
        wmma_n_ = WMMA_N;
        wmma_k_ = WMMA_K;
    }

    void getInformation(const Options& options) {
        inputFile_ = options.inputFile();
        K_ = options.K();
        numITER_ = options.numIterations();
        alpha_ = options.similarityThresholdAlpha();
        delta_ = options.blockDensityThresholdDelta();
    }

    void getInformation(const sparseMatrix::DataBase& matrix) {
        M_ = matrix.row();
        N_ = matrix.col();
        NNZ_ = matrix.nnz();
        sparsity_ = matrix.getSparsity();
    }

    template <typename T>
    void getInformation(const Matrix<T>& matrixA, const Matrix<T>& matrixB) {
        K_ = matrixA.col();
        matrixA_storageOrder_ = matrixA.storageOrder() == row_major ? "row_major" : "col_major";
        matrixB_storageOrder_ = matrixB.storageOrder() == row_major ? "row_major" : "col_major";
    }

    inline void printLogInformation(std::ostream& out = std::cout) const;

    std::string inputFile_;
    std::string checkData_;
    float errorRate_ = 0.0f;
    std::string gpu_;
    std::string buildType_;
    size_t wmma_m_ = 0, wmma_n_ = 0, wmma_k_ = 0;
    std::string matrixA_type_, matrixB_type_, matrixC_type_;
    std::string matrixA_storageOrder_, matrixB_storageOrder_;
    size_t M_ = 0, N_ = 0, K_ = 0, NNZ_ = 0;
    float sparsity_ = 0.0f;
    Dim3 gridDim_dense_, gridDim_sparse_, blockDim_dense_, blockDim_sparse_;
    int numRowPanels_ = 0;
    int numDenseBlock_ = 0;
    float averageDensity_ = 0.0f;
    int originalNumDenseBlock_ = 0;
    float originalAverageDensity_ = 0.0f;
    int numDenseThreadBlocks_ = 0;
    int numSparseThreadBlocks_ = 0;
    int numDenseData_ = 0;
    int numSparseData_ = 0;
    int numITER_ = 10;
    float alpha_ = 0.0f;
    float delta_ = 0.0f;
    int numClusters_ = 1;
    float sddmmTime_ = 0.0f;
    float shardComputeTime_ = 0.0f, shardGatherTime_ = 0.0f;   // sddmm_multi_gpu: one un-pipelined step taken apart (ms)
    float rowReorderingTime_ = 0.0f;
    float colReorderingTime_ = 0.0f;
    float reorderingTime_ = 0.0f;
    // MI355X additions (printed after the reference's keys)
    std::string computeMode_ = "f16";
    std::string denseEngine_ = "stream";   // dense engine of the timed calls (bsmr_plan_tune, BSMR_DENSE_ENGINE=tuned)
    float convertTime_ = 0.0f;
    float denseTime_ = 0.0f;
    float sparseTime_ = 0.0f;
    int status_ = 0;              // bsmr_hip.h status of the device call behind the numbers above (0 = success)
};

void Logger::printLogInformation(std::ostream& out) const {
    auto dims = [&](const char* key, const Dim3& d) {
        out << "[" << key << " : " << d.x << ", " << d.y << ", " << d.z << "]\n";
    };
    out << "[File : " << inputFile_ << "]\n";
    out << "[Build type : " << buildType_ << "]\n";
    out << "[Device : " << gpu_ << "]\n";
    out << "[WMMA_M : " << wmma_m_ << "], [WMMA_N : " << wmma_n_ << "], [WMMA_K : " << wmma_k_ << "]\n";
    out << "[K : " << K_ << "], [M : " << M_ << "], [N : " << N_ << "], [NNZ : " << NNZ_ << "], ";
    out << "[sparsity : " << std::fixed << std::setprecision(2)
        << (std::floor(sparsity_ * 10000) / 100.0) << "%]\n";
    out << "[matrixA type : " << matrixA_type_ << "]\n";
    out << "[matrixB type : " << matrixB_type_ << "]\n";
    out << "[matrixC type : " << matrixC_type_ << "]\n";
    out << "[matrixA storageOrder : " << matrixA_storageOrder_ << "]\n";
    out << "[matrixB storageOrder : " << matrixB_storageOrder_ << "]\n";
    out << "[Num iterations : " << numITER_ << "]\n";
    out << "[NumRowPanel : " << numRowPanels_ << "]\n";
    out << "[original_numDenseBlock : " << originalNumDenseBlock_ << "]\n";
    out << "[original_averageDensity : " << originalAverageDensity_ << "]\n";
    out << "[bsmr_alpha : " << alpha_ << "]\n";
    out << "[bsmr_delta : " << delta_ << "]\n";
    out << "[bsmr_numClusters : " << numClusters_ << "]\n";
    out << "[bsmr_numDenseBlock : " << numDenseBlock_ << "]\n";
    out << "[bsmr_averageDensity : " << averageDensity_ << "]\n";
    out << "[bsmr_rowReordering : " << rowReorderingTime_ << "]\n";
    out << "[bsmr_colReordering : " << colReorderingTime_ << "]\n";
    out << "[bsmr_reordering : " << reorderingTime_ << "]\n";
    dims("gridDim_dense", gridDim_dense_);
    dims("blockDim_dense", blockDim_dense_);
    dims("gridDim_sparse", gridDim_sparse_);
    dims("blockDim_sparse", blockDim_sparse_);
    out << "[bsmr_numDenseThreadBlocks : " << numDenseThreadBlocks_ << "]\n";
    out << "[bsmr_numSparseThreadBlocks : " << numSparseThreadBlocks_ << "]\n";
    out << "[bsmr_threadBlockRatio : " << std::fixed << std::setprecision(2)
        << static_cast<float>(numDenseThreadBlocks_) / numSparseThreadBlocks_ << "]\n";
    out << "[bsmr_numDenseData : " << numDenseData_ << "]\n";
    out << "[bsmr_numSparseData : " << numSparseData_ << "]\n";
    out << "[bsmr_dataRatio: " << std::fixed << std::setprecision(2)
        << static_cast<float>(numDenseData_) / numSparseData_ << "]\n";
    const size_t flops = 2 * NNZ_ * K_;
    out << "[bsmr_gflops : " << (flops / (sddmmTime_ * 1e6)) << "]\n";
    out << "[bsmr_sddmm : " << sddmmTime_ << "]\n";
    // MI355X additions: the reference's two-decimal milliseconds are too coarse for
    // 10-microsecond kernels, so the same times are repeated in microseconds.
    out << "[mi355x_compute : " << computeMode_ << "]\n";
    out << std::fixed << std::setprecision(3);
    out << "[mi355x_sddmm_us : " << sddmmTime_ * 1e3f << "]\n";
    out << "[mi355x_status : " << status_ << "]\n";
    out << "[mi355x_convert_us : " << convertTime_ * 1e3f << "]\n";
    out << "[mi355x_dense_us : " << denseTime_ * 1e3f << "]\n";
    out << "[mi355x_sparse_us : " << sparseTime_ * 1e3f << "]\n";
    out << "[mi355x_dense_engine : " << denseEngine_ << "]\n";
    out << std::setprecision(2);
    if (errorRate_ > 0)
        out << "[checkResults : NO PASS Error rate : " << std::fixed << std::setprecision(2)
            << errorRate_ << "%]\n";
}
