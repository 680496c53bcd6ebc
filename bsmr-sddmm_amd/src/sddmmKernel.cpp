// Host launchers: reference src/sddmmKernel.cu:2518-2762 (sddmm_gpu overloads and
// sddmm_gpu_k32).  All device work happens behind include/bsmr_hip.h.
#include "sddmmKernel.hpp"

#include <chrono>
#include <string>

#include <cstdio>

#include "bsmr_hip.h"
#include "BSMR.hpp"
#include "devVector.hpp"

namespace {
int g_computeMode = BSMR_COMPUTE_F16;
const char* modeName(int m) {
    return m == BSMR_COMPUTE_BF16 ? "bf16" : m == BSMR_COMPUTE_F32 ? "f32" : "f16";
}
}  // namespace

void setSddmmComputeMode(int mode) { g_computeMode = mode; }
int sddmmComputeMode() { return g_computeMode; }

std::string bsmrDeviceName() {
    char buf[256] = {0};
    if (bsmr_device_name(pipelineDevice(), buf, sizeof(buf)) != BSMR_OK) return "no device";
    return buf;
}

void sddmm_gpu(UIN M, UIN N, UIN K, const float* matrixA, const float* matrixB, const RPHM& rphm,
               float* matrixP, Logger& logger) {
    (void)M;
    (void)N;
    if (!rphm.plan()) {
        fprintf(stderr, "sddmm_gpu: RPHM has no device plan (status %d: %s)\n", rphm.planStatus(),
                bsmr_strerror(rphm.planStatus()));
        logger.status_ = rphm.planStatus() != BSMR_OK ? rphm.planStatus() : BSMR_ERR_INVALID_ARG;
        return;
    }
    // a tunable plan (BSMR_DENSE_ENGINE=tuned) first measures its dense engines on these operands; any other plan
    // answers BSMR_ERR_INVALID_ARG and runs as it was built
    bsmr_tune_report tuned{};
    if (bsmr_plan_tune(rphm.plan(), K, matrixA, matrixB, matrixP, g_computeMode, nullptr, &tuned) == BSMR_OK) {
        static const char* const names[] = {"stream", "tiles", "shared"};
        logger.denseEngine_ = std::string(names[tuned.chosen_engine >= 0 && tuned.chosen_engine <= 2 ? tuned.chosen_engine : 0]) + " (tuned)";
    }
    bsmr_timing t{};
    const int iters = logger.numITER_ > 0 ? logger.numITER_ : 1;
    const int st = bsmr_sddmm_timed(rphm.plan(), K, matrixA, matrixB, matrixP, g_computeMode, nullptr,
                                    1, iters, &t);
    if (st != BSMR_OK) {
        fprintf(stderr, "sddmm_gpu: %s (%s)\n", bsmr_strerror(st), bsmr_last_hip_error());
        logger.status_ = st;
        return;
    }
    logger.status_ = BSMR_OK;
    bsmr_plan_stats s{};
    bsmr_plan_get_stats(rphm.plan(), &s);
    logger.gridDim_dense_ = Dim3{static_cast<unsigned>(s.dense_work_items), 1, 1};
    logger.blockDim_dense_ = Dim3{256, 1, 1};
    logger.gridDim_sparse_ = Dim3{static_cast<unsigned>(s.sparse_work_items), 1, 1};
    logger.blockDim_sparse_ = Dim3{256, 1, 1};
    logger.sddmmTime_ = t.total_ms;
    logger.convertTime_ = t.convert_ms;
    logger.denseTime_ = t.dense_ms;
    logger.sparseTime_ = t.sparse_ms;
    logger.computeMode_ = modeName(g_computeMode);
}

void sddmm_gpu_k32(UIN M, UIN N, UIN K, const float* matrixA, const float* matrixB,
                   const RPHM& rphm, float* matrixP, Logger& logger) {
    sddmm_gpu(M, N, K, matrixA, matrixB, rphm, matrixP, logger);
}

void sddmm_gpu(const Matrix<float>& matrixA, const Matrix<float>& matrixB, const RPHM& rphm,
               sparseMatrix::CSR<float>& matrixP, Logger& logger) {
    dev::vector<float> A(matrixA.values(), rphm.device());
    dev::vector<float> B(matrixB.values(), rphm.device());
    dev::vector<float> P(matrixP.nnz(), 0, rphm.device());
    if (!A.ok() || !B.ok() || !P.ok()) {
        fprintf(stderr, "sddmm_gpu: device allocation failed\n");
        logger.status_ = BSMR_ERR_OOM;
        return;
    }
    sddmm_gpu(matrixP.row(), matrixP.col(), matrixA.col(), A.data(), B.data(), rphm, P.data(), logger);
    matrixP.setValues() = d2h(P);
}

void sddmm_gpu_batch(const UIN numBatch, const UIN M, const UIN N, const UIN K, const UIN nnz,
                     const float* matrixA, const float* matrixB, const RPHM& rphm, float* matrixP, float& time) {
    (void)M;
    (void)N;
    (void)nnz;
    time = 0.0f;
    if (!rphm.plan()) {
        fprintf(stderr, "sddmm_gpu_batch: RPHM has no device plan (status %d: %s)\n", rphm.planStatus(),
                bsmr_strerror(rphm.planStatus()));
        return;
    }
    // one untimed call (allocates the operand workspace), then the timed one
    int st = bsmr_sddmm_batch(rphm.plan(), K, matrixA, matrixB, matrixP, numBatch, g_computeMode, nullptr);
    if (st == BSMR_OK) st = bsmr_device_synchronize(rphm.device());
    const auto t0 = std::chrono::steady_clock::now();
    if (st == BSMR_OK) st = bsmr_sddmm_batch(rphm.plan(), K, matrixA, matrixB, matrixP, numBatch, g_computeMode, nullptr);
    if (st == BSMR_OK) st = bsmr_device_synchronize(rphm.device());
    if (st != BSMR_OK) {
        fprintf(stderr, "sddmm_gpu_batch: %s (%s)\n", bsmr_strerror(st), bsmr_last_hip_error());
        return;
    }
    time = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void batchedMatrixTranspose(const UIN width, const UIN height, const UIN numBatches, const float* d_input,
                            float* d_output) {
    const int st = bsmr_batched_transpose(width, height, numBatches, d_input, d_output, nullptr);
    if (st != BSMR_OK) fprintf(stderr, "batchedMatrixTranspose: %s (%s)\n", bsmr_strerror(st), bsmr_last_hip_error());
}
