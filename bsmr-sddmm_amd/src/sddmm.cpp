// Orchestration: reference src/sddmm.cu:10-118.
#include "sddmm.hpp"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "BSMR.hpp"
#include "checkData.hpp"
#include "host.hpp"
#include "sddmmKernel.hpp"

namespace {
bool g_validate = false;

void copyReorderingTimes(const BSMR& bsmr, Logger& logger) {
    logger.rowReorderingTime_ = bsmr.rowReorderingTime();
    logger.colReorderingTime_ = bsmr.colReorderingTime();
    logger.reorderingTime_ = bsmr.reorderingTime();
    logger.numRowPanels_ = bsmr.numRowPanels();
    logger.numClusters_ = bsmr.numClusters();
}
}  // namespace

void setSddmmValidate(bool on) { g_validate = on; }

void sddmm(const Options& options, const Matrix<float>& matrixA, const Matrix<float>& matrixB,
           sparseMatrix::CSR<float>& matrixP, Logger& logger) {
    BSMR bsmr(options.similarityThresholdAlpha(), options.blockDensityThresholdDelta(), matrixP, 1);
    copyReorderingTimes(bsmr, logger);

    RPHM rphm(matrixP, bsmr, pipelineDevice());
    sddmm_gpu(matrixA, matrixB, rphm, matrixP, logger);
    evaluationReordering(matrixP, bsmr, logger);

    if (g_validate || std::getenv("BSMR_VALIDATE")) {
        check_rphm(matrixP, bsmr, rphm, options.blockDensityThresholdDelta());
        sparseMatrix::CSR<float> expected(matrixP);
        sddmm_cpu(matrixA, matrixB, matrixP, expected);
        size_t numError = 0;
        printf("check cpu sddmm and BSMR sddmm: \n");
        if (!checkData(expected.values(), matrixP.values(), numError)) {
            logger.errorRate_ = static_cast<float>(numError) / static_cast<float>(matrixP.nnz()) * 100;
            printf("[checkData : NO PASS Error rate : %2.2f%%]\n", logger.errorRate_);
        }
    }
}

bool checkSddmm(const Matrix<float>& matrixA, const Matrix<float>& matrixB,
                const sparseMatrix::CSR<float>& matrixS, const sparseMatrix::CSR<float>& matrixP) {
    sparseMatrix::CSR<float> expected(matrixS);
    sddmm_cpu(matrixA, matrixB, matrixS, expected);
    printf("check cpu sddmm and BSMR sddmm: \n");
    size_t numError = 0;
    if (!checkData(expected.values(), matrixP.values(), numError)) {
        printf("[checkData : NO PASS Error rate : %2.2f%%]\n",
               static_cast<float>(numError) / static_cast<float>(matrixP.values().size()) * 100);
        return false;
    }
    return true;
}

void sddmm_testMode(const Options& options, sparseMatrix::CSR<float>& matrixP) {
    const std::vector<float> alphas = {0.1f, 0.3f, 0.5f, 0.7f, 0.9f};
    const std::vector<float> deltas = {0.0f, 0.1f, 0.3f, 0.5f, 0.7f, 0.9f, 1.1f};
    const std::vector<UIN> Ks = {32, 64, 128, 256};

    BSMR bsmr;
    for (const float alpha : alphas) {
        bsmr.rowReordering(alpha, matrixP);  // once per alpha; delta only moves the split
        for (const float delta : deltas) {
            bsmr.colReordering(delta, matrixP);
            RPHM rphm(matrixP, bsmr, pipelineDevice());
            for (const UIN k : Ks) {
                Matrix<float> matrixA(matrixP.row(), k, row_major);
                matrixA.makeData();
                Matrix<float> matrixB(k, matrixP.col(), col_major);
                matrixB.makeData();

                Logger logger;
                logger.getInformation(options);
                logger.getInformation(matrixP);
                logger.getInformation(matrixA, matrixB);
                logger.alpha_ = alpha;
                logger.delta_ = delta;
                copyReorderingTimes(bsmr, logger);

                sddmm_gpu(matrixA, matrixB, rphm, matrixP, logger);
                evaluationReordering(matrixP, bsmr, logger);

                const std::string logFile = options.outputLogDirectory() + "BSMR_k_" +
                                            util::to_trimmed_string(k) + "_a_" +
                                            util::to_trimmed_string(alpha) + "_d_" +
                                            util::to_trimmed_string(delta) + ".log";
                std::ofstream fout(logFile, std::ios::app);
                if (fout.fail()) {
                    fprintf(stderr, "Error, failed to open log file: %s\n", logFile.c_str());
                    return;
                }
                fout << "\n---New data---\n";
                logger.printLogInformation(fout);
            }
        }
    }
}
