// Orchestration: reference src/sddmm.cu:10-118.
#include "sddmm.hpp"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <fstream>
#include <memory>
#include <exception>
#include <thread>
#include <vector>

#include "bsmr_hip.h"

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

// ---------------------------------------------------------------------------
// several GPUs of one node
// ---------------------------------------------------------------------------
std::vector<UIN> partitionRowsByCost(const sparseMatrix::CSR<float>& matrix, int world) {
    // cost of a row = its entries + its sixteenth of a panel's fixed work (24 entry-equivalents per panel): the same
    // model as python/shard.py row_costs / partition_by_cost
    const UIN rows = matrix.row();
    std::vector<double> prefix(static_cast<size_t>(rows) + 1, 0.0);
    for (UIN r = 0; r < rows; ++r) {
        const UIN d = matrix.rowOffsets()[r + 1] - matrix.rowOffsets()[r];
        prefix[r + 1] = prefix[r] + static_cast<double>(d) + (d ? 24.0 / 16.0 : 0.0);
    }
    std::vector<UIN> bounds{0};
    for (int r = 1; r < world; ++r) {
        const double target = prefix[rows] * r / world;
        UIN cut = static_cast<UIN>(std::lower_bound(prefix.begin(), prefix.end(), target) - prefix.begin());
        cut = std::min<UIN>(rows, std::max<UIN>((cut + 8) / 16 * 16, bounds.back()));
        bounds.push_back(cut);
    }
    bounds.push_back(rows);
    return bounds;
}

void sddmm_multi_gpu(const Options& options, const Matrix<float>& matrixA, const Matrix<float>& matrixB,
                     sparseMatrix::CSR<float>& matrixP, const std::vector<int>& devices, Logger& logger) {
    const int world = static_cast<int>(devices.size());
    logger.status_ = BSMR_ERR_INVALID_ARG;
    if (world <= 0) return;
    const std::vector<UIN> bounds = partitionRowsByCost(matrixP, world);
    const UIN K = matrixA.col();
    std::vector<sparseMatrix::CSR<float>> slices;
    std::vector<BSMR> bsmrs(world);
    std::vector<std::unique_ptr<RPHM>> rphms(world);
    std::vector<bsmr_rphm_desc> descs(world);
    std::vector<const bsmr_rphm_desc*> descPtrs(world);
    slices.reserve(world);
    float reordering = 0.0f;
    for (int i = 0; i < world; ++i) {
        const UIN r0 = bounds[i], r1 = bounds[i + 1];
        const UIN e0 = matrixP.rowOffsets()[r0], e1 = matrixP.rowOffsets()[r1];
        std::vector<UIN> ro(static_cast<size_t>(r1 - r0) + 1), ci(matrixP.colIndices().begin() + e0, matrixP.colIndices().begin() + e1);
        for (UIN r = r0; r <= r1; ++r) ro[r - r0] = matrixP.rowOffsets()[r] - e0;
        slices.emplace_back(r1 - r0, matrixP.col(), e1 - e0, ro, ci);
    }
    // The shards' pipelines are independent (the row ranges are disjoint, reference src/BSMR.cpp:678-711): one host thread per
    // DISTINCT device builds the shards of that device one after another - eight devices cluster their slices side by side
    // instead of 8 x ~0.4 s in a row.  BSMR_SHARD_BUILD_THREADS=n: n threads whatever the devices are (several shards of one
    // device then build side by side too; their device work shares that device's default stream).
    std::vector<float> shardTime(world, 0.0f);
    auto buildShard = [&](int i) {
        struct DeviceScope {   // clustering of this slice on the slice's device
            int before = pipelineDevice();
            explicit DeviceScope(int d) { setPipelineDevice(d); }
            ~DeviceScope() { setPipelineDevice(before); }
        } scope(devices[i]);
        bsmrs[i] = BSMR(options.similarityThresholdAlpha(), options.blockDensityThresholdDelta(), slices[i], 1);
        rphms[i].reset(new RPHM(slices[i], bsmrs[i], -1));   // host arrays; the device side is bsmr_sharded_create
        shardTime[i] = bsmrs[i].reorderingTime() + rphms[i]->time();
        bsmr_rphm_desc& d = descs[i];
        d = bsmr_rphm_desc{};
        d.M = slices[i].row();
        d.N = slices[i].col();
        d.nnz = slices[i].nnz();
        d.num_row_panels = static_cast<uint32_t>(bsmrs[i].numRowPanels());
        d.num_nonzero_rows = static_cast<uint32_t>(bsmrs[i].reorderedRows().size());
        d.reordered_rows = bsmrs[i].reorderedRows().data();
        d.dense_cols = bsmrs[i].denseCols().data();
        d.block_offsets = rphms[i]->blockOffsets().data();
        d.block_values = rphms[i]->blockValues().data();
        d.sparse_value_offsets = bsmrs[i].sparseValueOffsets().data();
        d.sparse_values = rphms[i]->sparseValues().data();
        d.sparse_relative_rows = rphms[i]->sparseRelativeRows().data();
        d.sparse_col_indices = rphms[i]->sparseColIndices().data();
        descPtrs[i] = &d;
    };
    {
        std::vector<std::vector<int>> lanes;   // shard numbers per builder thread
        int forced = 0;
        if (const char* env = std::getenv("BSMR_SHARD_BUILD_THREADS")) forced = std::max(0, std::atoi(env));
        if (forced > 0) {
            lanes.assign(static_cast<size_t>(std::min(forced, world)), {});
            for (int i = 0; i < world; ++i) lanes[static_cast<size_t>(i) % lanes.size()].push_back(i);
        } else {
            std::vector<int> seen;
            for (int i = 0; i < world; ++i) {
                size_t u = 0;
                while (u < seen.size() && seen[u] != devices[i]) ++u;
                if (u == seen.size()) {
                    seen.push_back(devices[i]);
                    lanes.emplace_back();
                }
                lanes[u].push_back(i);
            }
        }
        if (lanes.size() <= 1) {
            for (int i = 0; i < world; ++i) buildShard(i);
        } else {
            std::vector<std::thread> pool;
            std::vector<std::exception_ptr> failed(lanes.size());
            for (size_t t = 0; t < lanes.size(); ++t)
                pool.emplace_back([&, t]() {
                    try {
                        for (const int i : lanes[t]) buildShard(i);
                    } catch (...) {
                        failed[t] = std::current_exception();
                    }
                });
            for (std::thread& th : pool) th.join();
            for (const std::exception_ptr& f : failed)
                if (f) std::rethrow_exception(f);
        }
    }
    for (int i = 0; i < world; ++i) reordering = std::max(reordering, shardTime[i]);
    logger.reorderingTime_ = reordering;
    bsmr_plan_options opts;
    bsmr_plan_options_from_env(&opts);
    bsmr_sharded* sharded = nullptr;
    int st = bsmr_sharded_create(&sharded, devices.data(), static_cast<uint32_t>(world), descPtrs.data(), bounds.data(), &opts);
    if (st == BSMR_OK) {
        bsmr_sharded_timing t{};
        st = bsmr_sharded_sddmm_host(sharded, K, matrixA.data(), matrixB.data(), matrixP.setValues().data(), sddmmComputeMode(),
                                     logger.numITER_ > 0 ? logger.numITER_ : 1, &t);
        logger.sddmmTime_ = t.step_ms;
        logger.shardComputeTime_ = t.compute_ms;
        logger.shardGatherTime_ = t.gather_ms;
    }
    if (st != BSMR_OK) fprintf(stderr, "sddmm_multi_gpu: %s (%s)\n", bsmr_strerror(st), bsmr_last_hip_error());
    bsmr_sharded_destroy(sharded);
    logger.status_ = st;
    logger.computeMode_ = sddmmComputeMode() == BSMR_COMPUTE_BF16 ? "bf16" : sddmmComputeMode() == BSMR_COMPUTE_F32 ? "f32" : "f16";
}
