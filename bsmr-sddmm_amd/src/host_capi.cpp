// extern "C" wrappers over the C++ host pipeline (include/bsmr_host.h).
#include "bsmr_host.h"

#include <cstring>
#include <memory>
#include <new>
#include <sstream>

#include "BSMR.hpp"
#include "Logger.hpp"
#include "Matrix.hpp"
#include "checkData.hpp"
#include "host.hpp"
#include "sddmm.hpp"
#include "sddmmKernel.hpp"

struct bsmr_csr {
    sparseMatrix::CSR<float> m;
};

struct bsmr_pipeline {
    BSMR bsmr;
    std::unique_ptr<RPHM> rphm;
    int clusters = -1;  // set when the clustering ran outside BSMR (explicit bin width)
};

namespace {
template <typename F>
auto guarded(F f, decltype(f()) onError) -> decltype(f()) {
    try {
        return f();
    } catch (...) {
        return onError;
    }
}
}  // namespace

extern "C" {

bsmr_csr* bsmr_csr_from_file(const char* path) {
    if (!path) return nullptr;
    return guarded([&]() -> bsmr_csr* {
        std::unique_ptr<bsmr_csr> h(new bsmr_csr);
        if (!h->m.initializeFromMatrixFile(path)) return nullptr;
        return h.release();
    }, nullptr);
}

bsmr_csr* bsmr_csr_from_arrays(uint32_t rows, uint32_t cols, uint32_t nnz, const uint32_t* ro,
                               const uint32_t* ci) {
    if (!ro || (!ci && nnz)) return nullptr;
    return guarded([&]() -> bsmr_csr* {
        std::vector<UIN> r(ro, ro + rows + 1), c(ci, ci + nnz);
        std::unique_ptr<bsmr_csr> h(new bsmr_csr{sparseMatrix::CSR<float>(rows, cols, nnz, r, c)});
        return h.release();
    }, nullptr);
}

void bsmr_csr_free(bsmr_csr* m) { delete m; }
uint32_t bsmr_csr_rows(const bsmr_csr* m) { return m ? m->m.row() : 0; }
uint32_t bsmr_csr_cols(const bsmr_csr* m) { return m ? m->m.col() : 0; }
uint32_t bsmr_csr_nnz(const bsmr_csr* m) { return m ? m->m.nnz() : 0; }
const uint32_t* bsmr_csr_row_offsets(const bsmr_csr* m) { return m ? m->m.rowOffsets().data() : nullptr; }
const uint32_t* bsmr_csr_col_indices(const bsmr_csr* m) { return m ? m->m.colIndices().data() : nullptr; }
float* bsmr_csr_values(bsmr_csr* m) { return m ? m->m.setValues().data() : nullptr; }
int bsmr_csr_check(const bsmr_csr* m) { return m && checkMatrixData(m->m) ? 1 : 0; }
int bsmr_csr_write_mtx(const bsmr_csr* m, const char* path) {
    return m && path && m->m.outputToMarketMatrixFile(path) ? 1 : 0;
}

void bsmr_make_data(float* out, size_t count, uint32_t seed) {
    if (!out || count == 0) return;
    Matrix<float> tmp(1, static_cast<UIN>(count), row_major);
    tmp.makeDataSeeded(seed);
    memcpy(out, tmp.data(), count * sizeof(float));
}

uint32_t bsmr_calculate_block_size(const bsmr_csr* m, size_t freeBytes) {
    return m ? calculateBlockSize(m->m, freeBytes) : 0;
}

bsmr_pipeline* bsmr_pipeline_create(const bsmr_csr* m, float alpha, float delta, int row_mode,
                                    uint32_t block_size, int device) {
    if (!m) return nullptr;
    return guarded([&]() -> bsmr_pipeline* {
        std::unique_ptr<bsmr_pipeline> p(new bsmr_pipeline);
        struct DeviceScope {   // clustering, block size and plan of this pipeline all on `device`
            int before = pipelineDevice();
            explicit DeviceScope(int d) { if (d >= 0) setPipelineDevice(d); }
            ~DeviceScope() { setPipelineDevice(before); }
        } scope(device);
        if (row_mode == BSMR_ROWS_IDENTITY) {
            std::vector<UIN> rows;
            float t = 0;
            noReorderRow(m->m, rows, t);
            if (rows.empty()) return nullptr;
            p->bsmr.colReordering(delta, m->m, rows);
        } else if (block_size != 0) {
            int clusters = 0;
            float t = 0;
            std::vector<UIN> rows;
            if (device < 0 || !bsa_rowReordering_device(m->m, alpha, block_size, device, rows, clusters, t))
                rows = bsa_rowReordering_host(m->m, alpha, block_size, clusters, t);
            if (rows.empty()) return nullptr;
            p->clusters = clusters;
            p->bsmr.colReordering(delta, m->m, rows);
        } else {
            const int before = clusteringDevice();
            if (device < 0) setClusteringDevice(-1);  // host-only pipeline: no GPU touched
            p->bsmr = BSMR(alpha, delta, m->m, 1);
            setClusteringDevice(before);
        }
        p->rphm.reset(new RPHM(m->m, p->bsmr, device));
        return p.release();
    }, nullptr);
}

void bsmr_pipeline_free(bsmr_pipeline* p) { delete p; }

int bsmr_pipeline_resplit(bsmr_pipeline* p, const bsmr_csr* m, float delta, int device) {
    if (!p || !m) return BSMR_ERR_INVALID_ARG;
    return guarded([&]() -> int {
        p->bsmr.colReordering(delta, m->m);
        p->rphm.reset(new RPHM(m->m, p->bsmr, device));
        return device >= 0 ? p->rphm->planStatus() : BSMR_OK;
    }, BSMR_ERR_OOM);
}

int bsmr_pipeline_array(const bsmr_pipeline* p, int which, const uint32_t** data, size_t* len) {
    if (!p || !data || !len || !p->rphm) return BSMR_ERR_INVALID_ARG;
    const std::vector<UIN>* v = nullptr;
    switch (which) {
    case BSMR_ARR_REORDERED_ROWS: v = &p->bsmr.reorderedRows(); break;
    case BSMR_ARR_DENSE_COLS: v = &p->bsmr.denseCols(); break;
    case BSMR_ARR_DENSE_COL_OFFSETS: v = &p->bsmr.denseColOffsets(); break;
    case BSMR_ARR_SPARSE_COLS: v = &p->bsmr.sparseCols(); break;
    case BSMR_ARR_SPARSE_COL_OFFSETS: v = &p->bsmr.sparseColOffsets(); break;
    case BSMR_ARR_SPARSE_VALUE_OFFSETS: v = &p->bsmr.sparseValueOffsets(); break;
    case BSMR_ARR_BLOCK_OFFSETS: v = &p->rphm->blockOffsets(); break;
    case BSMR_ARR_BLOCK_VALUES: v = &p->rphm->blockValues(); break;
    case BSMR_ARR_SPARSE_VALUES: v = &p->rphm->sparseValues(); break;
    case BSMR_ARR_SPARSE_RELATIVE_ROWS: v = &p->rphm->sparseRelativeRows(); break;
    case BSMR_ARR_SPARSE_COL_INDICES: v = &p->rphm->sparseColIndices(); break;
    case BSMR_ARR_DENSE_ROW_PANEL_IDS: v = &p->rphm->denseRowPanelIds(); break;
    case BSMR_ARR_DENSE_COL_BLOCK_ITERS: v = &p->rphm->denseColBlockIters(); break;
    case BSMR_ARR_SPARSE_ROW_PANEL_IDS: v = &p->rphm->sparseRowPanelIds(); break;
    case BSMR_ARR_SPARSE_COL_BLOCK_ITERS: v = &p->rphm->sparseColBlockIters(); break;
    default: return BSMR_ERR_INVALID_ARG;
    }
    *data = v->data();
    *len = v->size();
    return BSMR_OK;
}

int bsmr_pipeline_num_row_panels(const bsmr_pipeline* p) { return p ? p->bsmr.numRowPanels() : 0; }
int bsmr_pipeline_num_clusters(const bsmr_pipeline* p) {
    if (!p) return 0;
    return p->clusters >= 0 ? p->clusters : p->bsmr.numClusters();
}
float bsmr_pipeline_row_reordering_ms(const bsmr_pipeline* p) { return p ? p->bsmr.rowReorderingTime() : 0; }
float bsmr_pipeline_col_reordering_ms(const bsmr_pipeline* p) { return p ? p->bsmr.colReorderingTime() : 0; }
float bsmr_pipeline_rphm_ms(const bsmr_pipeline* p) { return p && p->rphm ? p->rphm->time() : 0; }

int bsmr_pipeline_check(const bsmr_pipeline* p, const bsmr_csr* m, float delta) {
    if (!p || !m || !p->rphm) return 0;
    return guarded([&]() -> int { return check_rphm(m->m, p->bsmr, *p->rphm, delta) ? 1 : 0; }, 0);
}

int bsmr_pipeline_evaluate(const bsmr_pipeline* p, const bsmr_csr* m, float delta, bsmr_reordering_report* out) {
    if (!p || !m || !out || !p->rphm) return BSMR_ERR_INVALID_ARG;
    return guarded([&]() -> int {
        Logger logger;
        logger.delta_ = delta;
        evaluationReordering(m->m, p->bsmr, logger);
        out->original_num_dense_blocks = logger.originalNumDenseBlock_;
        out->original_average_density = logger.originalAverageDensity_;
        out->num_dense_blocks = logger.numDenseBlock_;
        out->average_density = logger.averageDensity_;
        out->num_dense_thread_blocks = logger.numDenseThreadBlocks_;
        out->num_sparse_thread_blocks = logger.numSparseThreadBlocks_;
        out->num_dense_data = logger.numDenseData_;
        out->num_sparse_data = logger.numSparseData_;
        out->max_dense_blocks_per_panel = p->rphm->maxNumDenseColBlocksInRowPanel();
        out->max_sparse_blocks_per_panel = p->rphm->maxNumSparseColBlocksInRowPanel();
        return BSMR_OK;
    }, BSMR_ERR_OOM);
}

bsmr_plan* bsmr_pipeline_plan(const bsmr_pipeline* p) { return p && p->rphm ? p->rphm->plan() : nullptr; }
int bsmr_pipeline_plan_status(const bsmr_pipeline* p) {
    return p && p->rphm ? p->rphm->planStatus() : BSMR_ERR_INVALID_ARG;
}

void bsmr_host_sddmm_cpu(const bsmr_csr* m, uint32_t K, const float* A, const float* B, float* P) {
    if (!m || !A || !B || !P) return;
    Matrix<float> a(m->m.row(), K, row_major, A);
    Matrix<float> b(K, m->m.col(), col_major, B);
    sparseMatrix::CSR<float> out(m->m);
    sddmm_cpu(a, b, m->m, out);
    memcpy(P, out.values().data(), out.values().size() * sizeof(float));
}

size_t bsmr_host_check_data(size_t n, const float* x, const float* y) {
    size_t errors = 0;
    for (size_t i = 0; i < n; ++i) errors += !checkOneData(x[i], y[i]);
    return errors;
}

int bsmr_host_sddmm(const bsmr_csr* m, uint32_t K, float alpha, float delta, int compute_mode,
                    int num_iterations, const float* A, const float* B, float* P, char* log_buf,
                    size_t log_buf_len) {
    if (!m || !A || !B || !P) return BSMR_ERR_INVALID_ARG;
    if (K == 0 || K % 32) return BSMR_ERR_UNSUPPORTED_K;
    return guarded([&]() -> int {
        const std::string a = std::to_string(alpha), d = std::to_string(delta), k = std::to_string(K);
        const char* argv[] = {"bsmr_host_sddmm", "-k", k.c_str(), "-a", a.c_str(), "-d", d.c_str()};
        Options options(7, argv);
        Matrix<float> ma(m->m.row(), K, row_major, A);
        Matrix<float> mb(K, m->m.col(), col_major, B);
        Logger logger;
        logger.getInformation(options);
        logger.getInformation(m->m);
        logger.getInformation(ma, mb);
        if (num_iterations > 0) logger.numITER_ = num_iterations;
        sparseMatrix::CSR<float> p(m->m);
        const int before = sddmmComputeMode();
        setSddmmComputeMode(compute_mode);
        sddmm(options, ma, mb, p, logger);
        setSddmmComputeMode(before);
        memcpy(P, p.values().data(), p.values().size() * sizeof(float));
        if (log_buf && log_buf_len) {
            std::ostringstream os;
            logger.printLogInformation(os);
            const std::string s = os.str();
            const size_t n = std::min(s.size(), log_buf_len - 1);
            memcpy(log_buf, s.data(), n);
            log_buf[n] = 0;
        }
        return logger.status_;
    }, BSMR_ERR_OOM);
}

int bsmr_partition_rows_by_cost(const bsmr_csr* m, uint32_t world, uint32_t* bounds) {
    if (!m || !bounds || world == 0) return BSMR_ERR_INVALID_ARG;
    return guarded([&]() -> int {
        const std::vector<UIN> b = partitionRowsByCost(m->m, static_cast<int>(world));
        std::copy(b.begin(), b.end(), bounds);
        return BSMR_OK;
    }, BSMR_ERR_OOM);
}

int bsmr_host_sddmm_sharded(const bsmr_csr* m, uint32_t K, float alpha, float delta, int compute_mode, int num_iterations,
                            const int* devices, uint32_t num_devices, const float* A, const float* B, float* P,
                            float* step_ms) {
    float t[3] = {0.f, 0.f, 0.f};
    const int st = bsmr_host_sddmm_sharded_timed(m, K, alpha, delta, compute_mode, num_iterations, devices, num_devices, A, B, P, t);
    if (step_ms) *step_ms = t[0];
    return st;
}

int bsmr_host_sddmm_sharded_timed(const bsmr_csr* m, uint32_t K, float alpha, float delta, int compute_mode, int num_iterations,
                                  const int* devices, uint32_t num_devices, const float* A, const float* B, float* P,
                                  float* times_ms) {
    if (!m || !A || !B || !P || !devices || num_devices == 0) return BSMR_ERR_INVALID_ARG;
    if (K == 0 || K % 32) return BSMR_ERR_UNSUPPORTED_K;
    return guarded([&]() -> int {
        const std::string a = std::to_string(alpha), d = std::to_string(delta), k = std::to_string(K);
        const char* argv[] = {"bsmr_host_sddmm_sharded", "-k", k.c_str(), "-a", a.c_str(), "-d", d.c_str()};
        Options options(7, argv);
        Matrix<float> ma(m->m.row(), K, row_major, A);
        Matrix<float> mb(K, m->m.col(), col_major, B);
        Logger logger;
        if (num_iterations > 0) logger.numITER_ = num_iterations;
        sparseMatrix::CSR<float> p(m->m);
        const int before = sddmmComputeMode();
        setSddmmComputeMode(compute_mode);
        sddmm_multi_gpu(options, ma, mb, p, std::vector<int>(devices, devices + num_devices), logger);
        setSddmmComputeMode(before);
        memcpy(P, p.values().data(), p.values().size() * sizeof(float));
        if (times_ms) {
            times_ms[0] = logger.sddmmTime_;
            times_ms[1] = logger.shardComputeTime_;
            times_ms[2] = logger.shardGatherTime_;
        }
        return logger.status_;
    }, BSMR_ERR_OOM);
}

}  // extern "C"
