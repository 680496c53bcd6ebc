/*
 * bsmr_host.h -- C view of the host-side BSMR pipeline (libbsmr_host.so), for
 * FFI users (the Python tests and bench.py bind it with ctypes).  The C++ API
 * itself mirrors the reference's headers (the .hpp files under bsmr-sddmm_amd/include); each
 * function below names the C++ entry point it forwards to and the reference
 * interface that one replaces (paths relative to the reference checkout).
 */
#ifndef BSMR_HOST_H
#define BSMR_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "bsmr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bsmr_csr bsmr_csr;           /* sparseMatrix::CSR<float>  (include/Matrix.hpp:196-297) */
typedef struct bsmr_pipeline bsmr_pipeline; /* BSMR + RPHM               (include/BSMR.hpp:21-159)    */

/* CSR<float>::initializeFromMatrixFile (src/Matrix.cpp:279-294): .mtx/.mmio, .smtx, .txt.
 * Returns NULL where the reference returns false. */
bsmr_csr *bsmr_csr_from_file(const char *path);
/* CSR(row, col, nnz, rowOffsets, colIndices) constructor (include/Matrix.hpp:245-254). */
bsmr_csr *bsmr_csr_from_arrays(uint32_t rows, uint32_t cols, uint32_t nnz,
                               const uint32_t *row_offsets, const uint32_t *col_indices);
void bsmr_csr_free(bsmr_csr *m);
uint32_t bsmr_csr_rows(const bsmr_csr *m);
uint32_t bsmr_csr_cols(const bsmr_csr *m);
uint32_t bsmr_csr_nnz(const bsmr_csr *m);
const uint32_t *bsmr_csr_row_offsets(const bsmr_csr *m);
const uint32_t *bsmr_csr_col_indices(const bsmr_csr *m);
float *bsmr_csr_values(bsmr_csr *m);
/* checkMatrixData (include/Matrix.hpp:400-401) */
int bsmr_csr_check(const bsmr_csr *m);
/* CSR::outputToMarketMatrixFile(fileName) */
int bsmr_csr_write_mtx(const bsmr_csr *m, const char *path);

/* Matrix<float>::makeData with an explicit seed (src/Matrix.cpp:117-138): U[0,2),
 * x = 2 * (u >> 8) * 2^-24 with u drawn sequentially from std::mt19937(seed). */
void bsmr_make_data(float *out, size_t count, uint32_t seed);

/* calculateBlockSize (src/rowReordering.cu:1009-1025) with an explicit free-memory figure */
uint32_t bsmr_calculate_block_size(const bsmr_csr *m, size_t free_device_bytes);

#define BSMR_ROWS_CLUSTER   0  /* bsa_rowReordering_* (src/rowReordering.cu:1027-1095) */
#define BSMR_ROWS_IDENTITY  1  /* noReorderRow        (src/rowReordering.cu:15-46)     */

/* BSMR(alpha, delta, S) + RPHM(S, bsmr) (src/BSMR.cpp:16-25, 83-265).  block_size 0
 * = calculateBlockSize().  device < 0 builds the host arrays only. */
bsmr_pipeline *bsmr_pipeline_create(const bsmr_csr *m, float alpha, float delta, int row_mode,
                                    uint32_t block_size, int device);
void bsmr_pipeline_free(bsmr_pipeline *p);
/* BSMR::colReordering(delta, S) again on the same row order, then a fresh RPHM. */
int bsmr_pipeline_resplit(bsmr_pipeline *p, const bsmr_csr *m, float delta, int device);

#define BSMR_ARR_REORDERED_ROWS        0
#define BSMR_ARR_DENSE_COLS            1
#define BSMR_ARR_DENSE_COL_OFFSETS     2
#define BSMR_ARR_SPARSE_COLS           3
#define BSMR_ARR_SPARSE_COL_OFFSETS    4
#define BSMR_ARR_SPARSE_VALUE_OFFSETS  5
#define BSMR_ARR_BLOCK_OFFSETS         6
#define BSMR_ARR_BLOCK_VALUES          7
#define BSMR_ARR_SPARSE_VALUES         8
#define BSMR_ARR_SPARSE_RELATIVE_ROWS  9
#define BSMR_ARR_SPARSE_COL_INDICES   10
#define BSMR_ARR_DENSE_ROW_PANEL_IDS  11
#define BSMR_ARR_DENSE_COL_BLOCK_ITERS 12
#define BSMR_ARR_SPARSE_ROW_PANEL_IDS 13
#define BSMR_ARR_SPARSE_COL_BLOCK_ITERS 14
/* Accessors of BSMR (include/BSMR.hpp:38-50) and RPHM (:85-102). */
int bsmr_pipeline_array(const bsmr_pipeline *p, int which, const uint32_t **data, size_t *len);
int bsmr_pipeline_num_row_panels(const bsmr_pipeline *p);
int bsmr_pipeline_num_clusters(const bsmr_pipeline *p);
float bsmr_pipeline_row_reordering_ms(const bsmr_pipeline *p);
float bsmr_pipeline_col_reordering_ms(const bsmr_pipeline *p);
float bsmr_pipeline_rphm_ms(const bsmr_pipeline *p);
/* check_rphm (src/BSMR.cpp:932-953): 1 = all invariants hold */
int bsmr_pipeline_check(const bsmr_pipeline *p, const bsmr_csr *m, float delta);
/* evaluationReordering (src/BSMR.cpp:826-930) +
 * calculateNumDenseBlocksAndAverageDensityInOriginalMatrix (:955-994): the reordering
 * statistics the reference logs per (alpha, delta) - `original_*`, `bsmr_numDenseBlock`,
 * `bsmr_averageDensity`, `bsmr_num{Dense,Sparse}ThreadBlocks`, `bsmr_num{Dense,Sparse}Data`
 * - plus RPHM::maxNumDenseColBlocksInRowPanel (include/BSMR.hpp:124), whose quarter is
 * the reference's `gridDim_dense.y` (src/sddmmKernel.cu:2570-2574). */
typedef struct bsmr_reordering_report {
    int32_t  original_num_dense_blocks;
    float    original_average_density;
    int32_t  num_dense_blocks;
    float    average_density;
    int32_t  num_dense_thread_blocks;
    int32_t  num_sparse_thread_blocks;
    int32_t  num_dense_data;
    int32_t  num_sparse_data;
    uint32_t max_dense_blocks_per_panel;
    uint32_t max_sparse_blocks_per_panel;
} bsmr_reordering_report;
int bsmr_pipeline_evaluate(const bsmr_pipeline *p, const bsmr_csr *m, float delta,
                           bsmr_reordering_report *out);
/* Device plan of the RPHM (NULL when built host-only); status of its creation. */
bsmr_plan *bsmr_pipeline_plan(const bsmr_pipeline *p);
int bsmr_pipeline_plan_status(const bsmr_pipeline *p);

/* sddmm_cpu, CSR overload (src/host.cpp:44-76).  A: rows x K row-major, B: K x cols
 * column-major (column j at B + j*K). */
void bsmr_host_sddmm_cpu(const bsmr_csr *m, uint32_t K, const float *A, const float *B, float *P);
/* checkData (include/checkData.hpp:44-79) without the printing; returns #errors. */
size_t bsmr_host_check_data(size_t n, const float *x, const float *y);

/* The operator itself: sddmm(options, A, B, P, logger) (src/sddmm.cu:10-39) on host
 * operands.  P receives nnz floats in S's CSR order.  If log_buf != NULL the
 * `[key : value]` record (Logger::printLogInformation) is copied into it.
 * Returns 0 on success, a bsmr_hip.h status otherwise. */
int bsmr_host_sddmm(const bsmr_csr *m, uint32_t K, float alpha, float delta, int compute_mode,
                    int num_iterations, const float *A, const float *B, float *P,
                    char *log_buf, size_t log_buf_len);

/* sddmm_multi_gpu (bsmr-sddmm_amd/include/sddmm.hpp; no reference counterpart - the reference is single-GPU): the
 * operator over `num_devices` GPUs of one node from one process - rows cut by cost, the whole pipeline per slice,
 * bsmr_sharded_* underneath (one RCCL gather-v of P per step).  step_ms (may be NULL) receives the device time of one
 * step.  bsmr_partition_rows_by_cost writes the world + 1 row boundaries of that cut. */
int bsmr_host_sddmm_sharded(const bsmr_csr *m, uint32_t K, float alpha, float delta, int compute_mode,
                            int num_iterations, const int *devices, uint32_t num_devices, const float *A,
                            const float *B, float *P, float *step_ms);
/* ... with the step taken apart: times_ms[3] (may be NULL) = {one pipelined step, SDDMM alone (max over devices), gather-v alone} */
int bsmr_host_sddmm_sharded_timed(const bsmr_csr *m, uint32_t K, float alpha, float delta, int compute_mode,
                                  int num_iterations, const int *devices, uint32_t num_devices, const float *A,
                                  const float *B, float *P, float *times_ms);
int bsmr_partition_rows_by_cost(const bsmr_csr *m, uint32_t world, uint32_t *bounds);

#ifdef __cplusplus
}
#endif
#endif /* BSMR_HOST_H */
