#pragma once
// Operator entry points: the reference's include/sddmm.hpp:8-21.

#include "Logger.hpp"
#include "Matrix.hpp"
#include "Options.hpp"

// Reorder (BSMR), build the block format (RPHM), run the SDDMM on the device and
// collect statistics.  On entry matrixP carries S's pattern; on return its
// values are P = (A*B) sampled at that pattern, in S's CSR order.
void sddmm(const Options& options, const Matrix<float>& matrixA, const Matrix<float>& matrixB,
           sparseMatrix::CSR<float>& matrixP, Logger& logger);

// alpha x delta x K sweep of the reference's test mode; appends one record per
// configuration to <logdir>/BSMR_k_<K>_a_<alpha>_d_<delta>.log.
void sddmm_testMode(const Options& options, sparseMatrix::CSR<float>& matrixP);

// Recompute on the CPU and compare with the reference's tolerance.
bool checkSddmm(const Matrix<float>& matrixA, const Matrix<float>& matrixB,
                const sparseMatrix::CSR<float>& matrixS, const sparseMatrix::CSR<float>& matrixP);

// Run-time replacement for the reference's compile-time `#define VALIDATE`
// (src/sddmm.cu:7): when on, sddmm() also runs check_rphm and checkSddmm.
void setSddmmValidate(bool on);

// The same operator over several GPUs of one node, from one process (new - the reference is
// single-GPU).  The rows of S are cut into devices.size() contiguous ranges of equal COST (entries + a panel share per
// non-empty row, partitionRowsByCost), every range runs the whole BSMR pipeline on its own slice, the device side is
// bsmr_sharded_* (include/bsmr_hip.h): shard i on devices[i], B replicated, one RCCL gather-v of P to devices[0].
// logger.sddmmTime_ = one step (SDDMM on every device + gather); logger.status_ = the device status.
void sddmm_multi_gpu(const Options& options, const Matrix<float>& matrixA, const Matrix<float>& matrixB,
                     sparseMatrix::CSR<float>& matrixP, const std::vector<int>& devices, Logger& logger);

// world + 1 row boundaries of contiguous ranges of nearly equal cost, cut at multiples of 16 rows.
std::vector<UIN> partitionRowsByCost(const sparseMatrix::CSR<float>& matrix, int world);
