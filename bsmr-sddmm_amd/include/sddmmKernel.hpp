#pragma once
// Host launchers of the device SDDMM: interface of the reference's
// include/sddmmKernel.cuh:19-39.  They forward to the C ABI (include/bsmr_hip.h).

#include "BSMR.hpp"
#include "Logger.hpp"
#include "Matrix.hpp"

// Dense-path arithmetic for subsequent sddmm_gpu calls of this process
// (BSMR_COMPUTE_F16 / _BF16 / _F32; default F16).  The reference has a
// compile-time switch instead (include/TensorCoreConfig.cuh:17-20).
void setSddmmComputeMode(int mode);
int sddmmComputeMode();

// Host operands in, matrixP.values() out (upload, timed loop, download).
void sddmm_gpu(const Matrix<float>& matrixA, const Matrix<float>& matrixB, const RPHM& rphm,
               sparseMatrix::CSR<float>& matrixP, Logger& logger);

// Device pointers in/out; runs logger.numITER_ timed iterations after one
// warm-up and records logger.sddmmTime_ (ms per SDDMM) and the launch geometry.
void sddmm_gpu(UIN M, UIN N, UIN K, const float* matrixA, const float* matrixB, const RPHM& rphm,
               float* matrixP, Logger& logger);

// K <= 32 entry point of the reference; the HIP kernels handle every K that is
// a multiple of 32 through one path, so this forwards to sddmm_gpu.
void sddmm_gpu_k32(UIN M, UIN N, UIN K, const float* matrixA, const float* matrixB,
                   const RPHM& rphm, float* matrixP, Logger& logger);

// Batched form (include/sddmmKernel.cuh:41-47): numBatch problems over one plan, device operands
// stored back to back (A: [b][M][K], B: [b][N][K], P: [b][nnz]); `time` receives ms per batched call.
void sddmm_gpu_batch(const UIN numBatch, const UIN M, const UIN N, const UIN K, const UIN nnz,
                     const float* matrixA, const float* matrixB, const RPHM& rphm, float* matrixP, float& time);

// out[b] = transpose(in[b]) for numBatches row-major height x width device matrices
// (include/sddmmKernel.cuh:49-51).
void batchedMatrixTranspose(const UIN width, const UIN height, const UIN numBatches, const float* d_input,
                            float* d_output);
