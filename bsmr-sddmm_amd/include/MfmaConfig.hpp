#pragma once
// Tile geometry and index types of the BSMR-SDDMM engine on gfx950.
//
// Plays the role of the reference's include/TensorCoreConfig.cuh:10-70 (UIN,
// NULL_VALUE, WMMA_M/N/K, operand types).  The reference multiplies TF32
// operands with m16n16k8 WMMA; gfx950 has no TF32, so the dense path here is
// v_mfma_f32_16x16x32_{f16,bf16} (same 16x16 output tile, 32-deep K step) with
// an exact v_mfma_f32_16x16x4_f32 mode.  Panel and block sizes stay 16 so the
// BSMR outputs are identical to the reference's.

#include <cstdint>
#include <limits>

using UIN = uint32_t;
constexpr UIN MAX_UIN = std::numeric_limits<UIN>::max();
constexpr UIN NULL_VALUE = MAX_UIN;

// Output tile of one MFMA and its K step (fp16 / bf16 modes).
constexpr int MFMA_M = 16;
constexpr int MFMA_N = 16;
constexpr int MFMA_K = 32;

// Names kept for log compatibility with the reference's `[WMMA_M : ..]` record.
constexpr int WMMA_M = MFMA_M;
constexpr int WMMA_N = MFMA_N;
constexpr int WMMA_K = MFMA_K;

using MATRIX_A_TYPE = float;
using MATRIX_B_TYPE = float;
using MATRIX_C_TYPE = float;

constexpr int WAVE_SIZE = 64;

// Bytes of LDS the reference assumed when it sized the clustering histogram
// (include/TensorCoreConfig.cuh:14); kept because calculateBlockSize() must
// return the same bin width for the same matrix.
constexpr UIN maxSharedMemoryPerBlock = 49152;
