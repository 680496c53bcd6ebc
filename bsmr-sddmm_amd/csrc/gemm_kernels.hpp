// Output-stationary GEMM engine of the dense part (round 4) for gfx950: format csrc/gemm_format.hpp.
//
// What it computes is the dense part of the reference's SDDMM (reference src/sddmmKernel.cu:213-351; K loop :274-330,
// masked scatter :331-348): for every stored entry (i, j) that the plan keeps on the dense path, P[e] = sum_k A[i,k] B[k,j],
// fp16 / bf16 operands, fp32 accumulation - bit for bit what convertOperands + denseStream produce (same casts, same MFMA
// instruction, the k steps of an entry accumulated in the same order into one accumulator).
//
// One workgroup = one macro-tile of C: TM = 16 PM reordered rows x TN = 16 NB natural columns, 8 waves as 2 (M) x 4 (N),
// wave (wm, wn) owns the m x n = (PM / 2) x (NB / 4) block of 16 x 16 tiles at rows wm TM / 2, columns wn TN / 4 and keeps
// its accumulators (4 m n registers) for the whole K loop.
//   * K is cut into slices of 64.  A slice of the operands - TM rows of A (gathered by row id: a row is K contiguous
//     elements) and TN columns of the column-major B (contiguous) x 64 k = 128-byte rows - is an LDS STAGE; two stages.
//     Every wave issues its share of a stage's LDS-DMA (buffer_load ... lds, 1 KiB = 8 rows per instruction, no registers):
//     PM / 4 + NB / 4 instructions per wave and slice.  The 16-byte pieces of a row are XOR-swizzled on the SOURCE address
//     (piece ^ ((row >> 1) & 7)), the image itself is lane-linear as LDS-DMA requires; the fragment reads apply the same XOR
//     and are bank-conflict free (a ds_read_b128 is served in groups of 16 lanes: rows 0-3, 12-15 at k-group g and rows
//     4-11 at g + 1 land on 16 different 16-byte slots of the 256-byte bank row).
//   * Per slice a wave reads m + n fragments per 32-k step (ds_read_b128) and issues m n MFMAs (v_mfma_f32_16x16x32): one
//     fragment of A feeds n MFMAs, one of B feeds m (the sweep engine of round 3: at most 4 and 1).  The slice is worked off
//     in four quadrants of the wave's block (rows top / bottom x columns left / right), so that 32 + 16 registers hold the
//     fragments; the loads of slice t + 1 are issued before the quadrants of slice t and waited for (counted, this wave's
//     own) behind them; ONE workgroup barrier per slice publishes the landed stage and frees the other one.
//   * Epilogue = the sparse mask: the stages are dead, every wave takes 16 KiB of them as its private slab.  Per pass of 16
//     tiles: accumulators -> slab (ds_write_b128 per tile), then one lane per STORED entry: entry word -> slab value and the
//     row's first index (a table of the macro-tile in LDS) -> one store.  The words of a (wave, pass) list are contiguous
//     and ordered by (row, column): 64 lanes read 256 contiguous bytes and write runs of P.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_format.hpp"
#include "sddmm_kernels.hpp"

namespace bsmr {

constexpr uint32_t kGemmBK = 64;                                   // k per LDS stage
constexpr uint32_t kGemmRowBytes = kGemmBK * 2u;                   // one row / column of a stage
constexpr uint32_t gemmStageBytes(int PM, int NB) { return (uint32_t)(PM + NB) * 16u * kGemmRowBytes; }
constexpr uint32_t kGemmSlabBytes = kGemmPassTiles * 1024u;         // a wave's slab: 16 tiles of 64 lanes x 16 bytes
// two stages, reused as the eight waves' slabs by the epilogue
constexpr uint32_t gemmRingBytes(int PM, int NB) {
    return 2u * gemmStageBytes(PM, NB) > kGemmWaves * kGemmSlabBytes ? 2u * gemmStageBytes(PM, NB) : kGemmWaves * kGemmSlabBytes;
}
constexpr size_t gemmLdsBytes(int PM, int NB) { return gemmRingBytes(PM, NB) + (size_t)PM * 16u * 4u; }

#if defined(BSMR_GEMM_LAB)
#define GEMM_LAB_ARG , uint32_t labSkip   /* bit 0 MFMAs, 1 fragment reads, 2 DMAs, 3 slab writes, 4 entry loads, 5 stores */
#define GEMM_LAB_SKIP(bit) (labSkip & (1u << (bit)))
#else
#define GEMM_LAB_ARG
#define GEMM_LAB_SKIP(bit) false
#endif

// lab builds: a value that feeds a part left out stays computed (the host pass does not know the register constraint)
__device__ __forceinline__ void gemmKeep(const u32x4& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(x));
#else
    (void)x;
#endif
}

// This wave's share of one stage: ADMAS + BDMAS LDS-DMA instructions of 1 KiB (8 rows of 128 bytes), voff = the lane's
// byte offset of its piece in slice 0, kOff = the slice's byte offset inside a row.  (A function of its own: a buffer
// resource cannot be captured by a lambda of a kernel that the host pass instantiates too.)
template <uint32_t ADMAS, uint32_t BDMAS>
__device__ __forceinline__ void gemmStage(const uint16_t* A, uint32_t aBytes, const uint16_t* B, uint32_t bBytes, uint8_t* stage, uint32_t bAt,
                                          uint32_t wave, const uint32_t* voffA, const uint32_t* voffB, uint32_t kOff) {
    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A), 0, aBytes, 0x00020000);
    const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(B), 0, bBytes, 0x00020000);
#pragma unroll
    for (uint32_t j = 0; j < ADMAS; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(stage + (wave * ADMAS + j) * 1024u), 16, voffA[j],
                                                 kOff, 0, 0);
#pragma unroll
    for (uint32_t j = 0; j < BDMAS; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (__attribute__((address_space(3))) void*)(stage + bAt + (wave * BDMAS + j) * 1024u), 16,
                                                 voffB[j], kOff, 0, 0);
}

template <int KT, int PM, int NB, int MODE>
__global__ void __launch_bounds__(kGemmWaves * kWave, 2)
denseGemm(const uint16_t* __restrict__ A16, const uint16_t* __restrict__ B16, uint32_t aBytes, uint32_t bBytes,
          const uint32_t* __restrict__ panelRows, const GemmItem* __restrict__ items, const uint32_t* __restrict__ rowStart,
          const uint32_t* __restrict__ lists, const uint32_t* __restrict__ words, float* __restrict__ P, uint32_t N,
          Batch batch GEMM_LAB_ARG) {
    constexpr uint32_t K = kGemmBK * KT, TM = PM * 16u, TN = NB * 16u;
    constexpr uint32_t m = PM / kGemmWavesM, n = NB / kGemmWavesN, Q = m * n / kGemmPassTiles;
    constexpr uint32_t MH = m / 2 ? m / 2 : 1, NH = n / 2 ? n / 2 : 1;          // tiles per quadrant side
    constexpr uint32_t QM = m / MH, QN = n / NH;                                // quadrants per side (1 or 2)
    constexpr uint32_t ADMAS = PM / 4u, BDMAS = NB / 4u;                        // LDS-DMA instructions per wave and slice
    static_assert(PM % 4 == 0 && NB % 4 == 0, "a stage is filled in whole 1-KiB pieces per wave");
    constexpr uint32_t stageBytes = gemmStageBytes(PM, NB), bAt = TM * kGemmRowBytes;
    constexpr uint32_t rowTableAt = gemmRingBytes(PM, NB);

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t itemId = xcdContiguous(blockIdx.x, gridDim.x);
    const GemmItem item = items[itemId];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t wm = wave / kGemmWavesN, wn = wave % kGemmWavesN;
    const uint32_t r = lane & 15u, g = lane >> 4;

    const uint16_t* Ab = A16 + (size_t)blockIdx.y * batch.strideA;
    const uint16_t* Bb = B16 + (size_t)blockIdx.y * batch.strideB;
    P += (size_t)blockIdx.y * batch.strideP;

    // the macro-tile's table of first indices, one row per thread
    if (threadIdx.x < TM) reinterpret_cast<uint32_t*>(lds + rowTableAt)[threadIdx.x] = rowStart[(size_t)item.rowStartBase * TM + threadIdx.x];

    // my pieces of a stage: DMA instruction i = wave * DMAS + j moves rows 8 i .. 8 i + 7, lane l the 16-byte slot l & 7 of
    // row 8 i + (l >> 3); the piece that belongs in that slot is slot ^ ((row >> 1) & 7)
    uint32_t voffA[ADMAS], voffB[BDMAS];
#pragma unroll
    for (uint32_t j = 0; j < ADMAS; ++j) {
        const uint32_t row = 8u * (wave * ADMAS + j) + (lane >> 3);
        const uint32_t id = panelRows[(size_t)item.group * TM + row];
        voffA[j] = id * (K * 2u) + (((lane & 7u) ^ ((row >> 1) & 7u)) << 4);
    }
#pragma unroll
    for (uint32_t j = 0; j < BDMAS; ++j) {
        const uint32_t col = 8u * (wave * BDMAS + j) + (lane >> 3);
        const uint32_t id = min(item.firstBlock * 16u + col, N - 1u);            // the last block of B may be ragged
        voffB[j] = id * (K * 2u) + (((lane & 7u) ^ ((col >> 1) & 7u)) << 4);
    }
    // fragment addresses: row / column r of a tile, k-group g; piece 4 s + g of the 128-byte row sits at slot (4 s + g) ^ (r >> 1)
    const uint32_t fragOff = r * kGemmRowBytes + ((g ^ (r >> 1)) << 4);        // s = 0; s = 1 is this ^ 64
    const uint32_t aRead = wm * (TM / 2u) * kGemmRowBytes + fragOff;
    const uint32_t bRead = bAt + wn * (TN / 4u) * kGemmRowBytes + fragOff;

    f32x4 acc[m][n];
#pragma unroll
    for (uint32_t i = 0; i < m; ++i)
#pragma unroll
        for (uint32_t j = 0; j < n; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (!GEMM_LAB_SKIP(2)) gemmStage<ADMAS, BDMAS>(Ab, aBytes, Bb, bBytes, lds, bAt, wave, voffA, voffB, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#pragma unroll
    for (uint32_t t = 0; t < (uint32_t)KT; ++t) {
        if (t + 1u < (uint32_t)KT && !GEMM_LAB_SKIP(2))   // slice t + 1 -> the other stage
            gemmStage<ADMAS, BDMAS>(Ab, aBytes, Bb, bBytes, lds + ((t + 1u) & 1u) * stageBytes, bAt, wave, voffA, voffB, (t + 1u) * kGemmRowBytes);
        const uint8_t* base = lds + (t & 1u) * stageBytes;
        // quadrants in the order (top, left) (top, right) (bottom, right) (bottom, left): only one side's fragments change
        // between two quadrants, the other side's stay in registers
        u32x4 af[MH][2], bf[NH][2];
#pragma unroll
        for (uint32_t qi = 0; qi < QM * QN; ++qi) {
            const uint32_t qm = QN == 1 ? qi : qi / 2u, qn = QN == 1 ? 0u : (qi == 1u || qi == 2u ? 1u : 0u);
            const bool newA = qi == 0u || (QN == 1 ? true : qi == 2u), newB = qi != 2u || QN == 1;
            if (!GEMM_LAB_SKIP(1)) {
                if (newB) {
#pragma unroll
                    for (uint32_t j = 0; j < NH; ++j)
#pragma unroll
                        for (uint32_t s = 0; s < 2; ++s)
                            bf[j][s] = *reinterpret_cast<const u32x4*>(base + ((bRead + (qn * NH + j) * 16u * kGemmRowBytes) ^ (s << 6)));
                }
                if (newA) {
#pragma unroll
                    for (uint32_t i = 0; i < MH; ++i)
#pragma unroll
                        for (uint32_t s = 0; s < 2; ++s)
                            af[i][s] = *reinterpret_cast<const u32x4*>(base + ((aRead + (qm * MH + i) * 16u * kGemmRowBytes) ^ (s << 6)));
                }
            } else {
#pragma unroll
                for (uint32_t j = 0; j < NH; ++j) bf[j][0] = bf[j][1] = u32x4{lane, t, j, 1u};
#pragma unroll
                for (uint32_t i = 0; i < MH; ++i) af[i][0] = af[i][1] = u32x4{lane, t, i, 2u};
            }
            if (!GEMM_LAB_SKIP(0)) {
#pragma unroll
                for (uint32_t s = 0; s < 2; ++s)
#pragma unroll
                    for (uint32_t i = 0; i < MH; ++i)
#pragma unroll
                        for (uint32_t j = 0; j < NH; ++j)
                            acc[qm * MH + i][qn * NH + j] = mfma16<MODE>(af[i][s], bf[j][s], acc[qm * MH + i][qn * NH + j]);
            } else {
#pragma unroll
                for (uint32_t i = 0; i < MH; ++i)
#pragma unroll
                    for (uint32_t j = 0; j < NH; ++j) { gemmKeep(af[i][0]); gemmKeep(af[i][1]); gemmKeep(bf[j][0]); gemmKeep(bf[j][1]); }
            }
        }
        // slice t + 1 has landed (this wave's pieces), everybody has read slice t: the stages change roles
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- the sparse mask: slab passes over the wave's tiles
    float* slab = reinterpret_cast<float*>(lds) + wave * (kGemmPassTiles * 256u);
    const uint32_t* rowTable = reinterpret_cast<const uint32_t*>(lds + rowTableAt) + wm * (TM / 2u);
    const uint32_t* myLists = lists + item.listBase + wave * Q;
#pragma unroll
    for (uint32_t q = 0; q < Q; ++q) {
        if (!GEMM_LAB_SKIP(3)) {
#pragma unroll
            for (uint32_t tp = 0; tp < kGemmPassTiles; ++tp) {
                const uint32_t tIdx = q * kGemmPassTiles + tp;
                *reinterpret_cast<f32x4*>(slab + (tp * 64u + lane) * 4u) = acc[tIdx / n][tIdx % n];
            }
        }
        const uint32_t first = myLists[q], last = myLists[q + 1u];
        for (uint32_t e = first; e < last; e += 4u * kWave) {
            uint32_t w[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) w[u] = GEMM_LAB_SKIP(4) ? kGemmNoEntry : words[e + u * kWave + lane];   // (reads past the list stay inside `words`: its slack)
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                if (e + u * kWave + lane < last && w[u] != kGemmNoEntry) {
                    const float val = slab[w[u] & 4095u];
                    const uint32_t dst = rowTable[(w[u] >> 12) & 127u] + (w[u] >> 19);
                    if (!GEMM_LAB_SKIP(5)) P[dst] = val;
                }
            }
        }
    }
}

}  // namespace bsmr
