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
//   * Per slice and 32-k step a wave reads m + n fragments (ds_read_b128) and issues m n MFMAs (v_mfma_f32_16x16x32): one
//     fragment of A feeds n MFMAs, one of B feeds m (the sweep engine of round 3: at most 4 and 1).  The loads of slice
//     t + 1 are issued at the top of slice t and waited for (this wave's own) behind its MFMAs; ONE workgroup barrier per
//     slice publishes the landed stage and frees the other one.  (What hipcc makes of it, and it is what one wants: the
//     fragment reads of slice t stay in front of the barrier, its MFMAs move behind it - register-only instructions are not
//     ordered by the barrier - and run beside the DMA issue and the fragment reads of slice t + 1.)
//   * Epilogue = the sparse mask: the stages are dead, every wave takes 16 KiB of them as its private slab.  Per pass of 16
//     tiles: accumulators -> slab (ds_write_b128 per tile), then one lane per STORED entry: entry word -> slab value and the
//     row's first index (a table of the macro-tile in LDS) -> one store.  The words of a (wave, pass) list are contiguous
//     and ordered by (row, column): 64 lanes read 256 contiguous bytes and write runs of P.  Entry words are read once per
//     launch, i.e. from HBM: the first eight batches of a pass are requested one phase ahead (pass 0's in front of the last
//     slice's MFMAs).
//   * SRC32: the caller's fp32 operands, no conversion pass (K <= 128).  A stage then holds 32 k (rows of 128 bytes again:
//     same DMA shape, K / 32 slices); a fragment is two ds_read_b128 (8 consecutive k as fp32) rounded in registers with the
//     casts of convertOperands, so the MFMA operands - and P - are bit for bit those of conversion pass + 16-bit kernel.
//     The XOR of the 16-byte pieces is another one (a lane's two pieces are neighbours; the 16 lanes a read is served in
//     hold rows 0-3, 12-15 at pieces 2 g, 2 g + 1 and rows 4-11 at 2 g + 2, 2 g + 3): piece ^ swz32(row), below.
// Measured (tools/probes/gemm_probe.hip, MI355X): 4096^2 Bernoulli(0.1), K = 512, bf16: the K loop runs at 1.5 us per slice
// = 1.4 PFLOP/s executed, the rate of the guide's 256^2 8-phase GEMM template on random data; what is left is the fixed
// part (launch, first stage, epilogue).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_format.hpp"
#include "sddmm_kernels.hpp"

namespace bsmr {

constexpr uint32_t kGemmBK = 64;                                   // k per LDS stage (16-bit operands; fp32 operands: 32)
constexpr uint32_t kGemmRowBytes = kGemmBK * 2u;                   // one row / column of a stage: 128 bytes either way
// the XOR that spreads a row's eight 16-byte pieces over the LDS banks, 16-bit and fp32 stages
__host__ __device__ constexpr uint32_t gemmSwz16(uint32_t row) { return (row >> 1) & 7u; }
__host__ __device__ constexpr uint32_t gemmSwz32(uint32_t row) { return ((row >> 1) & 1u) | (((row >> 3) & 1u) * 6u); }
constexpr uint32_t gemmStageBytes(int PM, int NB) { return (uint32_t)(PM + NB) * 16u * kGemmRowBytes; }
constexpr uint32_t kGemmSlabBytes = kGemmPassTiles * 1024u;         // a wave's slab: 16 tiles of 64 lanes x 16 bytes
// two stages, reused as the eight waves' slabs by the epilogue
constexpr uint32_t gemmRingBytes(int PM, int NB) {
    return 2u * gemmStageBytes(PM, NB) > kGemmWaves * kGemmSlabBytes ? 2u * gemmStageBytes(PM, NB) : kGemmWaves * kGemmSlabBytes;
}
constexpr size_t gemmLdsBytes(int PM, int NB) { return gemmRingBytes(PM, NB) + (size_t)PM * 16u * 4u; }

#if defined(BSMR_GEMM_LAB)
// labSkip: bit 0 MFMAs, 1 fragment reads (denseGemmCvt: the rounding step), 2 DMAs, 3 slab writes, 4 entry loads, 5 stores,
// 6 denseGemmCvt's fragment reads.  labStamps (may be null): per (workgroup, wave) eight readings of the 100 MHz clock -
// 0 entry, 1 first stage requested, 2 own share of it landed, 3 first barrier passed, 4 K loop left, 5 + q pass q done.
#define GEMM_LAB_ARG , uint32_t labSkip, unsigned long long* labStamps
#define GEMM_LAB_SKIP(bit) (labSkip & (1u << (bit)))
#define GEMM_LAB_STAMPS unsigned long long labStamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define GEMM_LAB_STAMP(i) labStamp[i] = __builtin_amdgcn_s_memrealtime()
#define GEMM_LAB_STAMPS_OUT                                                                              \
    if (labStamps && lane == 0)                                                                          \
        for (uint32_t i = 0; i < 8u; ++i) labStamps[((size_t)blockIdx.x * kGemmWaves + wave) * 8u + i] = labStamp[i]
#else
#define GEMM_LAB_ARG
#define GEMM_LAB_SKIP(bit) false
#define GEMM_LAB_STAMPS
#define GEMM_LAB_STAMP(i)
#define GEMM_LAB_STAMPS_OUT
#endif

// lab builds: a value that feeds a part left out stays computed (the host pass does not know the register constraint)
__device__ __forceinline__ void gemmKeep(const u32x4& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(x));
#else
    (void)x;
#endif
}

// The epilogue's dump of one accumulator tile into the wave's slab (gemmSlabSlot: register j of the 64 lanes = the 256 bytes at
// 256 j).  ds_write_addtid_b32 stores 4 bytes per lane at M0[15:0] + offset + 4 lane in 2 cycles per wave-instruction - four of
// them 8 cycles where one ds_write_b128 takes 13 (MI355X_MICROARCH.md, LDS): the slab passes are bound by exactly this
// transfer.  M0 holds 16 bits: tiles at 64 KiB and above (the slabs of waves 4 - 7) are reached through the immediate
// (kGemmDumpHigh + 256 j <= 65532).  One wait state between the write of M0 and its use (the assembler does not see into
// the statement); LDS operations of a wave complete in order, so the entry reads that follow need no wait of their own.
constexpr uint32_t kGemmDumpHigh = 64764u;
__device__ __forceinline__ void gemmDumpTile(const f32x4& a, uint32_t tileAddr, bool high) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    if (!high)
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tds_write_addtid_b32 %0\n\tds_write_addtid_b32 %1 offset:256\n\t"
                     "ds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768"
                     ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "s"(tileAddr) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:64764\n\tds_write_addtid_b32 %1 offset:65020\n\t"
                     "ds_write_addtid_b32 %2 offset:65276\n\tds_write_addtid_b32 %3 offset:65532"
                     ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "s"(tileAddr - kGemmDumpHigh) : "memory", "m0");
#pragma clang diagnostic pop
#else
    (void)a; (void)tileAddr; (void)high;
#endif
}
// LDS byte address of a pointer into the dynamic allocation
__device__ __forceinline__ uint32_t gemmLdsAddress(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// This wave's share of one stage: ADMAS + BDMAS LDS-DMA instructions of 1 KiB (8 rows of 128 bytes), voff = the lane's
// byte offset of its piece in slice 0, kOff = the slice's byte offset inside a row.  (A function of its own: a buffer
// resource cannot be captured by a lambda of a kernel that the host pass instantiates too.)
template <uint32_t DMAS>
__device__ __forceinline__ void gemmStagePart(const uint8_t* src, uint32_t srcBytes, uint8_t* dst, uint32_t wave, const uint32_t* voff, uint32_t kOff) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, srcBytes, 0x00020000);
#pragma unroll
    for (uint32_t j = 0; j < DMAS; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (wave * DMAS + j) * 1024u), 16, voff[j], kOff, 0, 0);
}
// The epilogue's store of one entry: a buffer store whose byte offset is out of range for the lanes without an entry (the
// hardware drops those), so that the pass is straight-line code.  Under `if (entry) P[dst] = val` every batch of 64 entries
// was a branch around a store: the compiler could not count the stores between a batch of entry words and its use any
// more, waited with vmcnt(2), (1), (0) for words that had landed long before - and with them for the acknowledgements of
// the stores issued in between (stores and loads share the counter on gfx9).
#if !defined(BSMR_GEMM_BRANCHY_STORES)
__device__ __forceinline__ void gemmStoreEntry(float* P, bool live, uint32_t index, float val) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(P, 0, 0xFFFFFFFC, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, val), rsrc, live ? index << 2 : 0xFFFFFFFFu, 0, 0);
}
#endif
template <uint32_t ADMAS, uint32_t BDMAS>
__device__ __forceinline__ void gemmStage(const uint8_t* A, uint32_t aBytes, const uint8_t* B, uint32_t bBytes, uint8_t* stage, uint32_t bAt,
                                          uint32_t wave, const uint32_t* voffA, const uint32_t* voffB, uint32_t kOff) {
    gemmStagePart<BDMAS>(B, bBytes, stage + bAt, wave, voffB, kOff);
    gemmStagePart<ADMAS>(A, aBytes, stage, wave, voffA, kOff);
}
// Row id of this lane's row in a chunk of 8 rows whose ids lie at `rows8` (a wave-uniform address): eight SCALAR loads and a
// select.  A vector load here would sit in the same in-order counter as the LDS-DMAs issued before it, and hipcc waits for
// everything (vmcnt(0)) at its first use: the first stage's B columns, which do not depend on the row ids, would have to land
// before the A rows are even requested.  Scalar loads count on lgkmcnt.
__device__ __forceinline__ uint32_t gemmRowOfLane(const uint32_t* __restrict__ rows8, uint32_t sub) {
    uint32_t id = rows8[0];
#pragma unroll
    for (uint32_t k = 1; k < 8; ++k) id = sub == k ? rows8[k] : id;
    return id;
}

// grid: numItems x batches.  gridG / gridS: row groups / column strips of the format; fullGrid != 0: every macro-tile is an
// item and item i's place follows from i alone (gemmItemPlace), so nothing waits for the item record.
// KT: slices of K (64 k each on 16-bit operands, 32 k each on fp32 operands).
template <int KT, int PM, int NB, int MODE, bool SRC32 = false>
__global__ void __launch_bounds__(kGemmWaves * kWave, 2)
denseGemm(const void* __restrict__ Aop, const void* __restrict__ Bop, uint32_t aBytes, uint32_t bBytes,
          const uint32_t* __restrict__ panelRows, const uint32_t* __restrict__ colOf, const GemmItem* __restrict__ items,
          const uint32_t* __restrict__ rowStart,
          const uint32_t* __restrict__ lists, const uint32_t* __restrict__ words, float* __restrict__ P, uint32_t N,
          uint32_t gridG, uint32_t gridS, uint32_t fullGrid, Batch batch GEMM_LAB_ARG) {
    constexpr uint32_t ESZ = SRC32 ? 4u : 2u, K = (kGemmRowBytes / ESZ) * KT, TM = PM * 16u, TN = NB * 16u;
    constexpr uint32_t KSUB = SRC32 ? 1u : 2u;                                  // 32-k MFMA steps per slice
    constexpr uint32_t m = PM / kGemmWavesM, n = NB / kGemmWavesN, Q = (m * n + kGemmPassTiles - 1u) / kGemmPassTiles, L = kGemmWaves * Q;
    constexpr uint32_t ADMAS = PM / 4u, BDMAS = NB / 4u;                        // LDS-DMA instructions per wave and slice
    static_assert(PM % 4 == 0 && NB % 4 == 0, "a stage is filled in whole 1-KiB pieces per wave");
    constexpr uint32_t stageBytes = gemmStageBytes(PM, NB), bAt = TM * kGemmRowBytes;
    constexpr uint32_t rowTableAt = gemmRingBytes(PM, NB);
    // batches of 64 entry words requested together (fewer where the accumulators leave few registers)
    constexpr uint32_t kGemmWordChunk = m * n > 32u ? 4u : 8u;
    // how the epilogue dumps the accumulators (below): the 8 x 5 tile blocks of this kernel keep to plain 4-byte stores
    constexpr bool kPlainDump = m * n > 32u;

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t itemId = xcdContiguous(blockIdx.x, gridDim.x);
    uint32_t group, firstBlock, listBase;
    if (fullGrid) {
        gemmItemPlace(itemId, gridG, gridS, group, firstBlock);
        firstBlock *= (uint32_t)NB;
        listBase = itemId * (L + 1u);
    } else {
        const GemmItem item = items[itemId];
        group = item.group;
        firstBlock = item.firstBlock;
        listBase = item.listBase;
    }
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t wm = wave / kGemmWavesN, wn = wave % kGemmWavesN;
    const uint32_t r = lane & 15u, g = lane >> 4;
    GEMM_LAB_STAMPS;
    GEMM_LAB_STAMP(0);

    const uint8_t* Ab = static_cast<const uint8_t*>(Aop) + (size_t)blockIdx.y * batch.strideA * ESZ;
    const uint8_t* Bb = static_cast<const uint8_t*>(Bop) + (size_t)blockIdx.y * batch.strideB * ESZ;
    P += (size_t)blockIdx.y * batch.strideP;

    // my pieces of a stage: DMA instruction i = wave * DMAS + j moves rows 8 i .. 8 i + 7, lane l the 16-byte slot l & 7 of
    // row 8 i + (l >> 3); the piece that belongs in that slot is slot ^ swz(row)
    uint32_t voffA[ADMAS], voffB[BDMAS];
#pragma unroll
    for (uint32_t j = 0; j < BDMAS; ++j) {
        const uint32_t col = 8u * (wave * BDMAS + j) + (lane >> 3);
        const uint32_t id = gemmRowOfLane(colOf + (size_t)firstBlock * 16u + 8u * (wave * BDMAS + j), lane >> 3);   // the column in this slot
        voffB[j] = id * (K * ESZ) + (((lane & 7u) ^ (SRC32 ? gemmSwz32(col) : gemmSwz16(col))) << 4);
    }
    if (!GEMM_LAB_SKIP(2)) gemmStagePart<BDMAS>(Bb, bBytes, lds + bAt, wave, voffB, 0u);
#pragma unroll
    for (uint32_t j = 0; j < ADMAS; ++j) {
        const uint32_t row = 8u * (wave * ADMAS + j) + (lane >> 3);
        const uint32_t id = gemmRowOfLane(panelRows + (size_t)group * TM + 8u * (wave * ADMAS + j), lane >> 3);
        voffA[j] = id * (K * ESZ) + (((lane & 7u) ^ (SRC32 ? gemmSwz32(row) : gemmSwz16(row))) << 4);
    }
    if (!GEMM_LAB_SKIP(2)) gemmStagePart<ADMAS>(Ab, aBytes, lds, wave, voffA, 0u);
    GEMM_LAB_STAMP(1);
    // the macro-tile's table of first indices, one row per thread
    if (threadIdx.x < TM) reinterpret_cast<uint32_t*>(lds + rowTableAt)[threadIdx.x] = rowStart[(size_t)itemId * TM + threadIdx.x];
    // entry lists of this wave: words[myList[q] .. myList[q + 1]) is pass q's
    uint32_t myList[Q + 1];
#pragma unroll
    for (uint32_t q = 0; q <= Q; ++q) myList[q] = lists[listBase + wave * Q + q];

    // fragment addresses: row / column r of a tile, k-group g.  16-bit: piece 4 s + g of the 128-byte row sits at slot
    // (4 s + g) ^ (r >> 1) (s = 1 is s = 0's address ^ 64).  fp32: pieces 2 g and 2 g + 1 at (2 g) ^ swz32(r) and that ^ 1.
    const uint32_t fragOff = r * kGemmRowBytes + ((SRC32 ? (2u * g) ^ gemmSwz32(r) : g ^ gemmSwz16(r)) << 4);
    const uint32_t aRead = wm * (TM / 2u) * kGemmRowBytes + fragOff;
    const uint32_t bRead = bAt + wn * (TN / 4u) * kGemmRowBytes + fragOff;

    f32x4 acc[m][n];
#pragma unroll
    for (uint32_t i = 0; i < m; ++i)
#pragma unroll
        for (uint32_t j = 0; j < n; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // entry words, kGemmWordChunk batches of 64 at a time (reads past a list stay inside `words`: its slack)
    auto loadWords = [&](uint32_t e0, uint32_t (&w)[kGemmWordChunk]) {
#pragma unroll
        for (uint32_t u = 0; u < kGemmWordChunk; ++u) w[u] = GEMM_LAB_SKIP(4) ? kGemmNoEntry : words[e0 + u * kWave + lane];
    };
    // the first words of the passes' lists are requested two passes ahead where the registers allow it (a pass is shorter
    // than the words' way from HBM: 0.1-0.2 us per launch at 256 x 256; the 8 x 5 tile blocks have no registers left for
    // it - 3 to 8 spilled, 0.25 us lost - and stay one pass ahead)
    constexpr bool kDeepWords = Q >= 2u && m * n <= 32u;
    uint32_t wAhead[2][kGemmWordChunk];

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GEMM_LAB_STAMP(2);
    __builtin_amdgcn_s_barrier();
    GEMM_LAB_STAMP(3);
    // pass 0's first entry words: an HBM round trip (they are read once per launch).  A short K loop does not cover it from
    // the top of its last slice, so up to four slices request them before the loop and carry the registers through it.
    constexpr bool kEarlyWords = KT <= 4;
    if (kEarlyWords) {
        loadWords(myList[0], wAhead[0]);
        if (kDeepWords) loadWords(myList[1], wAhead[1]);
    }

#pragma unroll
    for (uint32_t t = 0; t < (uint32_t)KT; ++t) {
        if (t + 1u < (uint32_t)KT && !GEMM_LAB_SKIP(2))   // slice t + 1 -> the other stage
            gemmStage<ADMAS, BDMAS>(Ab, aBytes, Bb, bBytes, lds + ((t + 1u) & 1u) * stageBytes, bAt, wave, voffA, voffB, (t + 1u) * kGemmRowBytes);
        if (!kEarlyWords && t + 1u == (uint32_t)KT) {   // ... a long loop: behind its last slice
            loadWords(myList[0], wAhead[0]);
            if (kDeepWords) loadWords(myList[1], wAhead[1]);
        }
        const uint8_t* base = lds + (t & 1u) * stageBytes;
#pragma unroll
        for (uint32_t s = 0; s < KSUB; ++s) {
            // fragment of row / column tile `tile` at base address `at`: 16-bit operands as they lie; fp32 operands rounded here
            auto fragment = [&](uint32_t at, uint32_t tile, uint32_t tag) -> u32x4 {
                if (GEMM_LAB_SKIP(1)) return u32x4{lane, t, tile, s + tag};
                const uint8_t* src = base + ((at + tile * 16u * kGemmRowBytes) ^ (s << 6));
                if constexpr (SRC32) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(src) ^ 16u));
                    return packLowp<MODE>(lo, hi);
                } else {
                    return *reinterpret_cast<const u32x4*>(src);
                }
            };
            u32x4 bf[n];
#pragma unroll
            for (uint32_t j = 0; j < n; ++j) bf[j] = fragment(bRead, j, 0u);
#pragma unroll
            for (uint32_t i = 0; i < m; ++i) {
                const u32x4 af = fragment(aRead, i, 2u);
                if (!GEMM_LAB_SKIP(0)) {
#pragma unroll
                    for (uint32_t j = 0; j < n; ++j) acc[i][j] = mfma16<MODE>(af, bf[j], acc[i][j]);
                } else {
                    gemmKeep(af);
#pragma unroll
                    for (uint32_t j = 0; j < n; ++j) gemmKeep(bf[j]);
                }
            }
        }
        // slice t + 1 has landed (this wave's pieces), everybody has read slice t: the stages change roles
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- the sparse mask: slab passes over the wave's tiles
    GEMM_LAB_STAMP(4);
    float* slab = reinterpret_cast<float*>(lds) + wave * (kGemmPassTiles * 256u);
    const uint32_t slabAddr = __builtin_amdgcn_readfirstlane(gemmLdsAddress(slab));
    const uint32_t* rowTable = reinterpret_cast<const uint32_t*>(lds + rowTableAt) + wm * (TM / 2u);
#pragma unroll
    for (uint32_t q = 0; q < Q; ++q) {
        if (!GEMM_LAB_SKIP(3)) {
            if (kPlainDump) {
                // (the two forms of the ds_write_addtid dump below, laid out one after the other, cost the 8 x 5 tile blocks of
                // this kernel 5 to 14 spilled registers - 256 x 320 at K = 512: 21.5 us against 20.3 with one plain 4-byte
                // store per register, which keeps the slab's layout; denseGemmCvt has the registers)
#pragma unroll
                for (uint32_t tp = 0; tp < kGemmPassTiles; ++tp) {
                    const uint32_t tIdx = q * kGemmPassTiles + tp;
                    if (tIdx < m * n) {
#pragma unroll
                        for (uint32_t j = 0; j < 4u; ++j) slab[tp * 256u + j * 64u + lane] = acc[tIdx / n][tIdx % n][j];
                    }
                }
            } else if (slabAddr < 0x10000u) {
#pragma unroll
                for (uint32_t tp = 0; tp < kGemmPassTiles; ++tp) {
                    const uint32_t tIdx = q * kGemmPassTiles + tp;
                    if (tIdx < m * n) gemmDumpTile(acc[tIdx / n][tIdx % n], slabAddr + tp * 1024u, false);
                }
            } else {
#pragma unroll
                for (uint32_t tp = 0; tp < kGemmPassTiles; ++tp) {
                    const uint32_t tIdx = q * kGemmPassTiles + tp;
                    if (tIdx < m * n) gemmDumpTile(acc[tIdx / n][tIdx % n], slabAddr + tp * 1024u, true);
                }
            }
        }
        const uint32_t first = myList[q], last = myList[q + 1u];
        uint32_t w[kGemmWordChunk];
#pragma unroll
        for (uint32_t u = 0; u < kGemmWordChunk; ++u) w[u] = wAhead[q & 1u][u];
        if (kDeepWords) {
            if (q + 2u < Q) loadWords(myList[q + 2u], wAhead[q & 1u]);
        } else if (q + 1u < Q) {
            loadWords(last, wAhead[(q + 1u) & 1u]);                   // the next pass's first words (lists follow each other)
        }
        for (uint32_t e = first; e < last; e += kGemmWordChunk * kWave) {
            if (e != first) loadWords(e, w);                         // (a list of more than 512 words: rare)
#pragma unroll
            for (uint32_t u = 0; u < kGemmWordChunk; ++u) {
#if defined(BSMR_GEMM_BRANCHY_STORES)   // lab: the form with a branch per batch
                if (e + u * kWave + lane < last && w[u] != kGemmNoEntry) {
                    const float val = slab[w[u] & 4095u];
                    const uint32_t dst = rowTable[(w[u] >> 12) & 127u] + (w[u] >> 19);
                    if (!GEMM_LAB_SKIP(5)) P[dst] = val;
                }
#else
                // (a padding word reads slot 4095 and row TM / 2 - 1: inside the slab and the table)
                const bool live = e + u * kWave + lane < last && w[u] != kGemmNoEntry;
                const float val = slab[w[u] & 4095u];
                const uint32_t dst = rowTable[(w[u] >> 12) & (TM / 2u - 1u)] + (w[u] >> 19);
                if (!GEMM_LAB_SKIP(5)) gemmStoreEntry(P, live, dst, val);
#endif
            }
        }
        GEMM_LAB_STAMP(5u + q);
    }
    GEMM_LAB_STAMPS_OUT;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same macro-tile kernel on the caller's fp32 operands with ONE shared rounding per element (K = 64, 128).
//
// denseGemm<..., SRC32> lets every wave round the fragments it reads: the four waves of a row half round the same rows of A,
// the two waves of a column quarter the same columns of B - 2.9 times the casts the operands need, all of them in front of
// the MFMAs of an in-order wave (measured, nips-like K = 128: 13.4 us against 9.1 us for the kernel on 16-bit copies).  Here
// a slice of 32 k goes through two LDS images:
//   F  (TM + TN) rows x 128 bytes of fp32, filled by LDS-DMA, lane-linear (piece = slot: nothing reads it by fragment).  A
//      wave's share of F is PRIVATE to it: the wave that issued the DMAs of a 1-KiB chunk is the one that rounds it, so
//      its own counted wait orders everything and no barrier guards F;
//   H  two stages of (TM + TN) rows x 64 bytes of fp16 / bf16 (rows of 32 k), written by the rounding step - 64 lanes read
//      1 KiB of F linearly, round with the casts of convertOperands, write 8 bytes each - and read as MFMA fragments
//      (ds_read_b128 of piece g of row r at slot g ^ swzH(r): conflict-free for the 16-lane groups a read is served in,
//      as are the 8-byte writes).
// Per slice: MFMAs of slice t from H[t & 1]; wait for the own DMAs of slice t + 1, round them into H[(t + 1) & 1], re-issue
// the DMAs of slice t + 2 into the same chunks of F, ONE barrier (publishes H[(t + 1) & 1]; everybody's reads of it, two
// slices ago, ended before the barrier in between).  Same casts, same MFMA, same order of k steps: bit-identical to the
// conversion pass + 16-bit kernels.
constexpr uint32_t kGemmHalfRowBytes = 64u;                        // a row of an H stage: 32 k of 16 bits
__host__ __device__ constexpr uint32_t gemmSwzH(uint32_t row) { return (4u - ((row >> 2) & 3u)) & 3u; }   // 0, 3, 2, 1
constexpr uint32_t gemmCvtRingBytes(int PM, int NB) {
    const uint32_t used = (uint32_t)(PM + NB) * 16u * (kGemmRowBytes + 2u * kGemmHalfRowBytes);
    return used > kGemmWaves * kGemmSlabBytes ? used : kGemmWaves * kGemmSlabBytes;
}
constexpr size_t gemmCvtLdsBytes(int PM, int NB) { return gemmCvtRingBytes(PM, NB) + (size_t)PM * 16u * 4u; }

template <int KT, int PM, int NB, int MODE>
__global__ void __launch_bounds__(kGemmWaves * kWave, 2)
denseGemmCvt(const float* __restrict__ Aop, const float* __restrict__ Bop, uint32_t aBytes, uint32_t bBytes,
             const uint32_t* __restrict__ panelRows, const uint32_t* __restrict__ colOf, const GemmItem* __restrict__ items,
             const uint32_t* __restrict__ rowStart,
             const uint32_t* __restrict__ lists, const uint32_t* __restrict__ words, float* __restrict__ P, uint32_t N,
             uint32_t gridG, uint32_t gridS, uint32_t fullGrid, Batch batch GEMM_LAB_ARG) {
    constexpr uint32_t K = 32u * KT, TM = PM * 16u, TN = NB * 16u, ROWS = TM + TN;
    constexpr uint32_t m = PM / kGemmWavesM, n = NB / kGemmWavesN, Q = (m * n + kGemmPassTiles - 1u) / kGemmPassTiles, L = kGemmWaves * Q;
    constexpr uint32_t ADMAS = PM / 4u, BDMAS = NB / 4u, DMAS = ADMAS + BDMAS;   // 1-KiB chunks of F per wave: 8 rows each
    constexpr uint32_t fBytes = ROWS * kGemmRowBytes, hBytes = ROWS * kGemmHalfRowBytes, bAtF = TM * kGemmRowBytes, bAtH = TM * kGemmHalfRowBytes;
    constexpr uint32_t hAt = fBytes, rowTableAt = gemmCvtRingBytes(PM, NB);
    constexpr uint32_t kGemmWordChunk = m * n > 32u ? 4u : 8u;

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t itemId = xcdContiguous(blockIdx.x, gridDim.x);
    uint32_t group, firstBlock, listBase;
    if (fullGrid) {
        gemmItemPlace(itemId, gridG, gridS, group, firstBlock);
        firstBlock *= (uint32_t)NB;
        listBase = itemId * (L + 1u);
    } else {
        const GemmItem item = items[itemId];
        group = item.group;
        firstBlock = item.firstBlock;
        listBase = item.listBase;
    }
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t wm = wave / kGemmWavesN, wn = wave % kGemmWavesN;
    const uint32_t r = lane & 15u, g = lane >> 4;
    GEMM_LAB_STAMPS;
    GEMM_LAB_STAMP(0);
    const uint8_t* Ab = reinterpret_cast<const uint8_t*>(Aop) + (size_t)blockIdx.y * batch.strideA * 4u;
    const uint8_t* Bb = reinterpret_cast<const uint8_t*>(Bop) + (size_t)blockIdx.y * batch.strideB * 4u;
    P += (size_t)blockIdx.y * batch.strideP;

    // my chunks of F: chunk c = wave * DMAS + j holds rows 8 c .. 8 c + 7 (A's chunks first, then B's), lane l the piece l & 7
    // (k = 4 (l & 7) .. + 3) of row 8 c + (l >> 3); where the rounded piece goes in H: row * 64 + 16 (slot) + 8 (half)
    uint32_t voffA[ADMAS], voffB[BDMAS];
    const uint32_t piece = lane & 7u;
#pragma unroll
    for (uint32_t j = 0; j < BDMAS; ++j)
        voffB[j] = gemmRowOfLane(colOf + (size_t)firstBlock * 16u + 8u * (wave * BDMAS + j), lane >> 3) * (K * 4u) + (piece << 4);
    if (!GEMM_LAB_SKIP(2)) gemmStagePart<BDMAS>(Bb, bBytes, lds + bAtF, wave, voffB, 0u);
#pragma unroll
    for (uint32_t j = 0; j < ADMAS; ++j)
        voffA[j] = gemmRowOfLane(panelRows + (size_t)group * TM + 8u * (wave * ADMAS + j), lane >> 3) * (K * 4u) + (piece << 4);
    if (!GEMM_LAB_SKIP(2)) gemmStagePart<ADMAS>(Ab, aBytes, lds, wave, voffA, 0u);
    GEMM_LAB_STAMP(1);
    if (threadIdx.x < TM) reinterpret_cast<uint32_t*>(lds + rowTableAt)[threadIdx.x] = rowStart[(size_t)itemId * TM + threadIdx.x];
    uint32_t myList[Q + 1];
#pragma unroll
    for (uint32_t q = 0; q <= Q; ++q) myList[q] = lists[listBase + wave * Q + q];

    // rounding step: my chunks of F -> stage `to` of H.  The reads of a batch of chunks are issued together, then rounded and
    // written: written chunk by chunk (read, round, write, read ...) every read waited for the write in front of it - the
    // compiler cannot know that F and H do not overlap - and a wave paid nine LDS round trips in a row per slice.
#if defined(BSMR_GEMM_ROUND_SERIAL)
    constexpr uint32_t kRoundBatch = 1u;      // lab: one chunk at a time
#else
    constexpr uint32_t kRoundBatch = 5u;
#endif
    // (addresses: a chunk's place in F is one register per operand plus j KiB; its rows' places in H are two registers per
    // operand - the swizzle of row 8 c + (lane >> 3) depends on c's parity only - plus j * 512 + the stage: nine source and
    // nine destination registers, live over the whole kernel, were what the 8 x 5 tile blocks had no room for)
    const uint32_t fOfA = wave * ADMAS * 1024u + lane * 16u, fOfB = bAtF + wave * BDMAS * 1024u + lane * 16u;
    auto hOf = [&](uint32_t part, uint32_t firstChunk, uint32_t odd) {
        const uint32_t quarter = (2u * (firstChunk + odd) + (lane >> 5)) & 3u;            // (row >> 2) & 3 of the lane's row
        return hAt + part + firstChunk * 512u + (lane >> 3) * kGemmHalfRowBytes + (((piece >> 1) ^ ((4u - quarter) & 3u)) << 4) + ((piece & 1u) << 3);
    };
    const uint32_t hOfA[2] = {hOf(0u, wave * ADMAS, 0u), hOf(0u, wave * ADMAS, 1u)};
    const uint32_t hOfB[2] = {hOf(bAtH, wave * BDMAS, 0u), hOf(bAtH, wave * BDMAS, 1u)};
    auto roundChunks = [&](uint32_t to) {
        if (GEMM_LAB_SKIP(1)) return;
#pragma unroll
        for (uint32_t j0 = 0; j0 < DMAS; j0 += kRoundBatch) {
            f32x4 v[kRoundBatch];
#pragma unroll
            for (uint32_t b = 0; b < kRoundBatch; ++b) {
                const uint32_t j = j0 + b;
                if (j >= DMAS) continue;
                v[b] = j < ADMAS ? *reinterpret_cast<const f32x4*>(lds + fOfA + j * 1024u)
                                 : *reinterpret_cast<const f32x4*>(lds + fOfB + (j - ADMAS) * 1024u);
            }
#pragma unroll
            for (uint32_t b = 0; b < kRoundBatch; ++b) {
                const uint32_t j = j0 + b;
                if (j >= DMAS) continue;
                const uint32_t jj = j < ADMAS ? j : j - ADMAS;
                uint8_t* dst = lds + (j < ADMAS ? hOfA[jj & 1u] : hOfB[jj & 1u]) + jj * 512u + to * hBytes;
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                u32x2 h;
                if constexpr (MODE == 0) {
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    f16x4 o;
                    o[0] = (_Float16)v[b][0]; o[1] = (_Float16)v[b][1]; o[2] = (_Float16)v[b][2]; o[3] = (_Float16)v[b][3];
                    h = __builtin_bit_cast(u32x2, o);
                } else {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 o;
                    o[0] = (__bf16)v[b][0]; o[1] = (__bf16)v[b][1]; o[2] = (__bf16)v[b][2]; o[3] = (__bf16)v[b][3];
                    h = __builtin_bit_cast(u32x2, o);
                }
                *reinterpret_cast<u32x2*>(dst) = h;
            }
        }
    };

    // fragment addresses in an H stage: piece g of row / column r at slot g ^ swzH(r)
    const uint32_t fragOff = r * kGemmHalfRowBytes + ((g ^ gemmSwzH(r)) << 4);
    const uint32_t aRead = wm * (TM / 2u) * kGemmHalfRowBytes + fragOff;
    const uint32_t bRead = bAtH + wn * (TN / 4u) * kGemmHalfRowBytes + fragOff;

    f32x4 acc[m][n];
#pragma unroll
    for (uint32_t i = 0; i < m; ++i)
#pragma unroll
        for (uint32_t j = 0; j < n; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto loadWords = [&](uint32_t e0, uint32_t (&w)[kGemmWordChunk]) {
#pragma unroll
        for (uint32_t u = 0; u < kGemmWordChunk; ++u) w[u] = GEMM_LAB_SKIP(4) ? kGemmNoEntry : words[e0 + u * kWave + lane];
    };
    // the first words of the passes' lists are requested two passes ahead where the registers allow it (a pass is shorter
    // than the words' way from HBM: 0.1-0.2 us per launch at 256 x 256; the 8 x 5 tile blocks have no registers left for
    // it - 3 to 8 spilled, 0.25 us lost - and stay one pass ahead)
    constexpr bool kDeepWords = Q >= 2u && m * n <= 32u;
    uint32_t wAhead[2][kGemmWordChunk];

    // slice 0: landed (my chunks), rounded into H[0], slice 1 requested, H[0] published
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GEMM_LAB_STAMP(2);
    roundChunks(0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (KT > 1 && !GEMM_LAB_SKIP(2)) gemmStage<ADMAS, BDMAS>(Ab, aBytes, Bb, bBytes, lds, bAtF, wave, voffA, voffB, kGemmRowBytes);
    __builtin_amdgcn_s_barrier();
    GEMM_LAB_STAMP(3);
    // pass 0's first entry words (an HBM round trip): a loop of up to four slices requests them here and carries them through
    constexpr bool kEarlyWords = KT <= 4;
    if (kEarlyWords) {
        loadWords(myList[0], wAhead[0]);
        if (kDeepWords) loadWords(myList[1], wAhead[1]);
    }

#pragma unroll
    for (uint32_t t = 0; t < (uint32_t)KT; ++t) {
        if (!kEarlyWords && t + 1u == (uint32_t)KT) {
            loadWords(myList[0], wAhead[0]);
            if (kDeepWords) loadWords(myList[1], wAhead[1]);
        }
        const uint8_t* base = lds + hAt + (t & 1u) * hBytes;
        u32x4 bf[n];
#pragma unroll
        for (uint32_t j = 0; j < n; ++j)
            bf[j] = GEMM_LAB_SKIP(6) ? u32x4{lane, t, j, 0u} : *reinterpret_cast<const u32x4*>(base + bRead + j * 16u * kGemmHalfRowBytes);
#pragma unroll
        for (uint32_t i = 0; i < m; ++i) {
            const u32x4 af = GEMM_LAB_SKIP(6) ? u32x4{lane, t, i, 2u} : *reinterpret_cast<const u32x4*>(base + aRead + i * 16u * kGemmHalfRowBytes);
            if (!GEMM_LAB_SKIP(0)) {
#pragma unroll
                for (uint32_t j = 0; j < n; ++j) acc[i][j] = mfma16<MODE>(af, bf[j], acc[i][j]);
            } else {
                gemmKeep(af);
#pragma unroll
                for (uint32_t j = 0; j < n; ++j) gemmKeep(bf[j]);
            }
        }
        if (t + 1u < (uint32_t)KT) {
            // my chunks of slice t + 1 have landed: round them into the other stage of H, then F takes slice t + 2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            roundChunks((t + 1u) & 1u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + 2u < (uint32_t)KT && !GEMM_LAB_SKIP(2))
                gemmStage<ADMAS, BDMAS>(Ab, aBytes, Bb, bBytes, lds, bAtF, wave, voffA, voffB, (t + 2u) * kGemmRowBytes);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }

    // ---- the sparse mask: slab passes over the wave's tiles (as denseGemm)
    GEMM_LAB_STAMP(4);
    float* slab = reinterpret_cast<float*>(lds) + wave * (kGemmPassTiles * 256u);
    const uint32_t slabAddr = __builtin_amdgcn_readfirstlane(gemmLdsAddress(slab));
    const uint32_t* rowTable = reinterpret_cast<const uint32_t*>(lds + rowTableAt) + wm * (TM / 2u);
#pragma unroll
    for (uint32_t q = 0; q < Q; ++q) {
        if (!GEMM_LAB_SKIP(3)) {
            if (slabAddr < 0x10000u) {
#pragma unroll
                for (uint32_t tp = 0; tp < kGemmPassTiles; ++tp) {
                    const uint32_t tIdx = q * kGemmPassTiles + tp;
                    if (tIdx < m * n) gemmDumpTile(acc[tIdx / n][tIdx % n], slabAddr + tp * 1024u, false);
                }
            } else {
#pragma unroll
                for (uint32_t tp = 0; tp < kGemmPassTiles; ++tp) {
                    const uint32_t tIdx = q * kGemmPassTiles + tp;
                    if (tIdx < m * n) gemmDumpTile(acc[tIdx / n][tIdx % n], slabAddr + tp * 1024u, true);
                }
            }
        }
        const uint32_t first = myList[q], last = myList[q + 1u];
        uint32_t w[kGemmWordChunk];
#pragma unroll
        for (uint32_t u = 0; u < kGemmWordChunk; ++u) w[u] = wAhead[q & 1u][u];
        if (kDeepWords) {
            if (q + 2u < Q) loadWords(myList[q + 2u], wAhead[q & 1u]);
        } else if (q + 1u < Q) {
            loadWords(last, wAhead[(q + 1u) & 1u]);
        }
        for (uint32_t e = first; e < last; e += kGemmWordChunk * kWave) {
            if (e != first) loadWords(e, w);
#pragma unroll
            for (uint32_t u = 0; u < kGemmWordChunk; ++u) {
#if defined(BSMR_GEMM_BRANCHY_STORES)   // lab: the form with a branch per batch
                if (e + u * kWave + lane < last && w[u] != kGemmNoEntry) {
                    const float val = slab[w[u] & 4095u];
                    const uint32_t dst = rowTable[(w[u] >> 12) & 127u] + (w[u] >> 19);
                    if (!GEMM_LAB_SKIP(5)) P[dst] = val;
                }
#else
                // (a padding word reads slot 4095 and row TM / 2 - 1: inside the slab and the table)
                const bool live = e + u * kWave + lane < last && w[u] != kGemmNoEntry;
                const float val = slab[w[u] & 4095u];
                const uint32_t dst = rowTable[(w[u] >> 12) & (TM / 2u - 1u)] + (w[u] >> 19);
                if (!GEMM_LAB_SKIP(5)) gemmStoreEntry(P, live, dst, val);
#endif
            }
        }
        GEMM_LAB_STAMP(5u + q);
    }
    GEMM_LAB_STAMPS_OUT;
}

}  // namespace bsmr
