// Dense kernels of the "tiles" format (csrc/tile_format.hpp): what the reference's dense-block kernel
// computes (src/sddmmKernel.cu:213-351 - P[e] = sum_k A[i,k] B[k,j] for the entries of the dense
// blocks), restructured for MI355X.
//
// denseTiles<KS, H, MODE>: one wave per workgroup, one work item (a run of <= 32 blocks of one row
// group of H panels) per wave.
//   * The A fragments of all H panels stay in registers for the whole item (H * KS * 4 VGPRs).
//   * A block's 16 B columns are gathered ONCE, by LDS-DMA, into a wave-private ring of images and
//     multiplied against every panel of the group that has an entry in the block (wave-uniform mask):
//     one ds_read_b128 of a B fragment feeds up to H MFMAs.  Grouping H panels cuts the L2 -> LDS gather
//     volume by the overlap of their column sets - the r01 kernels re-gathered every column once per
//     16-row panel (28x the operand on the nips-like matrix, 209x on 4096^2 10 %).
//   * The loop has no barrier.  Gathers run D-1 images ahead and are awaited with a counted vmcnt (gfx950
//     completes loads, stores and LDS-DMA in issue order), the column ids of the next gather are read from
//     LDS one iteration early, and the block's entry words arrive by LDS-DMA together with its image.
//   * Mask + write-back: the 16x16 results of a block go to an LDS slab (row stride 17 floats); then
//     every lane carries ONE stored entry from the slab to P[rowBase[row] + offset] - 64 entries per
//     store instruction, whatever the fill of the tiles (the r01 epilogue issued 4 masked stores per
//     tile for ~10-34 entries).
// Numerics: v_mfma_f32_16x16x32_{f16,bf16}, k ascending - the same instruction sequence per entry as the
// r01 kernels (operands rounded RNE, products exact, fp32 accumulate).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sddmm_kernels.hpp"
#include "tile_format.hpp"

namespace bsmr {

constexpr uint32_t kSlabStride = 17;   // floats per slab row: a row's entries hit consecutive banks

// A block is gathered as SPLIT images of at most four 32-wide k steps (4 KiB): a wave's ring stays small, so that
// 10-16 waves fit a CU - the waves are in-order, and what hides the chain gather -> fragments -> MFMA -> slab ->
// entries -> store of one wave is the other waves (measured: a 20 KiB ring with 5-6 waves per CU ran 2x slower than
// the r01 kernel at equal gather volume).
constexpr uint32_t tileKsl(int KS) { return KS > 4 ? 4u : (uint32_t)KS; }
constexpr uint32_t tileDefaultDepth(int KS) { return KS == 1 ? 4u : KS == 2 ? 3u : 2u; }
// entry-word slots: blocks whose entry words are in LDS at the same time
constexpr uint32_t tileEntrySlots(int KS, uint32_t D) {
    const uint32_t split = (uint32_t)KS / tileKsl(KS);
    return (D - 1u + split - 1u) / split + 1u;
}

// LDS bytes of one wave = one workgroup
constexpr uint32_t tileLdsBytes(int KS, int H, uint32_t D, uint32_t entryCap) {
    return D * 1024u * tileKsl(KS) + tileEntrySlots(KS, D) * 4u * entryCap + kTileMaxItemBlocks * 64u +
           (16u * H < 64u ? 64u : 16u * H) * 4u + 16u * H * kSlabStride * 4u;
}

// s_waitcnt takes its count as an immediate: n is wave-uniform; counts above 31 wait for 31 (stricter, still correct)
__device__ __forceinline__ void waitVmcnt(uint32_t n) {
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    }
}

// Lab builds (make LAB=1): per-wave time stamps of the phases, written to `stamps` ([item][8] cycles).
#ifdef BSMR_LAB_STAMPS
#define BSMR_STAMP(slot)                                                          \
    do {                                                                          \
        if (stamps) {                                                             \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
            const uint64_t now_ = __builtin_readcyclecounter();                   \
            if (lane == 0) stamps[(size_t)itemId * 8 + (slot)] = now_;            \
        }                                                                         \
    } while (0)
#define BSMR_STAMP_ACC(slot, t0_)                                                 \
    do {                                                                          \
        if (stamps) {                                                             \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
            const uint64_t now_ = __builtin_readcyclecounter();                   \
            accum[slot] += now_ - (t0_);                                          \
            (t0_) = now_;                                                         \
        }                                                                         \
    } while (0)
#else
#define BSMR_STAMP(slot) do {} while (0)
#define BSMR_STAMP_ACC(slot, t0_) do {} while (0)
#endif

template <int KS, int H, int MODE, int DEPTH>
__global__ void __launch_bounds__(kWave)
denseTiles(const uint16_t* __restrict__ A16, const uint16_t* __restrict__ B16,
           const uint32_t* __restrict__ groupRows, const uint32_t* __restrict__ blockCols,
           const uint4* __restrict__ blockInfo, const uint32_t* __restrict__ entries,
           const TileItem* __restrict__ items, const uint32_t* __restrict__ itemRowBase,
           float* __restrict__ P, uint32_t entryCap, Batch batch
#ifdef BSMR_LAB_STAMPS
           , uint64_t* __restrict__ stamps
#endif
           ) {
    A16 += blockIdx.y * batch.strideA;   // batched call: problem blockIdx.y of a strided batch
    B16 += blockIdx.y * batch.strideB;
    P += blockIdx.y * batch.strideP;
    constexpr uint32_t K = 32u * KS;
    constexpr uint32_t KSL = tileKsl(KS);              // 32-wide k steps per image
    constexpr uint32_t SPLIT = (uint32_t)KS / KSL;     // images per block (the accumulators carry over)
    constexpr uint32_t PC = 4u * KSL;                  // 16-byte pieces per column of an image
    constexpr uint32_t SW = PC - 1u < 15u ? PC - 1u : 15u;
    constexpr uint32_t rowBytes = 64u * KSL;
    constexpr uint32_t imgBytes = 16u * rowBytes;
    constexpr uint32_t D = (uint32_t)DEPTH;
    constexpr uint32_t DE = tileEntrySlots(KS, D);
    constexpr uint32_t R = 16u * H;
    static_assert(H >= 1 && H <= 8 && D >= 2, "panels per group / ring depth");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* const ring = lds;                                                   // [D] images
    uint32_t* const entRing = reinterpret_cast<uint32_t*>(lds + D * imgBytes);   // [DE][entryCap] entry words
    uint32_t* const colsLds = entRing + DE * entryCap;                           // [kTileMaxItemBlocks][16]
    uint32_t* const rowBaseLds = colsLds + kTileMaxItemBlocks * 16u;             // [max(R, 64)]
    float* const slab = reinterpret_cast<float*>(rowBaseLds + (R < 64u ? 64u : R));   // [R][17]

    const uint32_t itemId = xcdContiguous(blockIdx.x, gridDim.x);
    const TileItem item = items[itemId];
    const uint32_t lane = threadIdx.x;
    const uint32_t r = lane & 15u, g = lane >> 4;
    const uint32_t count = item.count;
    const uint32_t units = count * SPLIT;
#ifdef BSMR_LAB_STAMPS
    uint64_t accum[3] = {0, 0, 0};
    uint64_t tPhase = 0;
#endif
    BSMR_STAMP(0);

    // ---- prologue: the item's column ids and row bases to LDS, block records and row ids to registers ----
    {
        const uint32_t words = count * 16u;
        const uint32_t* src = blockCols + (size_t)item.first * 16u;
        for (uint32_t j = 0; j * 64u < words; ++j) {
            const uint32_t i = j * 64u + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (i < words ? i : 0u)),
                                             (__attribute__((address_space(3))) void*)(colsLds + j * 64u), 4, 0, 0);
        }
        const uint32_t* rb = itemRowBase + (size_t)itemId * R;
#pragma unroll
        for (uint32_t j = 0; j * 64u < R; ++j) {
            const uint32_t i = j * 64u + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rb + (i < R ? i : 0u)),
                                             (__attribute__((address_space(3))) void*)(rowBaseLds + j * 64u), 4, 0, 0);
        }
    }
    const uint4 info = blockInfo[item.first + (lane < count ? lane : 0u)];   // lane m: block m of the item
    uint32_t myRow[H];
#pragma unroll
    for (int h = 0; h < H; ++h) myRow[h] = groupRows[(size_t)item.group * R + 16u * h + r];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BSMR_STAMP(1);
    uint32_t infoStart = info.x, infoMask = info.y;
    asm volatile("" : "+v"(infoStart), "+v"(infoMask));
#pragma unroll
    for (int h = 0; h < H; ++h) asm volatile("" : "+v"(myRow[h]));

    // ---- A fragments of the H panels: lane (r, g) holds k = 32 s + 8 g .. + 7 of row r (16 bytes) ----
    // issued through inline assembly and waited for on the spot, before the first gather is issued: loads, wait and
    // the statements that pin the registers behind the wait are one straight line.  (Round 2 left them in flight under
    // the first gathers, covered by the first image's counted wait; the compiler, for which an asm load's registers
    // are written when the statement ends, copied some of them in between - `make check-isa`.  The round trip costs
    // a work item ~500 cycles once.)
    u32x4 a[H][KS];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const uint16_t* aRow = A16 + (size_t)myRow[h] * K + g * 8u;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[h][s]) : "v"(aRow + s * 32) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(a[h][s]));

    // per-lane constants of the gather: DMA instruction j moves piece slots [64 j, 64 j + 64) of an image;
    // slot f = column f / PC, piece f % PC, XOR-swizzled on the SOURCE so that fragment reads are conflict-free
    uint32_t gCol[KSL], gOff[KSL];
#pragma unroll
    for (uint32_t j = 0; j < KSL; ++j) {
        const uint32_t f = 64u * j + lane;
        const uint32_t col = f / PC, t = f % PC;
        gCol[j] = col;
        gOff[j] = (t ^ (col & SW)) << 3;
    }
    uint32_t cid[KSL];   // column ids of the unit gathered next
    auto readCids = [&](uint32_t u) {
        const uint32_t m = u / SPLIT;
#pragma unroll
        for (uint32_t j = 0; j < KSL; ++j) cid[j] = colsLds[m * 16u + gCol[j]];
    };
    uint32_t entWrite = 0;   // entry-word slot of the next block gathered
    // Exact accounting of the vector-memory queue (loads, stores and LDS-DMA complete in issue order): `issued`
    // counts this wave's operations, seq[i] is its value right after the gather of the i-th image in flight.
    // Waiting for `issued - seq[0]` outstanding operations is waiting for exactly that image - not for the entry
    // words and stores issued behind it (a wait that ignored those stalled every block on the DMA just issued).
    uint32_t issued = 0, seq[D - 1];
    auto gather = [&](uint32_t u, uint32_t slot) {   // unit u = (block u / SPLIT, k part u % SPLIT), ids in cid[]
        if (u % SPLIT == 0) {   // the block's entry words ride along, 4 per lane, only the lanes that hold any
            const uint32_t m = u / SPLIT;
            const uint32_t start = __builtin_amdgcn_readlane(infoStart, m);
            const uint32_t n = (uint32_t)__builtin_amdgcn_readlane(infoMask, m) >> 16;
            uint32_t* dst = entRing + entWrite * entryCap;
            if (lane * 4u < n)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(entries + start + lane * 4u),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
#pragma unroll 1
            for (uint32_t c = 1; c * kTileEntryChunk < n; ++c)   // more than 256 entries in the block
                if (c * kTileEntryChunk + lane * 4u < n)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(entries + start + c * kTileEntryChunk + lane * 4u),
                        (__attribute__((address_space(3))) void*)(dst + c * kTileEntryChunk), 16, 0, 0);
            issued += (n + kTileEntryChunk - 1u) / kTileEntryChunk;
            entWrite = entWrite + 1 == DE ? 0 : entWrite + 1;
        }
        uint8_t* dst = ring + slot * imgBytes;
        const uint32_t part = u % SPLIT;
#pragma unroll
        for (uint32_t j = 0; j < KSL; ++j) {
            const uint16_t* src = B16 + (size_t)cid[j] * K + part * (32u * KSL) + gOff[j];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + j * 1024u), 16, 0, 0);
        }
        issued += KSL;
    };

    // the first D-1 images
#pragma unroll
    for (uint32_t u = 0; u + 1 < D; ++u) {
        if (u < units) {
            readCids(u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            gather(u, u);
        }
        seq[u] = issued;
    }
    if (D - 1 < units) readCids(D - 1);

    const uint32_t slabLane = (4u * g) * kSlabStride + r;   // slab word of accumulator register 0 of panel 0
    // entry rounds kept in registers: the words of a block's first RMAX * 64 entries are read together with the
    // last image's fragments, their row bases while the MFMAs run, so that the write-back costs two LDS round
    // trips (fragments + words, slab) instead of five
    constexpr uint32_t RMAX = H <= 2 ? 2u : 4u;
    f32x4 acc[H];
    uint32_t slotRead = 0, slotWrite = D - 1, entRead = 0;
    for (uint32_t m = 0; m < count; ++m) {
        const uint32_t mask = (uint32_t)__builtin_amdgcn_readlane(infoMask, m);
        const uint32_t n = mask >> 16;
        const uint32_t* entWords = entRing + entRead * entryCap;
        entRead = entRead + 1 == DE ? 0 : entRead + 1;
        uint32_t w[RMAX];
#pragma unroll
        for (uint32_t part = 0; part < SPLIT; ++part) {
            const uint32_t u = m * SPLIT + part;
            // A. image u has landed when all but the operations issued after its gather are complete
            const uint32_t landed = seq[0];
#pragma unroll
            for (uint32_t i = 0; i + 2 < D; ++i) seq[i] = seq[i + 1];
            // ... next image into the slot unit u-1 was read from (its reads returned before its MFMAs)
            if (u + D - 1 < units) {
                gather(u + D - 1, slotWrite);
                slotWrite = slotWrite + 1 == D ? 0 : slotWrite + 1;
                if (u + D < units) readCids(u + D);
            }
            seq[D - 2] = issued;
            waitVmcnt(issued - landed);
            if (u == 0) {
                BSMR_STAMP(2);
#ifdef BSMR_LAB_STAMPS
                if (stamps) tPhase = __builtin_readcyclecounter();
#endif
            } else {
                BSMR_STAMP_ACC(0, tPhase);   // issue of the next gather + wait for this image
            }
            // C. B fragments of the image: lane (c = r, g) reads piece 4 s + g of column c
            const uint8_t* img = ring + slotRead * imgBytes;
            u32x4 b[KSL];
#pragma unroll
            for (uint32_t s = 0; s < KSL; ++s)
                b[s] = *reinterpret_cast<const u32x4*>(img + r * rowBytes + (((4u * s + g) ^ (r & SW)) << 4));
            if (part + 1 == SPLIT) {
#pragma unroll
                for (uint32_t q = 0; q < RMAX; ++q)
                    if (q * kWave < n) w[q] = entWords[q * kWave + lane];
            }
            // D. every panel that has an entry in the block
#pragma unroll
            for (int h = 0; h < H; ++h) {
                if (!(mask & (1u << h))) continue;   // wave-uniform
                f32x4 c = acc[h];
                if (part == 0) c = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (uint32_t s = 0; s < KSL; ++s) c = mfma16<MODE>(a[h][part * KSL + s], b[s], c);
                acc[h] = c;
            }
            slotRead = slotRead + 1 == D ? 0 : slotRead + 1;
        }
#ifdef BSMR_LAB_STAMPS
        if (stamps) {   // the MFMA results are in registers
#pragma unroll
            for (int h = 0; h < H; ++h) asm volatile("" : "+v"(acc[h]));
        }
#endif
        BSMR_STAMP_ACC(1, tPhase);           // fragments + MFMAs
        // E. row bases of the entries, results to the slab, one entry per lane from the slab to P
        uint32_t base[RMAX], src[RMAX];
#pragma unroll
        for (uint32_t q = 0; q < RMAX; ++q)
            if (q * kWave < n) {
                w[q] = q * kWave + lane < n ? w[q] : 0u;   // lanes past the end: a harmless slot
                const uint32_t row = w[q] & 255u, col = (w[q] >> 8) & 15u;
                src[q] = row * kSlabStride + col;
                base[q] = rowBaseLds[row];
            }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            if (!(mask & (1u << h))) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) slab[slabLane + (16u * h + i) * kSlabStride] = acc[h][i];
        }
        float val[RMAX];
#pragma unroll
        for (uint32_t q = 0; q < RMAX; ++q)
            if (q * kWave < n) val[q] = slab[src[q]];
#pragma unroll
        for (uint32_t q = 0; q < RMAX; ++q)
            if (q * kWave + lane < n) P[base[q] + (w[q] >> 12)] = val[q];
#pragma unroll 1
        for (uint32_t e = RMAX * kWave + lane; e < n; e += kWave) {   // blocks with more entries (rare)
            const uint32_t word = entWords[e];
            const uint32_t row = word & 255u, col = (word >> 8) & 15u;
            P[rowBaseLds[row] + (word >> 12)] = slab[row * kSlabStride + col];
        }
        issued += (n + kWave - 1u) / kWave;   // one store instruction per round of 64 entries
        BSMR_STAMP_ACC(2, tPhase);           // slab + entries + store issue
    }
    BSMR_STAMP(3);
#ifdef BSMR_LAB_STAMPS
    if (stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BSMR_STAMP(4);
        if (lane == 0) {
            stamps[(size_t)itemId * 8 + 5] = accum[0];
            stamps[(size_t)itemId * 8 + 6] = accum[1];
            stamps[(size_t)itemId * 8 + 7] = accum[2];
        }
    }
#endif
}

}  // namespace bsmr
