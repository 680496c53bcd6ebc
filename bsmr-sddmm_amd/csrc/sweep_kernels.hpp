// Panel-sweep dense kernel (round 3) for gfx950: the masked product computed the way a GEMM is, without a gather.
//
// What it computes is the dense part of the reference's SDDMM (src/sddmmKernel.cu:213-351): for every stored
// entry (i, j) that the plan keeps on the dense path, P[e] = sum_k A[i,k] * B[k,j], fp16 / bf16 operands, fp32
// accumulation, bit for bit what convertOperands + denseStream produce (same casts, same MFMA, same k order).
//
// Work item = (row group of 16 * W * PW reordered rows) x (strip of consecutive 16-column blocks of B), format
// csrc/sweep_format.hpp.  A workgroup is W CONSUMER waves and NL LOADER waves:
//   * the loader waves stream a sequence of 16 x K images through an LDS ring by LDS-DMA: first the W * PW row
//     panels of the group (16 rows of A gathered by row id: a row is K contiguous elements; fragment-shaped
//     loads of the same rows straight into the consumers' registers - 16 rows x 64 bytes per instruction -
//     took 11 000 cycles per workgroup, measured), then the item's entry words (one contiguous run, into a
//     region of their own: they are read once per launch, so they always come from beyond the L2 and must not
//     sit in the delivery of every block), then the blocks of the strip (16 columns of B = ONE contiguous run
//     of 16 K elements: every DMA instruction moves eight whole 128-byte lines).  A loader's only vector-memory
//     operations are these DMAs, the same number per image, so "my pieces of image u have landed" is an exact
//     counted s_waitcnt vmcnt;
//   * fp32 operands (SRC32): each loader then rounds ITS pieces to fp16 / bf16 (the casts of convertOperands) and
//     writes them into a second, shallow ring of 16-bit images in the XOR-swizzled layout the MFMA fragment reads
//     want.  Measured (tools/probes): a wave alone on its SIMD issues one vector instruction per 4-5 cycles, and
//     with four consumers each converting the same block the loop was issue-bound at 3-4 times the MFMA time;
//     the loaders otherwise wait at the barrier.  The DMA ring is then private to its loader (a slot is reused
//     once the loader itself has converted it), so all of it but one group is in flight;
//   * the workgroup barrier of a group of NBB images publishes them (counted vmcnt + barrier: the only ordering
//     LDS-DMA data has);
//   * consumer wave w owns panels w * PW .. w * PW + PW - 1 of the group: it takes their A fragments from the
//     16-bit images and keeps them in registers.  Per block of B it reads the B fragments (ds_read_b128), issues
//     PW * KS MFMAs, drops the accumulators into its private LDS slab and lets one lane per stored entry carry a
//     value from the slab to P: the sparse mask costs one store per ENTRY instead of a test per cell.  Its only
//     vector-memory operations in the loop are those stores, which nothing waits for.  A consumer is alone on its
//     SIMD and issues in order, so nothing overlaps unless the program order says so (measured: fragments -> MFMAs
//     -> slab -> entries -> store run block after block took ~1000 cycles per block, the MFMAs 256 of them): the
//     loop is software-pipelined by hand, block b - 1's write-back and block b + 1's fragment requests sit
//     BETWEEN block b's MFMAs (sched_barrier fences keep them there).
// Ring accounting.  16-bit operands: consumers read the DMA ring itself, which holds DG + 2 groups: the loaders
// refill the slots of group g - 2 after passing barrier g - 1, which every consumer reaches only after its reads
// of group g - 2 have returned.  fp32 operands: the 16-bit ring holds two groups (written for g while g - 1 is
// read), the DMA ring RG groups of which RG - 1 are in flight behind the one being converted.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "sddmm_kernels.hpp"
#include "sweep_format.hpp"

namespace bsmr {

constexpr uint32_t kSweepLdsBudget = 152u * 1024u;     // of the CU's 160 KiB
constexpr uint32_t kSweepItemWords = 8192u;            // entry words of one item that fit their LDS region (32 KiB)

constexpr uint32_t sweepImageBytes(int KS, bool src32) { return 16u * 32u * KS * (src32 ? 4u : 2u); }
// slabs, row bases, row ids, entry words of the item, and for fp32 operands the ring of two groups of 16-bit images
constexpr uint32_t sweepFixedBytes(int KS, bool src32, int W, int PW, int NBB) {
    return (uint32_t)W * PW * 4u * 64u * 4u + (uint32_t)W * 16u * PW * 4u + (uint32_t)W * PW * 16u * 4u + kSweepItemWords * 4u +
           (src32 ? 2u * NBB * sweepImageBytes(KS, false) : 0u);
}
// groups of NBB slots in the DMA ring: as many as fit for `perCu` workgroups per CU, at most 12 slots
constexpr uint32_t sweepRingGroups(int KS, bool src32, int W, int PW, int perCu, int NBB) {
    const uint32_t room = kSweepLdsBudget / perCu - sweepFixedBytes(KS, src32, W, PW, NBB);
    uint32_t s = room / sweepImageBytes(KS, src32);
    if (s > 12u) s = 12u;
    return s / NBB;
}
constexpr uint32_t sweepSlots(int KS, bool src32, int W, int PW, int perCu, int NBB) {
    return sweepRingGroups(KS, src32, W, PW, perCu, NBB) * NBB;
}
constexpr size_t sweepLdsBytes(int KS, int PW, bool src32, int W, int perCu, int NBB) {
    return (size_t)sweepSlots(KS, src32, W, PW, perCu, NBB) * sweepImageBytes(KS, src32) + sweepFixedBytes(KS, src32, W, PW, NBB);
}

template <uint32_t N>
__device__ __forceinline__ void sweepWaitVm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// all but the `younger` (<= MAXY) most recently issued images of this loader have landed (OPS operations each)
template <uint32_t OPS, uint32_t MAXY>
__device__ __forceinline__ void sweepWaitYounger(uint32_t younger) {
    if constexpr (MAXY == 0) {
        sweepWaitVm<0>();
    } else {
        if (younger >= MAXY) sweepWaitVm<(OPS * MAXY < 63u ? OPS * MAXY : 63u)>();
        else sweepWaitYounger<OPS, MAXY - 1>(younger);
    }
}
// the workgroup barrier of a group: this wave's LDS operations have completed, then all waves meet
__device__ __forceinline__ void sweepBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// lab builds (tools/probes/sweep_probe.hip).  BSMR_SWEEP_STAMPS: clock stamps of consumer wave 0 and loader wave 0, kept
// in registers and written after the loops (a store inside the loader's loop would change its vmcnt accounting; the
// stamps cost hundreds of cycles per barrier, so that build's run time is not the kernel's).  BSMR_SWEEP_LAB: a mask
// argument that leaves parts of the kernel out (timing only: results are wrong).
#if defined(BSMR_SWEEP_STAMPS)
#define SWEEP_STAMP_ARG , uint64_t* __restrict__ stamps, uint32_t labSkip   /* bit 1 MFMAs, 2 slab writes, 3 entries, 4 stores, 5 DMAs, 6 conversion */
#define SWEEP_LAB_SKIP(bit) (labSkip & (1u << (bit)))
#define SWEEP_CLOCK(x) const uint64_t x = __builtin_amdgcn_s_memtime()
#elif defined(BSMR_SWEEP_LAB)
#define SWEEP_STAMP_ARG , uint32_t labSkip
#define SWEEP_LAB_SKIP(bit) (labSkip & (1u << (bit)))
#define SWEEP_CLOCK(x)
#else
#define SWEEP_STAMP_ARG
#define SWEEP_CLOCK(x)
#define SWEEP_LAB_SKIP(bit) false
#endif

template <int KS, int PW, int MODE, bool SRC32, int W, int NL, int PERCU, int NBB>
__global__ void __launch_bounds__((W + NL) * kWave)
denseSweep(const void* __restrict__ Aop, const void* __restrict__ Bop, const uint32_t* __restrict__ panelRows,
           const SweepItem* __restrict__ items, const uint32_t* __restrict__ starts,
           const uint32_t* __restrict__ rowStart, const uint32_t* __restrict__ words, float* __restrict__ P,
           uint32_t N, Batch batch SWEEP_STAMP_ARG) {
    SWEEP_CLOCK(tStart);
    constexpr uint32_t K = 32u * KS, ESZ = SRC32 ? 4u : 2u;
    constexpr uint32_t colBytes = K * ESZ, imgBytes = 16u * colBytes;      // as the loaders fetch it
    constexpr uint32_t col16 = K * 2u, img16 = 16u * col16;                // as the consumers read it
    constexpr uint32_t PC = colBytes / 16u;                                // 16-byte pieces per fetched column
    constexpr uint32_t PC16 = col16 / 16u, SW16 = PC16 - 1u < 15u ? PC16 - 1u : 15u;
    constexpr uint32_t DMAS = imgBytes / 1024u;                            // LDS-DMA instructions per image
    constexpr uint32_t MYD = (DMAS + NL - 1) / NL;                         // ... per loader wave (the last ones may be clamped duplicates)
    constexpr uint32_t wordImageBytes = (uint32_t)NL * MYD * 1024u;        // entry words one pseudo-image carries
    constexpr uint32_t RG = sweepRingGroups(KS, SRC32, W, PW, PERCU, NBB), S = RG * NBB;
    static_assert(RG >= (SRC32 ? 2u : 3u), "the DMA ring is too small");
    constexpr uint32_t DG = SRC32 ? RG - 1u : RG - 2u;                     // groups in flight behind the one being waited for
    static_assert(MYD * DG * NBB <= 63, "vmcnt is 6 bits");
    constexpr uint32_t S16 = SRC32 ? 2u * NBB : S;                         // slots of the ring the consumers read
    constexpr uint32_t GP = W * PW;                                        // panels per row group
    static_assert(GP % NBB == 0, "the panels of a group fill whole barrier groups");
    constexpr uint32_t slabFloats = PW * 4u * 64u;
    static_assert(PW * 16 <= 64 && PW * 4 * 64 <= 1024, "entry word: 6 bits of row, 10 bits of slab slot");

    // LDS: DMA ring | W slabs | W rowStart tables | row ids of the group | entry words | (fp32 operands) 16-bit ring
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr uint32_t ringBytes = S * imgBytes;
    constexpr uint32_t slabAt = ringBytes, rowLdsAt = slabAt + W * slabFloats * 4u, rowIdsAt = rowLdsAt + W * 16u * PW * 4u;
    constexpr uint32_t wordsAt = rowIdsAt + GP * 16u * 4u, ring16At = SRC32 ? wordsAt + kSweepItemWords * 4u : 0u;
    const uint8_t* Ab = static_cast<const uint8_t*>(Aop) + (size_t)blockIdx.y * batch.strideA * ESZ;
    const uint8_t* Bb = static_cast<const uint8_t*>(Bop) + (size_t)blockIdx.y * batch.strideB * ESZ;
    P += (size_t)blockIdx.y * batch.strideP;

    const uint32_t itemId = xcdContiguous(blockIdx.x, gridDim.x);
    const SweepItem item = items[itemId];
    const uint32_t nb = item.numBlocks;
    // the item's entry words: one run [wordBase, wordEnd), moved as WI pseudo-images of wordImageBytes (whole barrier groups)
    const uint32_t wordBase = starts[item.startsBase], wordEnd = starts[item.startsBase + W * (nb + 1u) - 1u];
    const uint32_t WI = ((wordEnd - wordBase) * 4u + wordImageBytes * NBB - 1u) / (wordImageBytes * NBB) * NBB;
    const uint32_t firstBlockImage = GP + WI;
    const uint32_t total = firstBlockImage + nb;                           // images: panels of A, entry words, blocks of B
    const uint32_t numGroups = (total + NBB - 1u) / NBB;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;

    if (wave >= (uint32_t)W) {
        // ------------------------------------------------------------------ loader
        const uint32_t q = wave - W;
        uint32_t* rowIds = reinterpret_cast<uint32_t*>(lds + rowIdsAt);
        // the group's row ids (every loader writes the same table, and reads it only after its own writes)
#pragma unroll
        for (uint32_t t = 0; t < GP * 16u; t += 64u)
            if (t + lane < GP * 16u) rowIds[t + lane] = panelRows[(size_t)item.group * (GP * 16u) + t + lane];
        // my pieces of an image: DMA instruction d = q + NL m moves piece slots 64 d .. 64 d + 63; lane l's piece is
        // 16 bytes at `pieceOff` of column (or row) `colOf` (fp32: kept in fetch order, only this loader reads it back;
        // 16-bit: XOR-swizzled on the source address, the consumers read it)
        uint32_t colOf[MYD], pieceOff[MYD], dstOff[MYD], cvtOff[MYD];
#pragma unroll
        for (uint32_t m = 0; m < MYD; ++m) {
            const uint32_t d = min(q + (uint32_t)NL * m, DMAS - 1u);
            const uint32_t f = 64u * d + lane;
            colOf[m] = f / PC;
            const uint32_t t = f % PC;
            pieceOff[m] = SRC32 ? t << 4 : ((t ^ (colOf[m] & SW16)) << 4);
            dstOff[m] = d * 1024u;
            // fp32 piece t = elements 4 t .. 4 t + 3 -> half of the 16-bit piece t / 2, which sits at slot (t / 2) ^ (col & SW16)
            cvtOff[m] = colOf[m] * col16 + (((t >> 1) ^ (colOf[m] & SW16)) << 4) + ((t & 1u) << 3);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        SWEEP_CLOCK(tMeta);
        auto issue = [&](uint32_t u) {
            uint8_t* dst = lds + (u % S) * imgBytes;
            if (SWEEP_LAB_SKIP(5)) return;
            if (u < GP) {                                                    // a row panel of A
#pragma unroll
                for (uint32_t m = 0; m < MYD; ++m) {
                    const uint32_t row = rowIds[u * 16u + colOf[m]];
                    const uint8_t* src = Ab + (size_t)row * colBytes + pieceOff[m];
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dst + dstOff[m]), 16, 0, 0);
                }
            } else if (u < firstBlockImage) {                                // entry words (reads past the run stay inside `words`: its slack)
                const uint32_t piece = (u - GP) * wordImageBytes;
#pragma unroll
                for (uint32_t m = 0; m < MYD; ++m) {
                    const uint32_t off = min(piece + (q + (uint32_t)NL * m) * 1024u, kSweepItemWords * 4u - 1024u);
                    const uint8_t* src = reinterpret_cast<const uint8_t*>(words + wordBase) + off + lane * 16u;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(lds + wordsAt + off), 16, 0, 0);
                }
            } else {                                                         // a block of B
                const uint32_t c0 = (item.firstBlock + (u - firstBlockImage)) * 16u;
#pragma unroll
                for (uint32_t m = 0; m < MYD; ++m) {
                    const uint32_t c = min(c0 + colOf[m], N - 1u);          // the last block of B may be ragged
                    const uint8_t* src = Bb + (size_t)c * colBytes + pieceOff[m];
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(dst + dstOff[m]), 16, 0, 0);
                }
            }
        };
        // fp32 operands: round my pieces of image u into the 16-bit ring
        auto convert = [&](uint32_t u) {
            if constexpr (SRC32) {
                if (u >= GP && u < firstBlockImage) return;                  // (entry words)
                if (SWEEP_LAB_SKIP(6)) return;
                const uint8_t* src = lds + (u % S) * imgBytes;
                uint8_t* dst = lds + ring16At + (u % S16) * img16;
#pragma unroll
                for (uint32_t m = 0; m < MYD; ++m) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src + dstOff[m] + lane * 16u);
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    u32x2 h;
                    if constexpr (MODE == 0) {
                        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                        f16x4 o;
                        o[0] = (_Float16)v[0]; o[1] = (_Float16)v[1]; o[2] = (_Float16)v[2]; o[3] = (_Float16)v[3];
                        h = __builtin_bit_cast(u32x2, o);
                    } else {
                        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                        bf16x4 o;
                        o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
                        h = __builtin_bit_cast(u32x2, o);
                    }
                    *reinterpret_cast<u32x2*>(dst + cvtOff[m]) = h;
                }
            }
        };
#ifdef BSMR_SWEEP_STAMPS
        uint64_t tFirst = 0, tPanels = 0, waitSum = 0, barSum = 0;
#endif
        for (uint32_t g = 0; g < DG; ++g)
#pragma unroll
            for (uint32_t b = 0; b < (uint32_t)NBB; ++b)
                if (g < numGroups && g * NBB + b < total) issue(g * NBB + b);
        for (uint32_t g = 0; g < numGroups; ++g) {
            if (g + DG < numGroups) {
#pragma unroll
                for (uint32_t b = 0; b < (uint32_t)NBB; ++b)
                    if ((g + DG) * NBB + b < total) issue((g + DG) * NBB + b);
            }
            SWEEP_CLOCK(w0);
            const uint32_t issued = min(total, (g + DG + 1u) * NBB), needed = min(total, (g + 1u) * NBB);
            sweepWaitYounger<MYD, DG * NBB>(issued - needed);
            SWEEP_CLOCK(w1);
#pragma unroll
            for (uint32_t b = 0; b < (uint32_t)NBB; ++b)
                if (g * NBB + b < total) convert(g * NBB + b);
            sweepBarrier();                                                  // group g is in LDS for everyone
#ifdef BSMR_SWEEP_STAMPS
            const uint64_t w2 = __builtin_amdgcn_s_memtime();
            waitSum += w1 - w0;
            barSum += w2 - w1;
            if (g == 0) tFirst = w1;
            if (g == GP / NBB - 1u) tPanels = w2;
#endif
        }
#ifdef BSMR_SWEEP_STAMPS
        if (stamps && q == 0 && lane == 0) {
            uint64_t* o = stamps + (size_t)itemId * 32u + 16u;
            o[0] = tStart; o[1] = tMeta; o[2] = tMeta; o[3] = tFirst; o[4] = tPanels; o[5] = __builtin_amdgcn_s_memtime();
            o[6] = waitSum; o[7] = barSum;
        }
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumer
    const uint32_t r = lane & 15u, g = lane >> 4;
    float* slab = reinterpret_cast<float*>(lds + slabAt) + wave * slabFloats;
    uint32_t* rowLds = reinterpret_cast<uint32_t*>(lds + rowLdsAt) + wave * (16u * PW);
    const uint32_t* wordLds = reinterpret_cast<const uint32_t*>(lds + wordsAt);
    const uint8_t* ring16 = lds + ring16At;                                 // (16-bit operands: the DMA ring itself)

    // entry lists of this wave: lane b holds where block b's words start in the item's run (nb + 1 <= 64 values)
    const uint32_t st = starts[item.startsBase + wave * (nb + 1u) + min(lane, nb)] - wordBase;
    if (lane < 16u * PW) rowLds[lane] = rowStart[((size_t)itemId * W + wave) * (16u * PW) + lane];
    SWEEP_CLOCK(tMeta);

    // 16-bit image u -> the fragments of lane (r, g): 8 consecutive k of row / column r at every k step
    auto request = [&](uint32_t u, u32x4 (&frag)[KS]) {
        const uint8_t* col = ring16 + (u % S16) * img16 + r * col16;
#pragma unroll
        for (int s = 0; s < KS; ++s) frag[s] = *reinterpret_cast<const u32x4*>(col + (((4u * s + g) ^ (r & SW16)) << 4));
    };

    // A fragments of my panels, taken from the 16-bit images as the panels of the group stream by
    u32x4 a[PW][KS];
#pragma unroll
    for (uint32_t t = 0; t < GP; ++t) {
        if (t % NBB == 0) sweepBarrier();
        if (t / PW == wave) request(t, a[t % PW]);
    }
    // the entry words pass by (pseudo-images: nothing to take from the ring)
    for (uint32_t t = 0; t < WI; t += NBB) sweepBarrier();
    SWEEP_CLOCK(tPanels);
#ifdef BSMR_SWEEP_STAMPS
    uint64_t barSum = 0;
#endif

    // Block b's step, in program order (the wave issues in order; the fences keep the compiler from regrouping):
    //   B  the MFMAs of k step 0 (PW of them: 16 cycles of matrix pipe each)
    //   C  block b - 1's write-back, first half: its accumulators into the slab, then the slab value and the row
    //      base of each of its entries (LDS serves a wave in order: the reads see the writes, and the NEXT step's
    //      writes cannot overtake these reads)
    //   D  the MFMAs of k steps 1 .. KS - 1; behind k step s the fragment of k step s - 1 of block b + 1 is
    //      requested into the registers that k step has just released (first the barrier, if b + 1 starts a group)
    //   E  block b - 1's write-back, second half: one store per entry (the reads of C returned long ago), and the
    //      entry words of block b + 1
    uint32_t s0 = __builtin_amdgcn_readlane(st, 0);
    auto writeBack = [&](const f32x4 (&accPrev)[PW], uint32_t wPrev, float& val, uint32_t& base) {
        if (!SWEEP_LAB_SKIP(2)) {
#pragma unroll
            for (int j = 0; j < PW; ++j) *reinterpret_cast<f32x4*>(slab + (j * 64 + lane) * 4) = accPrev[j];
        }
        if (!SWEEP_LAB_SKIP(3)) {
            val = slab[wPrev & (slabFloats - 1u)];
            base = rowLds[(wPrev >> 10) & (16u * PW - 1u)];
        }
    };
    auto storeEntries = [&](uint32_t wPrev, float val, uint32_t base, uint32_t s0Prev, uint32_t nPrev) {
        if (lane < nPrev && wPrev != kSweepNoEntry && !SWEEP_LAB_SKIP(4)) P[base + (wPrev >> 16)] = val;
        if (nPrev > 64u) {                                                   // (rare: a step with more than 64 entries)
            for (uint32_t e = 64u + lane; e < nPrev; e += 64u) {
                const uint32_t w = wordLds[s0Prev + e];
                if (w != kSweepNoEntry) P[rowLds[(w >> 10) & 63u] + (w >> 16)] = slab[w & 1023u];
            }
        }
    };
    if (nb) {
        u32x4 frag[KS];
        f32x4 accA[PW], accB[PW];
        uint32_t wPrev = kSweepNoEntry, s0Prev = 0, nPrev = 0;
        sweepBarrier();                                                      // (the group of block 0)
        request(firstBlockImage, frag);
        uint32_t wHere = wordLds[s0 + lane];     // (64 words from the list's start: what lies past its end is not used)
        auto step = [&](uint32_t b, f32x4 (&acc)[PW], const f32x4 (&accPrev)[PW], auto firstTag) {
            constexpr bool FIRST = decltype(firstTag)::value;
            const uint32_t s1 = __builtin_amdgcn_readlane(st, b + 1u);
            const bool more = b + 1u < nb;
            float val = 0.f;
            uint32_t base = 0;
            if (!SWEEP_LAB_SKIP(1)) {
#pragma unroll
                for (int j = 0; j < PW; ++j) acc[j] = mfma16<MODE>(a[j][0], frag[0], f32x4{0.f, 0.f, 0.f, 0.f});
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!FIRST) writeBack(accPrev, wPrev, val, base);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 1; s < KS; ++s) {
                if (!SWEEP_LAB_SKIP(1)) {
#pragma unroll
                    for (int j = 0; j < PW; ++j) acc[j] = mfma16<MODE>(a[j][s], frag[s], acc[j]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    if (s == 1 && (firstBlockImage + b + 1u) % NBB == 0) {
                        SWEEP_CLOCK(b0);
                        sweepBarrier();
#ifdef BSMR_SWEEP_STAMPS
                        barSum += __builtin_amdgcn_s_memtime() - b0;
#endif
                    }
                    const uint8_t* col = ring16 + ((firstBlockImage + b + 1u) % S16) * img16 + r * col16;
                    frag[s - 1] = *reinterpret_cast<const u32x4*>(col + (((4u * (s - 1) + g) ^ (r & SW16)) << 4));
                    if (s == KS - 1)
                        frag[s] = *reinterpret_cast<const u32x4*>(col + (((4u * s + g) ^ (r & SW16)) << 4));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (KS == 1) {
                if (more) {
                    if ((firstBlockImage + b + 1u) % NBB == 0) sweepBarrier();
                    request(firstBlockImage + b + 1u, frag);
                }
            }
            if constexpr (!FIRST) storeEntries(wPrev, val, base, s0Prev, nPrev);
            wPrev = wHere;
            s0Prev = s0;
            nPrev = s1 - s0;
            s0 = s1;
            if (more) wHere = wordLds[s0 + lane];
        };
        step(0, accA, accB, std::true_type{});
        uint32_t b = 1;
        for (; b + 1u < nb; b += 2u) {
            step(b, accB, accA, std::false_type{});
            step(b + 1u, accA, accB, std::false_type{});
        }
        float val = 0.f;
        uint32_t base = 0;
        if (b < nb) {
            step(b, accB, accA, std::false_type{});
            writeBack(accB, wPrev, val, base);
        } else {
            writeBack(accA, wPrev, val, base);
        }
        storeEntries(wPrev, val, base, s0Prev, nPrev);
    }
#ifdef BSMR_SWEEP_STAMPS
    if (stamps && wave == 0 && lane == 0) {
        uint64_t* o = stamps + (size_t)itemId * 32u;
        o[0] = tStart; o[1] = tMeta; o[2] = tPanels; o[3] = __builtin_amdgcn_s_memtime(); o[4] = barSum;
    }
#endif
}

}  // namespace bsmr
