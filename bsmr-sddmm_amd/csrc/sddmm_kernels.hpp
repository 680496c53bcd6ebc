// Hand-written gfx950 (CDNA4, wave64) kernels of the BSMR-SDDMM device path.
//
// What they compute is what the reference's live kernels compute
// (src/sddmmKernel.cu:213-351 dense blocks, :1994-2104 sparse residue): for every
// stored entry e = (i, j) of S,  P[e] = sum_k A[i,k] * B[k,j].  How they do it is
// designed for MI355X and shares nothing with the CUDA code.
//
// Dense path (denseGroups):
//  * Measured on MI355X the dense path is bound by the per-CU vector-memory
//    front end (TA), not by MFMA (idle > 90 %) nor by L2: gathering B columns is
//    what costs.  So the plan groups H consecutive row panels (16*H rows) into a
//    row GROUP and takes the union of their dense columns; a gathered 16-column
//    B block is then multiplied against all H panels (H MFMA tiles).  MFMA work
//    grows, B traffic shrinks (nips-like, H=4: 88 MB -> 46 MB).
//  * B is column-major: a column is K contiguous elements.  A 16-column block is
//    gathered with LDS-DMA (global_load_lds_dwordx4): each wave-instruction
//    moves 1 KiB made of whole 128-byte-line runs of columns (fragment-shaped
//    register loads - 16 rows x 64 B per instruction - cost twice the TA time).
//    The LDS image is lane-linear; the 16-byte pieces of a column are
//    XOR-swizzled on the SOURCE address so that the MFMA fragment reads
//    (ds_read_b128, 16 lanes = 16 columns at one k offset) are conflict-free.
//  * The four waves of a workgroup split a group's work by panel: each wave
//    keeps its own panel's A fragments in registers; B blocks are shared via LDS.
//  * Mask + destinations: per (block, panel) one tile of 256 destinations in
//    accumulator (lane-major) order - 4 bytes per lane in the staged form (8-bit
//    offsets into a per-row window of P that the workgroup assembles in LDS and
//    writes out coalesced), 8 or 16 bytes per lane in the direct form; independent
//    of anything the wave computes (the reference stores a 1 KiB row-major tile of
//    absolute 32-bit indices per block).  Tiles without entries are skipped
//    through a per-block bit mask.  Layouts: csrc/plan_pack.hpp.
//
// Sparse residue (sparseEntries): K split over LPE lanes per entry (coalesced
// 16-byte loads of the B column), A rows from LDS, butterfly reduction; exact
// fp32 with a defined summation order (bit-level CPU twin:
// oracle/sddmm_oracle.c oracle_sparse_twin).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bsmr {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

constexpr int kWave = 64;
constexpr int kThreads = 256;           // 4 waves per workgroup
constexpr int kWavesPerWG = kThreads / kWave;
constexpr int kMaxGroup = 4;            // row panels per group (H)

// One unit of dense work: blocks [first, first+count) (global block ids) of row group `group`.
// Strided batch (sddmm_gpu_batch of the reference, src/sddmmKernel.cu:2764-2869): problem b reads
// A + b * strideA, B + b * strideB and writes P + b * strideP (element strides); b = blockIdx.y.
struct Batch {
    uint64_t strideA, strideB, strideP;
    uint32_t count;
};

struct DenseItem {
    uint32_t group;
    uint32_t first;
    uint32_t count;
    uint32_t pad;
};

// One unit of sparse work: entries [start, start+count) of `panel`.
struct SparseItem {
    uint32_t panel;
    uint32_t start;
    uint32_t count;
    uint32_t pad;
};

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an
// L2).  Give each XCD a contiguous slice of the work list so that neighbouring
// items - consecutive blocks of one group, consecutive groups of one cluster,
// which share B columns - hit the same L2.  Speed only; any placement is correct.
__device__ __forceinline__ uint32_t xcdContiguous(uint32_t wg, uint32_t numWG) {
    const uint32_t xcd = wg & 7u, idx = wg >> 3;
    const uint32_t base = numWG >> 3, rem = numWG & 7u;  // XCD x owns base + (x < rem) items
    return xcd * base + (xcd < rem ? xcd : rem) + idx;
}

template <int MODE>
__device__ __forceinline__ u32x4 packLowp(const f32x4& lo, const f32x4& hi) {
    if constexpr (MODE == 0) {
        f16x8 o;
        o[0] = (_Float16)lo[0]; o[1] = (_Float16)lo[1]; o[2] = (_Float16)lo[2]; o[3] = (_Float16)lo[3];
        o[4] = (_Float16)hi[0]; o[5] = (_Float16)hi[1]; o[6] = (_Float16)hi[2]; o[7] = (_Float16)hi[3];
        return __builtin_bit_cast(u32x4, o);
    } else {
        bf16x8 o;
        o[0] = (__bf16)lo[0]; o[1] = (__bf16)lo[1]; o[2] = (__bf16)lo[2]; o[3] = (__bf16)lo[3];
        o[4] = (__bf16)hi[0]; o[5] = (__bf16)hi[1]; o[6] = (__bf16)hi[2]; o[7] = (__bf16)hi[3];
        return __builtin_bit_cast(u32x4, o);
    }
}

// ---------------------------------------------------------------------------
// fp32 -> fp16 / bf16 operand pass (8 elements per thread, 16-byte stores)
// ---------------------------------------------------------------------------
template <int MODE>  // 0 = fp16, 1 = bf16
__global__ void __launch_bounds__(kThreads)
convertOperands(const float* __restrict__ A, uint64_t nA8, const float* __restrict__ B, uint64_t nB8,
                uint16_t* __restrict__ A16, uint16_t* __restrict__ B16, bool slicedB) {
    // B is converted by the XCD that gathers it afterwards: the compute kernels give XCD x the x-th eighth of the
    // column range (items in column order, xcdContiguous), so workgroup w (XCD w & 7) takes the pieces of that
    // eighth of B and the converted lines are in the right L2 when the next kernel starts.  A goes first, in
    // plain order (every XCD reads all of it).  Grids of fewer than 8 workgroups convert B in plain order.
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < nA8; i += (uint64_t)gridDim.x * kThreads) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(A + i * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(A + i * 8 + 4);
        *reinterpret_cast<u32x4*>(A16 + i * 8) = packLowp<MODE>(lo, hi);
    }
    const bool sliced = slicedB && gridDim.x >= 8u;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint64_t slice = (nB8 + 7) / 8;
    const uint64_t b0 = sliced ? min(nB8, slice * xcd) : 0, b1 = sliced ? min(nB8, b0 + slice) : nB8;
    const uint64_t first = sliced ? blockIdx.x >> 3 : blockIdx.x;
    const uint64_t stride = sliced ? (gridDim.x + 7u - xcd) >> 3 : gridDim.x;   // workgroups of this XCD
    for (uint64_t i = b0 + first * kThreads + threadIdx.x; i < b1; i += stride * kThreads) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(B + i * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(B + i * 8 + 4);
        *reinterpret_cast<u32x4*>(B16 + i * 8) = packLowp<MODE>(lo, hi);
    }
}

// ---------------------------------------------------------------------------
// masked write-back of one 16x16 accumulator tile
//   lane l, register i  <->  tile row 4*(l>>4)+i, tile column l&15
//   tile element [4*l + i] = CSR index of the entry minus `rowBase` of its row,
//   or all-ones where S has no entry.  16-bit offsets serve every matrix whose
//   rows hold < 65535 entries; the 32-bit form is the fallback.
// ---------------------------------------------------------------------------
template <typename TileT> struct TileLoad;
template <> struct TileLoad<uint16_t> {
    typedef uint32_t raw __attribute__((ext_vector_type(2)));
    static constexpr bool windowed = false;
    static constexpr uint32_t kNull = 0xFFFFu;
    static __device__ __forceinline__ uint32_t get(const raw& v, int i) {
        const uint32_t w = i < 2 ? v[0] : v[1];
        return (i & 1) ? (w >> 16) : (w & 0xFFFFu);
    }
};
template <> struct TileLoad<uint8_t> {   // staged form: offsets into the item's per-row window
    typedef uint32_t raw;
    static constexpr bool windowed = true;
    static constexpr uint32_t kNull = 0xFFu;
    static __device__ __forceinline__ uint32_t get(const raw& v, int i) { return (v >> (8 * i)) & 0xFFu; }
};
// Mask form (the default whenever it applies): the sparse mask itself.  With the blocks in column-id order and sorted CSR
// rows the entries a tile holds of one row are consecutive positions of that row's window, so a tile is 16 x (16-bit
// column mask, 8-bit window offset of the row's first entry): 48 bytes instead of the 256 of the 8-bit form (the
// reference's tile: 1 KiB, src/BSMR.cpp:143-174).  Per 4 rows of a lane group: {mask01, mask23, four offsets}; lane
// (g, c) finds its destination as offset + popcount(mask below bit c).
struct TileMask {
    uint32_t word;
};
template <> struct TileLoad<TileMask> {
    typedef u32x3 raw;
    static constexpr bool windowed = true;
};
template <> struct TileLoad<uint32_t> {
    typedef u32x4 raw;
    static constexpr bool windowed = false;
    static constexpr uint32_t kNull = 0xFFFFFFFFu;
    static __device__ __forceinline__ uint32_t get(const raw& v, int i) { return v[i]; }
};

template <typename TileT>
__device__ __forceinline__ typename TileLoad<TileT>::raw loadTile(const TileT* __restrict__ tiles,
                                                                  size_t tileId, uint32_t lane) {
    if constexpr (sizeof(TileT) == sizeof(TileMask) && TileLoad<TileT>::windowed) {   // mask form: 12 words per tile
        const uint32_t* words = reinterpret_cast<const uint32_t*>(tiles) + tileId * 12u + (lane >> 4) * 3u;
        return u32x3{words[0], words[1], words[2]};
    } else {
        return *reinterpret_cast<const typename TileLoad<TileT>::raw*>(tiles + tileId * 256u + lane * 4u);
    }
}

template <typename TileT>
__device__ __forceinline__ void scatterTile(const f32x4& acc, const typename TileLoad<TileT>::raw& tile,
                                            const uint32_t* rowBase, float* __restrict__ P) {
    if constexpr (sizeof(TileT) == sizeof(TileMask) && TileLoad<TileT>::windowed) {
        const uint32_t c = threadIdx.x & 15u, below = (1u << c) - 1u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t m = ((i < 2 ? tile[0] : tile[1]) >> (16 * (i & 1))) & 0xFFFFu;
            const uint32_t first = (tile[2] >> (8 * i)) & 0xFFu;
            if ((m >> c) & 1u) P[rowBase[i] + first + __builtin_popcount(m & below)] = acc[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t off = TileLoad<TileT>::get(tile, i);
            if (off != TileLoad<TileT>::kNull) P[rowBase[i] + off] = acc[i];
        }
    }
}

template <int MODE>
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
    if constexpr (MODE == 0)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                      __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                       __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// dense kernel, 16-bit operands, K = 32*KS known at compile time (KS a power of 2).
//
// One workgroup = one DenseItem (a run of blocks of one row group).  The four
// waves split the work as (panel h = wave % H, block subset = wave / H): a wave
// keeps only ITS panel's A fragments in registers and multiplies them with every
// block of its subset, so a gathered B block is shared by the H panels of the
// group.  Blocks are processed in batches of NBW; the batch's B columns are
// gathered cooperatively (all four waves issue LDS-DMA) into one half of a
// double buffer while the previous batch is being multiplied out of the other.
//
// LDS image of a block: column c at c*2K bytes; its 16-byte piece w (k = 8w ..
// 8w+7) sits at piece slot w ^ (c & SW), SW = min(15, pieces-1).  LDS-DMA writes
// lane-linearly, so the swizzle is applied to the SOURCE address.  The MFMA
// fragment of lane (c = l&15, g = l>>4) at K step s is piece 4s+g of column c:
// the 16 lanes of a ds_read_b128 group hit 16 different 16-byte slots.
// ---------------------------------------------------------------------------
template <int KS, int H, int NBW, int MODE, typename TileT, bool LDS_STAGE = false>
__global__ void __launch_bounds__(kThreads)
denseGroups(const uint16_t* __restrict__ A16, const uint16_t* __restrict__ B16,
            const uint32_t* __restrict__ groupRows, const uint32_t* __restrict__ rowBaseTable,
            const uint16_t* __restrict__ winLen, const uint32_t* __restrict__ winMask,
            const uint32_t* __restrict__ blockCols, const TileT* __restrict__ tiles,
            const uint8_t* __restrict__ blockMask, const DenseItem* __restrict__ items, float* __restrict__ P, Batch batch) {
    A16 += blockIdx.y * batch.strideA;  // batched call: problem blockIdx.y of a strided batch
    B16 += blockIdx.y * batch.strideB;
    P += blockIdx.y * batch.strideP;
    // 8-bit tiles: offsets into a per-(item, row) window of P; rowBaseTable (window
    //   base), winLen and winMask are indexed by item.  With LDS_STAGE the windows are
    //   assembled in LDS and written out coalesced at the end (ownership bitmap =
    //   winMask); without it the results are scattered straight into the windows.
    // 16/32-bit tiles: offsets from the row's first dense entry; rowBaseTable is indexed
    //   by group; results are scattered straight to P.
    constexpr bool WINDOWED = TileLoad<TileT>::windowed;
    constexpr bool STAGED = WINDOWED && LDS_STAGE;
    static_assert(!LDS_STAGE || sizeof(TileT) == 1, "windows are staged in LDS from the 8-bit form only");
    constexpr uint32_t K = 32u * KS;
    constexpr uint32_t PC = 4u * KS;                 // 16-byte pieces per column
    constexpr uint32_t SW = PC - 1u < 15u ? PC - 1u : 15u;
    constexpr uint32_t rowBytes = 2u * K;
    constexpr uint32_t blkBytes = 16u * rowBytes;    // = KS KiB
    constexpr uint32_t NSUB = kWavesPerWG / H;       // block subsets (waves per panel)
    constexpr uint32_t MINE = (NBW + NSUB - 1) / NSUB;  // blocks of a batch one wave multiplies
    constexpr uint32_t CREG = (NBW + 3) / 4;         // registers holding the batch's column ids
    constexpr uint32_t DMAS = (NBW * KS + kWavesPerWG - 1) / kWavesPerWG;  // DMA instructions per wave and batch
    constexpr bool PRIVATE = H == 1 && NBW % kWavesPerWG == 0;  // wave-private B blocks, no barriers
    typedef typename TileLoad<TileT>::raw TileRaw;
    static_assert(NBW <= 16 && (NBW & (NBW - 1)) == 0, "NBW must be a power of two <= 16");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [2][NBW][blkBytes] (+ [16H][256] floats)
    float* stage = reinterpret_cast<float*>(lds + 2u * NBW * blkBytes);

    const uint32_t itemId = xcdContiguous(blockIdx.x, gridDim.x);
    const DenseItem item = items[itemId];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t r = lane & 15u;   // tile row for A, tile column for B and C
    const uint32_t g = lane >> 4;    // k group inside a 32-deep step / accumulator row group
    const uint32_t h = wave % H, sub = wave / H;
    const uint32_t rowSlot = item.group * (16u * H) + h * 16u;
    const uint32_t end = item.first + item.count;
    const uint32_t numBatches = (item.count + NBW - 1) / NBW;

    // this wave's panel: A fragments (once per item) and CSR row offsets
    u32x4 a[KS];
    {
        const uint16_t* aRow = A16 + (size_t)groupRows[rowSlot + r] * K + g * 8u;
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = *reinterpret_cast<const u32x4*>(aRow + s * 32);
    }
    uint32_t rowBase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        rowBase[i] = STAGED ? (h * 16u + 4u * g + i) * 256u   // float index of the row's LDS window
                     : WINDOWED ? rowBaseTable[(size_t)itemId * (16u * H) + h * 16u + 4u * g + i]
                                : rowBaseTable[rowSlot + 4u * g + i];

    // batch metadata: cols[q] of lane l = column (l & 15) of block b0 + 4q + (l >> 4);
    // maskReg of lane l = tile mask of block b0 + (l % NBW) (0 past the end)
    auto loadMeta = [&](uint32_t b0, uint32_t (&cols)[CREG], uint32_t& maskReg) {
#pragma unroll
        for (uint32_t q = 0; q < CREG; ++q)
            cols[q] = blockCols[(size_t)min(b0 + 4u * q + g, end - 1u) * 16u + r];
        const uint32_t mb = b0 + (lane & (NBW - 1u));
        maskReg = mb < end ? (uint32_t)blockMask[mb] : 0u;
    };
    // this wave's share of the batch's gather: DMA instruction i = n*KS + j moves
    // flat piece slots [64j, 64j+64) of block n
    auto issueGather = [&](uint32_t buf, const uint32_t (&cols)[CREG]) {
        uint8_t* base = lds + buf * (NBW * blkBytes);
#pragma unroll
        for (uint32_t d = 0; d < DMAS; ++d) {
            // H == 1: a wave gathers exactly the blocks it multiplies (n = wave, wave+4, ...),
            // so the waves never read each other's LDS and need no barrier.  H > 1: the
            // batch is shared by the panels, instructions are dealt round-robin.
            uint32_t n, j;
            if constexpr (PRIVATE) {
                n = wave + kWavesPerWG * (d / KS);
                j = d % KS;
            } else {
                const uint32_t i = d * kWavesPerWG + wave;
                if (NBW * KS % kWavesPerWG != 0 && i >= NBW * KS) break;
                n = i / KS;
                j = i % KS;
            }
            const uint32_t f = 64u * j + lane;
            const uint32_t col = f / PC, t = f % PC;
            uint32_t cid = 0;
#pragma unroll
            for (uint32_t q = 0; q < CREG; ++q) {  // n is wave-uniform: pick the register, then the lane
                const uint32_t v = __shfl(cols[q], ((n & 3u) << 4) + col);
                if ((n >> 2) == q) cid = v;
            }
            const uint16_t* src = B16 + (size_t)cid * K + ((t ^ (col & SW)) << 3);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)src,
                (__attribute__((address_space(3))) void*)(base + n * blkBytes + j * 1024u), 16, 0, 0);
        }
    };
    // destination tiles of the blocks this wave multiplies (block n = sub + NSUB*m)
    auto loadTiles = [&](uint32_t b0, uint32_t maskReg, TileRaw (&tile)[MINE]) {
#pragma unroll
        for (uint32_t m = 0; m < MINE; ++m) {
            const uint32_t n = sub + NSUB * m;
            const uint32_t mk = n < NBW ? (uint32_t)__builtin_amdgcn_readlane(maskReg, n & (NBW - 1u)) : 0u;
            if (mk & (1u << h)) tile[m] = loadTile<TileT>(tiles, (size_t)(b0 + n) * H + h, lane);
        }
    };

    uint32_t colsCur[CREG], colsNext[CREG];
    uint32_t maskCur = 0, maskNext = 0;
    TileRaw tileCur[MINE], tileNext[MINE];
    loadMeta(item.first, colsCur, maskCur);
    issueGather(0, colsCur);
    loadTiles(item.first, maskCur, tileCur);
    if (numBatches > 1) loadMeta(item.first + NBW, colsNext, maskNext);

    // The masked scatter of a batch is deferred by one batch: the stores are issued
    // right after the barrier, together with the next gather, so that their
    // completion (which the barrier's vmcnt(0) has to wait for - gfx950 has one
    // counter for loads and stores) overlaps the gather latency instead of adding to it.
    f32x4 pendAcc[MINE];
    TileRaw pendTile[MINE];
    uint32_t pendBits = 0;  // wave-uniform: bit m = pendAcc[m] holds results to write

    for (uint32_t it = 0; it < numBatches; ++it) {
        const uint32_t b0 = item.first + it * NBW;
        // batch `it` has landed (every wave drains its own DMA, then all meet); the
        // other buffer is free because every wave finished batch it-1 before arriving.
        if constexpr (PRIVATE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else __syncthreads();
        if constexpr (!STAGED) {
#pragma unroll
            for (uint32_t m = 0; m < MINE; ++m)
                if (pendBits & (1u << m)) scatterTile<TileT>(pendAcc[m], pendTile[m], rowBase, P);
            pendBits = 0;
        }
        if (it + 1 < numBatches) {
            issueGather((it + 1) & 1u, colsNext);
            loadTiles(b0 + NBW, maskNext, tileNext);
        }
        uint32_t colsAfter[CREG];
        uint32_t maskAfter = 0;
        if (it + 2 < numBatches) loadMeta(b0 + 2 * NBW, colsAfter, maskAfter);

        const uint8_t* buf = lds + (it & 1u) * (NBW * blkBytes);
#pragma unroll
        for (uint32_t m = 0; m < MINE; ++m) {
            const uint32_t n = sub + NSUB * m;
            if (n >= NBW) break;
            const uint32_t mk = __builtin_amdgcn_readlane(maskCur, n & (NBW - 1u));
            if (!(mk & (1u << h))) continue;  // wave-uniform: this panel has no entry in the block
            const uint8_t* bCol = buf + n * blkBytes + r * rowBytes;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const u32x4 bv = *reinterpret_cast<const u32x4*>(bCol + (((4u * s + g) ^ (r & SW)) << 4));
                acc = mfma16<MODE>(a[s], bv, acc);
            }
            if constexpr (STAGED) {
                scatterTile<TileT>(acc, tileCur[m], rowBase, stage);  // ds_write into the row windows
            } else {
                pendAcc[m] = acc;
                pendTile[m] = tileCur[m];
                pendBits |= 1u << m;
            }
        }
#pragma unroll
        for (uint32_t m = 0; m < MINE; ++m) tileCur[m] = tileNext[m];
#pragma unroll
        for (uint32_t q = 0; q < CREG; ++q) colsNext[q] = colsAfter[q];
        maskCur = maskNext;
        maskNext = maskAfter;
    }
    if constexpr (STAGED) {
        // write the windows out: every wave takes rows wave, wave+4, ...; lanes walk the
        // window, so a wave-instruction stores up to 256 contiguous bytes of P.  Window
        // positions this item does not own (sparse path, other items) are skipped.
        __syncthreads();
        for (uint32_t row = wave; row < 16u * H; row += kWavesPerWG) {
            const size_t slot = (size_t)itemId * (16u * H) + row;
            const uint32_t len = winLen[slot];
            const uint32_t base = rowBaseTable[slot];
            for (uint32_t j = lane; j < len; j += kWave) {
                const uint32_t word = winMask[slot * 8u + (j >> 5)];
                if ((word >> (j & 31u)) & 1u) P[base + j] = stage[row * 256u + j];
            }
        }
    } else {
#pragma unroll
        for (uint32_t m = 0; m < MINE; ++m)
            if (pendBits & (1u << m)) scatterTile<TileT>(pendAcc[m], pendTile[m], rowBase, P);
    }
}

// ---------------------------------------------------------------------------
// dense kernel for ungrouped plans (H = 1), K = 32*KS: the streaming form.
//
// Ablation of denseGroups on the nips-like matrix showed that more than half of
// its time is a fixed skeleton: every batch ends in `s_waitcnt vmcnt(0)`, which
// waits for that iteration's metadata loads and for the scattered stores (gfx950
// has ONE counter for loads and stores), i.e. one full memory round trip per batch.
// Here a wave owns up to MAXB blocks of the item (block wave + 4m) and
//   * loads ALL of its metadata (column ids, destination tiles, A fragments) in the
//     prologue and drains it before the first gather is issued,
//   * then runs a loop whose only vector-memory operations are the LDS-DMA gathers
//     of a wave-private double buffer, so the wait for block m is the counted
//     `vmcnt(KS)` (block m+1 stays in flight) - no barriers, no round trip per block,
//   * keeps the accumulators of all its blocks in registers and scatters them after
//     the loop, when nothing has to wait for the stores.
// ---------------------------------------------------------------------------
// block images in a wave's gather ring.  Three images (two gathers in flight) measured the
// same as two on MI355X (10.0 vs 10.1 us, nips-like K = 128): the loop is bound by gather
// throughput, not by the latency of one gather, so the smaller footprint is kept.
constexpr uint32_t streamSlots(int /*KS*/) { return 2u; }

// H > 1 (one-wave form only): the wave computes H consecutive panels of a row group for every block it
// gathers - one LDS read of a B fragment feeds H MFMAs, and the group's union of columns is gathered once.
template <int KS, int MODE, typename TileT, int MAXB = 8, int WAVES = kWavesPerWG, int H = 1>
__global__ void __launch_bounds__(WAVES * kWave)
denseStream(const uint16_t* __restrict__ A16, const uint16_t* __restrict__ B16,
            const uint32_t* __restrict__ groupRows, const uint32_t* __restrict__ rowBaseTable,
            const uint32_t* __restrict__ blockCols, const TileT* __restrict__ tiles,
            const DenseItem* __restrict__ items, float* __restrict__ P, Batch batch) {
    A16 += blockIdx.y * batch.strideA;  // batched call: problem blockIdx.y of a strided batch
    B16 += blockIdx.y * batch.strideB;
    P += blockIdx.y * batch.strideP;
    constexpr uint32_t K = 32u * KS;
    // A block is gathered as SPLIT images of KSL 32-wide k steps each (K = 512: two images of 8 KB), so
    // that a wave's ring stays at 16 KB and ten waves fit a CU; the accumulator carries over the images.
    constexpr uint32_t KSL = KS > 8 ? 8u : (uint32_t)KS;
    constexpr uint32_t SPLIT = (uint32_t)KS / KSL;
    constexpr uint32_t PC = 4u * KSL;                 // 16-byte pieces per column of an image
    constexpr uint32_t SW = PC - 1u < 15u ? PC - 1u : 15u;
    constexpr uint32_t rowBytes = 64u * KSL;          // one column of an image
    constexpr uint32_t blkBytes = 16u * rowBytes;     // one image
    constexpr bool WINDOWED = TileLoad<TileT>::windowed;
    constexpr uint32_t CREG = (MAXB + 3) / 4;
    constexpr uint32_t SLOTS = streamSlots(KS);
    typedef typename TileLoad<TileT>::raw TileRaw;

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [4 waves][SLOTS][blkBytes]
    const uint32_t itemId = xcdContiguous(blockIdx.x, gridDim.x);
    const DenseItem item = items[itemId];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t r = lane & 15u, g = lane >> 4;
    uint8_t* myLds = lds + wave * (SLOTS * blkBytes);
    // blocks of this wave: item.first + wave + WAVES * m, m < myCount
    const uint32_t myCount =
        item.count > wave ? min((item.count - wave + (uint32_t)WAVES - 1u) / (uint32_t)WAVES, (uint32_t)MAXB) : 0u;
    if (myCount == 0) return;
    const uint32_t myFirst = item.first + wave;

    // ---- prologue, two dependent round trips after the item record ----
    // 1. what the gathers and the A fragments are addressed with
    uint32_t cols[CREG];   // lane l, register q: column (l & 15) of my block 4q + (l >> 4)
#pragma unroll
    for (uint32_t q = 0; q < CREG; ++q)
        cols[q] = blockCols[(size_t)(myFirst + (uint32_t)WAVES * min(4u * q + g, myCount - 1u)) * 16u + r];
    static_assert(H == 1 || WAVES == 1, "several panels per wave only in the one-wave form");
    const uint32_t rowSlot = item.group * (16u * H);
    uint32_t myRow[H];
#pragma unroll
    for (int h = 0; h < H; ++h) myRow[h] = groupRows[rowSlot + 16u * h + r];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (uint32_t q = 0; q < CREG; ++q) asm volatile("" : "+v"(cols[q]));
#pragma unroll
    for (int h = 0; h < H; ++h) asm volatile("" : "+v"(myRow[h]));
    // ring of SLOTS block images per wave: SLOTS-1 gathers stay in flight (a gather takes
    // ~1200 cycles under load, a block's MFMAs ~400: measured with in-kernel stamps)
    auto gather = [&](uint32_t u) {  // image u = (my block u / SPLIT, k range u % SPLIT) -> slot u % SLOTS
        const uint32_t m = u / SPLIT, h = u % SPLIT;
        uint8_t* dst = myLds + (u % SLOTS) * blkBytes;
#pragma unroll
        for (int j = 0; j < (int)KSL; ++j) {
            const uint32_t f = 64u * j + lane;
            const uint32_t col = f / PC, t = f % PC;
            const uint32_t cid = __shfl(cols[m >> 2], ((m & 3u) << 4) + col);
            const uint16_t* src = B16 + (size_t)cid * K + h * (32u * KSL) + ((t ^ (col & SW)) << 3);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)src,
                (__attribute__((address_space(3))) void*)(dst + j * 1024u), 16, 0, 0);
        }
    };
    auto waitInFlight = [&](uint32_t images) {  // all but the youngest `images` gathers have landed
        if (images == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (images * KSL == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (images * KSL == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (images * KSL == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (images * KSL == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    };

    // 2. first gather, then destinations and row bases, the A fragments, and the second gather: all in
    //    flight together.  The A loads are issued through inline assembly so that the compiler, which
    //    cannot count past an LDS-DMA, does not put a vmcnt(0) in front of their first use; the counted
    //    wait of the first image (all but the youngest KSL operations) covers them.  The compiler takes
    //    an asm load's registers for written when the statement ends and may copy them before the data
    //    has landed (round 2's K = 512 fault; `make check-isa` found more instances of it): so from the
    //    A loads to the statement that pins the registers behind the wait the code is ONE straight line
    //    - loads, second gather (issued unconditionally: a wave with a single image gathers it twice),
    //    wait, pin - with nothing for the register allocator to split a live range at.
    f32x4 acc[MAXB][H];
    gather(0);
    u32x4 a[H][KS];
    TileRaw tile[MAXB][H];
#pragma unroll
    for (uint32_t m = 0; m < (uint32_t)MAXB; ++m)
        if (m < myCount) {
#pragma unroll
            for (int h = 0; h < H; ++h)
                tile[m][h] = loadTile<TileT>(tiles, (size_t)(myFirst + (uint32_t)WAVES * m) * H + h, lane);
        }
    uint32_t rowBase[H][4];
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rowBase[h][i] = WINDOWED ? rowBaseTable[(size_t)itemId * (16u * H) + 16u * h + 4u * g + i]
                                     : rowBaseTable[rowSlot + 16u * h + 4u * g + i];
    const uint32_t units = myCount * SPLIT;
    static_assert(SLOTS == 2, "the first wait below leaves exactly one image in flight");
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const uint16_t* aRow = A16 + (size_t)myRow[h] * K + g * 8u;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[h][s]) : "v"(aRow + s * 32) : "memory");
    }
    gather(min(1u, units - 1u));
    waitInFlight(1);
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(a[h][s]));
#pragma unroll
    for (uint32_t m = 0; m < (uint32_t)MAXB; ++m) {
        if (m >= myCount) break;  // wave-uniform
        f32x4 c[H];
#pragma unroll
        for (int h = 0; h < H; ++h) c[h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (uint32_t half = 0; half < SPLIT; ++half) {
            const uint32_t u = m * SPLIT + half;
            if (u != 0) {   // (image 0: waited for above, with the A fragments)
                // slot (u + SLOTS - 1) % SLOTS held image u-1, whose reads returned before its MFMAs
                if (u + SLOTS - 1 < units) gather(u + SLOTS - 1);
                const uint32_t younger = min(units - 1u - u, SLOTS - 1u);  // gathers issued after image u's
                if (younger == 0) waitInFlight(0);
                else waitInFlight(1);
            }
            const uint8_t* bCol = myLds + (u % SLOTS) * blkBytes + r * rowBytes;
#pragma unroll
            for (int s = 0; s < (int)KSL; ++s) {
                const u32x4 bv = *reinterpret_cast<const u32x4*>(bCol + (((4u * s + g) ^ (r & SW)) << 4));
#pragma unroll
                for (int h = 0; h < H; ++h) c[h] = mfma16<MODE>(a[h][half * KSL + s], bv, c[h]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int h = 0; h < H; ++h) acc[m][h] = c[h];
    }
#pragma unroll
    for (uint32_t m = 0; m < (uint32_t)MAXB; ++m)
        if (m < myCount) {
#pragma unroll
            for (int h = 0; h < H; ++h) scatterTile<TileT>(acc[m][h], tile[m][h], rowBase[h], P);
        }
}

// ---------------------------------------------------------------------------
// streaming dense kernel on fp32 operands for K = 32 and 64 (KS = 1, 2): the conversion pass and the kernel boundary
// behind it cost more than they save when a column is 128 or 256 bytes (nips-like K = 32: conversion 2.4 us + boundary
// against a dense kernel of 6 us).  Same structure as denseStream in its one-wave form: the fp32 columns of a block
// are gathered by LDS-DMA into a wave-private ring of two images (an image has the geometry of a 16-bit image of 2 K),
// a lane reads its 8 consecutive k as two 16-byte pieces and rounds them to fp16 / bf16 in registers with the casts of
// convertOperands (packLowp), so the MFMA operands - and the results - are bit for bit those of the two-kernel path.
// ---------------------------------------------------------------------------
template <int KS, int MODE, typename TileT, int MAXB = 8>
__global__ void __launch_bounds__(kWave)
denseStreamCvt(const float* __restrict__ A, const float* __restrict__ B, const uint32_t* __restrict__ groupRows,
               const uint32_t* __restrict__ rowBaseTable, const uint32_t* __restrict__ blockCols,
               const TileT* __restrict__ tiles, const DenseItem* __restrict__ items, float* __restrict__ P, Batch batch) {
    static_assert(KS == 1 || KS == 2 || KS == 4, "K = 32, 64 or 128");
    A += blockIdx.y * batch.strideA;
    B += blockIdx.y * batch.strideB;
    P += blockIdx.y * batch.strideP;
    constexpr uint32_t K = 32u * KS;
    constexpr uint32_t DMA = 2u * KS;                 // 1 KiB LDS-DMA instructions per image
    constexpr uint32_t PC = 8u * KS;                  // 16-byte pieces per fp32 column
    constexpr uint32_t SW = PC - 1u < 15u ? PC - 1u : 15u;
    constexpr uint32_t rowBytes = 128u * KS;          // one column of an image
    constexpr uint32_t blkBytes = 16u * rowBytes;     // one image
    constexpr bool WINDOWED = TileLoad<TileT>::windowed;
    constexpr uint32_t CREG = (MAXB + 3) / 4;
    typedef typename TileLoad<TileT>::raw TileRaw;

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [2][blkBytes]
    const uint32_t itemId = xcdContiguous(blockIdx.x, gridDim.x);
    const DenseItem item = items[itemId];
    const uint32_t lane = threadIdx.x & 63u, r = lane & 15u, g = lane >> 4;
    const uint32_t myCount = min(item.count, (uint32_t)MAXB);
    if (myCount == 0) return;
    const uint32_t myFirst = item.first;

    uint32_t cols[CREG];   // lane l, register q: column (l & 15) of block 4q + (l >> 4)
#pragma unroll
    for (uint32_t q = 0; q < CREG; ++q) cols[q] = blockCols[(size_t)(myFirst + min(4u * q + g, myCount - 1u)) * 16u + r];
    const uint32_t rowSlot = item.group * 16u;
    uint32_t myRow = groupRows[rowSlot + r];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (uint32_t q = 0; q < CREG; ++q) asm volatile("" : "+v"(cols[q]));
    asm volatile("" : "+v"(myRow));

    auto gather = [&](uint32_t m) {  // block m -> slot m % 2
        uint8_t* dst = lds + (m & 1u) * blkBytes;
#pragma unroll
        for (int j = 0; j < (int)DMA; ++j) {
            const uint32_t f = 64u * j + lane;
            const uint32_t col = f / PC, t = f % PC;
            const uint32_t cid = __shfl(cols[m >> 2], ((m & 3u) << 4) + col);
            const float* src = B + (size_t)cid * K + ((t ^ (col & SW)) << 2);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)src,
                (__attribute__((address_space(3))) void*)(dst + j * 1024u), 16, 0, 0);
        }
    };
    auto waitForAllButNewestGather = [&]() {
        if constexpr (DMA == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if constexpr (DMA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    };

    gather(0);
    f32x4 aLo[KS], aHi[KS];
    TileRaw tile[MAXB];
#pragma unroll
    for (uint32_t m = 0; m < (uint32_t)MAXB; ++m)
        if (m < myCount) tile[m] = loadTile<TileT>(tiles, (size_t)(myFirst + m), lane);
    uint32_t rowBase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        rowBase[i] = WINDOWED ? rowBaseTable[(size_t)itemId * 16u + 4u * g + i] : rowBaseTable[rowSlot + 4u * g + i];

    f32x4 acc[MAXB];
    u32x4 a[KS];
    // the A rows of the panel as fp32 fragments (inline assembly, and one straight line from the loads to the pin behind
    // the wait: see denseStream), rounded once they have landed
    {
        const float* aRow = A + (size_t)myRow * K + g * 8u;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(aLo[s]) : "v"(aRow + s * 32) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(aHi[s]) : "v"(aRow + s * 32 + 4) : "memory");
        }
        gather(min(1u, myCount - 1u));     // (a wave with one block gathers it twice: the wait below stays the same)
        waitForAllButNewestGather();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            asm volatile("" : "+v"(aLo[s]));
            asm volatile("" : "+v"(aHi[s]));
            a[s] = packLowp<MODE>(aLo[s], aHi[s]);
        }
    }
#pragma unroll
    for (uint32_t m = 0; m < (uint32_t)MAXB; ++m) {
        if (m >= myCount) break;  // wave-uniform
        if (m != 0) {
            if (m + 1 < myCount) {
                gather(m + 1);
                waitForAllButNewestGather();
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        const uint8_t* bCol = lds + (m & 1u) * blkBytes + r * rowBytes;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint32_t p0 = 8u * s + 2u * g;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(bCol + ((p0 ^ (r & SW)) << 4));
            const f32x4 hi = *reinterpret_cast<const f32x4*>(bCol + (((p0 + 1u) ^ (r & SW)) << 4));
            c = mfma16<MODE>(a[s], packLowp<MODE>(lo, hi), c);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc[m] = c;
    }
#pragma unroll
    for (uint32_t m = 0; m < (uint32_t)MAXB; ++m)
        if (m < myCount) scatterTile<TileT>(acc[m], tile[m], rowBase, P);
}

// ---------------------------------------------------------------------------
// dense fallback, 16-bit operands, any K (multiple of 32): fragments straight
// from global memory, run-time K loop.  One wave per DenseItem.
// ---------------------------------------------------------------------------
template <int MODE, typename TileT>
__global__ void __launch_bounds__(kThreads)
denseGroupsAnyK(const uint16_t* __restrict__ A16, const uint16_t* __restrict__ B16, uint32_t K, uint32_t H,
                const uint32_t* __restrict__ groupRows, const uint32_t* __restrict__ groupRowBase,
                const uint32_t* __restrict__ blockCols, const TileT* __restrict__ tiles,
                const uint8_t* __restrict__ blockMask, const DenseItem* __restrict__ items,
                uint32_t numItems, float* __restrict__ P, Batch batch) {
    A16 += blockIdx.y * batch.strideA;  // batched call: problem blockIdx.y of a strided batch
    B16 += blockIdx.y * batch.strideB;
    P += blockIdx.y * batch.strideP;
    const uint32_t itemId = blockIdx.x * kWavesPerWG + (threadIdx.x >> 6);
    if (itemId >= numItems) return;
    const DenseItem item = items[itemId];
    const uint32_t lane = threadIdx.x & 63u, r = lane & 15u, g = lane >> 4;
    const uint32_t steps = K >> 5;
    for (uint32_t b = item.first; b < item.first + item.count; ++b) {
        const uint32_t mask = blockMask[b];
        const uint16_t* bCol = B16 + (size_t)blockCols[(size_t)b * 16u + r] * K + g * 8u;
        for (uint32_t h = 0; h < H; ++h) {
            if (!(mask & (1u << h))) continue;
            const uint32_t slot = item.group * 16u * H + h * 16u;
            const uint16_t* aRow = A16 + (size_t)groupRows[slot + r] * K + g * 8u;
            const typename TileLoad<TileT>::raw tile = loadTile<TileT>(tiles, (size_t)b * H + h, lane);
            uint32_t rowBase[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rowBase[i] = groupRowBase[(TileLoad<TileT>::windowed ? itemId * 16u * H + h * 16u : slot) + 4u * g + i];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (uint32_t s = 0; s < steps; ++s) {
                const u32x4 av = *reinterpret_cast<const u32x4*>(aRow + s * 32u);
                const u32x4 bv = *reinterpret_cast<const u32x4*>(bCol + s * 32u);
                acc = mfma16<MODE>(av, bv, acc);
            }
            scatterTile<TileT>(acc, tile, rowBase, P);
        }
    }
}

// ---------------------------------------------------------------------------
// dense kernel for a SMALL dense part: fp32 operands converted to fp16 / bf16 in
// registers (round-to-nearest-even, the same values the convertOperands pass
// would produce), so the full-matrix conversion pass can be skipped when only a
// few blocks are dense.  One wave per DenseItem, fragments from global memory.
// ---------------------------------------------------------------------------

template <int MODE, typename TileT>
__global__ void __launch_bounds__(kThreads)
denseGroupsCvt(const float* __restrict__ A, const float* __restrict__ B, uint32_t K, uint32_t H,
               const uint32_t* __restrict__ groupRows, const uint32_t* __restrict__ groupRowBase,
               const uint32_t* __restrict__ blockCols, const TileT* __restrict__ tiles,
               const uint8_t* __restrict__ blockMask, const DenseItem* __restrict__ items,
               uint32_t numItems, float* __restrict__ P, Batch batch) {
    A += blockIdx.y * batch.strideA;  // batched call: problem blockIdx.y of a strided batch
    B += blockIdx.y * batch.strideB;
    P += blockIdx.y * batch.strideP;
    const uint32_t itemId = blockIdx.x * kWavesPerWG + (threadIdx.x >> 6);
    if (itemId >= numItems) return;
    const DenseItem item = items[itemId];
    const uint32_t lane = threadIdx.x & 63u, r = lane & 15u, g = lane >> 4;
    const uint32_t steps = K >> 5;
    for (uint32_t b = item.first; b < item.first + item.count; ++b) {
        const uint32_t mask = blockMask[b];
        const float* bCol = B + (size_t)blockCols[(size_t)b * 16u + r] * K + g * 8u;
        for (uint32_t h = 0; h < H; ++h) {
            if (!(mask & (1u << h))) continue;
            const uint32_t slot = item.group * 16u * H + h * 16u;
            const float* aRow = A + (size_t)groupRows[slot + r] * K + g * 8u;
            const typename TileLoad<TileT>::raw tile = loadTile<TileT>(tiles, (size_t)b * H + h, lane);
            uint32_t rowBase[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rowBase[i] = groupRowBase[(TileLoad<TileT>::windowed ? itemId * 16u * H + h * 16u : slot) + 4u * g + i];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (uint32_t s = 0; s < steps; ++s) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(aRow + s * 32u);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(aRow + s * 32u + 4u);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bCol + s * 32u);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bCol + s * 32u + 4u);
                acc = mfma16<MODE>(packLowp<MODE>(a0, a1), packLowp<MODE>(b0, b1), acc);
            }
            scatterTile<TileT>(acc, tile, rowBase, P);
        }
    }
}

// ---------------------------------------------------------------------------
// dense kernel, exact fp32: v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain.
// Lane (r, g) loads float4 chunks [16t + 4g, +4) of its row / column; MFMA
// number (t, j) multiplies element j of every chunk, so inside it lane group g
// supplies k = 16t + 4g + j.  Chain order of k: for t, for j, for g.
// (CPU twin: oracle_dense_f32_twin.)  One wave per DenseItem.
// ---------------------------------------------------------------------------
template <typename TileT>
__global__ void __launch_bounds__(kThreads)
denseGroupsF32(const float* __restrict__ A, const float* __restrict__ B, uint32_t K, uint32_t H,
               const uint32_t* __restrict__ groupRows, const uint32_t* __restrict__ groupRowBase,
               const uint32_t* __restrict__ blockCols, const TileT* __restrict__ tiles,
               const uint8_t* __restrict__ blockMask, const DenseItem* __restrict__ items,
               uint32_t numItems, float* __restrict__ P, Batch batch) {
    A += blockIdx.y * batch.strideA;  // batched call: problem blockIdx.y of a strided batch
    B += blockIdx.y * batch.strideB;
    P += blockIdx.y * batch.strideP;
    const uint32_t itemId = blockIdx.x * kWavesPerWG + (threadIdx.x >> 6);
    if (itemId >= numItems) return;
    const DenseItem item = items[itemId];
    const uint32_t lane = threadIdx.x & 63u, r = lane & 15u, g = lane >> 4;
    const uint32_t steps = K >> 4;
    for (uint32_t b = item.first; b < item.first + item.count; ++b) {
        const uint32_t mask = blockMask[b];
        const float* bCol = B + (size_t)blockCols[(size_t)b * 16u + r] * K + g * 4u;
        for (uint32_t h = 0; h < H; ++h) {
            if (!(mask & (1u << h))) continue;
            const uint32_t slot = item.group * 16u * H + h * 16u;
            const float* aRow = A + (size_t)groupRows[slot + r] * K + g * 4u;
            const typename TileLoad<TileT>::raw tile = loadTile<TileT>(tiles, (size_t)b * H + h, lane);
            uint32_t rowBase[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rowBase[i] = groupRowBase[(TileLoad<TileT>::windowed ? itemId * 16u * H + h * 16u : slot) + 4u * g + i];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (uint32_t t = 0; t < steps; ++t) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(aRow + t * 16u);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bCol + t * 16u);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc, 0, 0, 0);
            }
            scatterTile<TileT>(acc, tile, rowBase, P);
        }
    }
}

// ---------------------------------------------------------------------------
// residual sparse kernel (fp32).  One workgroup = one SparseItem.  The panel's
// 16 A rows are staged in LDS (row stride K+4 floats).  LPE lanes share one
// entry: lane t accumulates the float4 chunks q = t, t+LPE, ... of the dot
// product with one fmaf chain (ascending k), then a butterfly over the LPE lanes.
// ---------------------------------------------------------------------------
constexpr int kSparseLdsPad = 4;  // floats

// CPL > 0: K = 4 * LPE * CPL, every lane's CPL column chunks are requested before the first
// is used (the residue is a gather: what matters is bytes in flight); CPL = 0: any K.
// FREE: the residue in global (column, row) order - no panels; `panelRows` then holds one row id per entry
// and A rows are gathered like B columns (plan_pack.hpp, "Panel form or free form?").
template <int LPE, bool A_IN_LDS, int CPL = 0, bool FREE = false>
__global__ void __launch_bounds__(kThreads)
sparseEntries(const float* __restrict__ A, const float* __restrict__ B, uint32_t K,
              const uint32_t* __restrict__ panelRows, const uint32_t* __restrict__ entryCol,
              const uint32_t* __restrict__ entryDst, const uint8_t* __restrict__ entryRow,
              const SparseItem* __restrict__ items, float* __restrict__ P, Batch batch) {
    A += blockIdx.y * batch.strideA;  // batched call: problem blockIdx.y of a strided batch
    B += blockIdx.y * batch.strideB;
    P += blockIdx.y * batch.strideP;
    extern __shared__ __attribute__((aligned(16))) float panelA[];
    const SparseItem item = items[xcdContiguous(blockIdx.x, gridDim.x)];
    const uint32_t chunks = K >> 2;  // float4 chunks per row
    const uint32_t ldsStride = K + kSparseLdsPad;

    constexpr uint32_t groups = kThreads / LPE;
    const uint32_t group = threadIdx.x / LPE;
    const uint32_t t = threadIdx.x % LPE;
    // every group runs the same number of rounds so that all lanes of a wave
    // reach the shuffles together
    const uint32_t rounds = (item.count + groups - 1) / groups;

    // Tuned shapes over a staged panel: the chain item -> panel rows -> A rows -> barrier -> (entry -> B column)
    // per round is what a workgroup's time consists of, so the entries of two rounds and the B chunks of the
    // next round are requested ahead: round 0's before the panel is staged, round r+1's before round r is summed.
    // Same per-lane arithmetic in the same order as the plain loop below.
    // (with 8 chunks per lane a second B buffer costs occupancy - Trefethen_20000 K=256 25.4 -> 29.5 us - so those
    // shapes only keep the entries ahead and request round r+1's chunks right after round r is summed)
    if constexpr (CPL > 0 && A_IN_LDS && !FREE) {
        constexpr bool TWO_BUFFERS = CPL <= 4;
        auto entryOf = [&](uint32_t round, uint32_t& idx, uint32_t& col, uint32_t& row) {
            const uint32_t e = round * groups + group;
            idx = item.start + (e < item.count ? e : 0u);
            col = entryCol[idx];
            row = entryRow[idx];
        };
        auto request = [&](f32x4 (&bv)[CPL], uint32_t col) {
            const float* bCol = B + (size_t)col * K;
#pragma unroll
            for (int c = 0; c < CPL; ++c) bv[c] = *reinterpret_cast<const f32x4*>(bCol + (t + c * LPE) * 4u);
        };
        auto finish = [&](const f32x4 (&bv)[CPL], uint32_t round, uint32_t idx, uint32_t row) {
            const float* aRow = panelA + row * ldsStride;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(aRow + (t + c * LPE) * 4u);
                acc = __builtin_fmaf(av[0], bv[c][0], acc);
                acc = __builtin_fmaf(av[1], bv[c][1], acc);
                acc = __builtin_fmaf(av[2], bv[c][2], acc);
                acc = __builtin_fmaf(av[3], bv[c][3], acc);
            }
#pragma unroll
            for (int off = LPE / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, LPE);
            if (round * groups + group < item.count && t == 0) P[entryDst[idx]] = acc;
        };
        uint32_t idx0, col0, row0, idx1, col1, row1;
        entryOf(0, idx0, col0, row0);
        entryOf(1, idx1, col1, row1);
        f32x4 bvA[CPL], bvB[TWO_BUFFERS ? CPL : 1];
        request(bvA, col0);
        for (uint32_t i = threadIdx.x; i < 16u * chunks; i += kThreads) {
            const uint32_t row = i / chunks, q = i - row * chunks;
            const f32x4 v = *reinterpret_cast<const f32x4*>(
                A + (size_t)panelRows[item.panel * 16u + row] * K + q * 4u);
            *reinterpret_cast<f32x4*>(panelA + row * ldsStride + q * 4u) = v;
        }
        __syncthreads();
        if constexpr (TWO_BUFFERS) {
            for (uint32_t round = 0; round < rounds; round += 2) {
                uint32_t idx2 = 0, col2 = 0, row2 = 0, idx3 = 0, col3 = 0, row3 = 0;
                if (round + 1 < rounds) request(bvB, col1);
                if (round + 2 < rounds) entryOf(round + 2, idx2, col2, row2);
                finish(bvA, round, idx0, row0);
                if (round + 1 >= rounds) break;
                if (round + 2 < rounds) request(bvA, col2);
                if (round + 3 < rounds) entryOf(round + 3, idx3, col3, row3);
                finish(bvB, round + 1, idx1, row1);
                idx0 = idx2; col0 = col2; row0 = row2;
                idx1 = idx3; col1 = col3; row1 = row3;
            }
        } else {
            for (uint32_t round = 0; round < rounds; ++round) {
                uint32_t idx2 = 0, col2 = 0, row2 = 0;
                if (round + 2 < rounds) entryOf(round + 2, idx2, col2, row2);
                finish(bvA, round, idx0, row0);
                if (round + 1 < rounds) request(bvA, col1);
                idx0 = idx1; col0 = col1; row0 = row1;
                idx1 = idx2; col1 = col2; row1 = row2;
            }
        }
        return;
    }

    if constexpr (A_IN_LDS) {
        for (uint32_t i = threadIdx.x; i < 16u * chunks; i += kThreads) {
            const uint32_t row = i / chunks, q = i - row * chunks;
            const f32x4 v = *reinterpret_cast<const f32x4*>(
                A + (size_t)panelRows[item.panel * 16u + row] * K + q * 4u);
            *reinterpret_cast<f32x4*>(panelA + row * ldsStride + q * 4u) = v;
        }
        __syncthreads();
    }
    for (uint32_t round = 0; round < rounds; ++round) {
        const uint32_t e = round * groups + group;
        const bool live = e < item.count;
        const uint32_t idx = item.start + (live ? e : 0u);
        const uint32_t col = entryCol[idx];
        const uint32_t row = FREE ? panelRows[idx] : (uint32_t)entryRow[idx];
        const float* bCol = B + (size_t)col * K;
        const float* aRow = FREE       ? A + (size_t)row * K
                            : A_IN_LDS ? panelA + row * ldsStride
                                       : A + (size_t)panelRows[item.panel * 16u + row] * K;
        float acc = 0.f;
        if constexpr (CPL > 0) {
            f32x4 bv[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) bv[c] = *reinterpret_cast<const f32x4*>(bCol + (t + c * LPE) * 4u);
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(aRow + (t + c * LPE) * 4u);
                acc = __builtin_fmaf(av[0], bv[c][0], acc);
                acc = __builtin_fmaf(av[1], bv[c][1], acc);
                acc = __builtin_fmaf(av[2], bv[c][2], acc);
                acc = __builtin_fmaf(av[3], bv[c][3], acc);
            }
        } else {
            for (uint32_t q = t; q < chunks; q += LPE) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bCol + q * 4u);
                const f32x4 av = *reinterpret_cast<const f32x4*>(aRow + q * 4u);
                acc = __builtin_fmaf(av[0], bv[0], acc);
                acc = __builtin_fmaf(av[1], bv[1], acc);
                acc = __builtin_fmaf(av[2], bv[2], acc);
                acc = __builtin_fmaf(av[3], bv[3], acc);
            }
        }
#pragma unroll
        for (int off = LPE / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, LPE);
        if (live && t == 0) P[entryDst[idx]] = acc;
    }
}

// ---------------------------------------------------------------------------
// Sparse residue from the fp16 / bf16 operand copies.  Used when the conversion pass runs
// anyway (the plan has a dense part): the residue is bound by the rate at which B columns
// can be gathered from L2 (~17-19 TB/s chip-wide on MI355X, MI355X_MICROARCH.md "Indexed
// rows"), so halving the bytes per column is what makes it faster.  Same structure as
// sparseEntries; 16-byte chunks now hold 8 elements and go through v_dot2c_f32_{f16,bf16}
// (products exact, fp32 accumulation).  Accuracy class = the dense path's.
// ---------------------------------------------------------------------------
constexpr uint32_t kSparseLdsPad16 = 16;  // bytes

template <int MODE>
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
    if constexpr (MODE == 0) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), acc, false);
    } else {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, a), __builtin_bit_cast(b2, b), acc, false);
    }
}

// A_FP32 (panel form with LDS staging only): A16 points at the caller's fp32 A and the panel's rows are rounded
// while they are staged - plans without a dense part then convert B alone (half the conversion pass).
template <int LPE, int MODE, bool A_IN_LDS, int CPL = 0, bool FREE = false, bool A_FP32 = false>
__global__ void __launch_bounds__(kThreads)
sparseEntriesLowp(const uint16_t* __restrict__ A16, const uint16_t* __restrict__ B16, uint32_t K,
                  const uint32_t* __restrict__ panelRows, const uint32_t* __restrict__ entryCol,
                  const uint32_t* __restrict__ entryDst, const uint8_t* __restrict__ entryRow,
                  const SparseItem* __restrict__ items, float* __restrict__ P, Batch batch) {
    A16 += blockIdx.y * batch.strideA * (A_FP32 ? 2u : 1u);  // batched call: problem blockIdx.y of a strided batch
    B16 += blockIdx.y * batch.strideB;
    P += blockIdx.y * batch.strideP;
    extern __shared__ __attribute__((aligned(16))) uint8_t panelA16[];
    const SparseItem item = items[xcdContiguous(blockIdx.x, gridDim.x)];
    const uint32_t chunks = K >> 3;  // 16-byte chunks (8 elements) per row
    const uint32_t ldsStride = 2u * K + kSparseLdsPad16;
    constexpr uint32_t groups = kThreads / LPE;
    const uint32_t group = threadIdx.x / LPE;
    const uint32_t t = threadIdx.x % LPE;
    const uint32_t rounds = (item.count + groups - 1) / groups;
    constexpr bool PIPELINED = CPL > 0 && A_IN_LDS && !FREE;
    constexpr bool TWO_BUFFERS = CPL <= 4;
    auto entryOf = [&](uint32_t round, uint32_t& idx, uint32_t& col, uint32_t& row) {
        const uint32_t e = round * groups + group;
        idx = item.start + (e < item.count ? e : 0u);
        col = entryCol[idx];
        row = entryRow[idx];
    };
    auto request = [&](u32x4 (&bv)[CPL > 0 ? CPL : 1], uint32_t col) {
        const uint16_t* bCol = B16 + (size_t)col * K;
#pragma unroll
        for (int c = 0; c < CPL; ++c) bv[c] = *reinterpret_cast<const u32x4*>(bCol + (t + c * LPE) * 8u);
    };
    uint32_t idx0 = 0, col0 = 0, row0 = 0, idx1 = 0, col1 = 0, row1 = 0;
    u32x4 bvA[CPL > 0 ? CPL : 1], bvB[CPL > 0 && TWO_BUFFERS ? CPL : 1];
    if constexpr (PIPELINED) {  // round 0's entry and B chunks are on their way while the panel is staged
        entryOf(0, idx0, col0, row0);
        entryOf(1, idx1, col1, row1);
        request(bvA, col0);
    }

    if constexpr (A_IN_LDS) {
        static_assert(!A_FP32 || !FREE, "fp32 A is rounded while a panel is staged");
        for (uint32_t i = threadIdx.x; i < 16u * chunks; i += kThreads) {
            const uint32_t row = i / chunks, q = i - row * chunks;
            const size_t at = (size_t)panelRows[item.panel * 16u + row] * K + q * 8u;
            if constexpr (A_FP32) {
                const float* src = reinterpret_cast<const float*>(A16) + at;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(src + 4);
                if constexpr (MODE == 0) {
                    f16x8 o;
                    o[0] = (_Float16)lo[0]; o[1] = (_Float16)lo[1]; o[2] = (_Float16)lo[2]; o[3] = (_Float16)lo[3];
                    o[4] = (_Float16)hi[0]; o[5] = (_Float16)hi[1]; o[6] = (_Float16)hi[2]; o[7] = (_Float16)hi[3];
                    *reinterpret_cast<f16x8*>(panelA16 + row * ldsStride + q * 16u) = o;
                } else {
                    bf16x8 o;
                    o[0] = (__bf16)lo[0]; o[1] = (__bf16)lo[1]; o[2] = (__bf16)lo[2]; o[3] = (__bf16)lo[3];
                    o[4] = (__bf16)hi[0]; o[5] = (__bf16)hi[1]; o[6] = (__bf16)hi[2]; o[7] = (__bf16)hi[3];
                    *reinterpret_cast<bf16x8*>(panelA16 + row * ldsStride + q * 16u) = o;
                }
            } else {
                const u32x4 v = *reinterpret_cast<const u32x4*>(A16 + at);
                *reinterpret_cast<u32x4*>(panelA16 + row * ldsStride + q * 16u) = v;
            }
        }
        __syncthreads();
    }

    if constexpr (PIPELINED) {  // see sparseEntries: entries two rounds ahead, B chunks one round ahead
        auto finish = [&](const u32x4 (&bv)[CPL > 0 ? CPL : 1], uint32_t round, uint32_t idx, uint32_t row) {
            const uint8_t* aRow = panelA16 + row * ldsStride;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const u32x4 av = *reinterpret_cast<const u32x4*>(aRow + (t + c * LPE) * 16u);
                acc = dot2<MODE>(av[0], bv[c][0], acc);
                acc = dot2<MODE>(av[1], bv[c][1], acc);
                acc = dot2<MODE>(av[2], bv[c][2], acc);
                acc = dot2<MODE>(av[3], bv[c][3], acc);
            }
#pragma unroll
            for (int off = LPE / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, LPE);
            if (round * groups + group < item.count && t == 0) P[entryDst[idx]] = acc;
        };
        if constexpr (TWO_BUFFERS) {
            for (uint32_t round = 0; round < rounds; round += 2) {
                uint32_t idx2 = 0, col2 = 0, row2 = 0, idx3 = 0, col3 = 0, row3 = 0;
                if (round + 1 < rounds) request(bvB, col1);
                if (round + 2 < rounds) entryOf(round + 2, idx2, col2, row2);
                finish(bvA, round, idx0, row0);
                if (round + 1 >= rounds) break;
                if (round + 2 < rounds) request(bvA, col2);
                if (round + 3 < rounds) entryOf(round + 3, idx3, col3, row3);
                finish(bvB, round + 1, idx1, row1);
                idx0 = idx2; col0 = col2; row0 = row2;
                idx1 = idx3; col1 = col3; row1 = row3;
            }
        } else {
            for (uint32_t round = 0; round < rounds; ++round) {
                uint32_t idx2 = 0, col2 = 0, row2 = 0;
                if (round + 2 < rounds) entryOf(round + 2, idx2, col2, row2);
                finish(bvA, round, idx0, row0);
                if (round + 1 < rounds) request(bvA, col1);
                idx0 = idx1; col0 = col1; row0 = row1;
                idx1 = idx2; col1 = col2; row1 = row2;
            }
        }
        return;
    }

    for (uint32_t round = 0; round < rounds; ++round) {
        const uint32_t e = round * groups + group;
        const bool live = e < item.count;
        const uint32_t idx = item.start + (live ? e : 0u);
        const uint32_t col = entryCol[idx];
        const uint32_t row = FREE ? panelRows[idx] : (uint32_t)entryRow[idx];
        const uint16_t* bCol = B16 + (size_t)col * K;
        const uint8_t* aRow =
            FREE       ? reinterpret_cast<const uint8_t*>(A16 + (size_t)row * K)
            : A_IN_LDS ? panelA16 + row * ldsStride
                       : reinterpret_cast<const uint8_t*>(A16 + (size_t)panelRows[item.panel * 16u + row] * K);
        float acc = 0.f;
        if constexpr (CPL > 0) {  // K = 8 * LPE * CPL
            u32x4 bv[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) bv[c] = *reinterpret_cast<const u32x4*>(bCol + (t + c * LPE) * 8u);
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const u32x4 av = *reinterpret_cast<const u32x4*>(aRow + (t + c * LPE) * 16u);
                acc = dot2<MODE>(av[0], bv[c][0], acc);
                acc = dot2<MODE>(av[1], bv[c][1], acc);
                acc = dot2<MODE>(av[2], bv[c][2], acc);
                acc = dot2<MODE>(av[3], bv[c][3], acc);
            }
        } else {
            for (uint32_t q = t; q < chunks; q += LPE) {
                const u32x4 bv = *reinterpret_cast<const u32x4*>(bCol + q * 8u);
                const u32x4 av = *reinterpret_cast<const u32x4*>(aRow + q * 16u);
                acc = dot2<MODE>(av[0], bv[0], acc);
                acc = dot2<MODE>(av[1], bv[1], acc);
                acc = dot2<MODE>(av[2], bv[2], acc);
                acc = dot2<MODE>(av[3], bv[3], acc);
            }
        }
#pragma unroll
        for (int off = LPE / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, LPE);
        if (live && t == 0) P[entryDst[idx]] = acc;
    }
}

// ---------------------------------------------------------------------------
// batchedMatrixTranspose (reference src/sddmmKernel.cu:2486-2515): out[b][x][y] = in[b][y][x] for
// `height` x `width` row-major matrices; 32 x 32 tiles through LDS (padded), 256 threads.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads)
batchedTranspose(const float* __restrict__ in, float* __restrict__ out, uint32_t width, uint32_t height) {
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * width * height;
    const uint32_t x0 = blockIdx.x * 32u, y0 = blockIdx.y * 32u;
    const uint32_t tx = threadIdx.x & 31u, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (uint32_t j = 0; j < 32; j += 8) {
        const uint32_t x = x0 + tx, y = y0 + ty + j;
        if (x < width && y < height) tile[ty + j][tx] = in[base + (size_t)y * width + x];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < 32; j += 8) {
        const uint32_t x = y0 + tx, y = x0 + ty + j;  // coordinates in the transposed matrix (height wide)
        if (x < height && y < width) out[base + (size_t)y * height + x] = tile[tx][ty + j];
    }
}

}  // namespace bsmr
