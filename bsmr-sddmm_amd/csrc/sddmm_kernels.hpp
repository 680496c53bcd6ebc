// Hand-written gfx950 (CDNA4, wave64) kernels of the BSMR-SDDMM device path.
//
// What they compute is what the reference's live kernels compute
// (src/sddmmKernel.cu:213-351 dense blocks, :1994-2104 sparse residue): for every
// stored entry e = (i, j) of S,  P[e] = sum_k A[i,k] * B[k,j].  How they do it is
// designed for MI355X and shares nothing with the CUDA code:
//
//  * B is column-major, so a "column" is K contiguous elements.  The MFMA
//    16x16x32 operand maps put element j of lane l at A[row l&15][k = 8(l>>4)+j]
//    and B[k = 8(l>>4)+j][col l&15]: for both operands a lane's 8 elements are
//    16 contiguous bytes of one (gathered) row / column.  Fragments are therefore
//    loaded straight from global memory with one global_load_dwordx4 per lane and
//    K step - no LDS staging, no transposes, no bank conflicts.
//  * One wave owns one row panel x a run of dense blocks; the panel's A
//    fragments stay in registers across the run.
//  * The sparsity mask and the destinations are one tile of 256 row-relative
//    16-bit offsets per block, stored in accumulator (lane-major) order: 8 bytes
//    per lane, loaded together with the B fragments (the reference stores a 1 KiB
//    row-major tile of absolute 32-bit indices per block).
//  * The residual sparse path splits K over LPE lanes per entry (coalesced 16-byte
//    loads of the B column), takes the panel's A rows from LDS, and reduces with
//    a butterfly; it is exact fp32 with a defined summation order (bit-level CPU
//    twin: oracle/sddmm_oracle.c oracle_sparse_twin).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bsmr {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kThreads = 256;           // 4 waves per workgroup
constexpr int kWavesPerWG = kThreads / kWave;

// One unit of dense work: blocks [first, first+count) (global block ids) of `panel`.
struct DenseItem {
    uint32_t panel;
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
// items - consecutive blocks of one panel, consecutive panels of one cluster,
// which share B columns - hit the same L2.  Speed only; any placement is correct.
__device__ __forceinline__ uint32_t xcdContiguous(uint32_t wg, uint32_t numWG) {
    return (numWG & 7u) == 0 ? (wg & 7u) * (numWG >> 3) + (wg >> 3) : wg;
}

// ---------------------------------------------------------------------------
// fp32 -> fp16 / bf16 operand pass (8 elements per thread, 16-byte stores)
// ---------------------------------------------------------------------------
template <int MODE>  // 0 = fp16, 1 = bf16
__global__ void __launch_bounds__(kThreads)
convertOperands(const float* __restrict__ A, uint64_t nA8, const float* __restrict__ B, uint64_t nB8,
                uint16_t* __restrict__ A16, uint16_t* __restrict__ B16) {
    const uint64_t total = nA8 + nB8;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * kThreads) {
        const float* src = i < nA8 ? A + i * 8 : B + (i - nA8) * 8;
        uint16_t* dst = i < nA8 ? A16 + i * 8 : B16 + (i - nA8) * 8;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(src + 4);
        if constexpr (MODE == 0) {
            f16x8 o;
            o[0] = (_Float16)lo[0]; o[1] = (_Float16)lo[1]; o[2] = (_Float16)lo[2]; o[3] = (_Float16)lo[3];
            o[4] = (_Float16)hi[0]; o[5] = (_Float16)hi[1]; o[6] = (_Float16)hi[2]; o[7] = (_Float16)hi[3];
            *reinterpret_cast<f16x8*>(dst) = o;
        } else {
            bf16x8 o;
            o[0] = (__bf16)lo[0]; o[1] = (__bf16)lo[1]; o[2] = (__bf16)lo[2]; o[3] = (__bf16)lo[3];
            o[4] = (__bf16)hi[0]; o[5] = (__bf16)hi[1]; o[6] = (__bf16)hi[2]; o[7] = (__bf16)hi[3];
            *reinterpret_cast<bf16x8*>(dst) = o;
        }
    }
}

// ---------------------------------------------------------------------------
// masked write-back of one 16x16 accumulator tile
//   lane l, register i  <->  tile row 4*(l>>4)+i, tile column l&15
//   The plan stores, per block, 256 destinations in exactly that (lane-major)
//   order: element [4*l + i] = CSR index of the entry minus the CSR offset of its
//   row (`rowBase`), or all-ones where S has no entry.  16-bit offsets serve every
//   matrix whose rows hold < 65535 entries (8 bytes per lane and block); the
//   32-bit form is the fallback.  The load does not depend on anything computed
//   by the wave, so it is issued together with the B fragments.
// ---------------------------------------------------------------------------
template <typename TileT> struct TileLoad;
template <> struct TileLoad<uint16_t> {
    typedef uint32_t raw __attribute__((ext_vector_type(2)));
    static constexpr uint32_t kNull = 0xFFFFu;
    static __device__ __forceinline__ uint32_t get(const raw& v, int i) {
        const uint32_t w = i < 2 ? v[0] : v[1];
        return (i & 1) ? (w >> 16) : (w & 0xFFFFu);
    }
};
template <> struct TileLoad<uint32_t> {
    typedef u32x4 raw;
    static constexpr uint32_t kNull = 0xFFFFFFFFu;
    static __device__ __forceinline__ uint32_t get(const raw& v, int i) { return v[i]; }
};

template <typename TileT>
__device__ __forceinline__ typename TileLoad<TileT>::raw loadTile(const TileT* __restrict__ tiles,
                                                                  uint32_t block, uint32_t lane) {
    return *reinterpret_cast<const typename TileLoad<TileT>::raw*>(tiles + (size_t)block * 256u + lane * 4u);
}

template <typename TileT>
__device__ __forceinline__ void scatterTile(const f32x4& acc, const typename TileLoad<TileT>::raw& tile,
                                            const uint32_t (&rowBase)[4], float* __restrict__ P) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t off = TileLoad<TileT>::get(tile, i);
        if (off != TileLoad<TileT>::kNull) P[rowBase[i] + off] = acc[i];
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
// dense-block kernel, 16-bit operands.
//   KS = K/32 when known at compile time: the panel's A fragments live in
//   registers for the whole item, and the item's blocks are processed NB at a
//   time with every load of the batch (B fragments + destination tiles) issued
//   before the first MFMA - one memory round trip per batch instead of one per
//   block (the kernel is latency-bound, not bandwidth-bound, at L2-resident sizes).
//   KS = 0: run-time K loop (any multiple of 32).
// ---------------------------------------------------------------------------
template <int KS, int NB, int MODE, typename TileT>
__global__ void __launch_bounds__(kThreads)
denseBlocks16(const uint16_t* __restrict__ A16, const uint16_t* __restrict__ B16, uint32_t K,
              const uint32_t* __restrict__ panelRows, const uint32_t* __restrict__ panelRowBase,
              const uint32_t* __restrict__ blockCols, const TileT* __restrict__ tiles,
              const DenseItem* __restrict__ items, uint32_t numItems, float* __restrict__ P) {
    const uint32_t wg = xcdContiguous(blockIdx.x, gridDim.x);
    const uint32_t itemId = wg * kWavesPerWG + (threadIdx.x >> 6);
    if (itemId >= numItems) return;
    const DenseItem item = items[itemId];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t r = lane & 15u;   // tile row for A, tile column for B and C
    const uint32_t g = lane >> 4;    // k group inside a 32-deep step / accumulator row group

    const uint16_t* aRow = A16 + (size_t)panelRows[item.panel * 16u + r] * K + g * 8u;
    uint32_t rowBase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rowBase[i] = panelRowBase[item.panel * 16u + 4u * g + i];
    const uint32_t end = item.first + item.count;

    if constexpr (KS > 0) {
        u32x4 a[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = *reinterpret_cast<const u32x4*>(aRow + s * 32);

        for (uint32_t b0 = item.first; b0 < end; b0 += NB) {
            u32x4 bf[NB][KS];
            typename TileLoad<TileT>::raw tile[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const uint32_t b = min(b0 + n, end - 1u);  // tail of the batch re-reads the last block
                const uint16_t* bCol = B16 + (size_t)blockCols[b * 16u + r] * K + g * 8u;
#pragma unroll
                for (int s = 0; s < KS; ++s) bf[n][s] = *reinterpret_cast<const u32x4*>(bCol + s * 32);
                tile[n] = loadTile<TileT>(tiles, b, lane);
            }
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = mfma16<MODE>(a[s], bf[n][s], acc);
                if (b0 + n < end) scatterTile<TileT>(acc, tile[n], rowBase, P);
            }
        }
    } else {
        const uint32_t steps = K >> 5;
        for (uint32_t b = item.first; b < end; ++b) {
            const uint16_t* bCol = B16 + (size_t)blockCols[b * 16u + r] * K + g * 8u;
            const typename TileLoad<TileT>::raw tile = loadTile<TileT>(tiles, b, lane);
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
// dense-block kernel, exact fp32: v_mfma_f32_16x16x4_f32 is a k-ordered fmaf
// chain.  Lane (r, g) loads float4 chunks [16t + 4g, +4) of its row / column;
// MFMA number (t, j) multiplies element j of every chunk, so inside it lane
// group g supplies k = 16t + 4g + j.  Chain order of k: for t, for j, for g.
// (CPU twin: oracle_dense_f32_twin.)
// ---------------------------------------------------------------------------
template <typename TileT>
__global__ void __launch_bounds__(kThreads)
denseBlocks32(const float* __restrict__ A, const float* __restrict__ B, uint32_t K,
              const uint32_t* __restrict__ panelRows, const uint32_t* __restrict__ panelRowBase,
              const uint32_t* __restrict__ blockCols, const TileT* __restrict__ tiles,
              const DenseItem* __restrict__ items, uint32_t numItems, float* __restrict__ P) {
    const uint32_t wg = xcdContiguous(blockIdx.x, gridDim.x);
    const uint32_t itemId = wg * kWavesPerWG + (threadIdx.x >> 6);
    if (itemId >= numItems) return;
    const DenseItem item = items[itemId];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t r = lane & 15u;
    const uint32_t g = lane >> 4;
    const float* aRow = A + (size_t)panelRows[item.panel * 16u + r] * K + g * 4u;
    uint32_t rowBase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rowBase[i] = panelRowBase[item.panel * 16u + 4u * g + i];
    const uint32_t steps = K >> 4;
    for (uint32_t b = item.first; b < item.first + item.count; ++b) {
        const float* bCol = B + (size_t)blockCols[b * 16u + r] * K + g * 4u;
        const typename TileLoad<TileT>::raw tile = loadTile<TileT>(tiles, b, lane);
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

// ---------------------------------------------------------------------------
// residual sparse kernel (fp32).  One workgroup = one SparseItem.  The panel's
// 16 A rows are staged in LDS (row stride K+4 floats).  LPE lanes share one
// entry: lane t accumulates the float4 chunks q = t, t+LPE, ... of the dot
// product with one fmaf chain (ascending k), then a butterfly over the LPE lanes.
// ---------------------------------------------------------------------------
constexpr int kSparseLdsPad = 4;  // floats

template <int LPE, bool A_IN_LDS>
__global__ void __launch_bounds__(kThreads)
sparseEntries(const float* __restrict__ A, const float* __restrict__ B, uint32_t K,
              const uint32_t* __restrict__ panelRows, const uint32_t* __restrict__ entryCol,
              const uint32_t* __restrict__ entryDst, const uint8_t* __restrict__ entryRow,
              const SparseItem* __restrict__ items, float* __restrict__ P) {
    extern __shared__ __attribute__((aligned(16))) float panelA[];
    const SparseItem item = items[xcdContiguous(blockIdx.x, gridDim.x)];
    const uint32_t chunks = K >> 2;  // float4 chunks per row
    const uint32_t ldsStride = K + kSparseLdsPad;

    if constexpr (A_IN_LDS) {
        for (uint32_t i = threadIdx.x; i < 16u * chunks; i += kThreads) {
            const uint32_t row = i / chunks, q = i - row * chunks;
            const f32x4 v = *reinterpret_cast<const f32x4*>(
                A + (size_t)panelRows[item.panel * 16u + row] * K + q * 4u);
            *reinterpret_cast<f32x4*>(panelA + row * ldsStride + q * 4u) = v;
        }
        __syncthreads();
    }

    constexpr uint32_t groups = kThreads / LPE;
    const uint32_t group = threadIdx.x / LPE;
    const uint32_t t = threadIdx.x % LPE;
    // every group runs the same number of rounds so that all lanes of a wave
    // reach the shuffles together
    const uint32_t rounds = (item.count + groups - 1) / groups;
    for (uint32_t round = 0; round < rounds; ++round) {
        const uint32_t e = round * groups + group;
        const bool live = e < item.count;
        const uint32_t idx = item.start + (live ? e : 0u);
        const uint32_t col = entryCol[idx];
        const uint32_t row = entryRow[idx];
        const float* bCol = B + (size_t)col * K;
        const float* aRow = A_IN_LDS ? panelA + row * ldsStride
                                     : A + (size_t)panelRows[item.panel * 16u + row] * K;
        float acc = 0.f;
        for (uint32_t q = t; q < chunks; q += LPE) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bCol + q * 4u);
            const f32x4 av = *reinterpret_cast<const f32x4*>(aRow + q * 4u);
            acc = __builtin_fmaf(av[0], bv[0], acc);
            acc = __builtin_fmaf(av[1], bv[1], acc);
            acc = __builtin_fmaf(av[2], bv[2], acc);
            acc = __builtin_fmaf(av[3], bv[3], acc);
        }
#pragma unroll
        for (int off = LPE / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, LPE);
        if (live && t == 0) P[entryDst[idx]] = acc;
    }
}

}  // namespace bsmr
