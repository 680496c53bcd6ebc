// Row clustering on the device (BSA).
//
// What is computed is exactly what src/rowReordering.cpp computes on the host (and what
// the reference's bsa_clustering kernels compute, src/rowReordering.cu:235-432): greedy
// clusters over the rows in ascending-dispersion order, a row joins when its normalised
// weighted Jaccard similarity with the running representative exceeds alpha.  The
// similarity of one (representative, row) pair is evaluated by one workgroup of
// T = clusterThreads(bins) threads with the same per-thread bin assignment and the same
// block-wide sum as the reference (include/cudaUtil.cuh:14-45: 32-lane butterflies, then a
// fold of the per-warp values that skips some warps when T/32 is not a power of two), so the
// three implementations agree bit for bit (tests/test_gpu_cluster.py).
//
// How it is scheduled is different.  The reference chains one single-block kernel per cluster
// through device-side launches and per-row mutexes; gfx950 has no device-side launch.  Here the
// scan of a cluster is speculative: one pass judges the next `chunk` unassigned positions
// against the current representative in parallel (one workgroup each), the smallest accepted
// position wins (atomicMin), and the last workgroup to finish merges that row into the
// representative and moves the cursor behind it - everything before it was judged with the
// right representative, everything after it is judged again by the next pass.  The chunk
// adapts to the distance between hits.  All state lives in device memory, so the host only
// enqueues passes and polls a flag; a pass that finds the work finished is a no-op.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bsmr {

constexpr uint32_t kNoCluster = 0xFFFFFFFFu;
constexpr uint32_t kClusterMinChunk = 32;

struct ClusterState {
    uint32_t seed;       // position of the current cluster's first row
    uint32_t cursor;     // next position to judge
    uint32_t chunk;      // positions judged by the next pass
    uint32_t clusterId;  // id of the current cluster (1-based; 0 = empty rows)
    uint32_t firstHit;   // smallest accepted position of the running pass
    uint32_t arrived;    // workgroups of the running pass that have finished
    uint32_t done;       // every row has a cluster
    uint32_t sqRep;      // representative's sum of squares over the bins that count (mod 2^32)
    uint32_t passes;     // statistics
    uint32_t judged;     // statistics: similarities evaluated
};

// does bin b take part in the reference's block-wide sums?  liveWarps: bit w = warp w reaches the
// result of include/cudaUtil.cuh:37-43 for T threads
__device__ __forceinline__ bool binCounts(uint32_t b, uint32_t T, uint32_t liveWarps) {
    return (liveWarps >> ((b % T) >> 5)) & 1u;
}

// one workgroup per row: column-bin histogram (in LDS) -> dense table row, dispersion score
// (src/rowReordering.cu:49-93) and the row's sum of squares as the clustering block would see it
__global__ void clusterHistogram(const uint32_t* __restrict__ rowOffsets, const uint32_t* __restrict__ colIndices,
                                 uint32_t numBins, uint32_t binWidth, uint32_t T, uint32_t liveWarps,
                                 uint32_t* __restrict__ table, uint32_t* __restrict__ dispersion,
                                 uint32_t* __restrict__ rowSquares) {
    extern __shared__ uint32_t hist[];
    __shared__ uint32_t partial[3];
    const uint32_t row = blockIdx.x;
    const uint32_t b = rowOffsets[row], e = rowOffsets[row + 1];
    uint32_t* out = table + (size_t)row * numBins;
    if (b == e) {
        if (threadIdx.x == 0) {
            dispersion[row] = 0;
            rowSquares[row] = 0;
        }
        return;  // the table was zeroed
    }
    for (uint32_t i = threadIdx.x; i < numBins; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x < 3) partial[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) atomicAdd(&hist[colIndices[i] / binWidth], 1u);
    __syncthreads();
    uint32_t touched = 0, slack = 0, squares = 0;
    for (uint32_t i = threadIdx.x; i < numBins; i += blockDim.x) {
        const uint32_t v = hist[i];
        out[i] = v;
        if (v) {
            ++touched;
            slack += binWidth - v;
            if (binCounts(i, T, liveWarps)) squares += v * v;
        }
    }
    atomicAdd(&partial[0], touched);
    atomicAdd(&partial[1], slack);
    atomicAdd(&partial[2], squares);
    __syncthreads();
    if (threadIdx.x == 0) {
        dispersion[row] = partial[1] + (e - b) * partial[0];
        rowSquares[row] = partial[2];
    }
}

// include/cudaUtil.cuh:14-45 on 32-lane "warps" (two per wavefront), for the min-sum and the
// max-sum at once: the same additions in the same order as two separate sums
__device__ __forceinline__ void blockSumAsReference(float& a, float& b, float* shmA, float* shmB) {
    for (int w = 1; w < 32; w <<= 1) {
        a += __shfl_xor(a, w, 32);
        b += __shfl_xor(b, w, 32);
    }
    const uint32_t warp = threadIdx.x >> 5, lane = threadIdx.x & 31u;
    if (lane == 0) {
        shmA[warp] = a;
        shmB[warp] = b;
    }
    __syncthreads();
    for (uint32_t stride = blockDim.x / 64u; stride >= 1; stride >>= 1) {
        if (warp < stride && lane == 0) {
            shmA[warp] += shmA[warp + stride];
            shmB[warp] += shmB[warp + stride];
        }
        __syncthreads();
    }
    a = shmA[0];
    b = shmB[0];
    __syncthreads();  // the slots are reused by the next pair
}

// src/rowReordering.cu:235-293.  The two sums of squares are integers: their masked values are
// kept per row (clusterHistogram) and per representative (ClusterState::sqRep).
__device__ __forceinline__ float similarityAsReference(const uint32_t* __restrict__ rep, uint32_t sqRep,
                                                       const uint32_t* __restrict__ cmp, uint32_t sqCmp,
                                                       uint32_t numBins, float* shmA, float* shmB) {
    if (sqRep == 0 && sqCmp == 0) return 1.0f;
    if (sqRep == 0 || sqCmp == 0) return 0.0f;
    const float normRep = sqrtf((float)sqRep);
    const float normCmp = sqrtf((float)sqCmp);
    float minSum = 0.0f, maxSum = 0.0f;
    for (uint32_t i = threadIdx.x; i < numBins; i += blockDim.x) {
        const float x = (float)rep[i] / normRep;
        const float y = (float)cmp[i] / normCmp;
        minSum = minSum + fminf(x, y);
        maxSum = maxSum + fmaxf(x, y);
    }
    blockSumAsReference(minSum, maxSum, shmA, shmB);
    return minSum / maxSum;
}

// One speculative pass (see the header comment).  Workgroup g judges positions cursor + g,
// cursor + g + G, ... (G = gridDim.x) of the pass's `chunk` and stops once an earlier position
// has been accepted; block = T threads.
__global__ void clusterPass(const uint32_t* __restrict__ table, const uint32_t* __restrict__ rowSquares,
                            const uint32_t* __restrict__ order, uint32_t rows, uint32_t numBins, float alpha,
                            uint32_t maxChunk, uint32_t liveWarps, uint32_t* __restrict__ rep,
                            uint32_t* __restrict__ cluster, ClusterState* __restrict__ state) {
    __shared__ float shmA[32], shmB[32];
    __shared__ uint32_t shared[3];
    if (state->done) return;  // uniform over the grid: `done` only changes at the end of a pass
    const uint32_t cursor = state->cursor, chunk = state->chunk, id = state->clusterId, seed = state->seed;
    const uint32_t sqRep = state->sqRep;
    if (blockIdx.x >= chunk) return;  // nothing to judge, and nobody waits for this workgroup
    const uint32_t participants = chunk < gridDim.x ? chunk : gridDim.x;
    uint32_t judged = 0;
    for (uint32_t k = blockIdx.x; k < chunk; k += gridDim.x) {
        const uint32_t pos = cursor + k;
        if (pos >= rows) break;
        if (k != blockIdx.x) {  // later rounds: give up once an earlier position has been accepted
            if (threadIdx.x == 0)
                shared[2] = __hip_atomic_load(&state->firstHit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const uint32_t best = shared[2];
            __syncthreads();
            if (best < pos) break;
        }
        if (cluster[pos] != kNoCluster) continue;
        const uint32_t row = order[pos];
        const float sim =
            similarityAsReference(rep, sqRep, table + (size_t)row * numBins, rowSquares[row], numBins, shmA, shmB);
        ++judged;
        if (threadIdx.x == 0 && sim > alpha) atomicMin(&state->firstHit, pos);
    }
    // the last workgroup to arrive closes the pass
    if (threadIdx.x == 0) {
        if (judged) atomicAdd(&state->judged, judged);
        __threadfence();
        shared[0] = atomicAdd(&state->arrived, 1u) == participants - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!shared[0]) return;
    __threadfence();
    const uint32_t hit = __hip_atomic_load(&state->firstHit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t nextCursor, nextChunk;
    bool newRep = false;
    if (hit != kNoCluster) {
        const uint32_t* add = table + (size_t)order[hit] * numBins;
        for (uint32_t i = threadIdx.x; i < numBins; i += blockDim.x) rep[i] += add[i];
        if (threadIdx.x == 0) cluster[hit] = id;
        newRep = true;
        nextCursor = hit + 1;
        // judging costs time even when it runs beside the hit (measured: 21 us per pass at ~300
        // pairs against 11.5 us at ~30), so the next pass looks twice as far as this hit was
        const uint32_t gap = 2u * (hit - cursor + 1u);
        nextChunk = gap < kClusterMinChunk ? kClusterMinChunk : (gap > maxChunk ? maxChunk : gap);
    } else {
        nextCursor = cursor + chunk;
        nextChunk = 4u * chunk > maxChunk ? maxChunk : 4u * chunk;
    }
    __syncthreads();
    uint32_t nextSeed = seed, nextId = id, nextDone = 0;
    if (nextCursor >= rows) {
        // the cluster is complete: the first row it left behind seeds the next one
        if (threadIdx.x == 0) shared[1] = kNoCluster;
        __syncthreads();
        for (uint32_t base = seed + 1; base < rows; base += blockDim.x) {
            const uint32_t p = base + threadIdx.x;
            if (p < rows && cluster[p] == kNoCluster) atomicMin(&shared[1], p);
            __syncthreads();
            const uint32_t sofar = shared[1];
            __syncthreads();  // nobody updates the slot for the next stretch before everyone has read it
            if (sofar != kNoCluster) break;
        }
        const uint32_t found = shared[1];
        if (found == kNoCluster) {
            nextDone = 1;
        } else {
            nextSeed = found;
            nextId = id + 1;
            const uint32_t* first = table + (size_t)order[found] * numBins;
            for (uint32_t i = threadIdx.x; i < numBins; i += blockDim.x) rep[i] = first[i];
            if (threadIdx.x == 0) cluster[found] = nextId;
            newRep = true;
            nextCursor = found + 1;
            nextChunk = 2u * kClusterMinChunk;
            if (nextCursor >= rows) nextDone = 1;  // the last row forms its own cluster
        }
    }
    // the representative's sum of squares over the bins that count (UIN arithmetic: wraps)
    uint32_t nextSq = sqRep;
    if (newRep) {
        if (threadIdx.x == 0) shared[1] = 0;
        __syncthreads();
        uint32_t sq = 0;
        for (uint32_t i = threadIdx.x; i < numBins; i += blockDim.x)
            if (binCounts(i, blockDim.x, liveWarps)) sq += rep[i] * rep[i];
        atomicAdd(&shared[1], sq);
        __syncthreads();
        nextSq = shared[1];
    }
    if (threadIdx.x == 0) {
        state->seed = nextSeed;
        state->clusterId = nextId;
        state->cursor = nextCursor;
        state->chunk = nextChunk;
        state->sqRep = nextSq;
        state->firstHit = kNoCluster;
        state->arrived = 0;
        state->passes += 1;
        state->done = nextDone;
    }
}

// sum of squares of the first representative (same definition as in clusterPass)
__global__ void clusterInitSquares(const uint32_t* __restrict__ rep, uint32_t numBins, uint32_t liveWarps,
                                   ClusterState* __restrict__ state) {
    __shared__ uint32_t total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    uint32_t sq = 0;
    for (uint32_t i = threadIdx.x; i < numBins; i += blockDim.x)
        if (binCounts(i, blockDim.x, liveWarps)) sq += rep[i] * rep[i];
    atomicAdd(&total, sq);
    __syncthreads();
    if (threadIdx.x == 0) state->sqRep = total;
}

}  // namespace bsmr
