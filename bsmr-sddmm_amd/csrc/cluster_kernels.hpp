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
// through device-side launches and per-row mutexes; gfx950 has no device-side launch.  Here:
//   * Several clusters are in flight, oldest first.  Cluster j+1 is seeded by the first row
//     cluster j passed over and only ever judges positions cluster j has already decided
//     (cursor[j+1] <= cursor[j]) - the hand-over-hand order of the reference's mutexes, so every
//     row meets the clusters in the same order and the same state as in a sequential run.
//   * The scan of each cluster is speculative: one pass judges the next `chunk` unassigned
//     positions of every active cluster against that cluster's representative in parallel (one
//     workgroup per pair), the smallest accepted position of a cluster wins (atomicMin), and the
//     last workgroup to finish merges those rows, moves each cursor behind its hit - everything
//     before it was judged with the right representative, everything after it is judged again -
//     adapts the chunks, retires finished clusters and seeds the next one.
//   * A similarity is only compared with alpha.  One thread forms the same quotient from the row's sparse
//     (bin, count) list in double arithmetic - O(|row|) against O(bins) - which differs from the fp32
//     block sum by a few 1e-6 at most; only pairs within 1e-4 of alpha are re-evaluated by the whole
//     workgroup in the reference's exact order of operations (same rule as src/rowReordering.cpp).
// All state lives in device memory, so the host only enqueues passes and polls a flag; a pass
// that finds the work finished is a no-op.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bsmr {

constexpr uint32_t kNoCluster = 0xFFFFFFFFu;
// One histogram entry of the rows x bins table: a row has at most binWidth <= 65535 entries in a bin, so 16
// bits do (half the bytes of the reference's UIN table; the representatives, which add rows up, stay 32-bit).
typedef uint16_t ClusterCount;
constexpr uint32_t kClusterMinChunk = 32;

constexpr uint32_t kClusterMaxActive = 32;

struct ClusterSlot {
    uint32_t seed;      // position of the cluster's first row
    uint32_t cursor;    // next position to judge
    uint32_t chunk;     // positions judged by the next pass
    uint32_t id;        // cluster id (1-based; 0 = empty rows)
    uint32_t sq;        // representative's sum of squares over the bins that count (mod 2^32)
    uint32_t total;     // representative's sum of counts over the bins that count
    uint32_t firstHit;  // smallest accepted position of the running pass
    uint32_t scan;      // positions in [seed + 1, scan) are assigned (search for the successor's seed)
    uint32_t rep;       // which representative buffer the cluster owns
};

struct ClusterState {
    uint32_t numActive;  // slots [0, numActive), oldest first
    uint32_t nextId;     // id of the youngest cluster
    uint32_t floor;      // every position below is assigned
    uint32_t freeReps;   // bit r = representative buffer r is free
    uint32_t arrived;    // workgroups of the running pass that have finished
    uint32_t done;       // every row has a cluster
    uint32_t passes;     // statistics
    uint32_t judged;     // statistics: similarities decided by the O(|row|) evaluation
    uint32_t exact;      // statistics: similarities that needed the exact block-order evaluation
    ClusterSlot slot[kClusterMaxActive];
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
                                 ClusterCount* __restrict__ table, uint32_t* __restrict__ dispersion,
                                 uint32_t* __restrict__ rowSquares) {
    extern __shared__ uint32_t hist[];
    __shared__ uint32_t partial[3];
    const uint32_t row = blockIdx.x;
    const uint32_t b = rowOffsets[row], e = rowOffsets[row + 1];
    ClusterCount* out = table + (size_t)row * numBins;
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
        out[i] = (ClusterCount)v;
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
                                                       const ClusterCount* __restrict__ cmp, uint32_t sqCmp,
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

// One speculative pass (see the header comment).  The positions to judge of all active clusters form one
// list; thread t of workgroup g takes items g*T + t, g*T + t + G*T, ... (G = gridDim.x, T = blockDim.x) and
// skips an item once an earlier position of the same cluster has been accepted.
template <bool MANY>
__global__ void clusterPass(const ClusterCount* __restrict__ table, const uint32_t* __restrict__ rowSquares,
                            const uint32_t* __restrict__ encOffsets, const uint32_t* __restrict__ encBins,
                            const uint16_t* __restrict__ encCounts, const uint32_t* __restrict__ order, uint32_t rows,
                            uint32_t numBins, float alpha, uint32_t maxChunk, uint32_t maxActive, uint32_t liveWarps,
                            uint32_t* __restrict__ reps, uint32_t* __restrict__ cluster,
                            ClusterState* __restrict__ state) {
    __shared__ float shmA[32], shmB[32];
    __shared__ uint32_t shared[3];
    __shared__ uint32_t sCursor[kClusterMaxActive], sLen[kClusterMaxActive], sStart[kClusterMaxActive + 1];
    __shared__ uint32_t sHit[kClusterMaxActive], sSumSq[kClusterMaxActive], sSumTotal[kClusterMaxActive];
    __shared__ ClusterSlot sSlot[kClusterMaxActive + 1];
    // MANY only: items of this round within 1e-4 of alpha, (cluster << 27) | offset in its window
    __shared__ uint32_t sNear[MANY ? 1024 : 1];
    __shared__ uint32_t sNearCount;
    if (state->done) return;  // uniform over the grid: the state only changes at the end of a pass
    const uint32_t active = state->numActive;
    const uint32_t T = blockDim.x;
    if (threadIdx.x < active) sSlot[threadIdx.x] = state->slot[threadIdx.x];  // one slot per lane: one round trip
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t j = 0; j < active; ++j) {
            const ClusterSlot& c = sSlot[j];
            // a younger cluster only judges what the next older one has already decided
            const uint32_t limit = j == 0 ? rows : sSlot[j - 1].cursor;
            const uint32_t len = c.cursor < limit ? (c.chunk < limit - c.cursor ? c.chunk : limit - c.cursor) : 0u;
            sCursor[j] = c.cursor;
            sLen[j] = len;
            sStart[j] = total;
            total += len;
        }
        sStart[active] = total;
        sNearCount = 0;
    }
    __syncthreads();
    const uint32_t total = sStart[active];
    // one item per wave and round; the wave's lanes share the row's (bin, count) list
    const uint32_t wavesPerWG = (T + 63u) >> 6;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t lanes = T - 64u * wave < 64u ? T - 64u * wave : 64u;  // 32 in the last wave when T % 64 == 32
    // Two forms of one pass, picked by the host from the previous batch's statistics.  MANY = false: few
    // items per pass, latency counts - every item is evaluated exactly by a whole workgroup straight from the
    // dense table (two dependent loads instead of four, a small kernel).  MANY = true: throughput counts -
    // one wave per item decides from the sparse list, the workgroup only re-evaluates the near ones.
    const uint32_t wgsWithWork = MANY ? (total + wavesPerWG - 1) / wavesPerWG : total;
    if (blockIdx.x >= wgsWithWork) return;  // nothing to judge, and nobody waits for this workgroup
    const uint32_t participants = wgsWithWork < gridDim.x ? wgsWithWork : gridDim.x;
    uint32_t judged = 0, exact = 0;
    if constexpr (!MANY) {
        for (uint32_t item = blockIdx.x; item < total; item += gridDim.x) {
            uint32_t j = 0;
            while (item >= sStart[j + 1]) ++j;
            const uint32_t pos = sCursor[j] + (item - sStart[j]);
            if (item != blockIdx.x) {  // later rounds: skip what lies behind an accepted position
                if (threadIdx.x == 0)
                    shared[2] = __hip_atomic_load(&state->slot[j].firstHit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                const uint32_t best = shared[2];
                __syncthreads();
                if (best < pos) continue;
            }
            if (cluster[pos] != kNoCluster) continue;
            const uint32_t row = order[pos];
            const float sim = similarityAsReference(reps + (size_t)sSlot[j].rep * numBins, sSlot[j].sq,
                                                    table + (size_t)row * numBins, rowSquares[row], numBins, shmA, shmB);
            ++exact;
            if (threadIdx.x == 0 && sim > alpha) atomicMin(&state->slot[j].firstHit, pos);
        }
    }
    for (uint32_t base = blockIdx.x * wavesPerWG; MANY && base < total; base += gridDim.x * wavesPerWG) {  // uniform
        // -- decided from the row's sparse histogram unless it is close to alpha --
        const uint32_t item = base + wave;
        if (item < total) {
            uint32_t j = 0;
            while (item >= sStart[j + 1]) ++j;
            const uint32_t pos = sCursor[j] + (item - sStart[j]);
            const uint32_t best = __hip_atomic_load(&state->slot[j].firstHit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pos < best && cluster[pos] == kNoCluster) {  // (wave-uniform up to the race on firstHit, which only prunes)
                const uint32_t row = order[pos];
                const uint32_t sqRep = sSlot[j].sq, sqRow = rowSquares[row];
                if (lane == 0) ++judged;
                bool hit = false, near = false;
                if (sqRep == 0 || sqRow == 0) {  // src/rowReordering.cu:263-268
                    hit = (sqRep == 0 && sqRow == 0 ? 1.0f : 0.0f) > alpha;
                } else {
                    const uint32_t* rep = reps + (size_t)sSlot[j].rep * numBins;
                    const double normRep = (double)sqrtf((float)sqRep), normRow = (double)sqrtf((float)sqRow);
                    // min + max = x + y in every bin, so  sum max = |x|_1 + |y|_1 - sum min: only the min-sum and the
                    // row's own |y|_1 are accumulated, with reciprocals instead of two divisions per bin (the value is
                    // only compared with alpha, and anything within 1e-4 of it is evaluated exactly below)
                    const double invRep = 1.0 / normRep, invRow = 1.0 / normRow;
                    double minSum = 0.0, sumY = 0.0;
                    // eight entries per lane at a time: their (bin, count) loads are issued together, then the eight
                    // gathers from the representative - a lane's chain is two round trips per batch instead of two per
                    // entry (a pass of this form was bound by exactly that chain: 67 us median on the reddit-like shard)
                    const uint32_t rowEnd = encOffsets[row + 1];
                    for (uint32_t i0 = encOffsets[row] + lane; i0 < rowEnd; i0 += 8u * lanes) {
                        uint32_t bins[8], counts[8], repCounts[8];
#pragma unroll
                        for (uint32_t u = 0; u < 8; ++u) {
                            const uint32_t i = i0 + u * lanes;
                            const bool ok = i < rowEnd;
                            bins[u] = ok ? encBins[i] : 0xFFFFFFFFu;
                            counts[u] = ok ? (uint32_t)encCounts[i] : 0u;
                        }
#pragma unroll
                        for (uint32_t u = 0; u < 8; ++u) repCounts[u] = bins[u] != 0xFFFFFFFFu ? rep[bins[u]] : 0u;
#pragma unroll
                        for (uint32_t u = 0; u < 8; ++u) {
                            if (bins[u] == 0xFFFFFFFFu || !binCounts(bins[u], T, liveWarps)) continue;
                            const double x = (double)repCounts[u] * invRep, y = (double)counts[u] * invRow;
                            minSum += x < y ? x : y;
                            sumY += y;
                        }
                    }
                    for (uint32_t w = lanes >> 1; w >= 1; w >>= 1) {
                        minSum += __shfl_xor(minSum, w, 64);
                        sumY += __shfl_xor(sumY, w, 64);
                    }
                    const double approx = minSum / ((double)sSlot[j].total * invRep + sumY - minSum);
                    const double margin = approx - (double)alpha;
                    if (margin > 1e-4) hit = true;
                    else if (margin >= -1e-4) near = true;
                }
                if (lane == 0) {
                    if (hit) atomicMin(&state->slot[j].firstHit, pos);
                    if (near) sNear[atomicAdd(&sNearCount, 1u)] = (j << 27) | (item - sStart[j]);
                }
            }
        }
        __syncthreads();
        // -- the near ones again, by the whole workgroup, in the reference's order of operations --
        const uint32_t nearCount = sNearCount;
        for (uint32_t n = 0; n < nearCount; ++n) {
            const uint32_t j = sNear[n] >> 27, pos = sCursor[j] + (sNear[n] & 0x07FFFFFFu);
            const uint32_t row = order[pos];
            const float sim = similarityAsReference(reps + (size_t)sSlot[j].rep * numBins, sSlot[j].sq,
                                                    table + (size_t)row * numBins, rowSquares[row], numBins, shmA, shmB);
            if (threadIdx.x == 0 && sim > alpha) atomicMin(&state->slot[j].firstHit, pos);
        }
        exact += nearCount;
        __syncthreads();
        if (threadIdx.x == 0) sNearCount = 0;
        __syncthreads();
    }
    // the last workgroup to arrive closes the pass
    if constexpr (MANY) {  // lane 0 of every wave counted its items
        if (threadIdx.x == 0) shared[2] = 0;
        __syncthreads();
        if (judged) atomicAdd(&shared[2], judged);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (MANY && shared[2]) atomicAdd(&state->judged, shared[2]);
        if (exact) atomicAdd(&state->exact, exact);
        __threadfence();
        shared[0] = atomicAdd(&state->arrived, 1u) == participants - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!shared[0]) return;
    __threadfence();

    uint32_t sTotal = 0;
    // rep = first (assign) or rep += first (merge); returns the new sum of squares over the bins
    // that count (UIN arithmetic: wraps).  Every thread only touches the bins it owns.
    // Also leaves the new sum of counts over those bins in shared[2].
    auto absorb = [&](uint32_t* __restrict__ rep, const ClusterCount* __restrict__ first, bool merge) {
        if (threadIdx.x == 0) shared[1] = shared[2] = 0;
        __syncthreads();
        uint32_t sq = 0, tot = 0;
        for (uint32_t i = threadIdx.x; i < numBins; i += blockDim.x) {
            const uint32_t v = merge ? rep[i] + (uint32_t)first[i] : (uint32_t)first[i];
            rep[i] = v;
            if (binCounts(i, blockDim.x, liveWarps)) {
                sq += v * v;
                tot += v;
            }
        }
        for (uint32_t w = lanes >> 1; w >= 1; w >>= 1) {
            sq += __shfl_xor(sq, w, 64);
            tot += __shfl_xor(tot, w, 64);
        }
        if ((threadIdx.x & 63u) == 0) {
            atomicAdd(&shared[1], sq);
            atomicAdd(&shared[2], tot);
        }
        __syncthreads();
        const uint32_t sum = shared[1];
        sTotal = shared[2];
        __syncthreads();
        return sum;
    };
    // smallest unassigned position in [from, to), kNoCluster if there is none
    auto firstUnassigned = [&](uint32_t from, uint32_t to) {
        if (threadIdx.x == 0) shared[1] = kNoCluster;
        __syncthreads();
        uint32_t found = kNoCluster;
        for (uint32_t base = from; base < to; base += blockDim.x) {
            const uint32_t p = base + threadIdx.x;
            if (p < to && cluster[p] == kNoCluster) atomicMin(&shared[1], p);
            __syncthreads();
            found = shared[1];
            __syncthreads();  // nobody updates the slot for the next stretch before everyone has read it
            if (found != kNoCluster) break;
        }
        return found;
    };

    // 1. hits: every accepted row is merged into its cluster's representative.  The clusters are
    //    independent here, so all merges run in one sweep (per-cluster sums: a wave adds its lanes up
    //    and issues one LDS atomic), followed by the scalar bookkeeping of thread 0.
    if (threadIdx.x < active) {
        sHit[threadIdx.x] = __hip_atomic_load(&state->slot[threadIdx.x].firstHit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sSumSq[threadIdx.x] = 0;
        sSumTotal[threadIdx.x] = 0;
    }
    __syncthreads();
    for (uint32_t j = 0; j < active; ++j) {
        const uint32_t hit = sHit[j];
        if (hit == kNoCluster) continue;  // uniform
        uint32_t* rep = reps + (size_t)sSlot[j].rep * numBins;
        const ClusterCount* add = table + (size_t)order[hit] * numBins;
        uint32_t sq = 0, tot = 0;
        for (uint32_t i = threadIdx.x; i < numBins; i += blockDim.x) {
            const uint32_t v = rep[i] + (uint32_t)add[i];
            rep[i] = v;
            if (binCounts(i, blockDim.x, liveWarps)) {
                sq += v * v;
                tot += v;
            }
        }
        for (uint32_t w = lanes >> 1; w >= 1; w >>= 1) {  // `lanes` = 32 in a block's last, half-filled wave
            sq += __shfl_xor(sq, w, 64);
            tot += __shfl_xor(tot, w, 64);
        }
        if ((threadIdx.x & 63u) == 0) {
            atomicAdd(&sSumSq[j], sq);
            atomicAdd(&sSumTotal[j], tot);
        }
    }
    __syncthreads();
    if (threadIdx.x < active) {
        const uint32_t j = threadIdx.x, hit = sHit[j];
        const uint32_t cursor = sCursor[j], len = sLen[j], chunk = sSlot[j].chunk;
        if (hit != kNoCluster) {
            cluster[hit] = sSlot[j].id;
            sSlot[j].sq = sSumSq[j];
            sSlot[j].total = sSumTotal[j];
            sSlot[j].cursor = hit + 1;
            // judging costs time even beside the hit, so the next pass looks twice as far as this hit was
            const uint32_t gap = 2u * (hit - cursor + 1u);
            sSlot[j].chunk = gap < kClusterMinChunk ? kClusterMinChunk : (gap > maxChunk ? maxChunk : gap);
        } else if (len) {
            sSlot[j].cursor = cursor + len;
            if (len == chunk) sSlot[j].chunk = 4u * chunk > maxChunk ? maxChunk : 4u * chunk;  // not held back by the older one
        }
        sSlot[j].firstHit = kNoCluster;
    }
    __syncthreads();
    // 2. retire finished clusters (only the oldest can be finished: the others trail it)
    uint32_t retired = 0, floor = state->floor, freeReps = state->freeReps, nextId = state->nextId;
    while (retired < active && sSlot[retired].cursor >= rows) {
        freeReps |= 1u << sSlot[retired].rep;
        if (sSlot[retired].seed + 1 > floor) floor = sSlot[retired].seed + 1;
        ++retired;
    }
    uint32_t left = active - retired;
    __syncthreads();
    if (retired) {
        if (threadIdx.x == 0)
            for (uint32_t j = 0; j < left; ++j) sSlot[j] = sSlot[j + retired];
        __syncthreads();
    }
    // 3. the successor of the youngest cluster starts at the first row that cluster passed over;
    //    with nothing in flight, at the first row that has no cluster at all
    uint32_t done = 0;
    if (left < maxActive) {
        uint32_t from, to;
        if (left) {
            from = sSlot[left - 1].scan;
            to = sSlot[left - 1].cursor < rows ? sSlot[left - 1].cursor : rows;
        } else {
            from = floor;
            to = rows;
        }
        const uint32_t found = firstUnassigned(from, to);
        if (found == kNoCluster) {
            if (left == 0) done = 1;
            else if (threadIdx.x == 0 && to > from) sSlot[left - 1].scan = to;  // all of that stretch is assigned
        } else {
            uint32_t r = 0;
            while (!((freeReps >> r) & 1u)) ++r;
            freeReps &= ~(1u << r);
            ++nextId;
            const uint32_t sq = absorb(reps + (size_t)r * numBins, table + (size_t)order[found] * numBins, false);
            if (threadIdx.x == 0) {
                cluster[found] = nextId;
                if (left) sSlot[left - 1].scan = found + 1;
                ClusterSlot c;
                c.seed = found;
                c.cursor = found + 1;
                c.chunk = 2u * kClusterMinChunk;
                c.id = nextId;
                c.sq = sq;
                c.total = sTotal;
                c.firstHit = kNoCluster;
                c.scan = found + 1;
                c.rep = r;
                sSlot[left] = c;
            }
            // seeded by the very last position with nothing else in flight: that cluster is complete
            if (left == 0 && found + 1 >= rows) done = 1;
            ++left;
        }
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x < left) state->slot[threadIdx.x] = sSlot[threadIdx.x];
    if (threadIdx.x == 0) {
        state->numActive = left;
        state->nextId = nextId;
        state->floor = floor;
        state->freeReps = freeReps;
        state->arrived = 0;
        state->passes += 1;
        state->done = done;
    }
}

// first representative = the seed row's histogram; its sum of squares as in clusterPass
__global__ void clusterInitRepresentative(const ClusterCount* __restrict__ seedRow, uint32_t* __restrict__ rep,
                                          uint32_t numBins, uint32_t liveWarps, ClusterState* __restrict__ state) {
    __shared__ uint32_t total[2];
    if (threadIdx.x < 2) total[threadIdx.x] = 0;
    __syncthreads();
    uint32_t sq = 0, tot = 0;
    for (uint32_t i = threadIdx.x; i < numBins; i += blockDim.x) {
        const uint32_t v = seedRow[i];
        rep[i] = v;
        if (binCounts(i, blockDim.x, liveWarps)) {
            sq += v * v;
            tot += v;
        }
    }
    atomicAdd(&total[0], sq);
    atomicAdd(&total[1], tot);
    __syncthreads();
    if (threadIdx.x == 0) {
        state->slot[0].sq = total[0];
        state->slot[0].total = total[1];
    }
}

}  // namespace bsmr
