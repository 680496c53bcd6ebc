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
// through device-side launches and per-row mutexes; gfx950 has no device-side launch.  Here up to
// kClusterMaxActive clusters are in flight, oldest (smallest seed) first, and every one of them runs
// AHEAD of what is known:
//   * A cluster is seeded TENTATIVELY at the next position that has no cluster yet, without waiting for the
//     older clusters to pass over that position.  It is confirmed (the seed gets its id) once every older
//     cluster's cursor is behind the seed and the seed is still free; it is dropped when an older cluster
//     accepts its seed.  A tentative cluster has no members, so dropping it undoes nothing.  Many new
//     clusters start in one pass (round 2 seeded one per pass: a matrix of many small clusters needed one
//     pass per cluster at least).
//   * The scan of each cluster is speculative: one pass judges the next `chunk` unassigned positions of every
//     cluster against that cluster's representative in parallel, whether or not the older clusters have
//     decided those positions.  A rejection is final as long as the representative stands.  The smallest
//     accepted position wins (atomicMin); the last workgroup to finish closes the pass, oldest cluster first:
//     the hit is taken if the cluster is confirmed, every older cluster's cursor is behind the position and
//     no older cluster took the position in this pass - the row then meets the clusters in the same order and
//     the same state as in a sequential run.  A hit that cannot be taken yet parks the cursor in front of it.
//     Behind a hit everything is judged again with the new representative.
//   * The representative of a cluster of one row is that row's line of the histogram table; a buffer of
//     32-bit sums is taken at the first merge.
//   * A similarity is only compared with alpha.  One thread forms the same quotient from the row's sparse
//     (bin, count) list in double arithmetic - O(|row|) against O(bins) - which differs from the fp32
//     block sum by a few 1e-6 at most; only pairs within 1e-4 of alpha are re-evaluated by the whole
//     workgroup in the reference's exact order of operations (same rule as src/rowReordering.cpp).
// Cluster ids count the seeds in seed order, dropped ones included: the host renumbers them densely (the order
// is what the result depends on).  All state lives in device memory, so the host only enqueues passes and polls
// a flag; a pass that finds the work finished is a no-op.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace bsmr {

constexpr uint32_t kNoCluster = 0xFFFFFFFFu;
// One histogram entry of the rows x bins table: a row has at most binWidth <= 65535 entries in a bin, so 16
// bits do (half the bytes of the reference's UIN table; the representatives, which add rows up, stay 32-bit).
typedef uint16_t ClusterCount;
constexpr uint32_t kClusterMinChunk = 32;

constexpr uint32_t kClusterMaxActive = 32;
constexpr uint32_t kClusterSpeculateFrom = 8;
constexpr uint32_t kClusterOwnRow = 0xFFFFFFFFu;   // ClusterSlot::rep of a cluster of one row: the seed's table line

struct ClusterSlot {
    uint32_t seed;       // position of the cluster's first row
    uint32_t cursor;     // next position to judge: everything in (seed, cursor) is assigned or was rejected
    uint32_t chunk;      // positions judged by the next pass
    uint32_t id;         // cluster id (1-based, in seed order; 0 = empty rows)
    uint32_t sq;         // representative's sum of squares over the bins that count (mod 2^32)
    uint32_t total;      // representative's sum of counts over the bins that count
    uint32_t firstHit;   // smallest accepted position of the running pass
    uint32_t rep;        // which representative buffer the cluster owns, kClusterOwnRow before the first merge
    uint32_t confirmed;  // every older cluster has passed over the seed
    uint32_t seedRow;    // the seed's row id (its table line is the representative before the first merge)
    uint32_t len, start; // the next pass judges `len` unassigned positions from `cursor` on; they are items [start, start + len) of
                         // the pass (set by the workgroup that closes a pass, for the next one)
    uint32_t parked;     // the row at `cursor` was accepted, but an older cluster has not decided it yet: nothing is judged
                         // until it has (the representative does not change meanwhile, so the verdict stands)
};

// field by field, so that a slot held by a thread lives in registers (a whole-struct copy makes the compiler keep the struct in
// memory - it then put one per thread into LDS, 60 KB per workgroup, and every field access went there)
__device__ __forceinline__ ClusterSlot loadSlot(const ClusterSlot& s) {
    ClusterSlot c;
    c.seed = s.seed; c.cursor = s.cursor; c.chunk = s.chunk; c.id = s.id; c.sq = s.sq; c.total = s.total; c.firstHit = s.firstHit;
    c.rep = s.rep; c.confirmed = s.confirmed; c.seedRow = s.seedRow; c.len = s.len; c.start = s.start; c.parked = s.parked;
    return c;
}
__device__ __forceinline__ void storeSlot(ClusterSlot& d, const ClusterSlot& c) {
    d.seed = c.seed; d.cursor = c.cursor; d.chunk = c.chunk; d.id = c.id; d.sq = c.sq; d.total = c.total; d.firstHit = c.firstHit;
    d.rep = c.rep; d.confirmed = c.confirmed; d.seedRow = c.seedRow; d.len = c.len; d.start = c.start; d.parked = c.parked;
}

struct ClusterState {
    uint32_t numActive;  // slots [0, numActive), oldest first
    uint32_t nextId;     // id of the youngest seed
    uint32_t scanPos;    // every position below is assigned or is the seed of a slot
    uint32_t freeReps;   // bit r = representative buffer r is free
    uint32_t arrived;    // workgroups of the running pass that have finished
    uint32_t done;       // every row has a cluster
    uint32_t passes;     // statistics
    uint32_t judged;     // statistics: similarities decided by the O(|row|) evaluation
    uint32_t exact;      // statistics: similarities that needed the exact block-order evaluation
    uint32_t dropped;    // statistics: tentative clusters whose seed an older cluster accepted
    uint32_t tentative;  // how many clusters may run unconfirmed at a time: one more after a pass without a drop, half (down to
                         // none: only seeds that are certain) after a pass in which one was dropped
    uint32_t ahead;      // statistics: passes in which the clusters ran ahead of the older ones' decisions
    uint32_t startChunk; // first chunk of a new cluster: running mean of the chunks finished clusters ended with
    uint32_t totalItems; // what the next pass judges, all clusters together (a workgroup without an item leaves before it loads the slots:
                         // 512 workgroups reading the same 2 KB were a hot spot in one L2 channel)
    uint32_t assigned;   // rows that have their cluster for good (confirmed seeds + accepted rows): what the host measures the two
                         // scheduling rules by (bsmr_cluster_rows: rows per millisecond of a batch of passes)
    ClusterSlot slot[kClusterMaxActive];
#ifdef BSMR_LAB_STAMPS
    unsigned long long stamps[16];   // lab build: cycles of the closing workgroup per phase, summed over the passes
#endif
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
                                 uint32_t* __restrict__ rowSquares, uint32_t* __restrict__ rowTotals) {
    extern __shared__ uint32_t hist[];
    __shared__ uint32_t partial[4];
    const uint32_t row = blockIdx.x;
    const uint32_t b = rowOffsets[row], e = rowOffsets[row + 1];
    ClusterCount* out = table + (size_t)row * numBins;
    if (b == e) {
        if (threadIdx.x == 0) {
            dispersion[row] = 0;
            rowSquares[row] = 0;
            rowTotals[row] = 0;
        }
        return;  // the table was zeroed
    }
    for (uint32_t i = threadIdx.x; i < numBins; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x < 4) partial[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) atomicAdd(&hist[colIndices[i] / binWidth], 1u);
    __syncthreads();
    uint32_t touched = 0, slack = 0, squares = 0, counted = 0;
    for (uint32_t i = threadIdx.x; i < numBins; i += blockDim.x) {
        const uint32_t v = hist[i];
        out[i] = (ClusterCount)v;
        if (v) {
            ++touched;
            slack += binWidth - v;
            if (binCounts(i, T, liveWarps)) {
                squares += v * v;
                counted += v;
            }
        }
    }
    atomicAdd(&partial[0], touched);
    atomicAdd(&partial[1], slack);
    atomicAdd(&partial[2], squares);
    atomicAdd(&partial[3], counted);
    __syncthreads();
    if (threadIdx.x == 0) {
        dispersion[row] = partial[1] + (e - b) * partial[0];
        rowSquares[row] = partial[2];
        rowTotals[row] = partial[3];
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
template <typename RepT>
__device__ __forceinline__ float similarityAsReference(const RepT* __restrict__ rep, uint32_t sqRep,
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
// what the one-wave-per-item form needs to know about the row at a position, in one load: where its sparse (bin, count)
// list lies, its sum of squares, its id
__global__ void clusterPositionInfo(const uint32_t* __restrict__ order, const uint32_t* __restrict__ encOffsets,
                                    const uint32_t* __restrict__ rowSquares, uint32_t rows, uint4* __restrict__ posInfo) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= rows) return;
    const uint32_t row = order[p];
    posInfo[p] = uint4{encOffsets[row], encOffsets[row + 1], rowSquares[row], row};
}

// posInfo / encWords: MANY only.  encWords[i] = bin << 16 | count of the i-th nonzero bin (bins ascending inside a row).
#ifdef BSMR_LAB_STAMPS
#define CLUSTER_STAMP(k) do { if (threadIdx.x == 0) { const unsigned long long now = __builtin_readcyclecounter(); labT[k] = now; } } while (0)
#else
#define CLUSTER_STAMP(k) do { } while (0)
#endif

template <bool MANY>
__global__ void __launch_bounds__(1024, MANY ? 8 : 4)
clusterPass(const ClusterCount* __restrict__ table, const uint32_t* __restrict__ rowSquares,
                            const uint32_t* __restrict__ rowTotals, const uint4* __restrict__ posInfo,
                            const uint32_t* __restrict__ encWords,
                            const uint32_t* __restrict__ order, uint32_t rows, uint32_t numBins, float alpha,
                            uint32_t maxChunk, uint32_t maxActive, uint32_t liveWarps, uint32_t longRow, uint32_t* __restrict__ reps,
                            uint32_t* __restrict__ cluster, ClusterState* __restrict__ states, uint32_t ordinal, uint32_t tentativeCap) {
    // Two copies of the state, used in turn: launch number `ordinal` READS states[ordinal & 1] - nothing but the atomics of
    // its own participants (arrived, judged, exact, a slot's firstHit) touches that copy while the launch runs - and its
    // closing workgroup WRITES the next pass's state into the other copy.  A workgroup that has no item in this pass and is
    // scheduled after the closing workgroup has finished therefore still sees this pass's state and leaves; with one copy it
    // could read the next pass's totalItems, take itself for a participant of a pass that has not been launched and count
    // itself into its arrivals.
    ClusterState* __restrict__ state = states + (ordinal & 1u);
    ClusterState* __restrict__ next = states + ((ordinal + 1u) & 1u);
    __shared__ float shmA[32], shmB[32];
    __shared__ uint32_t shared[4];
    __shared__ uint32_t sCursor[kClusterMaxActive], sLen[kClusterMaxActive], sStart[kClusterMaxActive + 1];
    __shared__ uint32_t sHit[kClusterMaxActive], sSumSq[kClusterMaxActive], sSumTotal[kClusterMaxActive];
    __shared__ uint32_t sAccept[kClusterMaxActive], sNew[kClusterMaxActive];
    __shared__ uint8_t sDrop[kClusterMaxActive], sFirstMerge[kClusterMaxActive];
    __shared__ unsigned long long sBallot[16];
    __shared__ ClusterSlot sSlot[kClusterMaxActive];
    // MANY only: items of this round within 1e-4 of alpha, (cluster << 27) | offset in its window
    __shared__ uint32_t sNear[MANY ? 1024 : 1];
    __shared__ uint32_t sNearCount;
#ifdef BSMR_LAB_STAMPS
    unsigned long long labT[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    CLUSTER_STAMP(0);
    if (state->done) return;  // uniform over the grid: the state only changes at the end of a pass
    const uint32_t active = state->numActive;
    const uint32_t T = blockDim.x;
    // (what the closing workgroup needs of the state is read here, together with the rest: it changes at the end of a pass only)
    const uint32_t tentativeIn = state->tentative, scanPosIn = state->scanPos, nextIdIn = state->nextId;
    const uint32_t freeRepsIn = state->freeReps, startChunkIn = state->startChunk;
    const uint32_t passesIn = state->passes, aheadIn = state->ahead, droppedIn = state->dropped, assignedIn = state->assigned;
    const bool speculate = tentativeIn >= kClusterSpeculateFrom;
    {
        const uint32_t itemsIn = state->totalItems;
        if (blockIdx.x >= (itemsIn ? itemsIn : 1u)) return;  // nothing to judge, and nobody waits for this workgroup
    }
    if (threadIdx.x < active) {  // one slot per lane: one round trip
        const ClusterSlot c = loadSlot(state->slot[threadIdx.x]);
        storeSlot(sSlot[threadIdx.x], c);
        sCursor[threadIdx.x] = c.cursor;
        sLen[threadIdx.x] = c.len;
        sStart[threadIdx.x] = c.start;
        if (threadIdx.x + 1 == active) sStart[active] = c.start + c.len;
    }
    if (threadIdx.x == 0) {
        sNearCount = 0;
        if (active == 0) sStart[0] = 0;
    }
    __syncthreads();
    CLUSTER_STAMP(1);
    const uint32_t total = sStart[active];
    // one item per wave and round; the wave's lanes share the row's (bin, count) list
    const uint32_t wavesPerWG = (T + 63u) >> 6;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t lanes = T - 64u * wave < 64u ? T - 64u * wave : 64u;  // 32 in the last wave when T % 64 == 32
    // Two forms of one pass, picked by the host from the previous batch's statistics.  MANY = false: few
    // items per pass, latency counts - every item is evaluated exactly by a whole workgroup straight from the
    // dense table (two dependent loads instead of four, a small kernel).  MANY = true: throughput counts -
    // one wave per item decides from the sparse list, the workgroup only re-evaluates the near ones.
    uint32_t wgsWithWork = total;
    if (wgsWithWork == 0) wgsWithWork = 1;  // nothing to judge (no cluster in flight): workgroup 0 still closes the pass
    if (blockIdx.x >= wgsWithWork) return;  // nothing to judge, and nobody waits for this workgroup
    const uint32_t participants = wgsWithWork < gridDim.x ? wgsWithWork : gridDim.x;
    // the representative of cluster j, as one of two types
    auto ownRow = [&](uint32_t j) { return table + (size_t)sSlot[j].seedRow * numBins; };
    auto exactSimilarity = [&](uint32_t j, uint32_t row) {
        const ClusterCount* cmp = table + (size_t)row * numBins;
        return sSlot[j].rep == kClusterOwnRow
                   ? similarityAsReference(ownRow(j), sSlot[j].sq, cmp, rowSquares[row], numBins, shmA, shmB)
                   : similarityAsReference(reps + (size_t)sSlot[j].rep * numBins, sSlot[j].sq, cmp, rowSquares[row], numBins, shmA, shmB);
    };
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
            const float sim = exactSimilarity(j, order[pos]);
            ++exact;
            if (threadIdx.x == 0 && sim > alpha) atomicMin(&state->slot[j].firstHit, pos);
        }
    }
    // neighbouring items go to different workgroups: the rows with long lists sit next to each other in the dispersion order
    // (a workgroup per cluster window with the representative in LDS was tried: 780 against 590 ms on the reddit-like
    // shard - the pass waits for its slowest workgroup, and the gathers are not what it is bound by)
    // What is known about an item before its row's list can be read - which cluster, the smallest accepted position so far,
    // whether the position is taken, where the list lies - is fetched one item ahead: three loads that depend on nothing
    // but the item's number, issued together, in flight while the item before is judged.
    auto locate = [&](uint32_t item, uint32_t& j, uint32_t& pos) {
        uint32_t lo = 0, hi = active;   // the cluster whose window holds the item: sStart[lo] <= item < sStart[lo + 1]
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (item >= sStart[mid]) lo = mid;
            else hi = mid;
        }
        j = lo;
        pos = sCursor[lo] + (item - sStart[lo]);
    };
    uint32_t nextJ = 0, nextPos = 0, nextBest = 0, nextCluster = 0;
    uint4 nextInfo = {0, 0, 0, 0};
    auto fetch = [&](uint32_t item) {
        locate(item, nextJ, nextPos);
        // (a plain load of firstHit: it may see an old value out of the CU's L1, which only prunes less)
        nextBest = state->slot[nextJ].firstHit;
        nextCluster = cluster[nextPos];
        nextInfo = posInfo[nextPos];
    };
    const uint32_t itemStride = gridDim.x * wavesPerWG;
    if (MANY && blockIdx.x + gridDim.x * wave < total) fetch(blockIdx.x + gridDim.x * wave);
    for (uint32_t item = blockIdx.x + gridDim.x * wave; MANY && item < total; item += itemStride) {  // per wave
        // -- decided from the row's sparse histogram unless it is close to alpha --
        const uint32_t j = nextJ, pos = nextPos, best = nextBest, taken = nextCluster;
        const uint4 info = nextInfo;
        if (item + itemStride < total) fetch(item + itemStride);
        if (pos < best && taken == kNoCluster) {  // (wave-uniform up to the race on firstHit, which only prunes)
            const uint32_t sqRep = sSlot[j].sq, sqRow = info.z;
            if (lane == 0) ++judged;
            bool hit = false, near = false;
            if (sqRep == 0 || sqRow == 0) {  // src/rowReordering.cu:263-268
                hit = (sqRep == 0 && sqRow == 0 ? 1.0f : 0.0f) > alpha;
            } else if (info.y - info.x > longRow) {
                // a long list keeps one wave busy for many round trips while the pass waits for it: the whole
                // workgroup evaluates it from the dense table instead (O(bins / T) per thread)
                near = true;
            } else {
                const bool own = sSlot[j].rep == kClusterOwnRow;
                const uint32_t* rep32 = reps + (size_t)(own ? 0u : sSlot[j].rep) * numBins;
                const ClusterCount* rep16 = own ? ownRow(j) : table;
                const double normRep = (double)sqrtf((float)sqRep), normRow = (double)sqrtf((float)sqRow);
                // min + max = x + y in every bin, so  sum max = |x|_1 + |y|_1 - sum min: only the min-sum and the
                // row's own |y|_1 are accumulated, with reciprocals instead of two divisions per bin (the value is
                // only compared with alpha, and anything within 1e-4 of it is evaluated exactly below)
                const double invRep = 1.0 / normRep, invRow = 1.0 / normRow;
                double minSum = 0.0, sumY = 0.0;
                // (every bin counts when the reference's fold skips no warp, i.e. T / 32 is a power of two: no modulo then)
                const bool allLive = (uint32_t)__popc(liveWarps) == (T + 31u) / 32u;
                // eight entries per lane at a time: their words are loaded together, then the eight gathers from
                // the representative are issued together - a lane's chain is two round trips per batch instead of
                // two per entry
                const uint32_t rowEnd = info.y;
                for (uint32_t i0 = info.x + lane; i0 < rowEnd; i0 += 8u * lanes) {
                    uint32_t words[8], repCounts[8];
#pragma unroll
                    for (uint32_t u = 0; u < 8; ++u) {
                        const uint32_t i = i0 + u * lanes;
                        words[u] = i < rowEnd ? encWords[i] : 0xFFFFFFFFu;
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 8; ++u)
                        repCounts[u] = words[u] != 0xFFFFFFFFu ? (own ? (uint32_t)rep16[words[u] >> 16] : rep32[words[u] >> 16]) : 0u;
#pragma unroll
                    for (uint32_t u = 0; u < 8; ++u) {
                        if (words[u] == 0xFFFFFFFFu || !(allLive || binCounts(words[u] >> 16, T, liveWarps))) continue;
                        const double x = (double)repCounts[u] * invRep, y = (double)(words[u] & 0xFFFFu) * invRow;
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
    if constexpr (MANY) {
        // -- the near ones again, by the whole workgroup, in the reference's order of operations (after all rounds:
        //    the waves run through their items without waiting for each other; a workgroup has at most
        //    kClusterMaxActive * maxChunk / gridDim.x <= 1024 items, the host sees to that) --
        __syncthreads();
        const uint32_t nearCount = sNearCount;
        for (uint32_t n = 0; n < nearCount; ++n) {
            const uint32_t j = sNear[n] >> 27, pos = sCursor[j] + (sNear[n] & 0x07FFFFFFu);
            if (threadIdx.x == 0)
                shared[2] = __hip_atomic_load(&state->slot[j].firstHit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const uint32_t best = shared[2];
            __syncthreads();
            if (best < pos) continue;  // behind an accepted position: judged again by a later pass
            const float sim = exactSimilarity(j, order[pos]);
            if (threadIdx.x == 0 && sim > alpha) atomicMin(&state->slot[j].firstHit, pos);
            ++exact;
        }
        __syncthreads();
    }
    // the last workgroup to arrive closes the pass
    if constexpr (MANY) {  // lane 0 of every wave counted its items
        if (threadIdx.x == 0) shared[2] = 0;
        __syncthreads();
        if (judged) atomicAdd(&shared[2], judged);
        __syncthreads();
    }
    CLUSTER_STAMP(2);
    if (threadIdx.x == 0) {
        if (MANY && shared[2]) atomicAdd(&state->judged, shared[2]);
        if (exact) atomicAdd(&state->exact, exact);
        __threadfence();
        shared[0] = atomicAdd(&state->arrived, 1u) == participants - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!shared[0]) return;
    __threadfence();
    CLUSTER_STAMP(3);

    // 1. decisions, oldest cluster first.  Lane j of the first wave holds cluster j; the loop over the clusters reads the
    //    one it is at with v_readlane, so nothing but registers is touched (a single thread walking the slots in LDS
    //    cost more than everything else in a pass of few items).
    if (threadIdx.x < active) {
        sHit[threadIdx.x] = __hip_atomic_load(&state->slot[threadIdx.x].firstHit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sSumSq[threadIdx.x] = 0;
        sSumTotal[threadIdx.x] = 0;
        // a parked cluster's verdict on the row in front of it stands; has somebody else taken that row meanwhile?
        sNew[threadIdx.x] = 0;
        if (sSlot[threadIdx.x].parked) {
            const uint32_t at = sSlot[threadIdx.x].cursor;
            sHit[threadIdx.x] = at;
            sNew[threadIdx.x] = __hip_atomic_load(&cluster[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kNoCluster;
        }
    }
    __syncthreads();
    CLUSTER_STAMP(8);
    if (threadIdx.x < 32u) {
        const uint32_t j = threadIdx.x;
        const bool mine = j < active;
        ClusterSlot c = loadSlot(sSlot[mine ? j : 0u]);
        const uint32_t myHit = mine ? sHit[j] : kNoCluster, myLen = mine ? sLen[j] : 0u, myStart = mine ? sCursor[j] : 0u;
        const uint32_t myAssigned = mine ? sNew[j] : 0u;
        // which older clusters have (in this pass) a hit on the row I accepted / on my seed?
        const uint32_t hitLanes = (uint32_t)__ballot(mine && myHit != kNoCluster);
        uint32_t sameHit = 0, seedHit = 0;
        for (uint32_t todo = hitLanes; todo; todo &= todo - 1u) {
            const uint32_t i = (uint32_t)__builtin_ctz(todo);
            const uint32_t h = __builtin_amdgcn_readlane(myHit, i);
            if (i < j) {
                if (h == myHit) sameHit |= 1u << i;
                if (h == c.seed) seedHit |= 1u << i;
            }
        }
        CLUSTER_STAMP(10);
        uint32_t freeReps = freeRepsIn, dropped = 0, accepted = 0;   // uniform over the lanes
        uint32_t myAccept = kNoCluster, myDrop = 0, myFirstMerge = 0, myConfirmNow = 0;
        if (!speculate && (uint32_t)__ballot(mine && (c.parked || (!c.confirmed && myHit != kNoCluster))) == 0u) {
            // Nobody ran ahead in this pass: every hit lies behind the older clusters' cursors and can be taken at once - by the
            // oldest cluster that has it - and no unconfirmed cluster has a hit.  One step for all lanes, no walk.
            const bool hasHit = mine && myHit != kNoCluster;
            const bool takes = hasHit && sameHit == 0u && c.confirmed;
            accepted = (uint32_t)__ballot(takes);
            if (hasHit) {
                // judging costs time even beside the hit, so the next pass looks twice as far as this hit was
                const uint32_t gap = 2u * (myHit - myStart + 1u);
                c.chunk = gap < kClusterMinChunk ? kClusterMinChunk : (gap > maxChunk ? maxChunk : gap);
                c.cursor = myHit + 1u;
                if (takes) myAccept = myHit;
            } else if (mine && myLen) {
                c.cursor = myStart + myLen;
                if (myLen == c.chunk && c.confirmed) c.chunk = 4u * c.chunk > maxChunk ? maxChunk : 4u * c.chunk;
            }
            const uint32_t needs = (uint32_t)__ballot(takes && c.rep == kClusterOwnRow);   // first merges: one free buffer each, in order
            if (takes && c.rep == kClusterOwnRow) {
                uint32_t f = freeRepsIn;
                for (uint32_t n = __popc(needs & ((1u << j) - 1u)); n; --n) f &= f - 1u;
                c.rep = (uint32_t)__builtin_ctz(f);
                myFirstMerge = 1;
            }
            for (uint32_t n = __popc(needs); n; --n) freeReps &= freeReps - 1u;
            if (mine && !c.confirmed && (seedHit & accepted)) myDrop = 1;   // an older cluster took the seed
            uint32_t olderMin = mine && !myDrop ? c.cursor : 0xFFFFFFFFu;   // exclusive prefix minimum of the new cursors
            for (uint32_t w = 1; w < 32u; w <<= 1) {
                const uint32_t other = __shfl_up(olderMin, w, 32);
                if (j >= w) olderMin = olderMin < other ? olderMin : other;
            }
            olderMin = __shfl_up(olderMin, 1, 32);
            if (j == 0) olderMin = 0xFFFFFFFFu;
            if (mine && !c.confirmed && !myDrop && olderMin > c.seed) {
                c.confirmed = 1;
                myConfirmNow = 1;
            }
            dropped = __popc((uint32_t)__ballot(myDrop != 0u));
        } else {
            // A cluster without a hit moves on by itself.  For one with a hit the question is whether every older cluster's cursor
            // is behind the row.  The new cursor of an older cluster without a hit is known; that of one with a hit is its hit
            // or one more.  So with L = the smallest of those (hits counted as they are): a hit below L can be taken - and the
            // cluster confirmed on the way, L is then behind its seed too - and anything else waits in front of its row
            // (also a row an older cluster takes in this very pass: the next pass sees it assigned).  No lane waits for
            // another one's decision; waiting where a walk through the lanes would have found out more costs a pass at most.
            const bool hasHit = mine && myHit != kNoCluster;
            if (mine && !hasHit && myLen) {
                c.cursor = myStart + myLen;
                // (a tentative cluster looks no further per pass than it did at first: what it judges may be for nothing)
                if (myLen == c.chunk && c.confirmed) c.chunk = 4u * c.chunk > maxChunk ? maxChunk : 4u * c.chunk;
            }
            auto olderMinimum = [&](uint32_t mineOrNone) {   // exclusive prefix minimum over the lanes
                uint32_t m = mineOrNone;
                for (uint32_t w = 1; w < 32u; w <<= 1) {
                    const uint32_t other = __shfl_up(m, w, 32);
                    if (j >= w) m = m < other ? m : other;
                }
                m = __shfl_up(m, 1, 32);
                return j == 0 ? 0xFFFFFFFFu : m;
            };
            const uint32_t bound = olderMinimum(!mine ? 0xFFFFFFFFu : (hasHit ? myHit : c.cursor));
            CLUSTER_STAMP(11);
            bool takes = false;
            if (hasHit) {
                // judging costs time even beside the hit, so the next pass looks twice as far as this hit was
                if (!c.parked) {
                    const uint32_t gap = 2u * (myHit - myStart + 1u);
                    c.chunk = gap < kClusterMinChunk ? kClusterMinChunk : (gap > maxChunk ? maxChunk : gap);
                }
                if (c.parked && myAssigned) {  // somebody older took it meanwhile
                    c.cursor = myHit + 1u;
                    c.parked = 0;
                } else if (myHit < bound && (c.confirmed || bound > c.seed)) {  // every older cluster has passed over it
                    if (!c.confirmed) {
                        c.confirmed = 1;
                        myConfirmNow = 1;
                    }
                    takes = true;
                    myAccept = myHit;
                    c.cursor = myHit + 1u;
                    c.parked = 0;
                } else {  // an older cluster has not decided that position yet: wait in front of it
                    c.cursor = myHit;
                    c.parked = 1;
                }
            }
            accepted = (uint32_t)__ballot(takes);
            const uint32_t needs = (uint32_t)__ballot(takes && c.rep == kClusterOwnRow);   // first merges: one free buffer each, in order
            if (takes && c.rep == kClusterOwnRow) {
                uint32_t f = freeRepsIn;
                for (uint32_t n = __popc(needs & ((1u << j) - 1u)); n; --n) f &= f - 1u;
                c.rep = (uint32_t)__builtin_ctz(f);
                myFirstMerge = 1;
            }
            for (uint32_t n = __popc(needs); n; --n) freeReps &= freeReps - 1u;
            CLUSTER_STAMP(12);
            // the unconfirmed clusters: dropped when an older cluster took the seed in this pass, confirmed when every older
            // cluster that stays has its cursor behind the seed
            if (mine && !c.confirmed && (seedHit & accepted)) myDrop = 1;
            dropped = __popc((uint32_t)__ballot(myDrop != 0u));
            const uint32_t olderAll = olderMinimum(mine && !myDrop ? c.cursor : 0xFFFFFFFFu);
            if (mine && !c.confirmed && !myDrop && olderAll > c.seed) {
                c.confirmed = 1;
                myConfirmNow = 1;
            }
        }
        CLUSTER_STAMP(9);
        c.firstHit = kNoCluster;
        // 2. dropped and finished clusters leave (a tentative cluster that has judged everything waits for its confirmation)
        const bool finished = mine && !myDrop && c.confirmed && c.cursor >= rows;
        const bool keep = mine && !myDrop && !finished;
        const uint32_t keepMask = (uint32_t)__ballot(keep), finishedMask = (uint32_t)__ballot(finished);
        const uint32_t slotNow = __popc(keepMask & ((1u << j) - 1u));
        if (mine) {
            sAccept[j] = myAccept;
            sFirstMerge[j] = (uint8_t)myFirstMerge;
            sDrop[j] = (uint8_t)(keep ? slotNow : 0xFFu);   // where the slot goes (0xFF: nowhere)
            if (myConfirmNow && !myDrop) cluster[c.seed] = c.id;
            if (myAccept != kNoCluster) {
                cluster[myAccept] = c.id;
                sNew[j] = order[myAccept];   // the row to merge (all of them fetched at once)
                sHit[__popc(accepted & ((1u << j) - 1u))] = j;   // (the hits are in registers by now: the list of clusters that merge)
            }
            storeSlot(sSlot[j], c);   // (moved to its place after the merges, which read rep and seedRow by the old index)
        }
        // the first chunk of the next clusters: running mean over the finished ones (any order)
        uint32_t startChunk = startChunkIn, minCursor = keep ? c.cursor : rows;
        for (uint32_t f = finishedMask; f; f &= f - 1u) {
            const uint32_t k = (uint32_t)__builtin_ctz(f);
            startChunk = (3u * startChunk + __builtin_amdgcn_readlane(c.chunk, k)) / 4u;
            const uint32_t r = __builtin_amdgcn_readlane(c.rep, k);
            if (r != kClusterOwnRow) freeReps |= 1u << r;
        }
        for (uint32_t w = 16; w >= 1; w >>= 1) {
            const uint32_t other = __shfl_xor(minCursor, w, 32);
            minCursor = minCursor < other ? minCursor : other;
        }
        if (j == 0) {
            shared[0] = __popc(keepMask);
            shared[1] = freeReps;
            shared[2] = dropped;
            shared[3] = minCursor;
            sStart[0] = __popc((uint32_t)__ballot(keep && !c.confirmed));   // (the item list is not needed any more)
            sStart[1] = __popc((uint32_t)__ballot(mine && myConfirmNow && !myDrop));
            sStart[2] = startChunk < 2u * kClusterMinChunk ? 2u * kClusterMinChunk : (startChunk > maxChunk ? maxChunk : startChunk);
            sStart[3] = __popc(accepted);
            sStart[4] = __popc((uint32_t)__ballot(mine && myLen > 0));
            sStart[5] = __popc(accepted);
        }
    }
    __syncthreads();
    CLUSTER_STAMP(4);
    // 3. merges: every accepted row is added to its cluster's representative.  The clusters are independent
    //    here, so all merges run in one sweep (per-cluster sums: a wave adds its lanes up and issues one LDS atomic)
    const uint32_t merging = sStart[5];
    // (a thread's bins are threadIdx.x, threadIdx.x + T, ...: bin % T is the thread's number, so whether its bins take part in
    // the reference's block-wide sums is one bit per thread, not a modulo per bin)
    const bool myBinsCount = (liveWarps >> (threadIdx.x >> 5)) & 1u;
    // A thread owns few bins of a representative: the loads of several merges are issued together, all of a thread's bins at
    // once (one merge after the other, bin after bin, was the longest phase of a small pass).
    auto mergeBatches = [&](auto batchTag, auto binsTag) {
        constexpr uint32_t NB = decltype(batchTag)::value, NE = decltype(binsTag)::value;
        for (uint32_t a0 = 0; a0 < merging; a0 += NB) {  // uniform
            uint32_t have[NB][NE], plus[NB][NE];
#pragma unroll
            for (uint32_t b = 0; b < NB; ++b) {
                if (a0 + b >= merging) break;  // uniform
                const uint32_t j = sHit[a0 + b];
                const uint32_t* rep = reps + (size_t)sSlot[j].rep * numBins;
                const ClusterCount* add = table + (size_t)sNew[j] * numBins;
                const ClusterCount* own = sFirstMerge[j] ? ownRow(j) : nullptr;
#pragma unroll
                for (uint32_t k = 0; k < NE; ++k) {
                    const uint32_t i = threadIdx.x + k * T;
                    have[b][k] = i < numBins ? (own ? (uint32_t)own[i] : rep[i]) : 0u;
                    plus[b][k] = i < numBins ? (uint32_t)add[i] : 0u;
                }
            }
#pragma unroll
            for (uint32_t b = 0; b < NB; ++b) {
                if (a0 + b >= merging) break;  // uniform
                const uint32_t j = sHit[a0 + b];
                uint32_t* rep = reps + (size_t)sSlot[j].rep * numBins;
                uint32_t sq = 0, tot = 0;
#pragma unroll
                for (uint32_t k = 0; k < NE; ++k) {
                    const uint32_t i = threadIdx.x + k * T;
                    if (i >= numBins) continue;
                    const uint32_t v = have[b][k] + plus[b][k];
                    rep[i] = v;
                    sq += v * v;
                    tot += v;
                }
                if (!myBinsCount) sq = tot = 0;
                for (uint32_t w = lanes >> 1; w >= 1; w >>= 1) {  // `lanes` = 32 in a block's last, half-filled wave
                    sq += __shfl_xor(sq, w, 64);
                    tot += __shfl_xor(tot, w, 64);
                }
                if ((threadIdx.x & 63u) == 0) {   // (every lane adding by itself - same-address LDS atomics - was slower: 158 -> 196 ms on mycielskian15)
                    atomicAdd(&sSumSq[j], sq);
                    atomicAdd(&sSumTotal[j], tot);
                }
            }
        }
    };
    const uint32_t binsPerThread = (numBins + T - 1u) / T;
    if (binsPerThread <= 4u) {
        mergeBatches(std::integral_constant<uint32_t, 4>{}, std::integral_constant<uint32_t, 4>{});
    } else if (binsPerThread <= 8u) {
        mergeBatches(std::integral_constant<uint32_t, 2>{}, std::integral_constant<uint32_t, 8>{});
    } else {
        for (uint32_t a = 0; a < merging; ++a) {  // uniform
            const uint32_t j = sHit[a];
            uint32_t* rep = reps + (size_t)sSlot[j].rep * numBins;
            const ClusterCount* add = table + (size_t)sNew[j] * numBins;
            const ClusterCount* own = sFirstMerge[j] ? ownRow(j) : nullptr;
            uint32_t sq = 0, tot = 0;
            for (uint32_t i = threadIdx.x; i < numBins; i += blockDim.x) {
                const uint32_t v = (own ? (uint32_t)own[i] : rep[i]) + (uint32_t)add[i];
                rep[i] = v;
                if (myBinsCount) {
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
    }
    __syncthreads();
    {
        ClusterSlot c;
        const bool moves = threadIdx.x < active && sDrop[threadIdx.x] != 0xFFu;
        if (moves) {
            c = loadSlot(sSlot[threadIdx.x]);
            if (sAccept[threadIdx.x] != kNoCluster) {
                c.sq = sSumSq[threadIdx.x];
                c.total = sSumTotal[threadIdx.x];
            }
        }
        __syncthreads();
        if (moves) storeSlot(sSlot[sDrop[threadIdx.x]], c);
    }
    __syncthreads();
    uint32_t left = shared[0];
    const uint32_t freeReps = shared[1], dropped = shared[2], minCursor = shared[3];
    const uint32_t unconfirmed = sStart[0], confirmedNow = sStart[1], startChunk = sStart[2], hitsNow = sStart[3], judging = sStart[4];
    uint32_t tentative = tentativeIn;
    __syncthreads();
    CLUSTER_STAMP(5);
    // 4. new clusters: the next positions without a cluster, in order.  The first of them is a cluster for certain when
    //    everything in flight has passed over it (round 2 seeded only those); beyond it, as many as may run unconfirmed.
    uint32_t scanPos = scanPosIn, numNew = 0;
    const uint32_t room = left < maxActive ? maxActive - left : 0u;
    const uint32_t allowed = tentative > unconfirmed ? tentative - unconfirmed : 0u;
    // (with no tentative cluster allowed, only a position everything in flight has passed over can become a seed)
    const uint32_t cap = allowed == 0 && minCursor <= scanPos ? 0u : (room < allowed + 1u ? room : allowed + 1u);   // candidates to look for
    __threadfence();  // (the assignments of step 1 are read back here)
    for (uint32_t round = 0; round < 8u && numNew < cap && scanPos < rows; ++round) {  // uniform
        const uint32_t p = scanPos + threadIdx.x;
        const bool isFree = p < rows && __hip_atomic_load(&cluster[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kNoCluster;
        const unsigned long long mask = __ballot(isFree);
        if (lane == 0) sBallot[wave] = mask;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t n = numNew;
            for (uint32_t w = 0; w < wavesPerWG && n < cap; ++w) {
                unsigned long long m = sBallot[w];
                while (m && n < cap) {
                    const uint32_t bit = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1;
                    sNew[n++] = scanPos + 64u * w + bit;
                }
            }
            shared[0] = n;
            shared[1] = n == cap ? sNew[n - 1] + 1u : (scanPos + T < rows ? scanPos + T : rows);
        }
        __syncthreads();
        numNew = shared[0];
        scanPos = shared[1];
        __syncthreads();
    }
    uint32_t certain = 0;
    if (numNew) {  // uniform
        certain = minCursor > sNew[0] ? 1u : 0u;
        uint32_t take = allowed + certain < room ? allowed + certain : room;
        if (take < numNew) {  // the candidates not taken are found again by a later pass
            scanPos = sNew[take];
            numNew = take;
        }
    }
    const uint32_t nextId = nextIdIn;
    if (numNew == 0) certain = 0;
    // A dropped cluster halves the number that may run unconfirmed.  A pass without a drop adds one when it brought evidence
    // that rows end up in clusters of their own: a seed confirmed, or fewer than a quarter of the clusters that judged rows
    // took one (where clusters are large nearly every one does, every pass).  Confirmations alone got stuck: with every slot
    // taken by a cluster that scans hand over hand nothing new is seeded, so nothing is confirmed, and the passes stay of
    // that kind - the reddit-like shard spent 4 900 of 8 783 passes there.
    if (dropped) tentative >>= 1;
    else if ((confirmedNow || certain || hitsNow * 4u < judging) && tentative < maxActive) ++tentative;
    // (the host's choice between the two rules of step 5: a cap below kClusterSpeculateFrom keeps the passes to round 2's rule)
    if (tentative > tentativeCap) tentative = tentativeCap;
    if (threadIdx.x < numNew) {
        const uint32_t u = sNew[threadIdx.x], row = order[u];
        ClusterSlot c;
        c.seed = u;
        c.seedRow = row;
        c.cursor = u + 1;
        c.chunk = startChunk;
        c.id = nextId + 1u + threadIdx.x;
        c.sq = rowSquares[row];
        c.total = rowTotals[row];
        c.firstHit = kNoCluster;
        c.rep = kClusterOwnRow;
        c.parked = 0;
        c.confirmed = threadIdx.x == 0 && certain ? 1u : 0u;
        if (c.confirmed) cluster[u] = c.id;
        storeSlot(sSlot[left + threadIdx.x], c);
    }
    left += numNew;
    __syncthreads();
    CLUSTER_STAMP(6);
    // 5. what the next pass judges.  Running ahead of the older clusters pays when most rows end up in clusters of their
    //    own (what a younger cluster judges early is then rarely taken away by an older one) and costs when clusters are
    //    large.  The number of clusters that may run unconfirmed measures exactly that; below kClusterSpeculateFrom a
    //    cluster only judges what every older one has decided (round 2's rule: all it accepts can be taken at once, nothing
    //    is judged for nothing).
    if (threadIdx.x < 32u) {
        const uint32_t j = threadIdx.x;
        const bool mine = j < left;
        const uint32_t cursor = mine ? sSlot[j].cursor : 0xFFFFFFFFu, chunk = mine ? sSlot[j].chunk : 0u;
        const bool parked = mine && sSlot[j].parked;
        uint32_t olderMin = cursor;   // exclusive prefix minimum of the cursors
        for (uint32_t w = 1; w < 32u; w <<= 1) {
            const uint32_t other = __shfl_up(olderMin, w, 32);
            if (j >= w) olderMin = olderMin < other ? olderMin : other;
        }
        olderMin = __shfl_up(olderMin, 1, 32);
        uint32_t limit = rows;
        if (tentative < kClusterSpeculateFrom && j > 0 && olderMin < limit) limit = olderMin;
        const uint32_t len = mine && cursor < limit && !parked ? (chunk < limit - cursor ? chunk : limit - cursor) : 0u;
        uint32_t upTo = len;          // inclusive prefix sum of the lengths
        for (uint32_t w = 1; w < 32u; w <<= 1) {
            const uint32_t other = __shfl_up(upTo, w, 32);
            if (j >= w) upTo += other;
        }
        if (mine) {
            sSlot[j].len = len;
            sSlot[j].start = upTo - len;
        }
        if (j + 1 == left || (left == 0 && j == 0)) shared[3] = left ? upTo : 0u;
    }
    __syncthreads();
    if (threadIdx.x < left) storeSlot(next->slot[threadIdx.x], loadSlot(sSlot[threadIdx.x]));
    if (threadIdx.x == 0) {
        next->numActive = left;
        next->nextId = nextId + numNew;
        next->scanPos = scanPos;
        next->freeReps = freeReps;
        next->arrived = 0;
        next->passes = passesIn + 1;
        // (every participant added its counts before it arrived, and this workgroup arrived last)
        next->judged = __hip_atomic_load(&state->judged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        next->exact = __hip_atomic_load(&state->exact, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        next->ahead = aheadIn + (speculate ? 1u : 0u);
        next->dropped = droppedIn + dropped;
        next->tentative = tentative;
        next->assigned = assignedIn + confirmedNow + hitsNow + certain;
        next->startChunk = startChunk;
        next->totalItems = shared[3];
#ifdef BSMR_LAB_STAMPS
        labT[7] = __builtin_readcyclecounter();
        for (int k = 0; k < 16; ++k) next->stamps[k] = state->stamps[k];
        for (int k = 1; k <= 7; ++k) next->stamps[k] += labT[k] - labT[k - 1];
        next->stamps[8] += labT[8] - labT[3];
        next->stamps[9] += labT[9] - labT[8];
        if (labT[12]) {
            next->stamps[10] += labT[10] - labT[8];
            next->stamps[11] += labT[11] - labT[10];
            next->stamps[12] += labT[12] - labT[11];
            next->stamps[13] += labT[9] - labT[12];
            next->stamps[14] += 1;
        }
        next->stamps[0] += 1;
#endif
        const uint32_t done = left == 0 && scanPos >= rows ? 1u : 0u;
        next->done = done;
        // (the launches behind the last pass alternate between the two copies: both say that there is nothing left to do;
        // the copy with the larger `passes` holds the final state)
        if (done) state->done = 1u;
    }
}

}  // namespace bsmr
