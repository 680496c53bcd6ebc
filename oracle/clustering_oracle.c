/*
 * CPU restatement of the reference's row clustering - TEST INFRASTRUCTURE ONLY.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * What it follows (reference file:line, read as text):
 *   - column-bin histogram + dispersion score per row: src/rowReordering.cu:49-93
 *   - rows sorted by dispersion, ascending and stable:  src/rowReordering.cu:1056-1062
 *   - leading zero-dispersion rows -> cluster 0:         src/rowReordering.cu:939-951
 *   - one single-block kernel per cluster, seeded by the first row the previous cluster
 *     rejected, scanning the later unassigned rows in order and merging every row whose
 *     similarity exceeds alpha into the representative: src/rowReordering.cu:325-432
 *     (the kernels are chained through per-row mutexes, which is the same as finishing
 *     cluster c before c+1 starts)
 *   - normalised weighted Jaccard similarity on the integer histograms: :235-293
 *   - the block-wide sum every one of those quantities goes through:
 *     include/cudaUtil.cuh:14-45
 *   - thread count of the clustering block: src/rowReordering.cu:912-922
 *   - final stable sort by cluster id, the logged cluster count, and removal of the
 *     leading empty rows: src/rowReordering.cu:985-992, :1082-1090
 *
 * The block-wide sum is restated as the reference executes it, not as the mathematical sum:
 * thread t accumulates bins t, t+T, t+2T, ... in that order; a warp adds its 32 partials
 * in a balanced tree (shuffle-xor butterfly, lane 0); the per-warp values are then folded
 * with `for (s = T/64; s >= 1; s >>= 1) v[w] += v[w+s] (w < s)`.  When the warp count W = T/32
 * is not a power of two that loop never reads some warps (W = 6 folds warps {0,3,1,4} and
 * skips {2,5}; W = 10 skips {4,9}; W = 1 keeps warp 0), so the bins those warps own do not
 * enter the sum of squares, the min-sum or the max-sum.  Reproducing this is what makes the
 * cluster counts and dense-block statistics of the reference's published logs
 * (scripts/results_suiteSparse_dataset/BSMR_results) come out exactly; see
 * tests/golden/reference_logs.json and tests/test_reference_logs.py.
 *
 * Arithmetic: histogram entries and sums of squares are 32-bit unsigned (wrap-around kept),
 * norms are sqrtf of the float-converted sums, quotients are IEEE fp32 divisions, min/max
 * sums are fp32 in exactly the order above (compile with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NULL_CLUSTER 0xFFFFFFFFu

/* src/rowReordering.cu:912-922 */
static uint32_t cluster_block_threads(uint32_t numBins) {
    if (numBins < 32) return 32;
    uint32_t cand = 32u * (uint32_t)ceilf((float)(numBins / 4u) / 32.0f);
    if (cand < 32) cand = 32;
    return cand > 1024 ? 1024 : cand;
}

/* warps whose value reaches shm[0] in include/cudaUtil.cuh:37-43 */
static void folded_warps(uint32_t threads, uint8_t* used /* [threads/32] */) {
    const uint32_t W = threads / 32u;
    /* symbolic run of the fold: members[w] = set of warps summed into slot w */
    uint8_t* members = (uint8_t*)calloc((size_t)W * W, 1);
    for (uint32_t w = 0; w < W; ++w) members[(size_t)w * W + w] = 1;
    for (uint32_t s = threads / 64u; s >= 1; s >>= 1)
        for (uint32_t w = 0; w < s; ++w)
            for (uint32_t k = 0; k < W; ++k) members[(size_t)w * W + k] |= members[(size_t)(w + s) * W + k];
    memcpy(used, members, W);
    free(members);
}

/* exported for the tests: which bins enter the sums for a given histogram length */
uint32_t oracle_cluster_threads(uint32_t numBins) { return cluster_block_threads(numBins); }
void oracle_cluster_bin_mask(uint32_t numBins, uint8_t* mask /* [numBins] */) {
    const uint32_t T = cluster_block_threads(numBins);
    uint8_t used[32];
    folded_warps(T, used);
    for (uint32_t b = 0; b < numBins; ++b) mask[b] = used[(b % T) / 32u];
}

/* fp32 block sum of per-thread partials, in the reference's order (partials is clobbered) */
static float block_sum_f32(float* partial, uint32_t threads) {
    const uint32_t W = threads / 32u;
    float warp[32];
    for (uint32_t w = 0; w < W; ++w) {
        float* v = partial + 32u * w;
        for (uint32_t step = 1; step < 32; step <<= 1)
            for (uint32_t l = 0; l < 32; l += 2 * step) v[l] = v[l] + v[l + step];
        warp[w] = v[0];
    }
    for (uint32_t s = threads / 64u; s >= 1; s >>= 1)
        for (uint32_t w = 0; w < s; ++w) warp[w] = warp[w] + warp[w + s];
    return warp[0];
}

static uint32_t block_sum_u32(uint32_t* partial, uint32_t threads) {
    const uint32_t W = threads / 32u;
    uint32_t warp[32];
    for (uint32_t w = 0; w < W; ++w) {
        uint32_t acc = 0;
        for (uint32_t l = 0; l < 32; ++l) acc += partial[32u * w + l];
        warp[w] = acc;
    }
    for (uint32_t s = threads / 64u; s >= 1; s >>= 1)
        for (uint32_t w = 0; w < s; ++w) warp[w] += warp[w + s];
    return warp[0];
}

/* src/rowReordering.cu:235-293 through include/cudaUtil.cuh:27-45; scratch: 2*threads words */
static float similarity_as_executed(const uint32_t* rep, const uint32_t* cmp, uint32_t numBins, uint32_t threads,
                                    void* scratch) {
    uint32_t* ur = (uint32_t*)scratch;
    uint32_t* uc = ur + threads;
    memset(ur, 0, 2u * threads * sizeof(uint32_t));
    for (uint32_t b = 0; b < numBins; ++b) {
        const uint32_t t = b % threads;
        ur[t] += (uint32_t)((int)rep[b] * (int)rep[b]);
        uc[t] += (uint32_t)((int)cmp[b] * (int)cmp[b]);
    }
    const uint32_t sqRep = block_sum_u32(ur, threads);
    const uint32_t sqCmp = block_sum_u32(uc, threads);
    if (sqRep == 0 && sqCmp == 0) return 1.0f;
    if (sqRep == 0 || sqCmp == 0) return 0.0f;
    const float normRep = sqrtf((float)sqRep);
    const float normCmp = sqrtf((float)sqCmp);
    float* fmin_ = (float*)scratch;
    float* fmax_ = fmin_ + threads;
    for (uint32_t t = 0; t < 2u * threads; ++t) fmin_[t] = 0.0f;
    for (uint32_t b = 0; b < numBins; ++b) {
        const uint32_t t = b % threads;
        const float x = (float)rep[b] / normRep;
        const float y = (float)cmp[b] / normCmp;
        fmin_[t] = fmin_[t] + fminf(x, y);
        fmax_[t] = fmax_[t] + fmaxf(x, y);
    }
    const float minSum = block_sum_f32(fmin_, threads);
    const float maxSum = block_sum_f32(fmax_, threads);
    return minSum / maxSum;
}

/* exported so tests can probe single pairs */
float oracle_cluster_similarity(const uint32_t* rep, const uint32_t* cmp, uint32_t numBins) {
    const uint32_t T = cluster_block_threads(numBins);
    void* scratch = malloc(2u * T * sizeof(uint32_t));
    const float s = similarity_as_executed(rep, cmp, numBins, T, scratch);
    free(scratch);
    return s;
}

typedef struct {
    uint32_t key, pos;
} KeyPos;
static int cmp_keypos(const void* a, const void* b) {
    const KeyPos* x = (const KeyPos*)a;
    const KeyPos* y = (const KeyPos*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0); /* stable */
}

/*
 * Whole row reordering.  permutation: [rows] (the first *numOut entries are valid: the
 * reordered non-empty rows).  Returns 0, or -1 when the dense rows x bins table cannot be
 * allocated (the reference allocates the same table on the device).
 */
int oracle_bsa_row_reordering(uint32_t rows, uint32_t cols, const uint32_t* rowOffsets, const uint32_t* colIndices,
                              uint32_t binWidth, float alpha, uint32_t* permutation, uint32_t* numOut,
                              int32_t* numClusters) {
    const uint32_t numBins = (uint32_t)ceilf((float)cols / (float)binWidth);
    const uint32_t T = cluster_block_threads(numBins);
    *numOut = 0;
    *numClusters = 0;
    if (rows == 0) return 0;
    uint32_t* enc = (uint32_t*)calloc((size_t)rows * numBins, sizeof(uint32_t));
    KeyPos* byDisp = (KeyPos*)malloc((size_t)rows * sizeof(KeyPos));
    uint32_t* cluster = (uint32_t*)malloc((size_t)rows * sizeof(uint32_t));
    uint32_t* rep = (uint32_t*)malloc((size_t)numBins * sizeof(uint32_t));
    uint32_t* pending = (uint32_t*)malloc((size_t)rows * sizeof(uint32_t));
    if (!enc || !byDisp || !cluster || !rep || !pending) {
        free(enc); free(byDisp); free(cluster); free(rep); free(pending);
        return -1;
    }
    /* src/rowReordering.cu:49-93 */
    for (uint32_t r = 0; r < rows; ++r) {
        uint32_t* e = enc + (size_t)r * numBins;
        const uint32_t nz = rowOffsets[r + 1] - rowOffsets[r];
        uint32_t disp = 0;
        if (nz) {
            for (uint32_t i = rowOffsets[r]; i < rowOffsets[r + 1]; ++i) ++e[colIndices[i] / binWidth];
            uint32_t touched = 0, slack = 0;
            for (uint32_t b = 0; b < numBins; ++b)
                if (e[b]) {
                    ++touched;
                    slack += binWidth - e[b];
                }
            disp = slack + nz * touched;
        }
        byDisp[r].key = disp;
        byDisp[r].pos = r;
    }
    qsort(byDisp, rows, sizeof(KeyPos), cmp_keypos);

    uint32_t zeroRows = 0;
    for (uint32_t p = 0; p < rows; ++p) cluster[p] = NULL_CLUSTER;
    while (zeroRows < rows && byDisp[zeroRows].key == 0) cluster[zeroRows++] = 0;

    /* one short parallel region per scan step: more than a few threads only adds barrier time
     * (and on a CPU-quota-limited box the spinning workers starve each other) */
    int maxThreads = 1;
#ifdef _OPENMP
    maxThreads = omp_get_max_threads() < 8 ? omp_get_max_threads() : 8;
#endif
    void* scratch = malloc((size_t)maxThreads * 2u * T * sizeof(uint32_t));
    enum { CHUNK = 256 };
    uint32_t start = zeroRows, id = 0;
    while (start < rows) {
        ++id;
        cluster[start] = id;
        memcpy(rep, enc + (size_t)byDisp[start].pos * numBins, (size_t)numBins * sizeof(uint32_t));
        size_t np = 0;
        for (uint32_t p = start + 1; p < rows; ++p)
            if (cluster[p] == NULL_CLUSTER) pending[np++] = p;
        size_t i = 0;
        while (i < np) {
            const size_t n = np - i < CHUNK ? np - i : CHUNK;
            long long firstHit = (long long)n;
#pragma omp parallel for schedule(static) reduction(min : firstHit) num_threads(maxThreads)
            for (long long j = 0; j < (long long)n; ++j) {
                int tid = 0;
#ifdef _OPENMP
                tid = omp_get_thread_num();
#endif
                if (j > firstHit) continue;
                const uint32_t* cmp = enc + (size_t)byDisp[pending[i + j]].pos * numBins;
                const float s = similarity_as_executed(rep, cmp, numBins, T, (char*)scratch + (size_t)tid * 2u * T * 4u);
                if (s > alpha && j < firstHit) firstHit = j;
            }
            if (firstHit == (long long)n) {
                i += n;
                continue;
            }
            const uint32_t p = pending[i + firstHit];
            cluster[p] = id;
            const uint32_t* add = enc + (size_t)byDisp[p].pos * numBins;
            for (uint32_t b = 0; b < numBins; ++b) rep[b] += add[b];
            i += (size_t)firstHit + 1;
        }
        while (start < rows && cluster[start] != NULL_CLUSTER) ++start;
    }

    /* src/rowReordering.cu:985-992 */
    KeyPos* byCluster = (KeyPos*)malloc((size_t)rows * sizeof(KeyPos));
    for (uint32_t p = 0; p < rows; ++p) {
        byCluster[p].key = cluster[p];
        byCluster[p].pos = p;
    }
    qsort(byCluster, rows, sizeof(KeyPos), cmp_keypos);
    /* the count is read from the sorted id array at the pre-sort position of its last element */
    *numClusters = (int32_t)byCluster[byCluster[rows - 1].pos].key + (zeroRows != 0 ? 1 : 0);
    /* src/rowReordering.cu:1082-1090 */
    uint32_t out = 0;
    int leading = 1;
    for (uint32_t i2 = 0; i2 < rows; ++i2) {
        const uint32_t row = byDisp[byCluster[i2].pos].pos;
        if (leading && rowOffsets[row + 1] == rowOffsets[row]) continue;
        leading = 0;
        permutation[out++] = row;
    }
    *numOut = out;
    free(byCluster); free(scratch); free(enc); free(byDisp); free(cluster); free(rep); free(pending);
    return 0;
}
