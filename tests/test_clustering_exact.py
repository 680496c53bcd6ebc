"""Row clustering: the product's sparse / incremental evaluation against the plain dense
restatement (oracle/clustering_oracle.c), bit for bit.

The reference's block-wide sum (include/cudaUtil.cuh:14-45) skips some warps when the
thread count of the clustering block (src/rowReordering.cu:912-922) is not a power-of-two
number of warps; the cases below walk through 1, 3, 5, 6, 7, 10 and 32 warps, rows that
live entirely in skipped bins, and alphas outside (0, 1)."""
import numpy as np
import pytest

import synth

import bsmr_oracle as bo


def clustered_pattern(rows, cols, groups, per_row, seed, noise=0.3):
    """rows drawn around `groups` column prototypes so that clusters actually form"""
    rng = np.random.default_rng(seed)
    protos = [rng.choice(cols, size=min(cols, 3 * per_row), replace=False) for _ in range(groups)]
    per = []
    for r in range(rows):
        g = rng.integers(groups)
        k = max(1, int(rng.integers(per_row // 2, per_row + 1)))
        own = rng.choice(protos[g], size=min(k, protos[g].size), replace=False)
        extra = rng.choice(cols, size=max(0, int(noise * k)), replace=False)
        per.append(np.unique(np.concatenate([own, extra])))
    ro = np.zeros(rows + 1, dtype=np.uint32)
    ro[1:] = np.cumsum([c.size for c in per])
    return rows, cols, ro, np.concatenate(per).astype(np.uint32)


def test_thread_count_and_skipped_warps(oracle):
    # (bins, threads) from src/rowReordering.cu:912-922
    for nb, threads in ((1, 32), (31, 32), (32, 32), (160, 64), (300, 96), (520, 160), (768, 192),
                        (800, 224), (1250, 320), (4500, 1024), (100000, 1024)):
        assert oracle.lib.oracle_cluster_threads(nb) == threads == bo.cluster_threads(nb)
    live = {nb: sorted(set((np.nonzero(oracle.cluster_bin_mask(nb))[0] % t) // 32))
            for nb, t in ((300, 96), (520, 160), (768, 192), (800, 224), (1250, 320), (4500, 1024))}
    assert live[300] == [0, 1]                      # 3 warps: stride 1 only, warp 2 never read
    assert live[520] == [0, 1, 2, 3]                # 5 warps: strides 2, 1
    assert live[768] == [0, 1, 3, 4]                # 6 warps: strides 3, 1
    assert live[800] == [0, 1, 3, 4]                # 7 warps: strides 3, 1
    assert live[1250] == [0, 1, 2, 3, 5, 6, 7, 8]   # 10 warps: strides 5, 2, 1
    assert live[4500] == list(range(32))


def test_similarity_restatements_agree(oracle):
    rng = np.random.default_rng(5)
    for nb in (20, 300, 768, 1250):
        for _ in range(20):
            rep = (rng.random(nb) < 0.2) * rng.integers(1, 40, nb)
            row = (rng.random(nb) < 0.1) * rng.integers(1, 17, nb)
            a = oracle.cluster_similarity(rep, row)
            b = float(bo.similarity(rep.astype(np.int64), row.astype(np.int64)))
            assert a == b, (nb, a, b)
    # all-zero and one-sided cases (src/rowReordering.cu:263-268)
    z = np.zeros(40, np.uint32)
    one = z.copy()
    one[3] = 2
    assert oracle.cluster_similarity(z, z) == 1.0
    assert oracle.cluster_similarity(z, one) == 0.0 == oracle.cluster_similarity(one, z)
    # a row that only touches bins of a skipped warp looks empty to the block
    hidden = np.zeros(300, np.uint32)
    hidden[70] = 5                                   # thread 70 -> warp 2 of 3
    seen = np.zeros(300, np.uint32)
    seen[10] = 5
    assert oracle.cluster_similarity(hidden, hidden) == 1.0
    assert oracle.cluster_similarity(seen, hidden) == 0.0


CASES = [
    # rows, cols, bin width, groups, nnz per row, seed
    (300, 320, 16, 6, 12, 1),        # 20 bins, one warp
    (400, 4800, 16, 8, 20, 2),       # 300 bins, 3 warps
    (400, 8320, 16, 8, 24, 3),       # 520 bins, 5 warps
    (500, 12288, 16, 10, 30, 4),     # 768 bins, 6 warps
    (300, 16000, 20, 6, 40, 5),      # 800 bins, 7 warps
    (600, 20000, 16, 12, 16, 6),     # 1250 bins, 10 warps
    (200, 72000, 16, 4, 60, 7),      # 4500 bins, 1024 threads
]


@pytest.mark.parametrize("case", CASES, ids=[f"bins{-(-c[1] // c[2])}" for c in CASES])
@pytest.mark.parametrize("alpha", [0.1, 0.3, 0.6, 0.9])
def test_product_order_equals_dense_restatement(engine, oracle, case, alpha):
    rows, cols, bw, groups, per_row, seed = case
    rows, cols, ro, ci = clustered_pattern(rows, cols, groups, per_row, seed)
    perm, clusters = oracle.bsa_row_reordering(rows, cols, ro, ci, bw, alpha)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    pipe = engine.Pipeline(csr, alpha=alpha, delta=0.3, block_size=bw, device=-1)
    assert np.array_equal(pipe.array("reorderedRows"), perm)
    assert pipe.num_clusters == clusters
    if alpha == 0.1:
        assert clusters < rows        # the pattern does cluster


@pytest.mark.parametrize("alpha", [-0.5, 0.0, 0.3, 1.0, 1.5])
def test_rows_hidden_in_skipped_bins_and_extreme_alphas(engine, oracle, alpha):
    """300 bins / 3 warps: bins whose thread sits in warp 2 never enter a sum, so rows made
    only of them have 'zero' norm - similarity 1 among themselves, 0 with everything else."""
    rng = np.random.default_rng(11)
    cols, bw = 4800, 16
    hidden_bins = [b for b in range(300) if (b % 96) >= 64]
    per = []
    for r in range(120):
        if r % 3 == 0:                # hidden rows
            bins = rng.choice(hidden_bins, size=4, replace=False)
        elif r % 3 == 1:              # mixed
            bins = np.concatenate([rng.choice(hidden_bins, size=2, replace=False), rng.integers(0, 64, 3)])
        else:
            bins = rng.integers(0, 64, 5)
        per.append(np.unique(bins * bw + rng.integers(0, bw, bins.size)))
    per[7] = np.zeros(0, np.int64)   # an empty row -> cluster 0, dropped from the order
    ro = np.zeros(len(per) + 1, dtype=np.uint32)
    ro[1:] = np.cumsum([c.size for c in per])
    ci = np.concatenate(per).astype(np.uint32)
    perm, clusters = oracle.bsa_row_reordering(len(per), cols, ro, ci, bw, alpha)
    csr = engine.CSR.from_arrays(len(per), cols, ro, ci)
    pipe = engine.Pipeline(csr, alpha=alpha, delta=0.3, block_size=bw, device=-1)
    assert np.array_equal(pipe.array("reorderedRows"), perm)
    assert pipe.num_clusters == clusters
    assert 7 not in perm and perm.size == len(per) - 1
    if alpha == 0.3:
        # all hidden rows end up in one cluster: contiguous in the order
        pos = sorted(int(np.nonzero(perm == r)[0][0]) for r in range(0, 120, 3))
        assert pos[-1] - pos[0] == len(pos) - 1


def test_numpy_restatement_on_small_input(engine, oracle):
    rows, cols, ro, ci = clustered_pattern(60, 4800, 4, 10, 21)
    for alpha in (0.2, 0.5):
        perm, clusters = oracle.bsa_row_reordering(rows, cols, ro, ci, 16, alpha)
        rr, nc = bo.row_reordering(rows, cols, ro, ci, alpha, 16)
        assert np.array_equal(rr, perm) and nc == clusters
