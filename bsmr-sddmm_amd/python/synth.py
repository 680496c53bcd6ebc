"""Deterministic synthetic sparsity patterns for the BASELINE.json configurations.

None of the configuration matrices (nips, cop20k_A, reddit, DLMC) exists in the
container or on the GPU box (no network; the reference's dataset/nips.mtx is a
missing large blob), so every configuration has a seeded stand-in of the same
shape and nnz.  Each generator returns
(rows, cols, rowOffsets uint32[rows+1], colIndices uint32[nnz]) with ascending
column ids inside a row (what a sorted .mtx file would load as).
"""
from __future__ import annotations

import numpy as np


def _rows_to_csr(rows: int, cols: int, per_row_cols: list):
    ro = np.zeros(rows + 1, dtype=np.uint32)
    for r, c in enumerate(per_row_cols):
        ro[r + 1] = ro[r] + len(c)
    ci = np.concatenate(per_row_cols).astype(np.uint32) if rows else np.zeros(0, np.uint32)
    return rows, cols, ro, ci


def _fit_counts(raw: np.ndarray, total: int, cap: int) -> np.ndarray:
    """Scale non-negative weights to integers in [0, cap] that sum to `total`."""
    w = np.maximum(raw.astype(np.float64), 0)
    c = np.minimum(np.floor(w * (total / w.sum())).astype(np.int64), cap)
    i = 0
    order = np.argsort(-w, kind="stable")
    while c.sum() < total:  # hand the remainder to the heaviest rows
        j = order[i % len(order)]
        if c[j] < cap:
            c[j] += 1
        i += 1
    return c


def _fit_counts_fast(raw: np.ndarray, total: int, cap: int) -> np.ndarray:
    """_fit_counts for hundreds of thousands of rows: rescale the rows below the cap until the floor sum is within
    one per row of `total`, then hand the remainder to the heaviest rows below the cap, one each (vectorised)."""
    w = np.maximum(raw.astype(np.float64), 0)
    c = np.zeros(w.size, dtype=np.int64)
    free = np.ones(w.size, dtype=bool)
    for _ in range(64):
        left = total - int(c[~free].sum())
        scale = left / max(w[free].sum(), 1e-300)
        c[free] = np.floor(w[free] * scale).astype(np.int64)
        over = free & (c > cap)
        if not over.any():
            break
        c[over] = cap
        free &= ~over
    rem = total - int(c.sum())
    while rem > 0:
        order = np.argsort(-w, kind="stable")
        order = order[c[order] < cap]
        take = order[:rem]
        if take.size == 0:
            break
        c[take] += 1
        rem -= int(take.size)
    return c


def nips_like(rows=1500, cols=12419, nnz=746316, seed=1):
    """Bag-of-words shape (UCI NIPS dimensions): lognormal row lengths (mean ~498),
    Zipf(1.0) column popularity sampled without replacement inside a row."""
    rng = np.random.default_rng(seed)
    counts = _fit_counts(rng.lognormal(mean=0.0, sigma=0.6, size=rows), nnz, cols)
    logp = -np.log(np.arange(1, cols + 1, dtype=np.float64))
    perm = rng.permutation(cols)  # popular words are not the low column ids
    per_row = []
    for r in range(rows):
        k = int(counts[r])
        if k == 0:
            per_row.append(np.zeros(0, dtype=np.int64))
            continue
        keys = logp + rng.gumbel(size=cols)  # Gumbel top-k = weighted sampling w/o replacement
        top = np.argpartition(-keys, k - 1)[:k]
        per_row.append(np.sort(perm[top]))
    return _rows_to_csr(rows, cols, per_row)


def banded_mesh_like(n=121192, nnz=1362087, seed=2, empty_frac=0.18):
    """cop20k_A stand-in: strictly lower-triangular 2-D-mesh-like band (offsets
    within ~sqrt(n)) plus 5 % uniformly random long-range entries, ~18 % empty rows."""
    rng = np.random.default_rng(seed)
    w = rng.gamma(shape=4.0, scale=1.0, size=n)
    w[rng.random(n) < empty_frac] = 0
    w[0] = 0  # row 0 has no strictly-lower entries
    cap = np.minimum(np.arange(n), 64)
    counts = np.minimum(_fit_counts(w, nnz, 64), cap)
    deficit = nnz - int(counts.sum())
    r = n - 1
    while deficit > 0:  # refill from the bottom rows, which have the most room
        room = int(cap[r] - counts[r])
        if room > 0 and counts[r] > 0:
            take = min(room, deficit)
            counts[r] += take
            deficit -= take
        r -= 1
        if r < 1:
            r = n - 1
    band = int(np.sqrt(n))
    per_row = []
    for i in range(n):
        k = int(counts[i])
        if k == 0:
            per_row.append(np.zeros(0, dtype=np.int64))
            continue
        lo = max(0, i - band)
        chosen = set()
        while len(chosen) < k:
            need = k - len(chosen)
            near = rng.integers(lo, i, size=need)
            far = rng.integers(0, i, size=need)
            pick = np.where(rng.random(need) < 0.05, far, near)
            chosen.update(int(x) for x in pick)
            if i <= k:  # tiny rows: take everything available
                chosen = set(range(i))
                break
        per_row.append(np.sort(np.fromiter(list(chosen)[:k] if len(chosen) > k else chosen,
                                           dtype=np.int64)))
    rows, cols, ro, ci = _rows_to_csr(n, n, per_row)
    return rows, cols, ro, ci


def fem_node_blocks_like(n=121192, nnz=1362087, seed=2, block=16, block_frac=0.3):
    """Second cop20k_A stand-in (BASELINE configs[2], "hybrid dense + sparse path"): the band of banded_mesh_like plus
    the dense node blocks a finite-element matrix has - `block_frac` of the aligned groups of `block` consecutive
    unknowns are fully coupled (strictly lower triangle of a block x block clique, 120 entries for 16).  At the
    reference's default delta = 0.3 (>= 77 entries per 16 x 16 block) those cliques are dense blocks, the band
    stays in the residue, and with 11 entries per column nothing is promoted: the DEFAULT plan runs both kernels.
    (banded_mesh_like alone has no dense block at delta = 0.3.)  Same size and about the same nnz."""
    rng = np.random.default_rng(seed + 1000)
    groups = n // block
    chosen = np.nonzero(rng.random(groups) < block_frac)[0]
    tri = block * (block - 1) // 2
    rows, cols, ro, ci = banded_mesh_like(n=n, nnz=max(nnz - tri * int(chosen.size), n), seed=seed)
    li, lj = np.tril_indices(block, -1)
    br = (chosen[:, None] * block + li[None, :]).ravel().astype(np.int64)
    bc = (chosen[:, None] * block + lj[None, :]).ravel().astype(np.int64)
    r0 = np.repeat(np.arange(n, dtype=np.int64), np.diff(ro.astype(np.int64)))
    keys = np.unique(np.concatenate([r0 * n + ci.astype(np.int64), br * n + bc]))
    r, c = keys // n, keys % n
    ro2 = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ro2, r + 1, 1)
    return n, n, np.cumsum(ro2).astype(np.uint32), c.astype(np.uint32)


def bernoulli(rows=4096, cols=4096, density=0.1, seed=4):
    """DLMC-style unstructured mask: i.i.d. Bernoulli(density)."""
    rng = np.random.default_rng(seed)
    per_row = [np.nonzero(rng.random(cols) < density)[0] for _ in range(rows)]
    return _rows_to_csr(rows, cols, per_row)


def community_graph(n=4096, avg_degree=64, communities=8, inside=0.8, seed=3):
    """reddit-like shape at a configurable size: power-law out-degrees, `inside` of
    a row's edges fall in its own community (contiguous id range)."""
    rng = np.random.default_rng(seed)
    deg = _fit_counts(rng.pareto(1.5, size=n) + 0.2, n * avg_degree, n // 2)
    size = n // communities
    per_row = []
    for i in range(n):
        k = int(deg[i])
        if k == 0:
            per_row.append(np.zeros(0, dtype=np.int64))
            continue
        c0 = (i // size) * size
        c1 = min(c0 + size, n)
        chosen = set()
        while len(chosen) < k:
            need = k - len(chosen)
            a = rng.integers(c0, c1, size=need)
            b = rng.integers(0, n, size=need)
            chosen.update(int(x) for x in np.where(rng.random(need) < inside, a, b))
        per_row.append(np.sort(np.fromiter(list(chosen)[:k], dtype=np.int64)))
    return _rows_to_csr(n, n, per_row)


def _distinct(rng, lo, hi, k, taken=None):
    """k distinct integers of [lo, hi) (not in the sorted array `taken`), drawn with replacement until enough are distinct."""
    span = hi - lo
    k = min(k, span - (0 if taken is None else int(np.searchsorted(taken, hi) - np.searchsorted(taken, lo))))
    if k <= 0:
        return np.empty(0, dtype=np.int64)
    if k * 2 > span:                       # most of the range: a permutation is cheaper than rejection
        c = lo + rng.permutation(span)
        if taken is not None:
            c = c[~np.isin(c, taken)]
        return c[:k]
    got = np.empty(0, dtype=np.int64)
    while got.size < k:
        c = rng.integers(lo, hi, size=int((k - got.size) * 1.25) + 16)
        if taken is not None:
            c = c[~np.isin(c, taken)]
        both = np.concatenate([got, c])
        _, first = np.unique(both, return_index=True)
        got = both[np.sort(first)]         # distinct, in the order drawn
    return got[:k]


def reddit_shard_like(rows=29121, cols=232965, avg_degree=492, communities=41, inside=0.8, first_row=0, seed=3):
    """One row shard of a reddit-like graph (BASELINE configs[3]: 232 965^2, nnz 114.6 M, cut into 8 row ranges):
    rows [first_row, first_row + rows) of a graph over `cols` vertices with power-law out-degrees (mean
    `avg_degree`) and `communities` planted communities (contiguous id ranges; `inside` of a row's edges stay in
    its own).  A row stores exactly its degree (distinct columns are drawn until there are enough), so the shard holds
    rows * avg_degree = 14.3 M entries: an eighth of reddit's 114.6 M."""
    rng = np.random.default_rng(seed)
    deg = _fit_counts(rng.pareto(1.5, size=rows) + 0.2, rows * avg_degree, cols // 4)
    size = -(-cols // communities)
    per_row = []
    for i in range(rows):
        k = int(deg[i])
        c0 = ((first_row + i) // size) * size
        c1 = min(c0 + size, cols)
        k_in = min(int(round(k * inside)), (c1 - c0) * 9 // 10)
        own = np.sort(_distinct(rng, c0, c1, k_in))
        other = _distinct(rng, 0, cols, k - own.size, taken=own)
        per_row.append(np.sort(np.concatenate([own, other])).astype(np.uint32))
    return _rows_to_csr(rows, cols, per_row)


def reddit_like_degrees(n=232965, avg_degree=492, seed=3):
    """Out-degree of every vertex of the reddit-like graph (BASELINE configs[3]: 232 965^2, nnz 114.6 M): power law,
    mean `avg_degree`, capped at n / 4.  A function of (n, avg_degree, seed) alone, so every rank of a sharded run
    computes the same sequence and can cut the rows by cost before any row exists."""
    rng = np.random.default_rng(seed)
    return _fit_counts_fast(rng.pareto(1.5, size=n) + 0.2, n * avg_degree, n // 4)


def reddit_like_rows(first_row, rows, n=232965, avg_degree=492, communities=41, inside=0.8, seed=3, degrees=None):
    """Rows [first_row, first_row + rows) of THE reddit-like graph over n vertices (reddit_like_degrees; `communities`
    planted communities in contiguous id ranges, `inside` of a row's edges in its own).  Rows are generated in
    globally aligned chunks of 1024 from a generator seeded by (seed, chunk), so any cut of the rows gives the same
    graph: rank r of a sharded run builds only its own range.  A row STORES exactly its degree (distinct columns are
    drawn until there are enough, a community can be filled to nine tenths), so the graph holds n * avg_degree entries:
    114.6 M at the default size, reddit's count."""
    deg = reddit_like_degrees(n, avg_degree, seed) if degrees is None else degrees
    size = -(-n // communities)
    per_row = []
    last = min(first_row + rows, n)
    for chunk in range(first_row // 1024, -(-last // 1024)):
        rng = np.random.default_rng([seed, chunk])
        for i in range(chunk * 1024, min((chunk + 1) * 1024, n)):
            k = int(deg[i])
            c0 = (i // size) * size
            c1 = min(c0 + size, n)
            k_in = min(int(round(k * inside)), (c1 - c0) * 9 // 10)
            own = np.sort(_distinct(rng, c0, c1, k_in))
            other = _distinct(rng, 0, n, k - own.size, taken=own)
            if first_row <= i < last:
                per_row.append(np.sort(np.concatenate([own, other])).astype(np.uint32))
    return _rows_to_csr(last - first_row, n, per_row)


def random_pattern(rows, cols, nnz, seed, empty_rows=0):
    """Small uniformly random pattern for unit tests (optionally with empty rows)."""
    rng = np.random.default_rng(seed)
    live = np.ones(rows, dtype=bool)
    if empty_rows:
        live[rng.choice(rows, size=empty_rows, replace=False)] = False
    cells = np.flatnonzero(np.repeat(live, cols))
    pick = np.sort(rng.choice(cells, size=min(nnz, cells.size), replace=False))
    r, c = pick // cols, pick % cols
    ro = np.zeros(rows + 1, dtype=np.uint32)
    np.add.at(ro, r + 1, 1)
    return rows, cols, np.cumsum(ro).astype(np.uint32), c.astype(np.uint32)


def write_mtx(path, rows, cols, ro, ci, shuffle_seed=None, values=None):
    """MatrixMarket coordinate file of the pattern (optionally in shuffled line order)."""
    r = np.repeat(np.arange(rows), np.diff(ro.astype(np.int64)))
    order = np.arange(ci.size)
    if shuffle_seed is not None:
        order = np.random.default_rng(shuffle_seed).permutation(ci.size)
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{rows} {cols} {ci.size}\n")
        for i in order:
            v = 1 if values is None else values[i]
            f.write(f"{r[i] + 1} {ci[i] + 1} {v}\n")


# ---------------------------------------------------------------------------
# SuiteSparse matrices whose pattern is defined by a formula.  The collection stores
# symmetric matrices as their lower triangle in column-major order, and the reference's
# loader does not expand `symmetric` files (src/Matrix.cpp:398-480), so what the
# pipeline sees - and what these return - is the lower triangle as a CSR with ascending
# columns.  The `NNZ` the reference logged for each of them equals these counts
# (tests/golden/reference_logs.json).
# ---------------------------------------------------------------------------
def _lower_csr(n: int, r: np.ndarray, c: np.ndarray):
    keep = r >= c
    key = np.unique(r[keep].astype(np.int64) * n + c[keep].astype(np.int64))
    r, c = key // n, key % n
    ro = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ro, r + 1, 1)
    return n, n, np.cumsum(ro).astype(np.uint32), c.astype(np.uint32)


def trefethen_pattern(n=20000):
    """JGD_Trefethen/Trefethen_<n>: primes on the diagonal, ones where |i - j| is a power
    of two.  Trefethen_20000b (the same without the first row and column) has the pattern
    of n = 19999."""
    r, c = [np.arange(n)], [np.arange(n)]
    d = 1
    while d < n:
        r.append(np.arange(d, n))
        c.append(np.arange(d, n) - d)
        d *= 2
    return _lower_csr(n, np.concatenate(r), np.concatenate(c))


def mycielskian_pattern(k=14):
    """Mycielski/mycielskian<k>: M2 is one edge; M(j+1) = [M M 0; M 0 1; 0 1' 0] on
    2n+1 vertices (copies n..2n-1, apex 2n)."""
    e = np.array([[1, 0]], dtype=np.int64)
    n = 2
    for _ in range(k - 2):
        a, b = e[:, 0], e[:, 1]
        apex = np.stack([np.full(n, 2 * n), np.arange(n, 2 * n)], 1)
        e = np.concatenate([e, np.stack([a + n, b], 1), np.stack([b + n, a], 1), apex])
        n = 2 * n + 1
    return _lower_csr(n, e[:, 0], e[:, 1])


def wathen_pattern(nx=100, ny=100):
    """gallery('wathen', nx, ny): every pair of the 8 nodes of a serendipity element is
    coupled; node numbering of Higham's wathen.m.  GHS_psdef/wathen100 is (100, 100),
    wathen120 is (nx, ny) = (100, 120) - the other orientation has the same size and nnz
    but not the statistics the reference logged."""
    n = 3 * nx * ny + 2 * nx + 2 * ny + 1
    j, i = np.meshgrid(np.arange(1, ny + 1), np.arange(1, nx + 1), indexing="ij")
    j, i = j.ravel().astype(np.int64), i.ravel().astype(np.int64)
    nn = np.zeros((8, i.size), dtype=np.int64)
    nn[0] = 3 * j * nx + 2 * i + 2 * j + 1
    nn[1] = nn[0] - 1
    nn[2] = nn[1] - 1
    nn[3] = (3 * j - 1) * nx + 2 * j + i - 1
    nn[4] = 3 * (j - 1) * nx + 2 * i + 2 * j - 3
    nn[5] = nn[4] + 1
    nn[6] = nn[5] + 1
    nn[7] = nn[3] + 1
    nn -= 1
    return _lower_csr(n, np.repeat(nn, 8, axis=0).ravel(), np.tile(nn, (8, 1)).ravel())


def write_mtx_columnwise(path, rows, cols, ro, ci, header="%%MatrixMarket matrix coordinate pattern symmetric"):
    """The file layout of the collection: entries column by column, rows ascending."""
    r = np.repeat(np.arange(rows), np.diff(ro.astype(np.int64)))
    order = np.lexsort((r, ci))
    with open(path, "w") as f:
        f.write(header + "\n")
        f.write(f"{rows} {cols} {ci.size}\n")
        np.savetxt(f, np.stack([r[order] + 1, ci[order].astype(np.int64) + 1], 1), fmt="%d")


def outlier_row_pattern(groups=40, shared=64, long_row=2000, cols=8000, seed=7):
    """`groups` clusters of 16 identical rows over `shared` columns spread across [0, cols); the first row of the first
    cluster also has `long_row` columns of its own between them.  In that row's panel a block's 16 dense columns lie ~2 000
    ids apart and the long row has ~500 residue entries between two of them: (block, row) pairs that span more than 255
    entries of P - the outliers csrc/plan_evict.hpp moves to the residue."""
    rng = np.random.default_rng(seed)
    per_row = []
    for g in range(groups):
        common = np.sort(rng.choice(cols, size=shared, replace=False))
        for r in range(16):
            mine = common
            if g == 0 and r == 0:
                extra = rng.choice(np.setdiff1d(np.arange(cols), common), size=long_row, replace=False)
                mine = np.sort(np.concatenate([common, extra]))
            per_row.append(mine.astype(np.uint32))
    ro = np.zeros(len(per_row) + 1, dtype=np.uint32)
    ro[1:] = np.cumsum([len(c) for c in per_row])
    return len(per_row), cols, ro, np.concatenate(per_row)
