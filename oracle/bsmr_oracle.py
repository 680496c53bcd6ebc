"""oracle/bsmr_oracle.py -- TEST INFRASTRUCTURE ONLY.

Plain numpy / pure-Python restatement of the *host* side of the BSMR-SDDMM hot
path (MatrixMarket load -> row clustering -> per-panel column reordering ->
dense/sparse split -> RPHM index arrays), written for small inputs only.  The
product's C++ pipeline (bsmr-sddmm_amd/src) is checked against it; nothing in
bsmr-sddmm_amd/ may import this file.

Pinning: the reference has no tests and its host code cannot be built in this image
without stand-ins, but it ships the logs of its own sweep over SuiteSparse
(scripts/results_suiteSparse_dataset/BSMR_results).  Six of those matrices have a
closed-form pattern; tests/test_reference_logs.py rebuilds them and checks every logged
integer against the C restatement of the clustering (oracle/clustering_oracle.c) and
the product pipeline, and tests/test_oracle.py checks this file against that C
restatement on small inputs.  Citations are relative to the reference checkout.
"""
from __future__ import annotations

import math

import numpy as np

ROW_PANEL_SIZE = 16   # include/BSMR.hpp:8
BLOCK_COL_SIZE = 16   # include/BSMR.hpp:9
BLOCK_SIZE = 256      # include/BSMR.hpp:10
NULL_VALUE = 0xFFFFFFFF  # include/TensorCoreConfig.cuh:11-12


# --------------------------------------------------------------------------- #
# a1: MatrixMarket loader, src/Matrix.cpp:398-480 (+ :373-396, :236-250)      #
# --------------------------------------------------------------------------- #
def _words(line: str):
    # include/util.hpp:182-197 -- a word ends at ' ', '\t' or '\r'; then ALL
    # following separators are skipped.  A leading separator yields an empty
    # first word (begin == end), exactly like the reference tokenizer.
    out = []
    i, n = 0, len(line)
    while i < n:
        b = i
        while i < n and line[i] not in " \t\r":
            i += 1
        out.append(line[b:i])
        while i < n and line[i] in " \t\r":
            i += 1
    return out


def load_mtx(path: str):
    """Returns (rows, cols, nnz, rowOffsets, colIndices, values) or None on the
    conditions for which the reference returns false."""
    with open(path, "r", newline="\n") as f:
        lines = f.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()  # getline() does not produce a trailing empty line
    it = iter(lines)
    header = None
    for line in it:  # :410 -- skip while line[0] == '%'
        if not line.startswith("%"):
            header = line
            break
    if header is None:
        return None
    w = _words(header)
    rows, cols = int(w[0]), int(w[1])
    nnz = int(float(w[2])) if len(w) > 2 and w[2] != "" else 0  # third goes through stod (:385)
    ri, ci, va = [], [], []
    for line in it:
        if line == "":
            continue  # getOneLineThreeData returns false on empty lines (:375)
        w = _words(line)
        if len(ri) >= nnz:
            return None  # :430-433 too many elements
        ri.append(int(w[0]) - 1)
        ci.append(int(w[1]) - 1)
        va.append(float(w[2]) if len(w) > 2 and w[2] != "" else 0.0)
    if len(ri) < nnz:
        return None  # :443-446
    seen = set()
    for r, c in zip(ri, ci):
        if r < 0 or c < 0 or r >= rows or c >= cols:
            return None  # :451-454 (unsigned wrap of 0-1 is >= rows too)
        if (r, c) in seen:
            return None  # :455-460
        seen.add((r, c))
    if nnz <= 1:
        return None  # :462-465
    ri = np.asarray(ri, dtype=np.int64)
    order = np.argsort(ri, kind="stable")  # :467-470 stable sort by row ONLY
    ri = ri[order]
    ci = np.asarray(ci, dtype=np.uint32)[order]
    va = np.asarray(va, dtype=np.float32)[order]
    ro = np.zeros(rows + 1, dtype=np.uint32)
    np.add.at(ro, ri + 1, 1)
    ro = np.cumsum(ro).astype(np.uint32)  # :236-250
    return rows, cols, nnz, ro, ci, va


# --------------------------------------------------------------------------- #
# a3-a5: row clustering, src/rowReordering.cu:1009-1095, 49-93, 235-293,       #
#        325-432, 893-1007                           #
# --------------------------------------------------------------------------- #
def calculate_block_size(rows: int, cols: int, free_mem_bytes: int) -> int:
    # :1009-1025; maxSharedMemoryPerBlock = 49152 (include/TensorCoreConfig.cuh:14)
    g = math.ceil(np.float32(rows * rows * 4) / np.float32(free_mem_bytes // 2))
    s = math.ceil(np.float32(cols * 4) / np.float32(49152 // 2))
    return max(16, int(g), int(s))


def encodings(rows, cols, ro, ci, bs):
    nb = int(math.ceil(np.float32(cols) / np.float32(bs)))  # :1035
    enc = np.zeros((rows, nb), dtype=np.int64)
    disp = np.zeros(rows, dtype=np.int64)
    for r in range(rows):
        cs = ci[ro[r]:ro[r + 1]].astype(np.int64)
        if cs.size == 0:
            continue  # :62-64 empty rows keep dispersion 0
        np.add.at(enc[r], cs // bs, 1)
        nzb = enc[r] > 0
        disp[r] = int(np.sum(bs - enc[r][nzb])) + cs.size * int(nzb.sum())  # :80-88
    return enc, disp


def cluster_threads(nb: int) -> int:
    # src/rowReordering.cu:912-922
    if nb < 32:
        return 32
    cand = 32 * int(math.ceil(np.float32(nb // 4) / np.float32(32)))
    return min(1024, max(32, cand))


def block_sum(partials: np.ndarray):
    """include/cudaUtil.cuh:14-45 as executed: per-warp balanced tree (shuffle-xor, lane 0),
    then `for (s = T/64; s >= 1; s >>= 1) v[w] += v[w+s]` - warps that loop never reads are
    simply not part of the result (T/32 not a power of two)."""
    v = partials.reshape(-1, 32).copy()
    step = 1
    while step < 32:
        v[:, ::2 * step] = v[:, ::2 * step] + v[:, step::2 * step]
        step *= 2
    w = v[:, 0].copy()
    s = partials.size // 64
    while s >= 1:
        w[:s] = w[:s] + w[s:2 * s]
        s >>= 1
    return w[0]


def per_thread(values: np.ndarray, threads: int) -> np.ndarray:
    """thread t accumulates bins t, t+T, t+2T, ... in that order"""
    pad = (-values.size) % threads
    v = np.concatenate([values, np.zeros(pad, dtype=values.dtype)]).reshape(-1, threads)
    acc = np.zeros(threads, dtype=values.dtype)
    for row in v:
        acc = (acc + row).astype(values.dtype)
    return acc


def similarity(rep, cmp_):
    # :235-293 with every sum taken the way the block takes it.
    threads = cluster_threads(rep.size)
    sx = int(block_sum(per_thread((rep * rep).astype(np.uint32), threads)))   # UIN, wraps
    sy = int(block_sum(per_thread((cmp_ * cmp_).astype(np.uint32), threads)))
    if sx == 0 and sy == 0:
        return np.float32(1.0)
    if sx == 0 or sy == 0:
        return np.float32(0.0)
    a = rep.astype(np.float32) / np.sqrt(np.float32(sx))
    c = cmp_.astype(np.float32) / np.sqrt(np.float32(sy))
    mn = block_sum(per_thread(np.minimum(a, c), threads))
    mx = block_sum(per_thread(np.maximum(a, c), threads))
    return np.float32(mn / mx)


def row_reordering(rows, cols, ro, ci, alpha, bs):
    """Returns (reorderedRows, numClusters)."""
    alpha = np.float32(alpha)
    enc, disp = encodings(rows, cols, ro, ci, bs)
    order = np.argsort(disp, kind="stable")  # :1060-1062
    cluster = np.full(rows, -1, dtype=np.int64)
    z = 0
    while z < rows and disp[order[z]] == 0:  # :939-949
        cluster[z] = 0
        z += 1
    cid = 1
    start = z
    while start < rows:
        # one bsa_clustering launch (:325-432), run to completion
        cluster[start] = cid
        rep = enc[order[start]].copy()
        nxt = -1
        for p in range(start + 1, rows):
            if cluster[p] != -1:
                continue
            if similarity(rep, enc[order[p]]) > alpha:  # strict, :388
                cluster[p] = cid
                rep += enc[order[p]]
            elif nxt < 0:
                nxt = p  # first reject seeds the next cluster (:400-424)
        if nxt < 0:
            break
        start = nxt
        cid += 1
    pos = np.argsort(cluster, kind="stable")  # :988-995
    perm = order[pos]
    # :985-992: sort_by_key sorts cluster_ids in place, then cluster_ids[indices[rows-1]]
    # reads the SORTED ids at the unsorted position of the last element
    num_clusters = int(cluster[pos][pos[-1]]) + (1 if z != 0 else 0) if rows else 0
    # :1082-1090 drop leading empty rows
    k = 0
    while k < rows and ro[perm[k] + 1] - ro[perm[k]] == 0:
        k += 1
    return perm[k:].astype(np.uint32), num_clusters


def no_reorder_rows(rows, ro):
    # src/rowReordering.cu:15-46: identity order over the non-empty rows
    return np.asarray([r for r in range(rows) if ro[r + 1] > ro[r]], dtype=np.uint32)


# --------------------------------------------------------------------------- #
# a6: column reordering + dense/sparse split, src/colReordering.cu:244-404    #
# --------------------------------------------------------------------------- #
def dense_threshold(delta) -> int:
    # :246  ceil(float(delta) * 256) evaluated in fp32
    return int(math.ceil(np.float32(np.float32(delta) * np.float32(BLOCK_SIZE))))


def col_reordering(rows, cols, ro, ci, reordered_rows, delta):
    nrr = len(reordered_rows)
    num_panels = int(math.ceil(np.float32(nrr) / ROW_PANEL_SIZE))  # src/BSMR.cpp:48
    thr = dense_threshold(delta)
    dense_cols, sparse_cols = [], []
    dco, sco, svo = [0], [0], [0]
    for p in range(num_panels):
        cnt = {}
        for rr in reordered_rows[p * 16:min(p * 16 + 16, nrr)]:
            for c in ci[ro[rr]:ro[rr + 1]]:
                cnt[int(c)] = cnt.get(int(c), 0) + 1
        cs = sorted(cnt)                                   # ascending id (:318-331)
        cs.sort(key=lambda c: -cnt[c])                     # stable, count desc (:333-336)
        ns = [cnt[c] for c in cs]
        if len(cs) % 16:                                   # :338-343
            pad = 16 - len(cs) % 16
            cs += [cols] * pad
            ns += [0] * pad
        nd = 0
        for b in range(0, len(cs), 16):                    # :250-261
            if sum(ns[b:b + 16]) >= thr:
                nd += 16
        dense_cols += cs[:nd]
        sparse_cols += cs[nd:]
        dco.append(dco[-1] + nd)
        sco.append(sco[-1] + len(cs) - nd)
        svo.append(svo[-1] + sum(ns[nd:]))
    u = lambda x: np.asarray(x, dtype=np.uint32)
    return dict(numRowPanels=num_panels, denseCols=u(dense_cols), denseColOffsets=u(dco),
                sparseCols=u(sparse_cols), sparseColOffsets=u(sco), sparseValueOffsets=u(svo))


# --------------------------------------------------------------------------- #
# a7: RPHM, src/BSMR.cpp:83-265                                               #
# --------------------------------------------------------------------------- #
def rphm(rows, cols, ro, ci, reordered_rows, cr):
    P = cr["numRowPanels"]
    dco, sco, svo = cr["denseColOffsets"], cr["sparseColOffsets"], cr["sparseValueOffsets"]
    bo = [0]
    for p in range(P):
        bo.append(bo[-1] + int(math.ceil((int(dco[p + 1]) - int(dco[p])) / 16)))
    bv = np.full(bo[-1] * BLOCK_SIZE, NULL_VALUE, dtype=np.uint32)
    sv = np.zeros(int(svo[-1]), dtype=np.uint32)
    srr = np.zeros(int(svo[-1]), dtype=np.uint32)
    sci = np.zeros(int(svo[-1]), dtype=np.uint32)
    nrr = len(reordered_rows)
    for p in range(P):
        rws = reordered_rows[p * 16:min(p * 16 + 16, nrr)]
        maps = [{int(ci[e]): e for e in range(ro[r], ro[r + 1])} for r in rws]
        for t, col in enumerate(cr["denseCols"][dco[p]:dco[p + 1]]):     # :146-174
            for lr, m in enumerate(maps):
                e = m.get(int(col))
                if e is not None:
                    bv[(bo[p] + t // 16) * 256 + lr * 16 + t % 16] = e
        k = int(svo[p])
        for col in cr["sparseCols"][sco[p]:sco[p + 1]]:                  # :178-219
            for lr, m in enumerate(maps):
                e = m.get(int(col))
                if e is not None:
                    srr[k], sci[k], sv[k] = lr, col, e
                    k += 1
        assert k == int(svo[p + 1])
    return dict(blockOffsets=np.asarray(bo, dtype=np.uint32), blockValues=bv,
                sparseValues=sv, sparseRelativeRows=srr, sparseColIndices=sci)


def check_rphm_invariants(rows, cols, ro, ci, reordered_rows, cr, rp):
    """Restates check_rphm (src/BSMR.cpp:444-824) as assertions."""
    nnz = int(ro[-1])
    nonempty = [r for r in range(rows) if ro[r + 1] > ro[r]]
    assert sorted(int(r) for r in reordered_rows) == nonempty
    bv = rp["blockValues"]
    used = np.concatenate([bv[bv != NULL_VALUE], rp["sparseValues"]]).astype(np.int64)
    assert used.size == nnz and np.array_equal(np.sort(used), np.arange(nnz))
    return True
