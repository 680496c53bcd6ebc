"""Host-side BSMR pipeline (C++/OpenMP, bsmr-sddmm_amd/src) against the numpy oracle
(oracle/bsmr_oracle.py), hand-derived known answers and the committed fixtures.
No GPU needed: pipelines are built with device=-1."""
import json
from pathlib import Path

import numpy as np
import pytest

import bsmr_oracle as bo
import synth

GOLDEN = Path(__file__).parent / "golden"


# ------------------------------------------------------------------ loaders
MTX_CASES = {
    "general_sorted": "%%MatrixMarket matrix coordinate real general\n% comment\n3 4 4\n1 1 1.5\n1 3 2\n2 2 -1e3\n3 4 7\n",
    "unsorted_rows_keep_file_order": "%%MatrixMarket matrix coordinate real general\n3 5 6\n3 5 1\n1 4 2\n1 2 3\n2 1 4\n1 5 5\n3 1 6\n",
    "symmetric_banner_is_ignored": "%%MatrixMarket matrix coordinate real symmetric\n4 4 3\n2 1 1\n3 1 2\n4 4 3\n",
    "pattern_no_values": "%%MatrixMarket matrix coordinate pattern general\n3 3 3\n1 2\n2 3\n3 1\n",
    "blank_lines_and_tabs": "%c\n2 3 3\n\n1\t1\t5\n\n2 3  6\n1 2 7\n\n",
    "crlf": "%%MatrixMarket matrix coordinate real general\r\n2 2 2\r\n1 1 1\r\n2 2 2\r\n",
    "no_trailing_newline": "2 2 2\n1 2 3\n2 1 4",
}
MTX_ERRORS = {
    "duplicate": "2 2 3\n1 1 1\n2 2 1\n1 1 2\n",
    "too_many": "2 2 2\n1 1 1\n2 2 1\n1 2 1\n",
    "too_few": "2 2 3\n1 1 1\n2 2 1\n",
    "row_out_of_range": "2 2 2\n3 1 1\n1 1 1\n",
    "col_out_of_range": "2 2 2\n1 3 1\n1 1 1\n",
    "zero_based_index": "2 2 2\n0 1 1\n1 1 1\n",
    "single_entry": "2 2 1\n1 1 1\n",
}


@pytest.mark.parametrize("name", sorted(MTX_CASES))
def test_mtx_loader_matches_oracle(engine, tmp_path, name):
    f = tmp_path / f"{name}.mtx"
    f.write_bytes(MTX_CASES[name].encode())
    want = bo.load_mtx(str(f))
    assert want is not None
    rows, cols, nnz, ro, ci, va = want
    csr = engine.CSR.from_file(f)
    assert (csr.rows, csr.cols, csr.nnz) == (rows, cols, nnz)
    assert np.array_equal(csr.row_offsets, ro)
    assert np.array_equal(csr.col_indices, ci)
    assert np.array_equal(csr.values, va)
    assert csr.check()


def test_mtx_row_stable_order_known_answer(engine, tmp_path):
    # stable sort by row ONLY: inside a row the file order survives (src/Matrix.cpp:467-470)
    f = tmp_path / "o.mtx"
    f.write_text(MTX_CASES["unsorted_rows_keep_file_order"])
    csr = engine.CSR.from_file(f)
    assert csr.row_offsets.tolist() == [0, 3, 4, 6]
    assert csr.col_indices.tolist() == [3, 1, 4, 0, 4, 0]
    assert csr.values.tolist() == [2, 3, 5, 4, 1, 6]


@pytest.mark.parametrize("name", sorted(MTX_ERRORS))
def test_mtx_loader_rejects(engine, tmp_path, name):
    f = tmp_path / f"{name}.mtx"
    f.write_text(MTX_ERRORS[name])
    assert bo.load_mtx(str(f)) is None
    with pytest.raises(ValueError):
        engine.CSR.from_file(f)


def test_unknown_suffix_and_missing_file(engine, tmp_path):
    (tmp_path / "a.dat").write_text("2 2 2\n1 1 1\n2 2 1\n")
    with pytest.raises(ValueError):
        engine.CSR.from_file(tmp_path / "a.dat")
    with pytest.raises(ValueError):
        engine.CSR.from_file(tmp_path / "missing.mtx")


def test_smtx_loader(engine, tmp_path):
    # DLMC: "rows, cols, nnz" then row offsets then column indices; values become 1
    f = tmp_path / "m.smtx"
    f.write_text("3, 5, 4\n0 2 2 4\n1 4 0 3\n")
    csr = engine.CSR.from_file(f)
    assert (csr.rows, csr.cols, csr.nnz) == (3, 5, 4)
    assert csr.row_offsets.tolist() == [0, 2, 2, 4]
    assert csr.col_indices.tolist() == [1, 4, 0, 3]
    assert csr.values.tolist() == [1, 1, 1, 1]
    g = tmp_path / "dup.smtx"
    g.write_text("2, 3, 3\n0 2 3\n1 1 2\n")
    with pytest.raises(ValueError):
        engine.CSR.from_file(g)


def test_snap_edge_list_loader(engine, tmp_path):
    f = tmp_path / "g.txt"
    f.write_text("# Directed graph\n# Nodes: 4 Edges: 5\n# FromNodeId\tToNodeId\n10\t20\n10\t30\n20\t10\n40\t30\n30\t40\n")
    csr = engine.CSR.from_file(f)
    # ids are renumbered in order of first appearance: 10->0, 20->1, 30->2, 40->3
    assert (csr.rows, csr.cols, csr.nnz) == (4, 4, 5)
    assert csr.row_offsets.tolist() == [0, 2, 3, 4, 5]
    assert csr.col_indices.tolist() == [1, 2, 0, 3, 2]


@pytest.mark.parametrize("compressed", [False, True])
@pytest.mark.parametrize("index_dtype", [np.int32, np.int64])
def test_npz_graph_loader(engine, tmp_path, compressed, index_dtype):
    """Graph archives as the reference's scripts/convert_mtx_to_npz.py:9-41 writes them (np.savez: src_li, dst_li,
    num_nodes_src, num_nodes_dst, num_edges); also deflated members and 64-bit indices.  Rows keep file order."""
    rows, cols, ro, ci = synth.random_pattern(37, 53, 400, seed=12, empty_rows=3)
    src = np.repeat(np.arange(rows), np.diff(ro)).astype(index_dtype)
    dst = ci.astype(index_dtype)
    perm = np.random.default_rng(5).permutation(src.size)          # edges in arbitrary order, as from a COO
    f = tmp_path / "g.npz"
    save = np.savez_compressed if compressed else np.savez
    save(f, src_li=src[perm], dst_li=dst[perm], num_nodes_src=rows, num_nodes_dst=cols, num_edges=src.size)
    csr = engine.CSR.from_file(f)
    assert (csr.rows, csr.cols, csr.nnz) == (rows, cols, ci.size)
    assert np.array_equal(csr.row_offsets, ro)
    order = np.argsort(src[perm], kind="stable")                   # stable by row = the loader's contract
    assert np.array_equal(csr.col_indices, dst[perm][order].astype(np.uint32))
    assert not csr.values.any()


def test_npz_graph_loader_rejects(engine, tmp_path):
    ok = dict(src_li=np.array([0, 1, 1], np.int32), dst_li=np.array([1, 0, 2], np.int32), num_nodes_src=2,
              num_nodes_dst=3, num_edges=3)
    np.savez(tmp_path / "ok.npz", **ok)
    assert engine.CSR.from_file(tmp_path / "ok.npz").nnz == 3
    bad = {
        "missing": {k: v for k, v in ok.items() if k != "dst_li"},
        "range": dict(ok, src_li=np.array([0, 1, 2], np.int32)),
        "negative": dict(ok, dst_li=np.array([1, -1, 2], np.int32)),
        "count": dict(ok, num_edges=4),
        "duplicate": dict(ok, src_li=np.array([0, 1, 1], np.int32), dst_li=np.array([1, 2, 2], np.int32)),
    }
    for name, arrays in bad.items():
        np.savez(tmp_path / f"{name}.npz", **arrays)
        with pytest.raises(ValueError):
            engine.CSR.from_file(tmp_path / f"{name}.npz")
    (tmp_path / "junk.npz").write_bytes(b"PK\x05\x06" + bytes(30))
    with pytest.raises(ValueError):
        engine.CSR.from_file(tmp_path / "junk.npz")
    whole = (tmp_path / "ok.npz").read_bytes()
    (tmp_path / "cut.npz").write_bytes(whole[: len(whole) // 2])
    with pytest.raises(ValueError):
        engine.CSR.from_file(tmp_path / "cut.npz")


def test_write_then_read_mtx_roundtrip(engine, tmp_path):
    rows, cols, ro, ci = synth.random_pattern(23, 31, 150, seed=8)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    out = tmp_path / "rt.mtx"
    assert csr.write_mtx(out)
    back = engine.CSR.from_file(out)
    assert np.array_equal(back.row_offsets, ro) and np.array_equal(back.col_indices, ci)


def test_shuffled_file_loads_to_same_rows(engine, tmp_path):
    rows, cols, ro, ci = synth.random_pattern(19, 40, 200, seed=9, empty_rows=2)
    f = tmp_path / "s.mtx"
    synth.write_mtx(f, rows, cols, ro, ci, shuffle_seed=3)
    csr = engine.CSR.from_file(f)
    want = bo.load_mtx(str(f))
    assert np.array_equal(csr.row_offsets, ro)
    assert np.array_equal(csr.col_indices, want[4])
    for r in range(rows):  # same set per row, file order inside
        assert sorted(csr.col_indices[ro[r]:ro[r + 1]]) == sorted(ci[ro[r]:ro[r + 1]])


# ------------------------------------------------------------------ operands
def test_make_data_is_mt19937_sequence(engine):
    x = engine.make_data(6, 5489)
    # std::mt19937 default-seeded first outputs (C++ standard: 10000th is 4123659995)
    first = [3499211612, 581869302, 3890346734, 3586334585, 545404204, 4161255391]
    want = np.array([2.0 * (u >> 8) / 16777216.0 for u in first], dtype=np.float32)
    assert np.array_equal(x, want)
    assert (engine.make_data(4096, 1) < 2).all() and (engine.make_data(4096, 1) >= 0).all()
    assert np.array_equal(engine.make_data(100, 7), engine.make_data(100, 7))


# ------------------------------------------------------------------ pipeline
def test_block_size_formula(engine):
    for rows, cols in ((1500, 12419), (121192, 121192), (232965, 232965), (40, 64)):
        ro = np.zeros(rows + 1, dtype=np.uint32)
        csr = engine.CSR.from_arrays(rows, cols, ro, np.zeros(0, np.uint32))
        for free in (24 << 30, 288 << 30):
            assert csr.calculate_block_size(free) == bo.calculate_block_size(rows, cols, free)
    # nips -> 16, cop20k -> 20, reddit -> 38 (LDS term)
    ro = np.zeros(2, dtype=np.uint32)
    assert engine.CSR.from_arrays(1, 12419, ro, np.zeros(0, np.uint32)).calculate_block_size(288 << 30) == 16
    assert engine.CSR.from_arrays(1, 121192, ro, np.zeros(0, np.uint32)).calculate_block_size(288 << 30) == 20
    assert engine.CSR.from_arrays(1, 232965, ro, np.zeros(0, np.uint32)).calculate_block_size(288 << 30) == 38


def _compare_with_oracle(engine, rows, cols, ro, ci, alpha, delta, bw):
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    pipe = engine.Pipeline(csr, alpha=alpha, delta=delta, block_size=bw, device=-1)
    got = pipe.arrays()
    rr, nc = bo.row_reordering(rows, cols, ro, ci, alpha, bw)
    assert np.array_equal(got["reorderedRows"], rr)
    cr = bo.col_reordering(rows, cols, ro, ci, rr, delta)
    assert pipe.num_row_panels == cr["numRowPanels"]
    for k in ("denseCols", "denseColOffsets", "sparseCols", "sparseColOffsets", "sparseValueOffsets"):
        assert np.array_equal(got[k], cr[k]), k
    rp = bo.rphm(rows, cols, ro, ci, rr, cr)
    for k in rp:
        assert np.array_equal(got[k], rp[k]), k
    assert pipe.check()
    assert bo.check_rphm_invariants(rows, cols, ro, ci, got["reorderedRows"], cr, rp)
    return pipe, nc


@pytest.mark.parametrize("alpha", [0.1, 0.3, 0.5, 0.9])
@pytest.mark.parametrize("delta", [0.0, 0.1, 0.3, 1.1])
def test_pipeline_matches_oracle(engine, alpha, delta):
    rows, cols, ro, ci = synth.random_pattern(90, 130, 2600, seed=int(alpha * 10) + 17, empty_rows=5)
    _compare_with_oracle(engine, rows, cols, ro, ci, alpha, delta, 16)


@pytest.mark.parametrize("seed", range(6))
def test_pipeline_matches_oracle_structured(engine, seed):
    # community structure makes clusters of several rows (exercises the merge path)
    rows, cols, ro, ci = synth.community_graph(n=160, avg_degree=12, communities=5, seed=seed)
    pipe, nc = _compare_with_oracle(engine, rows, cols, ro, ci, 0.2, 0.05, 16)
    assert pipe.num_clusters == nc


def test_num_clusters_and_auto_block_size(engine):
    rows, cols, ro, ci = synth.random_pattern(60, 300, 900, seed=3, empty_rows=4)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    pipe = engine.Pipeline(csr, alpha=0.3, delta=0.3, device=-1)  # BSMR(alpha, delta, S)
    bw = bo.calculate_block_size(rows, cols, 288 << 30)
    rr, nc = bo.row_reordering(rows, cols, ro, ci, 0.3, bw)
    assert np.array_equal(pipe.array("reorderedRows"), rr)
    assert pipe.num_clusters == nc


def test_golden_fixtures(engine):
    files = sorted(GOLDEN.glob("pipeline_*.json"))
    assert files
    for f in files:
        g = json.loads(f.read_text())
        ro = np.asarray(g["rowOffsets"], np.uint32)
        ci = np.asarray(g["colIndices"], np.uint32)
        csr = engine.CSR.from_arrays(g["rows"], g["cols"], ro, ci)
        pipe = engine.Pipeline(csr, alpha=g["alpha"], delta=g["delta"], block_size=g["binWidth"], device=-1)
        got = pipe.arrays()
        for k in ("reorderedRows", "denseCols", "denseColOffsets", "sparseCols", "sparseColOffsets",
                  "sparseValueOffsets", "blockOffsets", "blockValues", "sparseValues",
                  "sparseRelativeRows", "sparseColIndices"):
            assert got[k].tolist() == g[k], (f.name, k)


def test_hand_derived_split_and_rphm(engine):
    """3 x 20 matrix, one panel, worked by hand from SURVEY.md appendix A.4/A.5."""
    rows, cols = 3, 20
    ro = np.array([0, 3, 6, 8], np.uint32)
    ci = np.array([0, 1, 2, 1, 2, 5, 2, 7], np.uint32)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    # column counts: c2:3 c1:2 c0:1 c5:1 c7:1 -> order [2,1,0,5,7], block sum 8
    order = [2, 1, 0, 5, 7] + [20] * 11
    dense = engine.Pipeline(csr, delta=0.03, row_mode=engine.ROWS_IDENTITY, device=-1)  # ceil(7.68) = 8 <= 8
    a = dense.arrays()
    assert a["reorderedRows"].tolist() == [0, 1, 2]
    assert a["denseCols"].tolist() == order and a["sparseCols"].size == 0
    tile = np.full(256, 0xFFFFFFFF, np.uint32)
    for (lr, slot), e in {(0, 0): 2, (0, 1): 1, (0, 2): 0, (1, 0): 4, (1, 1): 3, (1, 3): 5, (2, 0): 6, (2, 4): 7}.items():
        tile[lr * 16 + slot] = e
    assert np.array_equal(a["blockValues"], tile) and a["blockOffsets"].tolist() == [0, 1]
    sparse = engine.Pipeline(csr, delta=0.04, row_mode=engine.ROWS_IDENTITY, device=-1)  # ceil(10.24) = 11 > 8
    a = sparse.arrays()
    assert a["denseCols"].size == 0 and a["sparseCols"].tolist() == order
    assert a["sparseValueOffsets"].tolist() == [0, 8]
    assert a["sparseValues"].tolist() == [2, 4, 6, 1, 3, 0, 5, 7]
    assert a["sparseRelativeRows"].tolist() == [0, 1, 2, 0, 1, 0, 1, 2]
    assert a["sparseColIndices"].tolist() == [2, 2, 2, 1, 1, 0, 5, 7]
    assert a["sparseRowPanelIds"].tolist() == [0] and a["sparseColBlockIters"].tolist() == [0]


def test_hand_derived_clustering(engine):
    """Two groups of identical rows + one empty row; bins of 16 columns.
    rows 0,2,4 touch columns {0,1}; rows 1,3 touch columns {40,41,42}; row 5 is empty.
    dispersion: (16-2)+2*1 = 16 for the first kind, (16-3)+3*1 = 16 for the second: all tie,
    so the ascending order is row order; cluster 1 = {0,2,4} (similarity 1 > alpha),
    cluster 2 = {1,3}; the empty row is dropped."""
    rows, cols = 6, 64
    ro = np.array([0, 2, 5, 7, 10, 12, 12], np.uint32)
    ci = np.array([0, 1, 40, 41, 42, 0, 1, 40, 41, 42, 0, 1], np.uint32)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    pipe = engine.Pipeline(csr, alpha=0.3, delta=0.3, block_size=16, device=-1)
    assert pipe.array("reorderedRows").tolist() == [0, 2, 4, 1, 3]
    assert pipe.num_row_panels == 1
    rr, nc = bo.row_reordering(rows, cols, ro, ci, 0.3, 16)
    assert rr.tolist() == [0, 2, 4, 1, 3] and nc == 3  # two clusters + the empty-row cluster


def test_resplit_changes_only_the_split(engine):
    rows, cols, ro, ci = synth.random_pattern(70, 90, 2000, seed=12)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    pipe = engine.Pipeline(csr, alpha=0.3, delta=0.0, device=-1)
    rows_before = pipe.array("reorderedRows")
    assert pipe.array("sparseValues").size == 0           # delta = 0: everything dense
    pipe.resplit(1.1)
    assert np.array_equal(pipe.array("reorderedRows"), rows_before)
    assert pipe.array("blockValues").size == 0            # delta > 1: everything sparse
    assert pipe.array("sparseValues").size == csr.nnz and pipe.check()


def test_host_sddmm_cpu_matches_oracle(engine, oracle):
    rows, cols, ro, ci = synth.random_pattern(64, 80, 1000, seed=14, empty_rows=3)
    K = 96
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    A = engine.make_data(rows * K, 11)
    B = engine.make_data(cols * K, 12)
    got = engine.sddmm_cpu(csr, K, A, B)
    want = oracle.sddmm_cpu(rows, cols, K, ro, ci, A, B)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert engine.check_data(got, want) == 0
    bad = want.copy()
    bad[::7] *= 1.01
    assert engine.check_data(got, bad) == oracle.check_data(got, bad)[0] > 0


def test_nips_like_shape(engine):
    rows, cols, ro, ci = synth.nips_like()
    assert (rows, cols, ci.size) == (1500, 12419, 746316)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    assert csr.check()
    pipe = engine.Pipeline(csr, alpha=0.3, delta=0.3, device=-1)
    assert pipe.check() and pipe.num_row_panels == 94
