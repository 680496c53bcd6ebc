"""The host side of bsmr_plan_create without a GPU: residue promotion (csrc/plan_promote.hpp) and packing
(csrc/plan_pack.hpp) on RPHM arrays of the host pipeline, through tests/native/plancheck.hip, which checks that
every stored entry still has exactly one place - a cell of a block at its own row and column, or the residue."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

import synth

REPO = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def plancheck(engine):
    lib = C.CDLL(str(REPO / "tests" / "native" / "libplancheck.so"))
    lib.plancheck_promote.restype = C.c_int
    lib.plancheck_promote.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]

    def run(rows, cols, ro, ci, alpha, delta, min_average=16, min_entries=1_000_000, small_dense=32768, column_degree=32, head=0):
        csr = engine.CSR.from_arrays(rows, cols, ro, ci)
        pipe = engine.Pipeline(csr, alpha=alpha, delta=delta, device=-1)
        arrays = pipe.arrays()
        keep = {k: np.ascontiguousarray(arrays[k], dtype=np.uint32) for k in
                ("reorderedRows", "denseCols", "blockOffsets", "blockValues", "sparseValueOffsets",
                 "sparseValues", "sparseRelativeRows", "sparseColIndices")}
        d = engine.RphmDesc()
        d.M, d.N, d.nnz = rows, cols, csr.nnz
        d.num_nonzero_rows = keep["reorderedRows"].size
        d.num_row_panels = keep["blockOffsets"].size - 1
        cast = lambda a: a.ctypes.data_as(engine.u32p)
        d.reordered_rows, d.dense_cols = cast(keep["reorderedRows"]), cast(keep["denseCols"])
        d.block_offsets, d.block_values = cast(keep["blockOffsets"]), cast(keep["blockValues"])
        d.sparse_value_offsets, d.sparse_values = cast(keep["sparseValueOffsets"]), cast(keep["sparseValues"])
        d.sparse_relative_rows, d.sparse_col_indices = cast(keep["sparseRelativeRows"]), cast(keep["sparseColIndices"])
        out = (C.c_uint64 * 13)()
        rc = lib.plancheck_promote(C.byref(d), min_average, min_entries, small_dense, column_degree, head, out)
        names = ("promoted", "promoted_entries", "promoted_blocks", "blocks", "residue", "pack_status", "packed_dense",
                 "packed_residue", "promote_us", "pack_us", "union_columns", "union_columns_grouped4", "tile_bytes")
        res = dict(zip(names, (int(v) for v in out)))
        res["rphm_dense"] = int(csr.nnz - keep["sparseValues"].size)
        res["rphm_blocks"] = int(keep["blockOffsets"][-1])
        res["nnz"] = csr.nnz
        return rc, res
    return run


@pytest.fixture(scope="module")
def tilecheck(engine):
    lib = C.CDLL(str(REPO / "tests" / "native" / "libplancheck.so"))
    lib.plancheck_tiles.restype = C.c_int
    lib.plancheck_tiles.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]

    def run(rows, cols, ro, ci, alpha, delta, H, blocks_per_item):
        csr = engine.CSR.from_arrays(rows, cols, ro, ci)
        pipe = engine.Pipeline(csr, alpha=alpha, delta=delta, device=-1)
        arrays = pipe.arrays()
        keep = {k: np.ascontiguousarray(arrays[k], dtype=np.uint32) for k in
                ("reorderedRows", "denseCols", "blockOffsets", "blockValues", "sparseValueOffsets",
                 "sparseValues", "sparseRelativeRows", "sparseColIndices")}
        d = engine.RphmDesc()
        d.M, d.N, d.nnz = rows, cols, csr.nnz
        d.num_nonzero_rows = keep["reorderedRows"].size
        d.num_row_panels = keep["blockOffsets"].size - 1
        cast = lambda a: a.ctypes.data_as(engine.u32p)
        d.reordered_rows, d.dense_cols = cast(keep["reorderedRows"]), cast(keep["denseCols"])
        d.block_offsets, d.block_values = cast(keep["blockOffsets"]), cast(keep["blockValues"])
        d.sparse_value_offsets, d.sparse_values = cast(keep["sparseValueOffsets"]), cast(keep["sparseValues"])
        d.sparse_relative_rows, d.sparse_col_indices = cast(keep["sparseRelativeRows"]), cast(keep["sparseColIndices"])
        out = (C.c_uint64 * 11)()
        rc = lib.plancheck_tiles(C.byref(d), H, blocks_per_item, out)
        names = ("blocks", "tiles", "union_columns", "entries", "items", "entry_cap", "bytes", "census_blocks",
                 "census_tiles", "census_columns")
        res = dict(zip(names, (int(v) for v in out)))
        res["rphm_dense"] = int(csr.nnz - keep["sparseValues"].size)
        return rc, res
    return run


@pytest.mark.parametrize("H", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("blocks_per_item", [1, 5, 32])
def test_tiles_format_lists_every_dense_entry_once(tilecheck, H, blocks_per_item):
    """csrc/tile_format.hpp: the packed format, read the way the kernel reads it, reproduces the RPHM's dense part."""
    rows, cols, ro, ci = synth.nips_like(rows=330, cols=1500, nnz=42000, seed=3)   # 21 panels: a ragged last group
    for delta in (0.0, 0.1):
        rc, r = tilecheck(rows, cols, ro, ci, 0.3, delta, H, blocks_per_item)
        assert rc == 0, f"invariant {rc} violated: {r}"
        assert r["entries"] == r["rphm_dense"]
        assert r["blocks"] * 16 >= r["union_columns"] > (r["blocks"] - r["items"]) * 16 - 16 * r["items"]
    # grouping never gathers more columns, and never fewer than one panel's worth
    sizes = [tilecheck(rows, cols, ro, ci, 0.3, 0.0, h, 8)[1]["union_columns"] for h in (1, 2, 4, 8)]
    assert sizes == sorted(sizes, reverse=True) and sizes[3] * 8 >= sizes[0]


def test_tiles_format_cuts_blocks_at_the_entry_cap(tilecheck):
    """A block may hold at most 128 entries per panel of the group (the LDS room of the kernel): fuller blocks are
    cut at a column boundary."""
    rows, cols = 256, 64
    ro = np.arange(rows + 1, dtype=np.uint32) * cols
    ci = np.tile(np.arange(cols, dtype=np.uint32), rows)          # a full matrix: 256 entries per 16x16 tile
    for H in (1, 4, 8, 16):
        rc, r = tilecheck(rows, cols, ro, ci, 0.3, 0.0, H, 4)
        assert rc == 0, (rc, r)
        assert r["entries"] == rows * cols and r["entry_cap"] <= 128 * H + 16 * H + 255
        assert r["blocks"] > r["census_blocks"]                   # blocks of 8 columns instead of 16


@pytest.mark.parametrize("min_average,head", [(1, 0), (8, 0), (20, 0), (40, 0), (40, 12), (200, 6)])
@pytest.mark.parametrize("delta", [0.05, 0.3, 1.1])
def test_promotion_keeps_every_entry_exactly_once(plancheck, min_average, head, delta):
    rows, cols, ro, ci = synth.community_graph(n=900, avg_degree=70, communities=6, seed=5)
    rc, r = plancheck(rows, cols, ro, ci, 0.2, delta, min_average=min_average, min_entries=1000, small_dense=100,
                      column_degree=0, head=head)
    assert rc == 0, f"invariant {rc} violated: {r}"
    assert r["pack_status"] == 0
    assert r["packed_dense"] + r["packed_residue"] == r["nnz"]
    assert r["packed_dense"] == r["rphm_dense"] + r["promoted_entries"]
    assert r["blocks"] == r["rphm_blocks"] + r["promoted_blocks"]
    if min_average == 1 and (r["rphm_dense"] >= 100 or r["nnz"] - r["rphm_dense"] >= 1000):
        assert r["promoted"] == 1 and r["residue"] == 0          # every panel qualifies


def test_promotion_rules(plancheck):
    """Which plans change: a hybrid plan gives well-filled panels to the dense path and the rest too when little is
    left; a plan without a dense part only when at least `min_entries` entries move; threshold 0 switches it off."""
    rows, cols, ro, ci = synth.mycielskian_pattern(11)            # 1535 rows, hub columns shared by all panels
    rc, r = plancheck(rows, cols, ro, ci, 0.3, 0.3, small_dense=10_000)
    assert rc == 0 and r["rphm_dense"] > 10_000 and r["promoted"] == 1
    assert r["residue"] == 0 and r["packed_dense"] == r["nnz"]    # the whole residue follows
    rc, off = plancheck(rows, cols, ro, ci, 0.3, 0.3, min_average=0)
    assert rc == 0 and off["promoted"] == 0 and off["packed_dense"] == off["rphm_dense"]
    # all-sparse plans: below the entry threshold nothing moves, above it everything that qualifies does
    rc, r = plancheck(rows, cols, ro, ci, 0.3, 1.1)
    assert rc == 0 and r["rphm_dense"] == 0 and r["promoted"] == 0 and r["packed_residue"] == r["nnz"]
    rc, r = plancheck(rows, cols, ro, ci, 0.3, 1.1, min_entries=50_000)
    assert rc == 0 and r["promoted"] == 1 and r["promoted_entries"] >= 50_000
    # mesh-like patterns (11 entries per column of S: no reuse of a gathered column) stay in the residue
    rows, cols, ro, ci = synth.banded_mesh_like(n=20000, nnz=220000, seed=7)
    rc, r = plancheck(rows, cols, ro, ci, 0.3, 0.3, min_entries=1000)
    assert rc == 0 and r["promoted"] == 0
    rc, r = plancheck(rows, cols, ro, ci, 0.3, 0.3, min_entries=1000, column_degree=0)
    assert rc == 0 and r["promoted"] == 1                        # (their panels do fill 16 per block)


def test_promotion_with_repeated_entries(plancheck):
    """A CSR row may hold the same column twice: a block cell takes one copy, the other stays in the residue."""
    rows, cols = 64, 48
    rng = np.random.default_rng(3)
    ci, ro = [], [0]
    for _ in range(rows):
        c = np.sort(rng.choice(cols, size=20, replace=False))
        c = np.sort(np.concatenate([c, c[:3]]))                   # three repeated columns per row
        ci.extend(c.tolist())
        ro.append(len(ci))
    rc, r = plancheck(rows, cols, np.array(ro, dtype=np.uint32), np.array(ci, dtype=np.uint32), 0.3, 1.1,
                      min_average=1, min_entries=10, small_dense=10, column_degree=0)
    assert rc == 0, r
    assert r["promoted"] == 1 and r["packed_dense"] + r["packed_residue"] == r["nnz"]
    assert r["residue"] == 3 * rows                               # the second copies


@pytest.fixture(scope="module")
def sweepcheck(engine):
    lib = C.CDLL(str(REPO / "tests" / "native" / "libplancheck.so"))
    lib.plancheck_sweep.restype = C.c_int
    lib.plancheck_sweep.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]

    def run(rows, cols, ro, ci, alpha, delta, panels_per_wave, strip_blocks, waves=4):
        csr = engine.CSR.from_arrays(rows, cols, ro, ci)
        pipe = engine.Pipeline(csr, alpha=alpha, delta=delta, device=-1)
        arrays = pipe.arrays()
        keep = {k: np.ascontiguousarray(arrays[k], dtype=np.uint32) for k in
                ("reorderedRows", "denseCols", "blockOffsets", "blockValues", "sparseValueOffsets",
                 "sparseValues", "sparseRelativeRows", "sparseColIndices")}
        d = engine.RphmDesc()
        d.M, d.N, d.nnz = rows, cols, csr.nnz
        d.num_nonzero_rows = keep["reorderedRows"].size
        d.num_row_panels = keep["blockOffsets"].size - 1
        cast = lambda a: a.ctypes.data_as(engine.u32p)
        d.reordered_rows, d.dense_cols = cast(keep["reorderedRows"]), cast(keep["denseCols"])
        d.block_offsets, d.block_values = cast(keep["blockOffsets"]), cast(keep["blockValues"])
        d.sparse_value_offsets, d.sparse_values = cast(keep["sparseValueOffsets"]), cast(keep["sparseValues"])
        d.sparse_relative_rows, d.sparse_col_indices = cast(keep["sparseRelativeRows"]), cast(keep["sparseColIndices"])
        out = (C.c_uint64 * 6)()
        rc = lib.plancheck_sweep(C.byref(d), waves, panels_per_wave, strip_blocks, out)
        res = dict(zip(("items", "entries", "groups", "strips", "max_step_entries", "bytes"), (int(v) for v in out)))
        res["rphm_dense"] = int(csr.nnz - keep["sparseValues"].size)
        return rc, res
    return run


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("panels_per_wave", [1, 2, 4])
@pytest.mark.parametrize("strip_blocks", [1, 7, 63])
def test_sweep_format_lists_every_dense_entry_once(sweepcheck, panels_per_wave, strip_blocks, waves):
    """csrc/sweep_format.hpp, read the way denseSweep reads it: each dense entry of the RPHM once, in the list of its
    (row group, strip, wave, block), with the accumulator cell of its (row, column) - all-dense and hybrid plans, a
    ragged last row group (21 panels) and a ragged last column block (1500 = 93 * 16 + 12)."""
    rows, cols, ro, ci = synth.nips_like(rows=330, cols=1500, nnz=42000, seed=3)
    for delta in (0.0, 0.1):
        rc, r = sweepcheck(rows, cols, ro, ci, 0.3, delta, panels_per_wave, strip_blocks, waves)
        assert rc == 0, f"invariant {rc} violated: {r}"
        assert r["entries"] == r["rphm_dense"] > 0
        assert r["groups"] == -(-21 // (waves * panels_per_wave)) and r["strips"] == -(-94 // strip_blocks)
        assert r["items"] == r["groups"] * r["strips"]


def test_sweep_format_takes_unsorted_rows_and_full_tiles(sweepcheck):
    """CSR rows in file order (the reference's loader keeps it): the entry words carry explicit offsets, so the format
    does not need sorted rows; a full matrix fills every step with 256 entries per panel."""
    rng = np.random.default_rng(5)
    rows, cols, ro, ci = synth.random_pattern(150, 220, 5000, seed=11, empty_rows=9)
    ci = ci.copy()
    for i in range(rows):
        rng.shuffle(ci[ro[i]:ro[i + 1]])
    rc, r = sweepcheck(rows, cols, ro, ci, 0.3, 0.0, 2, 5)
    assert rc == 0 and r["entries"] == ci.size, (rc, r)
    rows, cols = 128, 64
    ro = np.arange(rows + 1, dtype=np.uint32) * cols
    ci = np.tile(np.arange(cols, dtype=np.uint32), rows)
    rc, r = sweepcheck(rows, cols, ro, ci, 0.3, 0.0, 2, 3)
    assert rc == 0 and r["max_step_entries"] == 512, (rc, r)


def test_sweep_format_rejects_shapes_it_cannot_hold(sweepcheck):
    rows, cols, ro, ci = synth.random_pattern(40, 60, 500, seed=2)
    assert sweepcheck(rows, cols, ro, ci, 0.3, 0.0, 3, 4)[0] == 201      # panels per wave: 1, 2 or 4
    assert sweepcheck(rows, cols, ro, ci, 0.3, 0.0, 2, 64)[0] == 201     # at most 63 blocks per strip
    assert sweepcheck(rows, cols, ro, ci, 0.3, 0.0, 2, 8, waves=6)[0] == 201   # 4 or 8 consumer waves


@pytest.fixture(scope="module")
def gemmcheck(engine):
    lib = C.CDLL(str(REPO / "tests" / "native" / "libplancheck.so"))
    lib.plancheck_gemm.restype = C.c_int
    lib.plancheck_gemm.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]

    def run(rows, cols, ro, ci, alpha, delta, panels, blocks, balance=1):
        csr = engine.CSR.from_arrays(rows, cols, ro, ci)
        pipe = engine.Pipeline(csr, alpha=alpha, delta=delta, device=-1)
        arrays = pipe.arrays()
        keep = {k: np.ascontiguousarray(arrays[k], dtype=np.uint32) for k in
                ("reorderedRows", "denseCols", "blockOffsets", "blockValues", "sparseValueOffsets",
                 "sparseValues", "sparseRelativeRows", "sparseColIndices")}
        d = engine.RphmDesc()
        d.M, d.N, d.nnz = rows, cols, csr.nnz
        d.num_nonzero_rows = keep["reorderedRows"].size
        d.num_row_panels = keep["blockOffsets"].size - 1
        cast = lambda a: a.ctypes.data_as(engine.u32p)
        d.reordered_rows, d.dense_cols = cast(keep["reorderedRows"]), cast(keep["denseCols"])
        d.block_offsets, d.block_values = cast(keep["blockOffsets"]), cast(keep["blockValues"])
        d.sparse_value_offsets, d.sparse_values = cast(keep["sparseValueOffsets"]), cast(keep["sparseValues"])
        d.sparse_relative_rows, d.sparse_col_indices = cast(keep["sparseRelativeRows"]), cast(keep["sparseColIndices"])
        out = (C.c_uint64 * 11)()
        rc = lib.plancheck_gemm(C.byref(d), panels, blocks, balance, out)
        res = dict(zip(("items", "entries", "groups", "strips", "full_grid", "bytes", "tiles", "longest_list", "fullest_strip", "emptiest_strip", "lopsided"),
                       (int(v) for v in out)))
        res["rphm_dense"] = int(csr.nnz - keep["sparseValues"].size)
        return rc, res
    return run


@pytest.mark.parametrize("panels,blocks", [(16, 16), (16, 20), (16, 12), (8, 16), (8, 20), (16, 8), (8, 8), (8, 12)])
def test_gemm_format_lists_every_dense_entry_once(gemmcheck, panels, blocks):
    """csrc/gemm_format.hpp, read the way denseGemm reads it: each dense entry of the RPHM once, in the list of its
    (macro-tile, wave, pass), with the accumulator cell of its (row, column); items in gemmItemPlace's order; lists in
    (row, column) order - all-dense and hybrid plans, a ragged last row group (21 panels) and a ragged last column
    block (1500 = 93 * 16 + 12)."""
    rows, cols, ro, ci = synth.nips_like(rows=330, cols=1500, nnz=42000, seed=3)
    for delta, balance in ((0.0, 1), (0.1, 1), (0.0, 0)):
        rc, r = gemmcheck(rows, cols, ro, ci, 0.3, delta, panels, blocks, balance)
        assert rc == 0, f"invariant {rc} violated: {r}"
        assert r["entries"] == r["rphm_dense"] > 0
        assert r["groups"] == -(-21 // panels) and r["strips"] == -(-94 // blocks)
        assert r["items"] <= r["groups"] * r["strips"] and r["full_grid"] == (r["items"] == r["groups"] * r["strips"])
        assert r["tiles"] == r["items"] * panels * blocks


def test_gemm_format_takes_unsorted_rows_full_tiles_and_empty_macro_tiles(gemmcheck):
    """CSR rows in file order (the reference's loader keeps it): the entry words carry explicit offsets, so the format
    does not need sorted rows; a full matrix makes lists of 4096 words per pass; macro-tiles without a dense entry are
    left out of the item list (the kernel then takes an item's place from its record)."""
    rng = np.random.default_rng(5)
    rows, cols, ro, ci = synth.random_pattern(150, 220, 5000, seed=11, empty_rows=9)
    ci = ci.copy()
    for i in range(rows):
        rng.shuffle(ci[ro[i]:ro[i + 1]])
    rc, r = gemmcheck(rows, cols, ro, ci, 0.3, 0.0, 8, 8)
    assert rc == 0 and r["entries"] == ci.size, (rc, r)
    rows, cols = 256, 256
    ro = np.arange(rows + 1, dtype=np.uint32) * cols
    ci = np.tile(np.arange(cols, dtype=np.uint32), rows)
    rc, r = gemmcheck(rows, cols, ro, ci, 0.3, 0.0, 16, 16)
    assert rc == 0 and r["longest_list"] == 4096 and r["items"] == 1, (rc, r)
    # a block-diagonal pattern: off-diagonal macro-tiles hold nothing
    blocks = [(i, j) for i in range(600) for j in range((i // 150) * 400, (i // 150) * 400 + 400, 7)]
    ro = np.zeros(601, dtype=np.uint32)
    for i, _ in blocks:
        ro[i + 1] += 1
    ro = np.cumsum(ro).astype(np.uint32)
    ci = np.array([j for _, j in blocks], dtype=np.uint32)
    rc, r = gemmcheck(600, 1600, ro, ci, 0.3, 0.0, 8, 8, balance=0)
    assert rc == 0 and r["full_grid"] == 0 and 0 < r["items"] < r["groups"] * r["strips"], (rc, r)
    # (balanced: no natural strip above twice the mean, so the columns stay in natural order; the ROWS are dealt over the row
    # halves by entry count, which spreads the diagonal blocks over all macro-tiles)
    rc, r = gemmcheck(600, 1600, ro, ci, 0.3, 0.0, 8, 8, balance=1)
    assert rc == 0 and r["lopsided"] == 0 and r["full_grid"] == 1, (rc, r)
    # hot columns FIRST (a vocabulary sorted by frequency): the first natural strip holds most entries, balancing deals them out
    rows, cols = 512, 4096
    per_row = [np.unique(np.concatenate([np.arange(0, 200, 1 + (i % 2)), (np.arange(40) * 97 + 13 * i) % cols])) for i in range(rows)]
    ro = np.zeros(rows + 1, dtype=np.uint32)
    ro[1:] = np.cumsum([c.size for c in per_row])
    ci = np.concatenate(per_row).astype(np.uint32)
    natural = gemmcheck(rows, cols, ro, ci, 0.3, 0.0, 16, 16, balance=0)
    dealt = gemmcheck(rows, cols, ro, ci, 0.3, 0.0, 16, 16, balance=1)
    assert natural[0] == 0 and dealt[0] == 0, (natural, dealt)
    assert dealt[1]["lopsided"] == 1 and dealt[1]["fullest_strip"] * 3 < natural[1]["fullest_strip"], (natural[1], dealt[1])


def test_gemm_format_rejects_shapes_it_cannot_hold(gemmcheck):
    rows, cols, ro, ci = synth.random_pattern(40, 60, 500, seed=2)
    assert gemmcheck(rows, cols, ro, ci, 0.3, 0.0, 12, 16)[0] == 201     # panels per macro-tile: 8 or 16
    assert gemmcheck(rows, cols, ro, ci, 0.3, 0.0, 16, 24)[0] == 201     # blocks per macro-tile: 8, 12, 16, 20


@pytest.fixture(scope="module")
def evictcheck(engine):
    lib = C.CDLL(str(REPO / "tests" / "native" / "libplancheck.so"))
    lib.plancheck_evict.restype = C.c_int
    lib.plancheck_evict.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]

    def run(rows, cols, ro, ci, alpha, delta):
        csr = engine.CSR.from_arrays(rows, cols, ro, ci)
        arrays = engine.Pipeline(csr, alpha=alpha, delta=delta, device=-1).arrays()
        keep = {k: np.ascontiguousarray(arrays[k], dtype=np.uint32) for k in
                ("reorderedRows", "denseCols", "blockOffsets", "blockValues", "sparseValueOffsets",
                 "sparseValues", "sparseRelativeRows", "sparseColIndices")}
        d = engine.RphmDesc()
        d.M, d.N, d.nnz = rows, cols, csr.nnz
        d.num_nonzero_rows = keep["reorderedRows"].size
        d.num_row_panels = keep["blockOffsets"].size - 1
        cast = lambda a: a.ctypes.data_as(engine.u32p)
        d.reordered_rows, d.dense_cols = cast(keep["reorderedRows"]), cast(keep["denseCols"])
        d.block_offsets, d.block_values = cast(keep["blockOffsets"]), cast(keep["blockValues"])
        d.sparse_value_offsets, d.sparse_values = cast(keep["sparseValueOffsets"]), cast(keep["sparseValues"])
        d.sparse_relative_rows, d.sparse_col_indices = cast(keep["sparseRelativeRows"]), cast(keep["sparseColIndices"])
        out = (C.c_uint64 * 7)()
        rc = lib.plancheck_evict(C.byref(d), out)
        return rc, dict(zip(("wide_before", "evicted", "wide_after", "tile_bytes", "dense", "residue", "dense_before"), (int(v) for v in out)))
    return run


def test_outlier_entries_are_evicted_and_the_windows_fit(evictcheck):
    """One long row with hundreds of residue entries between the dense columns of its panel (csrc/plan_evict.hpp): a few of
    its entries leave the dense part, every entry keeps its place in the matrix, and the plan packs with 8-bit windows."""
    rc, r = evictcheck(*synth.outlier_row_pattern(), 0.3, 0.3)
    assert rc == 0, (rc, r)
    assert r["wide_before"] == 1 and r["wide_after"] == 0 and r["tile_bytes"] == 1, r
    assert 0 < r["evicted"] * 16 <= r["dense_before"], r
    assert r["dense"] == r["dense_before"] - r["evicted"], r


def test_eviction_leaves_other_patterns_alone(evictcheck):
    """Nothing is too wide in a pattern with sorted rows and an all-dense split; with unsorted CSR rows nearly every block
    is, which is not a matter of outliers: the arrays stay as they are (the plan then takes direct offsets, as before)."""
    rc, r = evictcheck(*synth.nips_like(rows=320, cols=1500, nnz=40000, seed=1), 0.3, 0.0)
    assert rc == 0 and r["wide_before"] == 0 and r["evicted"] == 0, (rc, r)
    rng = np.random.default_rng(3)
    rows, cols = 64, 6000
    per_row = [rng.permutation(cols)[:3000] for _ in range(rows)]
    ro = np.zeros(rows + 1, dtype=np.uint32)
    ro[1:] = np.cumsum([len(c) for c in per_row])
    rc, r = evictcheck(rows, cols, ro, np.concatenate(per_row).astype(np.uint32), 0.3, 0.0)
    assert rc == 0 and r["wide_before"] == 1 and r["evicted"] == 0 and r["wide_after"] == 1, (rc, r)
