"""The host pipeline against the reference's own published outputs.

tests/golden/reference_logs.json holds what the reference (RTX 4090, sddmm_testMode)
logged for six SuiteSparse matrices over its alpha x delta x K sweep.  Their sparsity
patterns follow from a formula (synth.py), so the exact inputs can be rebuilt here and
every logged integer must come out again: NNZ after loading (the `symmetric` header is
not expanded), NumRowPanel, the cluster count of the row clustering for five alphas,
and for seven deltas the dense-block count, densities, dense / sparse entry counts and
the thread-block counts that size the reference's grids.

This pins, on real inputs: the loader's handling of symmetric files, the column-bin
histograms and dispersion order, the similarity as the reference's block executes it
(including the warps its block-wide sum skips), the greedy scan, the cluster-count
read-out, the per-panel column ordering, the dense / sparse cut and the work lists.
"""
import json
import math
from pathlib import Path

import numpy as np
import pytest

import synth

GOLDEN = json.loads((Path(__file__).parent / "golden" / "reference_logs.json").read_text())["matrices"]

PATTERNS = {
    "Trefethen_20000": lambda: synth.trefethen_pattern(20000),
    "Trefethen_20000b": lambda: synth.trefethen_pattern(19999),
    "mycielskian14": lambda: synth.mycielskian_pattern(14),
    "mycielskian15": lambda: synth.mycielskian_pattern(15),
    "wathen100": lambda: synth.wathen_pattern(100, 100),
    "wathen120": lambda: synth.wathen_pattern(100, 120),
}
# the reference's calculateBlockSize on its 24 GB card gives 16 for all six
# (src/rowReordering.cu:1009-1025: max(16, rows^2*4/(free/2), cols*4/24576))
BIN_WIDTH = 16
# alphas checked per matrix (the dense-row Mycielski graphs cost seconds per alpha)
ALPHAS = {
    "Trefethen_20000": (0.1, 0.3, 0.5, 0.7, 0.9),
    "Trefethen_20000b": (0.3, 0.7),
    "mycielskian14": (0.1, 0.3, 0.5, 0.7, 0.9),
    "mycielskian15": (0.3,),
    "wathen100": (0.1, 0.3, 0.5, 0.7, 0.9),
    "wathen120": (0.3, 0.9),
}


def fmt2(x: float) -> str:
    """std::fixed << std::setprecision(2) of a float"""
    if math.isnan(x):
        return "nan"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    return f"{x:.2f}"


def ratio(a: int, b: int) -> str:
    with np.errstate(divide="ignore", invalid="ignore"):
        return fmt2(float(np.float32(a) / np.float32(b)))


def runs_of(name, alpha):
    return [r for r in GOLDEN[name]["runs"] if abs(r["alpha"] - alpha) < 1e-6]


def test_fixture_is_complete():
    assert sorted(GOLDEN) == sorted(PATTERNS)
    for name, entry in GOLDEN.items():
        assert len(entry["runs"]) == 35, name          # 5 alphas x 7 deltas
        for run in entry["runs"]:
            assert sorted(run["gridDim_sparse"]) == ["128", "256", "32", "64"]


@pytest.mark.parametrize("name", sorted(PATTERNS))
def test_pattern_matches_logged_dimensions(name):
    rows, cols, ro, ci = PATTERNS[name]()
    dims = GOLDEN[name]["dims"]
    assert (rows, cols, int(ci.size)) == (dims["M"], dims["N"], dims["NNZ"])


def test_loader_keeps_symmetric_files_unexpanded(engine, tmp_path):
    """The logged NNZ of Trefethen_20000 is the stored lower triangle (src/Matrix.cpp:398-480)."""
    rows, cols, ro, ci = synth.trefethen_pattern(20000)
    path = tmp_path / "Trefethen_20000.mtx"
    synth.write_mtx_columnwise(path, rows, cols, ro, ci)
    csr = engine.CSR.from_file(path)
    assert csr.nnz == GOLDEN["Trefethen_20000"]["dims"]["NNZ"]
    # stable sort by row keeps the file's (ascending) column order inside a row
    assert np.array_equal(csr.row_offsets, ro) and np.array_equal(csr.col_indices, ci)


@pytest.mark.parametrize("name", sorted(PATTERNS))
def test_pipeline_reproduces_reference_logs(engine, name):
    rows, cols, ro, ci = PATTERNS[name]()
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    assert csr.calculate_block_size(24 << 30) == BIN_WIDTH
    for alpha in ALPHAS[name]:
        runs = runs_of(name, alpha)
        assert len(runs) == 7
        pipe = engine.Pipeline(csr, alpha=alpha, delta=runs[0]["delta"], block_size=BIN_WIDTH, device=-1)
        for run in runs:
            where = (name, alpha, run["delta"])
            if run is not runs[0]:
                pipe.resplit(run["delta"])
            rep = pipe.evaluate()
            arrays = pipe.arrays()
            assert pipe.num_clusters == run["bsmr_numClusters"], where
            assert pipe.num_row_panels == run["NumRowPanel"], where
            assert rep["original_num_dense_blocks"] == run["original_numDenseBlock"], where
            assert fmt2(rep["original_average_density"]) == run["original_averageDensity"], where
            assert rep["num_dense_blocks"] == run["bsmr_numDenseBlock"], where
            assert fmt2(rep["average_density"]) == run["bsmr_averageDensity"], where
            assert rep["num_dense_thread_blocks"] == run["bsmr_numDenseThreadBlocks"], where
            assert rep["num_sparse_thread_blocks"] == run["bsmr_numSparseThreadBlocks"], where
            assert rep["num_dense_data"] == run["bsmr_numDenseData"], where
            assert rep["num_sparse_data"] == run["bsmr_numSparseData"], where
            assert ratio(rep["num_dense_thread_blocks"], rep["num_sparse_thread_blocks"]) == run["bsmr_threadBlockRatio"], where
            assert ratio(rep["num_dense_data"], rep["num_sparse_data"]) == run["bsmr_dataRatio"], where
            # the RPHM work lists are what the reference sizes its grids with
            assert len(arrays["denseRowPanelIds"]) == run["bsmr_numDenseThreadBlocks"], where
            assert len(arrays["sparseRowPanelIds"]) == run["bsmr_numSparseThreadBlocks"], where
            assert len(arrays["sparseValues"]) == run["bsmr_numSparseData"], where
            # grid = (panels, ceil(max dense blocks of a panel / 4)) (src/sddmmKernel.cu:2570-2574)
            gx, gy, gz = (int(v) for v in run["gridDim_dense"].split(","))
            assert (gx, gy, gz) == (pipe.num_row_panels, -(-rep["max_dense_blocks_per_panel"] // 4), 1), where
            for k in ("64", "128", "256"):       # K > 32: one thread block per 128 sparse entries of a panel
                assert run["gridDim_sparse"][k] == f"{rep['num_sparse_thread_blocks']}, 1, 1", where
            assert run["gridDim_sparse"]["32"] == f"{pipe.num_row_panels}, 1, 1", where   # K <= 32: one per panel
            assert pipe.check(), where


@pytest.mark.parametrize("name,alphas", [("mycielskian14", (0.1, 0.3, 0.5)), ("wathen100", (0.1,))])
def test_oracle_clustering_reproduces_reference_logs(engine, oracle, name, alphas):
    """The plain restatement (dense histograms, every bin visited, sums in the reference's
    order) gives the logged cluster counts, and the product's row order is identical to it."""
    rows, cols, ro, ci = PATTERNS[name]()
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    for alpha in alphas:
        perm, clusters = oracle.bsa_row_reordering(rows, cols, ro, ci, BIN_WIDTH, alpha)
        assert clusters == runs_of(name, alpha)[0]["bsmr_numClusters"]
        pipe = engine.Pipeline(csr, alpha=alpha, delta=0.3, block_size=BIN_WIDTH, device=-1)
        assert np.array_equal(pipe.array("reorderedRows"), perm)


def test_cli_sweep_writes_the_reference_logs(engine, tmp_path):
    """`BSMR-sddmm -f Trefethen_20000.mtx -t 1 -l dir/` (sddmm_testMode, src/sddmm.cu:62-118) without a
    GPU: the 140 log files carry the reference's names (BSMR_k_<K>_a_<alpha>_d_<delta>.log) and, in
    the reference's `[key : value]` format, the same pipeline values the reference logged."""
    import re
    import subprocess
    rows, cols, ro, ci = synth.trefethen_pattern(20000)
    mtx = tmp_path / "Trefethen_20000.mtx"
    synth.write_mtx_columnwise(mtx, rows, cols, ro, ci)
    logs = tmp_path / "logs"
    logs.mkdir()
    exe = Path(__file__).resolve().parent.parent / "bsmr-sddmm_amd" / "bin" / "BSMR-sddmm"
    env = dict(__import__("os").environ, BSMR_CLUSTER="host")
    subprocess.run([str(exe), "-f", str(mtx), "-t", "1", "-l", str(logs) + "/"], capture_output=True, text=True,
                   timeout=600, env=env)
    names = sorted(p.name for p in logs.iterdir())
    assert len(names) == 140
    keys = ("NumRowPanel", "original_numDenseBlock", "original_averageDensity", "bsmr_numClusters",
            "bsmr_numDenseBlock", "bsmr_averageDensity", "bsmr_numDenseThreadBlocks", "bsmr_numSparseThreadBlocks",
            "bsmr_threadBlockRatio", "bsmr_numDenseData", "bsmr_numSparseData", "bsmr_dataRatio")
    checked = 0
    for run in GOLDEN["Trefethen_20000"]["runs"]:
        for k in (32, 64, 128, 256):
            a, d = (f"{v:g}" for v in (run["alpha"], run["delta"]))
            path = logs / f"BSMR_k_{k}_a_{a}_d_{d}.log"
            assert path.name in names, path.name
            text = path.read_text()
            assert text.count("---New data---") == 1
            got = {key.strip(): val.strip() for key, val in re.findall(r"\[([^\[\]:]+?)\s*:\s*([^\[\]]*)\]", text)}
            assert (got["K"], got["M"], got["N"], got["NNZ"]) == (str(k), "20000", "20000", "287233")
            assert (got["bsmr_alpha"], got["bsmr_delta"]) == (f"{run['alpha']:.2f}", f"{run['delta']:.2f}")
            for key in keys:
                assert got[key] == str(run[key]), (path.name, key, got[key], run[key])
            checked += 1
    assert checked == 140
