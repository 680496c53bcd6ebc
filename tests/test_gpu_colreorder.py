"""SURVEY.md 8f-2 on the GPU: bsmr_col_reorder (column reordering, dense / sparse split and the RPHM index arrays on the
device) against the host implementation - the same ten arrays, byte for byte - and inside the pipeline."""
import os
import time

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

ARRAYS = ("denseCols", "denseColOffsets", "sparseCols", "sparseColOffsets", "sparseValueOffsets", "blockOffsets",
          "blockValues", "sparseValues", "sparseRelativeRows", "sparseColIndices")


def _host(engine, csr, alpha, delta, row_mode=0):
    os.environ["BSMR_COLREORDER"] = "host"
    try:
        return engine.Pipeline(csr, alpha=alpha, delta=delta, row_mode=row_mode, device=-1)
    finally:
        del os.environ["BSMR_COLREORDER"]


@pytest.mark.parametrize("name,pattern", [
    ("random", lambda: synth.random_pattern(150, 220, 5000, seed=11, empty_rows=9)),
    ("one panel", lambda: synth.random_pattern(9, 40, 120, seed=3)),
    ("nips-like", lambda: synth.nips_like(rows=330, cols=1500, nnz=42000, seed=3)),
    ("community", lambda: synth.community_graph(n=900, avg_degree=70, communities=6, seed=5)),
    ("mesh", lambda: synth.banded_mesh_like(n=20000, nnz=220000, seed=7)),
    ("wathen100", lambda: synth.wathen_pattern(100, 100)),
])
@pytest.mark.parametrize("delta", [0.0, 0.1, 0.3, 0.5, 1.1])
def test_device_column_reordering_equals_the_host_arrays(engine, name, pattern, delta):
    rows, cols, ro, ci = pattern()
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    host = _host(engine, csr, 0.3, delta)
    want = host.arrays()
    st, got, ms = engine.col_reorder_device(rows, cols, ro, ci, want["reorderedRows"], delta, device=0)
    assert st == engine.OK
    for k in ARRAYS:
        assert got[k].shape == want[k].shape, (name, k, got[k].shape, want[k].shape)
        assert np.array_equal(got[k], want[k]), (name, k)


@pytest.mark.parametrize("delta", [0.0, 0.2, 1.1])
def test_device_column_reordering_equals_the_numpy_oracle(engine, delta):
    """The device arrays against oracle/bsmr_oracle.py directly (col_reordering + rphm restate src/colReordering.cu:274-404
    and src/BSMR.cpp:83-265 line by line), not through the host product: a row order that is NOT the pipeline's own
    (reversed identity with the empty rows left out) and a pattern whose last panel is ragged."""
    import bsmr_oracle
    rows, cols, ro, ci = synth.random_pattern(75, 130, 1900, seed=31, empty_rows=5)
    order = np.array([r for r in range(rows - 1, -1, -1) if ro[r + 1] > ro[r]], dtype=np.uint32)
    cr = bsmr_oracle.col_reordering(rows, cols, ro, ci, order, delta)
    rp = bsmr_oracle.rphm(rows, cols, ro, ci, order, cr)
    assert bsmr_oracle.check_rphm_invariants(rows, cols, ro, ci, order, cr, rp)
    st, got, _ = engine.col_reorder_device(rows, cols, ro, ci, order, delta, device=0)
    assert st == engine.OK
    want = {**cr, **rp}
    for k in ARRAYS:
        assert np.array_equal(got[k], want[k]), k


def test_device_column_reordering_in_the_pipeline_and_its_time(engine, oracle, capsys):
    """BSMR::colReordering on the device (BSMR_COLREORDER=device; automatic from 4 M entries): the pipeline's arrays,
    its check_rphm invariants and the SDDMM are unchanged; times of both paths are printed (mycielskian15: 5.6 M entries)."""
    rows, cols, ro, ci = synth.mycielskian_pattern(15)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    host = _host(engine, csr, 0.3, 0.3, row_mode=engine.ROWS_IDENTITY)
    os.environ["BSMR_COLREORDER"] = "device"
    try:
        t0 = time.perf_counter()
        dev = engine.Pipeline(csr, alpha=0.3, delta=0.3, row_mode=engine.ROWS_IDENTITY, device=0)
        wall = time.perf_counter() - t0
    finally:
        del os.environ["BSMR_COLREORDER"]
    a, b = host.arrays(), dev.arrays()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert dev.check()
    with capsys.disabled():
        print(f"\nmycielskian15 column reordering + RPHM arrays: host {host.col_reordering_ms:.1f} + {host.rphm_ms:.1f} ms, "
              f"device path {dev.col_reordering_ms:.1f} + {dev.rphm_ms:.1f} ms (pipeline wall {wall * 1e3:.0f} ms)")
    K = 64
    A, B = engine.make_data(rows * K, 5489), engine.make_data(cols * K, 5490)
    import torch
    d = torch.device("cuda:0")
    tA, tB = torch.from_numpy(A).to(d), torch.from_numpy(B).to(d)
    tP = torch.zeros(csr.nnz, dtype=torch.float32, device=d)
    engine.sddmm(dev.plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), engine.COMPUTE_F16, 0)
    torch.cuda.synchronize()
    bad, _ = oracle.check_data(oracle.sddmm_cpu(rows, cols, K, ro, ci, A, B), tP.cpu().numpy())
    assert bad == 0


def test_device_column_reordering_argument_checks(engine):
    rows, cols, ro, ci = synth.random_pattern(40, 50, 300, seed=2)
    st, got, _ = engine.col_reorder_device(rows, cols, ro, ci, np.array([0, 1, 99], np.uint32), 0.3)
    assert st == engine.ERR_BAD_PLAN            # a row id outside the matrix
    st, got, _ = engine.col_reorder_device(rows, cols, ro, ci, np.zeros(0, np.uint32), 0.3)
    assert st == engine.OK and got["blockOffsets"].tolist() == [0] and got["blockValues"].size == 0
    st, got, _ = engine.col_reorder_device(rows, cols, ro, ci, np.arange(rows, dtype=np.uint32), 0.3, device=99)
    assert st == engine.ERR_NO_DEVICE
