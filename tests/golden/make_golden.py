#!/usr/bin/env python3
"""Generates tests/golden/pipeline_*.json from the numpy oracle (oracle/bsmr_oracle.py).

These fixtures freeze the oracle's outputs on tiny seeded inputs so that later edits
cannot drift silently.  They are NOT reference output (that is reference_logs.json,
which pins the same oracle on real matrices: tests/test_reference_logs.py).
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO / "oracle"))
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
import bsmr_oracle as bo  # noqa: E402
import synth  # noqa: E402

CASES = [
    # name, rows, cols, nnz, seed, empty_rows, alpha, delta, binWidth
    ("a", 40, 64, 500, 101, 3, 0.3, 0.1, 16),
    ("b", 33, 200, 700, 102, 0, 0.5, 0.0, 16),
    ("c", 70, 90, 1200, 103, 6, 0.1, 0.3, 20),
    ("d", 18, 30, 120, 104, 1, 0.9, 1.1, 16),
]

for name, rows, cols, nnz, seed, empty, alpha, delta, bw in CASES:
    rows, cols, ro, ci = synth.random_pattern(rows, cols, nnz, seed=seed, empty_rows=empty)
    rr, nc = bo.row_reordering(rows, cols, ro, ci, alpha, bw)
    cr = bo.col_reordering(rows, cols, ro, ci, rr, delta)
    rp = bo.rphm(rows, cols, ro, ci, rr, cr)
    bo.check_rphm_invariants(rows, cols, ro, ci, rr, cr, rp)
    out = dict(rows=rows, cols=cols, alpha=alpha, delta=delta, binWidth=bw,
               rowOffsets=ro.tolist(), colIndices=ci.tolist(),
               reorderedRows=rr.tolist(), numClusters=nc)
    for k in ("denseCols", "denseColOffsets", "sparseCols", "sparseColOffsets", "sparseValueOffsets"):
        out[k] = cr[k].tolist()
    for k in ("blockOffsets", "blockValues", "sparseValues", "sparseRelativeRows", "sparseColIndices"):
        out[k] = rp[k].tolist()
    (Path(__file__).parent / f"pipeline_{name}.json").write_text(json.dumps(out, separators=(",", ":")))
    print(name, "panels", cr["numRowPanels"], "clusters", nc, "dense blocks", int(rp["blockOffsets"][-1]),
          "sparse", int(cr["sparseValueOffsets"][-1]))
