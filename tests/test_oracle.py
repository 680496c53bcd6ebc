"""The oracle against hand-derived known answers and an independent numpy
computation.  (The reference holds no golden output vectors, so the SDDMM values are
pinned to the cited semantics only; the host pipeline is pinned to the reference's
published logs in tests/test_reference_logs.py.)"""
import json
from pathlib import Path

import numpy as np
import pytest

import bsmr_oracle as bo
import synth

GOLDEN = Path(__file__).parent / "golden"


def test_sddmm_cpu_matches_numpy_sequential(oracle):
    rng = np.random.default_rng(0)
    rows, cols, ro, ci = synth.random_pattern(37, 53, 400, seed=1, empty_rows=3)
    K = 64
    A = rng.random((rows, K), dtype=np.float32) * 2
    B = rng.random((cols, K), dtype=np.float32) * 2
    got = oracle.sddmm_cpu(rows, cols, K, ro, ci, A.ravel(), B.ravel())
    r = np.repeat(np.arange(rows), np.diff(ro.astype(np.int64)))
    want = np.zeros(ci.size, dtype=np.float32)
    for k in range(K):  # strictly sequential fp32: val += a*b (src/host.cpp:62-71)
        want = (want + A[r, k] * B[ci, k]).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    f64 = oracle.sddmm_f64(rows, K, ro, ci, A.ravel(), B.ravel())
    assert np.allclose(f64, (A[r].astype(np.float64) * B[ci].astype(np.float64)).sum(1), rtol=1e-14)


def test_s_values_are_not_multiplied_in(oracle):
    # reference src/host.cpp:62-73: P = A*B sampled at S's pattern; S's values unused.
    rows, cols, ro, ci = synth.random_pattern(8, 8, 20, seed=2)
    A = np.ones(rows * 32, dtype=np.float32)
    B = np.ones(cols * 32, dtype=np.float32)
    assert np.array_equal(oracle.sddmm_cpu(rows, cols, 32, ro, ci, A, B), np.full(20, 32, np.float32))


@pytest.mark.parametrize("a,b,ok", [
    (1.0, 1.0, True),
    (1.0, 1.0009, True),       # 9e-4 relative
    (1.0, 1.0011, False),      # 1.1e-3 relative
    (0.0, 9e-6, True),         # absolute escape hatch 1e-5
    (0.0, 2e-5, False),        # denominator floored at 1e-3 -> 2e-2
    (5e-4, 5.000004e-4, True),
    (100.0, 100.09, True),
    (100.0, 100.11, False),
    (-3.0, 3.0, False),
])
def test_check_one_known_answers(oracle, a, b, ok):
    # include/checkData.hpp:21-30
    assert bool(oracle.lib.oracle_check_one(a, b)) is ok
    assert bool(oracle.lib.oracle_check_one(b, a)) is ok


def test_check_data_counts(oracle):
    x = np.array([1, 2, 3, 4], dtype=np.float32)
    y = np.array([1, 2.01, 3, 4.5], dtype=np.float32)
    assert oracle.check_data(x, y) == (2, 1)
    assert oracle.check_data(x, x) == (0, -1)


def test_rounding_known_answers(oracle):
    L = oracle.lib
    one = np.float32(1.0)
    ulp10 = np.float32(2.0 ** -10)
    tie = np.float32(1.0 + 2.0 ** -11)          # exactly half way between 1 and 1+2^-10
    assert L.oracle_round_tf32(tie) == one + ulp10            # ties away from zero (cvt.rna)
    assert L.oracle_round_fp16(tie) == one                    # ties to even
    assert L.oracle_round_fp16(np.float32(1.0 + 3 * 2.0 ** -11)) == one + 2 * ulp10
    assert L.oracle_round_tf32(np.float32(-tie)) == -(one + ulp10)
    assert L.oracle_round_bf16(np.float32(1.0 + 2.0 ** -8)) == one  # tie -> even
    assert L.oracle_round_bf16(np.float32(1.0 + 3 * 2.0 ** -8)) == np.float32(1.0 + 2.0 ** -6)
    assert L.oracle_round_fp16(np.float32(70000.0)) == np.float32(np.inf)
    assert L.oracle_round_fp16(np.float32(2.0 ** -24)) == np.float32(2.0 ** -24)  # smallest subnormal
    assert L.oracle_round_fp16(np.float32(2.0 ** -26)) == 0.0


def test_fp16_bf16_rounding_match_numpy_and_torch(oracle):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.random(20000, dtype=np.float32) * 2,
                        (rng.standard_normal(20000) * 10.0 ** rng.integers(-8, 5, 20000)).astype(np.float32)])
    assert np.array_equal(oracle.round_array(2, x), x.astype(np.float16).astype(np.float32))
    torch = pytest.importorskip("torch")
    t = torch.from_numpy(x)
    assert np.array_equal(oracle.round_array(3, x), t.to(torch.bfloat16).float().numpy())


def test_operand_range_keeps_tf32_and_fp16_identical(oracle):
    # for U[0,2) operands fp16 (RNE) and TF32 (RNA) round to the same
    # value except on exact ties and below fp16's normal range.
    import ctypes as C
    lib = C.CDLL(str(Path(__file__).parent.parent / "bsmr-sddmm_amd" / "lib" / "libbsmr_host.so"))
    x = np.empty(1 << 16, dtype=np.float32)
    lib.bsmr_make_data(x.ctypes.data_as(C.c_void_p), C.c_size_t(x.size), C.c_uint32(5489))
    a, b = oracle.round_array(1, x), oracle.round_array(2, x)
    diff = a != b
    # the generator emits 24-bit fractions, so ties (bit 12 set, lower bits clear) do occur
    small = x < 2.0 ** -14
    ties = (x.view(np.uint32) & 0x1FFF) == 0x1000
    assert not (diff & ~small & ~ties).any()


def test_ref_kernel_model_within_reference_tolerance(oracle):
    rows, cols, ro, ci = synth.random_pattern(60, 80, 1500, seed=4)
    K = 256
    rng = np.random.default_rng(5)
    A = (rng.random(rows * K, dtype=np.float32) * 2)
    B = (rng.random(cols * K, dtype=np.float32) * 2)
    cpu = oracle.sddmm_cpu(rows, cols, K, ro, ci, A, B)
    for flags in (np.ones(ci.size, np.uint8), np.zeros(ci.size, np.uint8)):
        model = oracle.ref_kernel_model(rows, K, ro, ci, flags, A, B)
        assert oracle.check_data(cpu, model)[0] == 0


def test_twins_agree_with_cpu_within_tolerance(oracle):
    rows, cols, ro, ci = synth.random_pattern(50, 70, 900, seed=6)
    K = 128
    rng = np.random.default_rng(7)
    A = rng.random(rows * K, dtype=np.float32) * 2
    B = rng.random(cols * K, dtype=np.float32) * 2
    cpu = oracle.sddmm_cpu(rows, cols, K, ro, ci, A, B)
    for lpe in (4, 8, 16):
        assert oracle.check_data(cpu, oracle.sparse_twin(rows, K, lpe, ro, ci, A, B))[0] == 0
    assert oracle.check_data(cpu, oracle.dense_f32_twin(rows, K, ro, ci, A, B))[0] == 0
    for mode in (2, 3):
        m = oracle.dense_lowp_model(mode, rows, K, ro, ci, A, B).astype(np.float32)
        assert oracle.check_data(cpu, m)[0] == 0


def test_dense_threshold_values():
    # ceil(delta*256) in fp32 (src/colReordering.cu:246)
    assert [bo.dense_threshold(d) for d in (0.0, 0.1, 0.3, 0.5, 0.7, 0.9, 1.1)] == [0, 26, 77, 128, 180, 231, 282]


def test_golden_pipeline_fixtures():
    """Committed outputs of the oracle on tiny inputs (tests/golden/make_golden.py).  They
    guard the restatement against drift; they are NOT reference output."""
    for f in sorted(GOLDEN.glob("pipeline_*.json")):
        g = json.loads(f.read_text())
        rows, cols = g["rows"], g["cols"]
        ro = np.asarray(g["rowOffsets"], np.uint32)
        ci = np.asarray(g["colIndices"], np.uint32)
        rr, nc = bo.row_reordering(rows, cols, ro, ci, g["alpha"], g["binWidth"])
        assert rr.tolist() == g["reorderedRows"] and nc == g["numClusters"]
        cr = bo.col_reordering(rows, cols, ro, ci, rr, g["delta"])
        rp = bo.rphm(rows, cols, ro, ci, rr, cr)
        for k in ("denseCols", "denseColOffsets", "sparseCols", "sparseColOffsets", "sparseValueOffsets"):
            assert cr[k].tolist() == g[k], k
        for k in ("blockOffsets", "blockValues", "sparseValues", "sparseRelativeRows", "sparseColIndices"):
            assert rp[k].tolist() == g[k], k
        assert bo.check_rphm_invariants(rows, cols, ro, ci, rr, cr, rp)
