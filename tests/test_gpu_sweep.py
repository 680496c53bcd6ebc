"""GPU parity tests of the sweep engine (BSMR_ENGINE_SWEEP, csrc/sweep_kernels.hpp): the dense part computed as a
masked GEMM over row groups x strips of B.  Same contract as the other dense engines - exact output indexing, the
dense-path error model against the oracle, zero checkData failures - and, stronger, results BIT-IDENTICAL to the
streaming engine's (same casts, same MFMA instruction, same order of the k steps)."""
import numpy as np
import pytest

import synth
from test_gpu_parity import _dev, check_case, run_hip

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _run(engine, rows, cols, nnz, arrays, K, A, B, mode, options):
    dev = _dev()
    st, plan = engine.plan_from_arrays(rows, cols, nnz, arrays, device=0, options=options)
    assert st == engine.OK, st
    tA, tB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    tP = torch.full((nnz,), float("nan"), dtype=torch.float32, device=dev)
    engine.sddmm(plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    choice = None
    engine.plan_destroy(plan)
    return tP.cpu().numpy(), choice


@pytest.mark.parametrize("K,mode", [(32, 0), (64, 1), (128, 0), (128, 1), (256, 0), (512, 1)])
def test_sweep_engine_matches_the_oracle(engine, oracle, monkeypatch, K, mode):
    """check_case through bsmr_plan_create (environment override): a ragged last row group (21 panels), a ragged
    last column block (1500 = 93 * 16 + 12), all-dense and hybrid plans, fp32 operands rounded in the kernel
    (K <= 128) and 16-bit copies, 4 and 8 consumer waves, strips of 1 to 63 blocks."""
    monkeypatch.setenv("BSMR_DENSE_ENGINE", "sweep")
    rows, cols, ro, ci = synth.nips_like(rows=330, cols=1500, nnz=42000, seed=3)
    for fp32, waves, panels, blocks in ((1, 4, 0, 0), (0, 4, 2, 7), (1, 8, 2, 63), (0, 8, 0, 1)):
        monkeypatch.setenv("BSMR_SWEEP_FP32", str(fp32))
        monkeypatch.setenv("BSMR_SWEEP_WAVES", str(waves))
        monkeypatch.setenv("BSMR_SWEEP_PANELS", str(panels))
        monkeypatch.setenv("BSMR_SWEEP_BLOCKS", str(blocks))
        check_case(engine, oracle, rows, cols, ro, ci, K, 0.3, 0.0, mode)
        # (hybrid plan: by rule the kernel takes the 16-bit copies, the residue kernel reads them too)
        monkeypatch.setenv("BSMR_SWEEP_FP32", "-1" if fp32 else "0")
        check_case(engine, oracle, rows, cols, ro, ci, K, 0.3, 0.1, mode)
    rows, cols, ro, ci = synth.random_pattern(150, 220, 5000, seed=7 + K, empty_rows=9)
    check_case(engine, oracle, rows, cols, ro, ci, K, 0.3, 0.1, mode)


@pytest.mark.parametrize("K,mode", [(32, 1), (128, 0), (512, 0)])
def test_sweep_engine_is_bit_identical_to_the_streaming_engine(engine, K, mode):
    rows, cols, ro, ci = synth.nips_like(rows=700, cols=2100, nnz=120000, seed=5)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    A = engine.make_data(rows * K, 5489)
    B = engine.make_data(cols * K, 5490)
    for delta in (0.0, 0.1):
        pipe = engine.Pipeline(csr, alpha=0.3, delta=delta, device=-1)
        arrays = pipe.arrays()
        ref, _ = _run(engine, rows, cols, csr.nnz, arrays, K, A, B, mode, engine.plan_options(fold_dense_below=0, convert_in_kernel=0))
        assert not np.isnan(ref).any()
        for fp32 in ((1, 0) if K <= 128 else (0,)):
            for waves in (4, 8):
                got, _ = _run(engine, rows, cols, csr.nnz, arrays, K, A, B, mode,
                              engine.plan_options(fold_dense_below=0, dense_engine=engine.ENGINE_SWEEP, sweep_fp32=fp32, sweep_waves=waves,
                                                  sparse_lowp=0 if fp32 else 1))
                dense = np.ones(csr.nnz, dtype=bool)
                if delta > 0:       # the residue of a call that rounds in the kernel runs its fp32 kernel: compare the dense entries
                    bv = pipe.array("blockValues")
                    dense[:] = False
                    dense[bv[bv != 0xFFFFFFFF]] = True
                assert np.array_equal(got[dense].view(np.uint32), ref[dense].view(np.uint32)), (delta, fp32, waves)
                assert not np.isnan(got).any()


def test_sweep_engine_output_indexing_and_unsorted_rows(engine, monkeypatch):
    """A = one-hot rows, B = column id: every entry's exact value identifies (row, col); CSR rows in file order (the
    reference's loader keeps it) are taken as they are - the entry words carry explicit offsets."""
    monkeypatch.setenv("BSMR_DENSE_ENGINE", "sweep")
    rows, cols, ro, ci = synth.random_pattern(130, 500, 4000, seed=21, empty_rows=4)
    rng = np.random.default_rng(9)
    shuffled = ci.copy()
    for i in range(rows):
        rng.shuffle(shuffled[ro[i]:ro[i + 1]])
    K = 32
    A = np.zeros((rows, K), dtype=np.float32)
    A[:, 0] = np.arange(1, rows + 1) % 64 + 1
    A[:, 1] = 1.0
    B = np.zeros((cols, K), dtype=np.float32)
    B[:, 0] = 1.0
    B[:, 1] = np.arange(cols) % 128
    r = np.repeat(np.arange(rows), np.diff(ro.astype(np.int64)))
    for cols_of in (ci, shuffled):
        csr = engine.CSR.from_arrays(rows, cols, ro, cols_of)
        want = (A[r, 0] + B[cols_of, 1]).astype(np.float32)
        for delta in (0.0, 0.05):
            for mode in (0, 1):
                pipe = engine.Pipeline(csr, alpha=0.3, delta=delta, device=0)
                got = run_hip(engine, pipe, K, A.ravel(), B.ravel(), mode)
                assert np.array_equal(got, want)


def test_sweep_engine_full_matrix_steps_of_more_than_64_entries(engine, oracle, monkeypatch):
    """A full 128 x 96 matrix: every (wave, block) step writes 256 entries per panel - four passes of the entry list."""
    monkeypatch.setenv("BSMR_DENSE_ENGINE", "sweep")
    monkeypatch.setenv("BSMR_SWEEP_PANELS", "2")
    rows, cols = 128, 96
    ro = np.arange(rows + 1, dtype=np.uint32) * cols
    ci = np.tile(np.arange(cols, dtype=np.uint32), rows)
    check_case(engine, oracle, rows, cols, ro, ci, 128, 0.3, 0.0, 0)
