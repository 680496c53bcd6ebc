"""GPU parity tests of the GEMM engine (BSMR_ENGINE_GEMM, csrc/gemm_kernels.hpp): the dense part computed as an
output-stationary masked GEMM over macro-tiles of C.  Same contract as the other dense engines - exact output indexing,
the dense-path error model against the oracle, zero checkData failures (reference include/checkData.hpp:14-30) - and,
stronger, results BIT-IDENTICAL to the streaming engine's (same casts, same MFMA instruction, same order of the k steps)."""
import numpy as np
import pytest

import synth
from test_gpu_parity import _dev, check_case, run_hip

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SHAPES = [(16, 16), (16, 20), (16, 12), (8, 16), (8, 20)]


def _run(engine, rows, cols, nnz, arrays, K, A, B, mode, options):
    dev = _dev()
    st, plan = engine.plan_from_arrays(rows, cols, nnz, arrays, device=0, options=options)
    assert st == engine.OK, st
    tA, tB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    tP = torch.full((nnz,), float("nan"), dtype=torch.float32, device=dev)
    engine.sddmm(plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    import ctypes as C
    g, t, u = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
    engine.hip().bsmr_plan_dense_choice(plan, K, C.byref(g), C.byref(t), C.byref(u))
    engine.plan_destroy(plan)
    return tP.cpu().numpy(), (g.value, t.value)


@pytest.mark.parametrize("K,mode", [(64, 1), (128, 0), (128, 1), (256, 0), (512, 1)])
def test_gemm_engine_matches_the_oracle(engine, oracle, monkeypatch, K, mode):
    """check_case through bsmr_plan_create (environment override): a ragged last row group (21 panels), a ragged last
    column block (1500 = 93 * 16 + 12), all-dense and hybrid plans, every macro-tile shape the kernels are built for
    and the shape the model picks."""
    monkeypatch.setenv("BSMR_DENSE_ENGINE", "gemm")
    rows, cols, ro, ci = synth.nips_like(rows=330, cols=1500, nnz=42000, seed=3)
    for panels, blocks in SHAPES + [(0, 0)]:
        monkeypatch.setenv("BSMR_GEMM_PANELS", str(panels))
        monkeypatch.setenv("BSMR_GEMM_BLOCKS", str(blocks))
        # (K = 64 / 128 on an all-dense plan: by default the kernel on the caller's fp32 operands; every second shape the 16-bit copies)
        monkeypatch.setenv("BSMR_GEMM_FP32", "0" if (panels + blocks) % 8 else "-1")
        pipe = check_case(engine, oracle, rows, cols, ro, ci, K, 0.3, 0.0, mode)
        assert pipe.dense_choice(K)["group_size"] in (8, 16)        # the engine ran (panels per macro-tile)
        check_case(engine, oracle, rows, cols, ro, ci, K, 0.3, 0.1, mode)      # hybrid: the residue kernel reads the 16-bit copies too
    rows, cols, ro, ci = synth.random_pattern(150, 220, 5000, seed=7 + K, empty_rows=9)
    check_case(engine, oracle, rows, cols, ro, ci, K, 0.3, 0.1, mode)


def test_gemm_engine_leaves_other_k_to_the_streaming_engine(engine, oracle, monkeypatch):
    """K = 32, 96: not a multiple of the 64-k slices the kernels are built for - the call streams, results stay right."""
    monkeypatch.setenv("BSMR_DENSE_ENGINE", "gemm")
    rows, cols, ro, ci = synth.nips_like(rows=330, cols=1500, nnz=42000, seed=3)
    for K in (32, 96):
        pipe = check_case(engine, oracle, rows, cols, ro, ci, K, 0.3, 0.0, 0)
        assert pipe.dense_choice(K)["group_size"] == 1


@pytest.mark.parametrize("K,mode", [(64, 0), (128, 0), (512, 1)])
def test_gemm_engine_is_bit_identical_to_the_streaming_engine(engine, K, mode):
    rows, cols, ro, ci = synth.nips_like(rows=700, cols=2100, nnz=120000, seed=5)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    A = engine.make_data(rows * K, 5489)
    B = engine.make_data(cols * K, 5490)
    for delta in (0.0, 0.1):
        pipe = engine.Pipeline(csr, alpha=0.3, delta=delta, device=-1)
        arrays = pipe.arrays()
        ref, _ = _run(engine, rows, cols, csr.nnz, arrays, K, A, B, mode, engine.plan_options(fold_dense_below=0, convert_in_kernel=0))
        assert not np.isnan(ref).any()
        for panels, blocks in SHAPES:
            got, (group, tiles) = _run(engine, rows, cols, csr.nnz, arrays, K, A, B, mode,
                                       engine.plan_options(fold_dense_below=0, dense_engine=engine.ENGINE_GEMM, gemm_panels=panels,
                                                           gemm_blocks=blocks, gemm_fp32=0))
            assert group == panels and tiles % (panels * blocks) == 0
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (delta, panels, blocks)
            if K <= 128 and blocks != 12:
                # the caller's fp32 operands rounded in the kernel (no conversion pass): the same MFMA operands; the residue
                # of such a call runs its fp32 kernel, so only the dense entries are compared on the hybrid plan
                got, (group, tiles) = _run(engine, rows, cols, csr.nnz, arrays, K, A, B, mode,
                                           engine.plan_options(fold_dense_below=0, dense_engine=engine.ENGINE_GEMM, gemm_panels=panels,
                                                               gemm_blocks=blocks, gemm_fp32=1, sparse_lowp=0))
                assert group == panels and not np.isnan(got).any()
                dense = np.ones(csr.nnz, dtype=bool)
                if delta > 0:
                    bv = pipe.array("blockValues")
                    dense[:] = False
                    dense[bv[bv != 0xFFFFFFFF]] = True
                assert np.array_equal(got[dense].view(np.uint32), ref[dense].view(np.uint32)), (delta, panels, blocks, "fp32 operands")


def test_gemm_engine_output_indexing_and_unsorted_rows(engine, monkeypatch):
    """A = one-hot rows, B = column id: every entry's exact value identifies (row, col); CSR rows in file order (the
    reference's loader keeps it) are taken as they are - the entry words carry explicit offsets."""
    monkeypatch.setenv("BSMR_DENSE_ENGINE", "gemm")
    rows, cols, ro, ci = synth.random_pattern(130, 500, 4000, seed=21, empty_rows=4)
    rng = np.random.default_rng(9)
    shuffled = ci.copy()
    for i in range(rows):
        rng.shuffle(shuffled[ro[i]:ro[i + 1]])
    K = 64
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
                assert pipe.dense_choice(K)["group_size"] in (8, 16)
                assert np.array_equal(got, want)


def test_gemm_engine_full_matrix_lists_of_more_than_512_entries(engine, oracle, monkeypatch):
    """A full 256 x 320 matrix: every (wave, pass) list holds 4096 entries - eight chunks of the entry loop; and a
    block-diagonal pattern whose off-diagonal macro-tiles are not items (places come from the item records)."""
    monkeypatch.setenv("BSMR_DENSE_ENGINE", "gemm")
    rows, cols = 256, 320
    ro = np.arange(rows + 1, dtype=np.uint32) * cols
    ci = np.tile(np.arange(cols, dtype=np.uint32), rows)
    check_case(engine, oracle, rows, cols, ro, ci, 128, 0.3, 0.0, 0)
    monkeypatch.setenv("BSMR_GEMM_PANELS", "8")
    monkeypatch.setenv("BSMR_GEMM_BLOCKS", "16")
    monkeypatch.setenv("BSMR_GEMM_BALANCE_COLUMNS", "0")     # natural row and column order: the off-diagonal macro-tiles stay empty
    entries = [(i, j) for i in range(600) for j in range((i // 150) * 400, (i // 150) * 400 + 400, 7)]
    ro = np.zeros(601, dtype=np.uint32)
    for i, _ in entries:
        ro[i + 1] += 1
    ro = np.cumsum(ro).astype(np.uint32)
    ci = np.array([j for _, j in entries], dtype=np.uint32)
    check_case(engine, oracle, 600, 1600, ro, ci, 64, 0.3, 0.0, 0, row_mode=engine.ROWS_IDENTITY)
    monkeypatch.setenv("BSMR_GEMM_BALANCE_COLUMNS", "1")
    check_case(engine, oracle, 600, 1600, ro, ci, 64, 0.3, 0.0, 0, row_mode=engine.ROWS_IDENTITY)


def test_tuned_choice_can_be_read_and_replayed(engine, oracle):
    """bsmr_plan_get_tuned / bsmr_plan_set_tuned: what bsmr_plan_tune chose for (K, mode) as plain numbers, installed in
    a second plan WITHOUT timing anything - the second plan then launches the first one's kernels (same dense choice,
    bit-identical results).  Choices a plan cannot run are refused with a status and leave the plan as it was."""
    rows, cols, ro, ci = synth.bernoulli(rows=1024, cols=2048, density=0.1, seed=9)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    arrays = engine.Pipeline(csr, alpha=0.3, delta=0.0, device=-1).arrays()
    dev = _dev()
    K, mode = 512, engine.COMPUTE_BF16
    A, B = engine.make_data(rows * K, 5489), engine.make_data(cols * K, 5490)
    want = oracle.sddmm_cpu(rows, cols, K, ro, ci, A, B)
    tA, tB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    tP = torch.full((csr.nnz,), float("nan"), dtype=torch.float32, device=dev)
    opts = engine.plan_options(dense_engine=engine.ENGINE_TUNED)
    st, first = engine.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0, options=opts)
    assert st == engine.OK
    with pytest.raises(engine.BsmrError):
        engine.plan_get_tuned(first, K, mode)                      # never tuned
    report = engine.plan_tune(first, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, 0)
    torch.cuda.synchronize()
    choice = engine.plan_get_tuned(first, K, mode)
    print(f"tuned: {report['chosen']} {choice}")
    assert engine.ENGINE_NAMES[choice["engine"]] == report["chosen"]
    tuned_result = tP.cpu().numpy()
    st, second = engine.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0, options=opts)
    assert st == engine.OK
    engine.plan_set_tuned(second, K, choice, mode)
    assert engine.plan_get_tuned(second, K, mode) == choice
    tP.fill_(float("nan"))
    engine.sddmm(second, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, 0)
    torch.cuda.synchronize()
    got = tP.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), tuned_result.view(np.uint32))
    bad, first_bad = oracle.check_data(want, got)
    assert bad == 0, (bad, first_bad)
    # every engine can be installed by hand; results stay the oracle's
    for eng_id, extra in ((engine.ENGINE_GEMM, dict(group=16, blocks_per_item=16)), (engine.ENGINE_GEMM, dict(group=8, blocks_per_item=20)),
                          (engine.ENGINE_SWEEP, dict(group=2, waves=4)), (engine.ENGINE_STREAM, {}), (engine.ENGINE_TILES, {})):
        c = dict(engine=eng_id, group=0, blocks_per_item=0, format=-1, b_only=-1, overlap=-1, cvt_in_kernel=-1, waves=0)
        c.update(extra)
        engine.plan_set_tuned(second, K, c, mode)
        tP.fill_(float("nan"))
        engine.sddmm(second, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, 0)
        torch.cuda.synchronize()
        bad, first_bad = oracle.check_data(want, tP.cpu().numpy())
        assert bad == 0, (eng_id, extra, bad, first_bad)
    # refused: an engine id that does not exist, a K the engine does not serve, a plan that is not tunable
    before = engine.plan_get_tuned(second, K, mode)
    import ctypes as C
    bad_choice = engine.TunedChoice()
    bad_choice.struct_size = C.sizeof(engine.TunedChoice)
    bad_choice.engine = 9
    assert engine.hip().bsmr_plan_set_tuned(second, K, mode, C.byref(bad_choice)) == engine.ERR_INVALID_ARG
    bad_choice.engine, bad_choice.format, bad_choice.b_only, bad_choice.overlap, bad_choice.cvt_in_kernel = engine.ENGINE_GEMM, -1, -1, -1, -1
    assert engine.hip().bsmr_plan_set_tuned(second, 96, mode, C.byref(bad_choice)) == engine.ERR_BAD_PLAN
    assert engine.plan_get_tuned(second, K, mode) == before
    st, plain = engine.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0, options=engine.plan_options())
    assert st == engine.OK
    assert engine.hip().bsmr_plan_set_tuned(plain, K, mode, C.byref(bad_choice)) == engine.ERR_INVALID_ARG
    for p in (first, second, plain):
        engine.plan_destroy(p)


@pytest.mark.parametrize("K,mode,fp32", [(128, 0, 1), (128, 1, 0), (256, 0, 0)])
def test_gemm_engine_batched_calls_and_graph_replay(engine, oracle, K, mode, fp32):
    """bsmr_sddmm_batch through the GEMM engine (grid y = the batch; the operand and output strides reach the kernel's
    LDS-DMA descriptors), on 16-bit copies and on the callers' fp32 operands: every batch equals the single call on its
    operands bit for bit and meets the oracle.  And the engine inside a captured graph: bsmr_plan_reserve builds the
    macro-tile format, the captured call allocates nothing, replays recompute P for new operand values."""
    rows, cols, ro, ci = synth.bernoulli(rows=700, cols=1300, density=0.08, seed=K + mode)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    arrays = engine.Pipeline(csr, alpha=0.3, delta=0.0, device=-1).arrays()
    st, plan = engine.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0,
                                       options=engine.plan_options(dense_engine=engine.ENGINE_GEMM, gemm_fp32=fp32))
    assert st == engine.OK
    try:
        nb, dev = 3, _dev()
        A = np.concatenate([engine.make_data(rows * K, 100 + b) for b in range(nb)])
        B = np.concatenate([engine.make_data(cols * K, 200 + b) for b in range(nb)])
        tA, tB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
        tP = torch.full((nb * csr.nnz,), float("nan"), dtype=torch.float32, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream
        engine.sddmm_batch(plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), nb, mode, s)
        torch.cuda.synchronize()
        got = tP.cpu().numpy().reshape(nb, csr.nnz)
        assert not np.isnan(got).any()
        one = torch.full((csr.nnz,), float("nan"), dtype=torch.float32, device=dev)
        for b in range(nb):
            a, bb = tA[b * rows * K:(b + 1) * rows * K], tB[b * cols * K:(b + 1) * cols * K]
            one.fill_(float("nan"))
            engine.sddmm(plan, K, a.data_ptr(), bb.data_ptr(), one.data_ptr(), mode, s)
            torch.cuda.synchronize()
            assert np.array_equal(got[b].view(np.uint32), one.cpu().numpy().view(np.uint32)), f"batch {b}"
            want = oracle.sddmm_cpu(rows, cols, K, ro, ci, A[b * rows * K:(b + 1) * rows * K], B[b * cols * K:(b + 1) * cols * K])
            if not (mode == 1 and K < 512):
                assert oracle.check_data(want, got[b])[0] == 0
        # captured: the single call on batch 0's operands
        assert engine.hip().bsmr_plan_reserve(plan, K) == engine.OK
        side = torch.cuda.Stream(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            engine.sddmm(plan, K, tA.data_ptr(), tB.data_ptr(), one.data_ptr(), mode, side.cuda_stream)
            side.synchronize()
            with torch.cuda.graph(graph, stream=side):
                engine.sddmm(plan, K, tA.data_ptr(), tB.data_ptr(), one.data_ptr(), mode, torch.cuda.current_stream(dev).cuda_stream)
        for _ in range(2):
            one.fill_(float("nan"))
            graph.replay()
            torch.cuda.synchronize()
            assert np.array_equal(one.cpu().numpy().view(np.uint32), got[0].view(np.uint32))
        tA[:rows * K].copy_(tA[rows * K:2 * rows * K])      # new values behind the same pointers
        tB[:cols * K].copy_(tB[cols * K:2 * cols * K])
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(one.cpu().numpy().view(np.uint32), got[1].view(np.uint32))
        import ctypes as C
        g, t, u = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
        engine.hip().bsmr_plan_dense_choice(plan, K, C.byref(g), C.byref(t), C.byref(u))
        assert g.value in (8, 16), "the GEMM engine must have served the calls"
    finally:
        engine.plan_destroy(plan)
