import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
import hostinfo  # noqa: E402

hostinfo.limit_openmp_threads()   # before any OpenMP runtime starts: honour the container's CPU quota

import numpy as np  # noqa: E402
import pytest  # noqa: E402

sys.path.insert(0, str(REPO / "oracle"))
sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "shipping_rules: the test builds its plans with the default bsmr_plan_options")


@pytest.fixture(autouse=True)
def plan_rules(request, monkeypatch):
    """Which rules bsmr_plan_create applies (it reads the BSMR_* environment as its override, bsmr_plan_options_from_env):
      "rphm"    - folding off (BSMR_FOLD_DENSE_BELOW=0): a dense part of any size runs on the dense kernels.  The
                  test matrices are small, and the dense kernels are what most tests are about; the default for
                  tests that do not ask otherwise.
      "default" - the shipping defaults (fold < 32768 dense entries, promotion, B-only conversion, ...): tests
                  marked `shipping_rules`, every BASELINE-size case, and the second leg of the tests parametrised
                  with `both_rules`."""
    mode = getattr(request, "param", None)
    if mode is None:
        mode = "default" if request.node.get_closest_marker("shipping_rules") else "rphm"
    if mode == "rphm":
        monkeypatch.setenv("BSMR_FOLD_DENSE_BELOW", "0")
    else:
        monkeypatch.delenv("BSMR_FOLD_DENSE_BELOW", raising=False)
    return mode


both_rules = pytest.mark.parametrize("plan_rules", ["rphm", "default"], indirect=True)


def _ensure_built():
    need = [REPO / "bsmr-sddmm_amd" / "lib" / "libbsmr_hip.so",
            REPO / "bsmr-sddmm_amd" / "lib" / "libbsmr_host.so",
            REPO / "oracle" / "liboracle.so",
            REPO / "tests" / "native" / "libplancheck.so"]
    if all(p.exists() for p in need):
        return
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture(scope="session")
def engine():
    _ensure_built()
    import bsmr_amd
    bsmr_amd.host()
    return bsmr_amd


class Oracle:
    """ctypes view of oracle/liboracle.so (the checker; never the thing under test)."""

    def __init__(self):
        self.lib = C.CDLL(str(REPO / "oracle" / "liboracle.so"))
        L = self.lib
        vp = C.c_void_p
        L.oracle_sddmm_cpu.argtypes = [C.c_uint32] * 3 + [vp] * 5
        L.oracle_sddmm_f64.argtypes = [C.c_uint32] * 2 + [vp] * 5
        L.oracle_check_one.argtypes = [C.c_float, C.c_float]
        L.oracle_check_one.restype = C.c_int
        L.oracle_check_data.argtypes = [C.c_uint64, vp, vp, C.POINTER(C.c_int64)]
        L.oracle_check_data.restype = C.c_uint64
        for f in ("oracle_round_tf32", "oracle_round_fp16", "oracle_round_bf16"):
            getattr(L, f).argtypes = [C.c_float]
            getattr(L, f).restype = C.c_float
        L.oracle_round_array.argtypes = [C.c_int, C.c_uint64, vp, vp]
        L.oracle_sddmm_ref_kernel_model.argtypes = [C.c_uint32] * 2 + [vp] * 6
        L.oracle_sparse_twin.argtypes = [C.c_uint32] * 3 + [vp] * 5
        L.oracle_dense_f32_twin.argtypes = [C.c_uint32] * 2 + [vp] * 5
        L.oracle_dense_lowp_model.argtypes = [C.c_int] + [C.c_uint32] * 2 + [vp] * 5
        L.oracle_num_threads.restype = C.c_int
        L.oracle_bsa_row_reordering.argtypes = [C.c_uint32] * 2 + [vp, vp, C.c_uint32, C.c_float, vp, vp, vp]
        L.oracle_bsa_row_reordering.restype = C.c_int
        L.oracle_cluster_similarity.argtypes = [vp, vp, C.c_uint32]
        L.oracle_cluster_similarity.restype = C.c_float
        L.oracle_cluster_threads.argtypes = [C.c_uint32]
        L.oracle_cluster_threads.restype = C.c_uint32
        L.oracle_cluster_bin_mask.argtypes = [C.c_uint32, vp]

    def bsa_row_reordering(self, rows, cols, ro, ci, bin_width, alpha):
        """oracle/clustering_oracle.c: (reorderedRows, numClusters) as the reference computes them"""
        ro = np.ascontiguousarray(ro, dtype=np.uint32)
        ci = np.ascontiguousarray(ci, dtype=np.uint32)
        perm = np.zeros(max(rows, 1), dtype=np.uint32)
        n_out, clusters = C.c_uint32(0), C.c_int32(0)
        st = self.lib.oracle_bsa_row_reordering(rows, cols, self._p(ro), self._p(ci), bin_width, alpha,
                                                self._p(perm), C.byref(n_out), C.byref(clusters))
        assert st == 0, "oracle could not allocate the rows x bins table"
        return perm[:n_out.value].copy(), clusters.value

    def cluster_similarity(self, rep, cmp_):
        rep = np.ascontiguousarray(rep, dtype=np.uint32)
        cmp_ = np.ascontiguousarray(cmp_, dtype=np.uint32)
        return float(self.lib.oracle_cluster_similarity(self._p(rep), self._p(cmp_), rep.size))

    def cluster_bin_mask(self, num_bins):
        mask = np.zeros(num_bins, dtype=np.uint8)
        self.lib.oracle_cluster_bin_mask(num_bins, self._p(mask))
        return mask.astype(bool)

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p)

    def sddmm_cpu(self, M, N, K, ro, ci, A, B):
        P = np.empty(ci.size, dtype=np.float32)
        self.lib.oracle_sddmm_cpu(M, N, K, self._p(ro), self._p(ci), self._p(A), self._p(B), self._p(P))
        return P

    def sddmm_f64(self, M, K, ro, ci, A, B):
        P = np.empty(ci.size, dtype=np.float64)
        self.lib.oracle_sddmm_f64(M, K, self._p(ro), self._p(ci), self._p(A), self._p(B), self._p(P))
        return P

    def check_data(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float32)
        first = C.c_int64(-1)
        n = self.lib.oracle_check_data(x.size, self._p(x), self._p(y), C.byref(first))
        return int(n), int(first.value)

    def round_array(self, mode, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.empty_like(a)
        self.lib.oracle_round_array(mode, a.size, self._p(a), self._p(out))
        return out

    def ref_kernel_model(self, M, K, ro, ci, is_dense, A, B):
        P = np.empty(ci.size, dtype=np.float32)
        d = np.ascontiguousarray(is_dense, dtype=np.uint8)
        self.lib.oracle_sddmm_ref_kernel_model(M, K, self._p(ro), self._p(ci), self._p(d), self._p(A),
                                               self._p(B), self._p(P))
        return P

    def sparse_twin(self, M, K, lpe, ro, ci, A, B):
        P = np.empty(ci.size, dtype=np.float32)
        self.lib.oracle_sparse_twin(M, K, lpe, self._p(ro), self._p(ci), self._p(A), self._p(B), self._p(P))
        return P

    def dense_f32_twin(self, M, K, ro, ci, A, B):
        P = np.empty(ci.size, dtype=np.float32)
        self.lib.oracle_dense_f32_twin(M, K, self._p(ro), self._p(ci), self._p(A), self._p(B), self._p(P))
        return P

    def dense_lowp_model(self, mode, M, K, ro, ci, A, B):
        P = np.empty(ci.size, dtype=np.float64)
        self.lib.oracle_dense_lowp_model(mode, M, K, self._p(ro), self._p(ci), self._p(A), self._p(B),
                                         self._p(P))
        return P


@pytest.fixture(scope="session")
def oracle():
    _ensure_built()
    return Oracle()
