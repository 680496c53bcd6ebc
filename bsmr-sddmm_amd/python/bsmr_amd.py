"""ctypes bindings of the MI355X BSMR-SDDMM engine.

Binds the two C ABIs declared in /include:
  * bsmr_hip.h  (lib/libbsmr_hip.so)  -- device plan + HIP kernels
  * bsmr_host.h (lib/libbsmr_host.so) -- host BSMR pipeline (C++/OpenMP)

This module is plumbing for tests and bench.py; it contains no arithmetic.  If
the libraries are missing it raises: there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent.parent
LIB_DIR = PKG_DIR / "lib"

OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED_K, ERR_OOM, ERR_BAD_PLAN = 1, 2, 3, 4, 5, 6
COMPUTE_F16, COMPUTE_BF16, COMPUTE_F32 = 0, 1, 2
ROWS_CLUSTER, ROWS_IDENTITY = 0, 1

ARRAY_IDS = {
    "reorderedRows": 0, "denseCols": 1, "denseColOffsets": 2, "sparseCols": 3,
    "sparseColOffsets": 4, "sparseValueOffsets": 5, "blockOffsets": 6, "blockValues": 7,
    "sparseValues": 8, "sparseRelativeRows": 9, "sparseColIndices": 10,
    "denseRowPanelIds": 11, "denseColBlockIters": 12, "sparseRowPanelIds": 13,
    "sparseColBlockIters": 14,
}

u32p = C.POINTER(C.c_uint32)
f32p = C.POINTER(C.c_float)


class RphmDesc(C.Structure):
    _fields_ = [("M", C.c_uint32), ("N", C.c_uint32), ("nnz", C.c_uint32),
                ("num_row_panels", C.c_uint32), ("num_nonzero_rows", C.c_uint32),
                ("reordered_rows", u32p), ("dense_cols", u32p), ("block_offsets", u32p),
                ("block_values", u32p), ("sparse_value_offsets", u32p), ("sparse_values", u32p),
                ("sparse_relative_rows", u32p), ("sparse_col_indices", u32p)]


# BSMR_VARIANT_* (include/bsmr_hip.h): the plan-time rules bsmr_plan_tune tries for a plan created with k_hint > 0
VARIANT_NAMES = ("rules", "as_rphm", "no_promotion", "promote_24", "promote_all", "all_residue")


class TuneReport(C.Structure):
    _fields_ = [("chosen_engine", C.c_int32), ("chosen_group", C.c_int32), ("chosen_blocks_per_item", C.c_int32),
                ("stream_us", C.c_float), ("grouped_us", C.c_float), ("tiles_us", C.c_float), ("shared_us", C.c_float),
                ("chosen_b_only", C.c_int32), ("fp32_residue_us", C.c_float), ("b_only_us", C.c_float),
                ("chosen_overlap", C.c_int32), ("one_stream_us", C.c_float), ("two_streams_us", C.c_float),
                ("chosen_cvt_in_kernel", C.c_int32), ("convert_pass_us", C.c_float), ("fp32_dense_us", C.c_float),
                ("sweep_us", C.c_float), ("lowp_call_us", C.c_float), ("sweep_fp32_call_us", C.c_float),
                ("chosen_sweep_fp32", C.c_int32), ("chosen_variant", C.c_int32), ("variant_us", C.c_float * 6),
                ("gemm_us", C.c_float), ("gemm_fp32_call_us", C.c_float), ("chosen_gemm_fp32", C.c_int32)]


class TunedChoice(C.Structure):
    """bsmr_tuned_choice: what a tuned plan keeps per (K, mode) (bsmr_plan_get_tuned / bsmr_plan_set_tuned)"""
    _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_int32) for n in (
        "engine", "group", "blocks_per_item", "format", "b_only", "overlap", "cvt_in_kernel", "waves")]


class PlanBuildMs(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("rules_ms", "pack_ms", "upload_ms", "second_format_ms", "total_ms")]


class PlanStats(C.Structure):
    _fields_ = [("num_row_panels", C.c_uint32), ("num_dense_blocks", C.c_uint64),
                ("num_dense_entries", C.c_uint64), ("num_sparse_entries", C.c_uint64),
                ("dense_work_items", C.c_uint64), ("sparse_work_items", C.c_uint64),
                ("device_index_bytes", C.c_uint64), ("group_size", C.c_uint32),
                ("num_dense_tiles", C.c_uint64), ("union_columns", C.c_uint64),
                ("grouped_group_size", C.c_uint32), ("grouped_dense_tiles", C.c_uint64),
                ("grouped_union_columns", C.c_uint64), ("sparse_lowp", C.c_uint64),
                ("folded_dense_entries", C.c_uint64), ("free_residue", C.c_uint64),
                ("promoted_sparse_entries", C.c_uint64)]


class PlanOptions(C.Structure):
    """bsmr_plan_options (include/bsmr_hip.h): every rule of plan construction that is not in the RPHM arrays"""
    _fields_ = [("struct_size", C.c_uint32)] + [(name, C.c_int32) for name in (
        "dense_engine", "fold_dense_below", "promote_average", "promote_min_entries_k", "promote_column_degree",
        "promote_head", "dense_group", "dense_blocks_per_item", "stream_waves", "output_mode", "force_tile32",
        "column_order", "dense_stream", "dense_batch", "tile_group", "tile_blocks_per_item", "tile_depth",
        "sparse_entries_per_item", "sparse_lowp", "sparse_lpe", "free_residue", "convert_in_kernel", "convert_sliced",
        "b_only", "b_only_work_m", "overlap_streams", "mask_tiles", "pack_on_device", "sweep_panels", "sweep_strip_blocks",
        "sweep_fp32", "sweep_waves", "sweep_per_cu", "k_hint", "promote_on_device", "gemm_panels", "gemm_blocks", "gemm_fp32", "gemm_balance_columns", "evict_wide_rows")]


ENGINE_STREAM, ENGINE_TILES, ENGINE_SHARED, ENGINE_TUNED, ENGINE_SWEEP, ENGINE_GEMM = 0, 1, 2, 3, 4, 5
ENGINE_NAMES = {0: "stream", 1: "tiles", 2: "shared", 4: "sweep", 5: "gemm"}


class ReorderingReport(C.Structure):
    _fields_ = [("original_num_dense_blocks", C.c_int32), ("original_average_density", C.c_float),
                ("num_dense_blocks", C.c_int32), ("average_density", C.c_float),
                ("num_dense_thread_blocks", C.c_int32), ("num_sparse_thread_blocks", C.c_int32),
                ("num_dense_data", C.c_int32), ("num_sparse_data", C.c_int32),
                ("max_dense_blocks_per_panel", C.c_uint32), ("max_sparse_blocks_per_panel", C.c_uint32)]


class ClusterStats(C.Structure):
    _fields_ = [("elapsed_ms", C.c_float), ("passes", C.c_uint32), ("similarities", C.c_uint32),
                ("exact_similarities", C.c_uint32), ("threads_per_pair", C.c_uint32), ("table_bytes", C.c_uint64),
                ("dropped_seeds", C.c_uint32), ("passes_ahead", C.c_uint32)]


class ColReorderSizes(C.Structure):
    _fields_ = [("num_row_panels", C.c_uint32), ("num_dense_cols", C.c_uint64), ("num_sparse_cols", C.c_uint64),
                ("num_blocks", C.c_uint64), ("num_sparse_entries", C.c_uint64), ("elapsed_ms", C.c_float)]


class ShardedTiming(C.Structure):
    _fields_ = [("step_ms", C.c_float), ("wall_ms", C.c_float), ("num_devices", C.c_uint32), ("compute_ms", C.c_float),
                ("gather_ms", C.c_float)]


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("convert_ms", C.c_float), ("dense_ms", C.c_float),
                ("sparse_ms", C.c_float)]


# every symbol the two headers declare: name -> (restype, argtypes)
HIP_SYMBOLS = {
    "bsmr_strerror": (C.c_char_p, [C.c_int]),
    "bsmr_last_hip_error": (C.c_char_p, []),
    "bsmr_abi_revision": (C.c_int, []),
    "bsmr_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "bsmr_mem_info": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "bsmr_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "bsmr_dev_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "bsmr_dev_free": (C.c_int, [C.c_void_p]),
    "bsmr_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "bsmr_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "bsmr_dev_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    "bsmr_device_synchronize": (C.c_int, [C.c_int]),
    "bsmr_plan_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(RphmDesc)]),
    "bsmr_plan_create_ex": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(RphmDesc), C.POINTER(PlanOptions)]),
    "bsmr_plan_options_default": (C.c_int, [C.POINTER(PlanOptions)]),
    "bsmr_plan_options_from_env": (C.c_int, [C.POINTER(PlanOptions)]),
    "bsmr_plan_destroy": (C.c_int, [C.c_void_p]),
    "bsmr_plan_promoted_on_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "bsmr_col_reorder_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "bsmr_plan_create_from_colreorder": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, u32p, C.c_uint32,
                                                   C.c_void_p]),
    "bsmr_plan_get_stats": (C.c_int, [C.c_void_p, C.POINTER(PlanStats)]),
    "bsmr_plan_build_times": (C.c_int, [C.c_void_p, C.POINTER(PlanBuildMs)]),
    "bsmr_plan_format_digest": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "bsmr_plan_entry_lists_digest": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "bsmr_plan_tune": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                C.POINTER(TuneReport)]),
    "bsmr_cluster_rows": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float,
                                    C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_void_p]),
    "bsmr_cluster_rows_sized": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float,
                                          C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_void_p, C.c_size_t]),
    "bsmr_plan_tune_sized": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.POINTER(TuneReport), C.c_size_t]),
    "bsmr_plan_get_tuned": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(TunedChoice)]),
    "bsmr_plan_set_tuned": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(TunedChoice)]),
    "bsmr_sddmm_batch": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int,
                                   C.c_void_p]),
    "bsmr_batched_transpose": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsmr_plan_sparse_choice": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint32)]),
    "bsmr_plan_dense_choice": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64)]),
    "bsmr_plan_reserve": (C.c_int, [C.c_void_p, C.c_uint32]),
    "bsmr_plan_dense_flags": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bsmr_sddmm": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                             C.c_void_p]),
    "bsmr_sddmm_timed": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(Timing)]),
    "bsmr_convert_operands": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_void_p]),
    "bsmr_sddmm_lowp": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "bsmr_sddmm_host": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "bsmr_col_reorder": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_uint32, C.c_float]),
    "bsmr_col_reorder_sizes": (C.c_int, [C.c_void_p, C.POINTER(ColReorderSizes)]),
    "bsmr_col_reorder_fetch": (C.c_int, [C.c_void_p] + [C.c_void_p] * 10),
    "bsmr_col_reorder_free": (C.c_int, [C.c_void_p]),
    "bsmr_sharded_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_uint32,
                                      C.POINTER(C.POINTER(RphmDesc)), C.POINTER(C.c_uint32), C.POINTER(PlanOptions)]),
    "bsmr_sharded_destroy": (C.c_int, [C.c_void_p]),
    "bsmr_sharded_num_entries": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bsmr_sharded_sddmm_host": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                          C.POINTER(ShardedTiming)]),
}

HOST_SYMBOLS = {
    "bsmr_csr_from_file": (C.c_void_p, [C.c_char_p]),
    "bsmr_csr_from_arrays": (C.c_void_p, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "bsmr_csr_free": (None, [C.c_void_p]),
    "bsmr_csr_rows": (C.c_uint32, [C.c_void_p]),
    "bsmr_csr_cols": (C.c_uint32, [C.c_void_p]),
    "bsmr_csr_nnz": (C.c_uint32, [C.c_void_p]),
    "bsmr_csr_row_offsets": (u32p, [C.c_void_p]),
    "bsmr_csr_col_indices": (u32p, [C.c_void_p]),
    "bsmr_csr_values": (f32p, [C.c_void_p]),
    "bsmr_csr_check": (C.c_int, [C.c_void_p]),
    "bsmr_csr_write_mtx": (C.c_int, [C.c_void_p, C.c_char_p]),
    "bsmr_make_data": (None, [C.c_void_p, C.c_size_t, C.c_uint32]),
    "bsmr_calculate_block_size": (C.c_uint32, [C.c_void_p, C.c_size_t]),
    "bsmr_pipeline_create": (C.c_void_p, [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_uint32, C.c_int]),
    "bsmr_pipeline_free": (None, [C.c_void_p]),
    "bsmr_pipeline_resplit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int]),
    "bsmr_pipeline_array": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(u32p), C.POINTER(C.c_size_t)]),
    "bsmr_pipeline_num_row_panels": (C.c_int, [C.c_void_p]),
    "bsmr_pipeline_num_clusters": (C.c_int, [C.c_void_p]),
    "bsmr_pipeline_row_reordering_ms": (C.c_float, [C.c_void_p]),
    "bsmr_pipeline_col_reordering_ms": (C.c_float, [C.c_void_p]),
    "bsmr_pipeline_rphm_ms": (C.c_float, [C.c_void_p]),
    "bsmr_pipeline_check": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float]),
    "bsmr_pipeline_evaluate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "bsmr_pipeline_plan": (C.c_void_p, [C.c_void_p]),
    "bsmr_pipeline_plan_status": (C.c_int, [C.c_void_p]),
    "bsmr_host_sddmm_cpu": (None, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bsmr_host_check_data": (C.c_size_t, [C.c_size_t, C.c_void_p, C.c_void_p]),
    "bsmr_host_sddmm": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]),
    "bsmr_host_sddmm_sharded": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_int,
                                          C.POINTER(C.c_int), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_float)]),
    "bsmr_host_sddmm_sharded_timed": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_int,
                                                C.POINTER(C.c_int), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.POINTER(C.c_float)]),
    "bsmr_partition_rows_by_cost": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
}


def _load(name: str, symbols: dict):
    path = LIB_DIR / name
    if not path.exists():
        raise RuntimeError(
            f"{path} is missing: build it with `make -C {PKG_DIR}` (or __graft_entry__.build()). "
            "The engine has no CPU fallback.")
    lib = C.CDLL(str(path), mode=C.RTLD_GLOBAL)
    for sym, (res, args) in symbols.items():
        fn = getattr(lib, sym)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    return lib


_hip = None
_host = None


def hip():
    global _hip
    if _hip is None:
        _hip = _load("libbsmr_hip.so", HIP_SYMBOLS)
    return _hip


def host():
    global _host
    if _host is None:
        hip()
        _host = _load("libbsmr_host.so", HOST_SYMBOLS)
    return _host


class BsmrError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        msg = hip().bsmr_strerror(status).decode()
        detail = hip().bsmr_last_hip_error().decode()
        super().__init__(f"{where}: {msg}" + (f" ({detail})" if detail else ""))


def _check(status: int, where: str):
    if status != OK:
        raise BsmrError(status, where)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    n = C.c_int(0)
    _check(hip().bsmr_device_count(C.byref(n)), "bsmr_device_count")
    return n.value


def make_data(count: int, seed: int) -> np.ndarray:
    """U[0,2) operands, bit-identical to Matrix<float>::makeDataSeeded(seed)."""
    out = np.empty(count, dtype=np.float32)
    host().bsmr_make_data(_ptr(out), count, seed)
    return out


class CSR:
    """sparseMatrix::CSR<float> (pattern owner)."""

    def __init__(self, handle):
        if not handle:
            raise ValueError("CSR construction failed (the loader returned false)")
        self._h = C.c_void_p(handle)

    @classmethod
    def from_file(cls, path) -> "CSR":
        return cls(host().bsmr_csr_from_file(str(path).encode()))

    @classmethod
    def from_arrays(cls, rows, cols, row_offsets, col_indices) -> "CSR":
        ro = np.ascontiguousarray(row_offsets, dtype=np.uint32)
        ci = np.ascontiguousarray(col_indices, dtype=np.uint32)
        assert ro.size == rows + 1
        return cls(host().bsmr_csr_from_arrays(rows, cols, int(ci.size), _ptr(ro), _ptr(ci)))

    def __del__(self):
        if getattr(self, "_h", None) and _host is not None:
            _host.bsmr_csr_free(self._h)
            self._h = None

    @property
    def handle(self):
        return self._h

    rows = property(lambda s: host().bsmr_csr_rows(s._h))
    cols = property(lambda s: host().bsmr_csr_cols(s._h))
    nnz = property(lambda s: host().bsmr_csr_nnz(s._h))

    @property
    def row_offsets(self) -> np.ndarray:
        return np.ctypeslib.as_array(host().bsmr_csr_row_offsets(self._h), (self.rows + 1,)).copy()

    @property
    def col_indices(self) -> np.ndarray:
        if self.nnz == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(host().bsmr_csr_col_indices(self._h), (self.nnz,)).copy()

    @property
    def values(self) -> np.ndarray:
        if self.nnz == 0:
            return np.zeros(0, dtype=np.float32)
        return np.ctypeslib.as_array(host().bsmr_csr_values(self._h), (self.nnz,)).copy()

    def check(self) -> bool:
        return bool(host().bsmr_csr_check(self._h))

    def write_mtx(self, path) -> bool:
        return bool(host().bsmr_csr_write_mtx(self._h, str(path).encode()))

    def calculate_block_size(self, free_device_bytes: int) -> int:
        return host().bsmr_calculate_block_size(self._h, free_device_bytes)


class Pipeline:
    """BSMR(alpha, delta, S) + RPHM(S, bsmr).  device=-1 keeps everything on the host."""

    def __init__(self, csr: CSR, alpha=0.3, delta=0.3, row_mode=ROWS_CLUSTER, block_size=0, device=-1):
        self.csr = csr
        self.delta = delta
        h = host().bsmr_pipeline_create(csr.handle, alpha, delta, row_mode, block_size, device)
        if not h:
            raise RuntimeError("bsmr_pipeline_create failed")
        self._h = C.c_void_p(h)
        if device >= 0:
            _check(host().bsmr_pipeline_plan_status(self._h), "bsmr_plan_create")

    def __del__(self):
        if getattr(self, "_h", None) and _host is not None:
            _host.bsmr_pipeline_free(self._h)
            self._h = None

    def resplit(self, delta: float, device=-1):
        _check(host().bsmr_pipeline_resplit(self._h, self.csr.handle, delta, device), "resplit")
        self.delta = delta

    def array(self, name: str) -> np.ndarray:
        data = u32p()
        n = C.c_size_t(0)
        _check(host().bsmr_pipeline_array(self._h, ARRAY_IDS[name], C.byref(data), C.byref(n)), name)
        if n.value == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(data, (n.value,)).copy()

    def arrays(self) -> dict:
        return {k: self.array(k) for k in ARRAY_IDS}

    num_row_panels = property(lambda s: host().bsmr_pipeline_num_row_panels(s._h))
    num_clusters = property(lambda s: host().bsmr_pipeline_num_clusters(s._h))
    row_reordering_ms = property(lambda s: host().bsmr_pipeline_row_reordering_ms(s._h))
    col_reordering_ms = property(lambda s: host().bsmr_pipeline_col_reordering_ms(s._h))
    rphm_ms = property(lambda s: host().bsmr_pipeline_rphm_ms(s._h))

    def check(self) -> bool:
        return bool(host().bsmr_pipeline_check(self._h, self.csr.handle, self.delta))

    def evaluate(self) -> dict:
        """evaluationReordering: the statistics the reference logs for this (alpha, delta)."""
        rep = ReorderingReport()
        _check(host().bsmr_pipeline_evaluate(self._h, self.csr.handle, self.delta, C.byref(rep)),
               "bsmr_pipeline_evaluate")
        return {name: getattr(rep, name) for name, _ in ReorderingReport._fields_}

    @property
    def plan(self):
        p = host().bsmr_pipeline_plan(self._h)
        if not p:
            raise RuntimeError("this pipeline has no device plan (built with device=-1?)")
        return C.c_void_p(p)

    def plan_stats(self) -> dict:
        s = PlanStats()
        _check(hip().bsmr_plan_get_stats(self.plan, C.byref(s)), "bsmr_plan_get_stats")
        return {k: getattr(s, k) for k, _ in PlanStats._fields_}

    def plan_build_ms(self) -> dict:
        """Host wall time of bsmr_plan_create for this pipeline's plan (rules, packing, upload, second format, total)."""
        t = PlanBuildMs()
        _check(hip().bsmr_plan_build_times(self.plan, C.byref(t)), "bsmr_plan_build_times")
        return {k: round(getattr(t, k), 3) for k, _ in PlanBuildMs._fields_}

    def dense_flags(self) -> np.ndarray:
        """uint8 per stored entry (CSR order): 1 = computed by the dense (MFMA) path of the plan, 0 = residue."""
        flags = np.zeros(self.csr.nnz, dtype=np.uint8)
        _check(hip().bsmr_plan_dense_flags(self.plan, flags.ctypes.data_as(C.c_void_p)), "dense_flags")
        return flags

    def sparse_choice(self, K: int, mode=COMPUTE_F16) -> dict:
        lanes, lowp = C.c_uint32(0), C.c_uint32(0)
        _check(hip().bsmr_plan_sparse_choice(self.plan, K, mode, C.byref(lanes), C.byref(lowp)), "sparse_choice")
        return {"lanes_per_entry": lanes.value, "low_precision": bool(lowp.value)}

    def dense_choice(self, K: int) -> dict:
        """Dense format used by a call with inner dimension K."""
        g, t, u = C.c_uint32(0), C.c_uint64(0), C.c_uint64(0)
        _check(hip().bsmr_plan_dense_choice(self.plan, K, C.byref(g), C.byref(t), C.byref(u)), "dense_choice")
        return {"group_size": g.value, "tiles": t.value, "union_columns": u.value}


def cluster_rows_device(rows, cols, row_offsets, col_indices, bin_width, alpha, device=0):
    """bsmr_cluster_rows: (status, reorderedRows, numClusters, stats) - the row clustering on the GPU"""
    ro = np.ascontiguousarray(row_offsets, dtype=np.uint32)
    ci = np.ascontiguousarray(col_indices, dtype=np.uint32)
    out = np.zeros(max(rows, 1), dtype=np.uint32)
    n, clusters, stats = C.c_uint32(0), C.c_int32(0), ClusterStats()
    st = hip().bsmr_cluster_rows(device, rows, cols, _ptr(ro), _ptr(ci), bin_width, alpha, _ptr(out),
                                 C.byref(n), C.byref(clusters), C.byref(stats))
    return st, out[:n.value].copy(), clusters.value, {k: getattr(stats, k) for k, _ in ClusterStats._fields_}


def sddmm_cpu(csr: CSR, K: int, A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """The engine's own OpenMP sddmm_cpu (host.cpp)."""
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    P = np.empty(csr.nnz, dtype=np.float32)
    host().bsmr_host_sddmm_cpu(csr.handle, K, _ptr(A), _ptr(B), _ptr(P))
    return P


def check_data(x: np.ndarray, y: np.ndarray) -> int:
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    assert x.size == y.size
    return int(host().bsmr_host_check_data(x.size, _ptr(x), _ptr(y)))


def sddmm_operator(csr: CSR, K: int, A, B, alpha=0.3, delta=0.3, mode=COMPUTE_F16, iters=1):
    """sddmm(options, A, B, P, logger) on host operands -> (P, log text)."""
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    P = np.empty(csr.nnz, dtype=np.float32)
    log = C.create_string_buffer(8192)
    _check(host().bsmr_host_sddmm(csr.handle, K, alpha, delta, mode, iters, _ptr(A), _ptr(B), _ptr(P),
                                  log, len(log)), "bsmr_host_sddmm")
    return P, log.value.decode()


def col_reorder_device(rows, cols, row_offsets, col_indices, reordered_rows, delta, device=0):
    """bsmr_col_reorder: (status, dict of the ten arrays named as in ARRAY_IDS, device ms)"""
    ro = np.ascontiguousarray(row_offsets, dtype=np.uint32)
    ci = np.ascontiguousarray(col_indices, dtype=np.uint32)
    rr = np.ascontiguousarray(reordered_rows, dtype=np.uint32)
    h = C.c_void_p()
    st = hip().bsmr_col_reorder(C.byref(h), device, rows, cols, _ptr(ro), _ptr(ci), _ptr(rr), rr.size, delta)
    if st != OK:
        return st, None, 0.0
    sz = ColReorderSizes()
    hip().bsmr_col_reorder_sizes(h, C.byref(sz))
    P1 = sz.num_row_panels + 1
    out = {"denseCols": np.zeros(sz.num_dense_cols, np.uint32), "denseColOffsets": np.zeros(P1, np.uint32),
           "sparseCols": np.zeros(sz.num_sparse_cols, np.uint32), "sparseColOffsets": np.zeros(P1, np.uint32),
           "sparseValueOffsets": np.zeros(P1, np.uint32), "blockOffsets": np.zeros(P1, np.uint32),
           "blockValues": np.zeros(sz.num_blocks * 256, np.uint32), "sparseValues": np.zeros(sz.num_sparse_entries, np.uint32),
           "sparseRelativeRows": np.zeros(sz.num_sparse_entries, np.uint32),
           "sparseColIndices": np.zeros(sz.num_sparse_entries, np.uint32)}
    order = ("denseCols", "denseColOffsets", "sparseCols", "sparseColOffsets", "sparseValueOffsets", "blockOffsets",
             "blockValues", "sparseValues", "sparseRelativeRows", "sparseColIndices")
    st = hip().bsmr_col_reorder_fetch(h, *[_ptr(out[k]) for k in order])
    hip().bsmr_col_reorder_free(h)
    return st, out, sz.elapsed_ms


def partition_rows_by_cost(csr: CSR, world: int) -> list:
    """partitionRowsByCost: world + 1 row boundaries of the C++ operator's cut"""
    b = (C.c_uint32 * (world + 1))()
    _check(host().bsmr_partition_rows_by_cost(csr.handle, world, b), "bsmr_partition_rows_by_cost")
    return [int(x) for x in b]


def sddmm_operator_sharded_timed(csr: CSR, K: int, A, B, devices, alpha=0.3, delta=0.3, mode=COMPUTE_F16, iters=1):
    """sddmm_multi_gpu on host operands -> (P, {step_ms (pipelined), compute_ms, gather_ms (one step taken apart)})"""
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    P = np.empty(csr.nnz, dtype=np.float32)
    devs = (C.c_int * len(devices))(*devices)
    t = (C.c_float * 3)()
    _check(host().bsmr_host_sddmm_sharded_timed(csr.handle, K, alpha, delta, mode, iters, devs, len(devices), _ptr(A), _ptr(B),
                                                _ptr(P), t), "bsmr_host_sddmm_sharded_timed")
    return P, {"step_ms": t[0], "compute_ms": t[1], "gather_ms": t[2]}


def sddmm_operator_sharded(csr: CSR, K: int, A, B, devices, alpha=0.3, delta=0.3, mode=COMPUTE_F16, iters=1):
    """sddmm_multi_gpu on host operands -> (P, device ms per step)"""
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    P = np.empty(csr.nnz, dtype=np.float32)
    devs = (C.c_int * len(devices))(*devices)
    ms = C.c_float(0)
    _check(host().bsmr_host_sddmm_sharded(csr.handle, K, alpha, delta, mode, iters, devs, len(devices), _ptr(A), _ptr(B),
                                          _ptr(P), C.byref(ms)), "bsmr_host_sddmm_sharded")
    return P, ms.value


# --- device entry points (pointers are integers, e.g. torch.Tensor.data_ptr()) ---
def sddmm(plan, K: int, A_ptr: int, B_ptr: int, P_ptr: int, mode=COMPUTE_F16, stream: int = 0):
    _check(hip().bsmr_sddmm(plan, K, A_ptr, B_ptr, P_ptr, mode, stream), "bsmr_sddmm")


def plan_tune(plan, K: int, A_ptr: int, B_ptr: int, P_ptr: int, mode=COMPUTE_F16, stream: int = 0) -> dict:
    """bsmr_plan_tune: time the dense engines for calls with this (K, mode) on these operands and keep the fastest
    (the plan must have been created with dense_engine = ENGINE_TUNED).  P holds the result afterwards."""
    r = TuneReport()
    _check(hip().bsmr_plan_tune(plan, K, A_ptr, B_ptr, P_ptr, mode, stream, C.byref(r)), "bsmr_plan_tune")
    out = {"chosen": ENGINE_NAMES[r.chosen_engine], "group": r.chosen_group, "blocks_per_item": r.chosen_blocks_per_item}
    for name in ("stream_us", "grouped_us", "tiles_us", "shared_us", "fp32_residue_us", "b_only_us", "one_stream_us", "two_streams_us", "convert_pass_us", "fp32_dense_us", "sweep_us", "lowp_call_us", "sweep_fp32_call_us", "gemm_us", "gemm_fp32_call_us"):
        out[name] = round(getattr(r, name), 2)
    out["b_only"], out["overlap"], out["cvt_in_kernel"] = r.chosen_b_only, r.chosen_overlap, r.chosen_cvt_in_kernel
    out["sweep_fp32"] = r.chosen_sweep_fp32
    out["gemm_fp32"] = r.chosen_gemm_fp32
    out["variant"] = VARIANT_NAMES[r.chosen_variant]
    out["variant_us"] = {VARIANT_NAMES[i]: round(r.variant_us[i], 2) for i in range(6) if r.variant_us[i] >= 0}
    return out


def plan_get_tuned(plan, K: int, mode=COMPUTE_F16) -> dict:
    """bsmr_plan_get_tuned: the choice bsmr_plan_tune kept for (K, mode), as a plain dict (JSON-able)"""
    c = TunedChoice()
    _check(hip().bsmr_plan_get_tuned(plan, K, mode, C.byref(c)), "bsmr_plan_get_tuned")
    return {n: getattr(c, n) for n, _ in TunedChoice._fields_ if n != "struct_size"}


def plan_set_tuned(plan, K: int, choice: dict, mode=COMPUTE_F16):
    """bsmr_plan_set_tuned: install a choice measured earlier (plan_get_tuned of another process) without timing anything"""
    c = TunedChoice()
    c.struct_size = C.sizeof(TunedChoice)
    for n, _ in TunedChoice._fields_:
        if n != "struct_size":
            setattr(c, n, int(choice[n]))
    _check(hip().bsmr_plan_set_tuned(plan, K, mode, C.byref(c)), "bsmr_plan_set_tuned")


def sddmm_batch(plan, K: int, A_ptr: int, B_ptr: int, P_ptr: int, num_batches: int, mode=COMPUTE_F16, stream: int = 0):
    _check(hip().bsmr_sddmm_batch(plan, K, A_ptr, B_ptr, P_ptr, num_batches, mode, stream), "bsmr_sddmm_batch")


def batched_transpose(width: int, height: int, num_batches: int, in_ptr: int, out_ptr: int, stream: int = 0):
    _check(hip().bsmr_batched_transpose(width, height, num_batches, in_ptr, out_ptr, stream), "bsmr_batched_transpose")


def sddmm_timed(plan, K, A_ptr, B_ptr, P_ptr, mode=COMPUTE_F16, stream=0, warmup=2, iters=20) -> dict:
    t = Timing()
    _check(hip().bsmr_sddmm_timed(plan, K, A_ptr, B_ptr, P_ptr, mode, stream, warmup, iters, C.byref(t)),
           "bsmr_sddmm_timed")
    return {k: getattr(t, k) for k, _ in Timing._fields_}


def convert_operands(plan, K, A_ptr, B_ptr, A16_ptr, B16_ptr, mode=COMPUTE_F16, stream=0):
    _check(hip().bsmr_convert_operands(plan, K, A_ptr, B_ptr, A16_ptr, B16_ptr, mode, stream),
           "bsmr_convert_operands")


def sddmm_lowp(plan, K, A16_ptr, B16_ptr, A_ptr, B_ptr, P_ptr, mode=COMPUTE_F16, stream=0):
    _check(hip().bsmr_sddmm_lowp(plan, K, A16_ptr, B16_ptr, A_ptr, B_ptr, P_ptr, mode, stream),
           "bsmr_sddmm_lowp")


def sddmm_host(plan, K, A: np.ndarray, B: np.ndarray, nnz: int, mode=COMPUTE_F16, iters=1):
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    P = np.empty(nnz, dtype=np.float32)
    ms = C.c_float(0)
    _check(hip().bsmr_sddmm_host(plan, K, _ptr(A), _ptr(B), _ptr(P), mode, iters, C.byref(ms)),
           "bsmr_sddmm_host")
    return P, ms.value


def plan_options(**changes) -> PlanOptions:
    """bsmr_plan_options_default with fields changed by keyword"""
    o = PlanOptions()
    _check(hip().bsmr_plan_options_default(C.byref(o)), "bsmr_plan_options_default")
    for k, v in changes.items():
        if k not in dict(PlanOptions._fields_):
            raise KeyError(k)
        setattr(o, k, v)
    return o


def plan_from_arrays(M, N, nnz, arrays: dict, device=0, options: PlanOptions = None):
    """bsmr_plan_create straight from RPHM-layout numpy arrays (what a reference
    maintainer would pass from RPHM's host vectors).  Returns the plan handle.
    options: a PlanOptions -> bsmr_plan_create_ex (the environment is then not consulted)."""
    keep = {k: np.ascontiguousarray(arrays[k], dtype=np.uint32) for k in
            ("reorderedRows", "denseCols", "blockOffsets", "blockValues", "sparseValueOffsets",
             "sparseValues", "sparseRelativeRows", "sparseColIndices")}
    d = RphmDesc()
    d.M, d.N, d.nnz = M, N, nnz
    d.num_nonzero_rows = keep["reorderedRows"].size
    d.num_row_panels = keep["blockOffsets"].size - 1
    cast = lambda a: a.ctypes.data_as(u32p)
    d.reordered_rows = cast(keep["reorderedRows"])
    d.dense_cols = cast(keep["denseCols"])
    d.block_offsets = cast(keep["blockOffsets"])
    d.block_values = cast(keep["blockValues"])
    d.sparse_value_offsets = cast(keep["sparseValueOffsets"])
    d.sparse_values = cast(keep["sparseValues"])
    d.sparse_relative_rows = cast(keep["sparseRelativeRows"])
    d.sparse_col_indices = cast(keep["sparseColIndices"])
    out = C.c_void_p()
    if options is not None:
        st = hip().bsmr_plan_create_ex(C.byref(out), device, C.byref(d), C.byref(options))
    else:
        st = hip().bsmr_plan_create(C.byref(out), device, C.byref(d))
    return st, out


def plan_destroy(plan):
    hip().bsmr_plan_destroy(plan)
