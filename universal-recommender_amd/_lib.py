"""ctypes binding of liburcco (include/urcco.h).

The library is hand-written HIP for gfx950 and has NO CPU fallback: if the shared object is missing, or no HIP
device is visible, everything here raises.  `use_library()` lets the CPU test-suite point the binding at a build
of the same sources against the test-only host simulator (tests/hostsim); nothing in the package does that.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "lib", "liburcco.so")

OK, BAD_ARG, OOM_HOST, OOM_DEVICE, HIP_ERROR, INTERNAL, NO_DEVICE, RCCL_ERROR, BUSY = range(9)
FLAG_SINGLE_STREAM = 1
FLAG_FORCE_EXCHANGE = 2
FLAG_UNORDERED_ROWS = 4
FLAG_EMULATE_RANKS = 8
UNIQUE_ID_BYTES = 128
ROW_RATE_MAHOUT_INT_DIV = 0
ROW_RATE_FRACTIONAL = 1
RNG_SPLITMIX53 = 0      # the down-sampling RNG (decision D10 of DESIGN.md), OR-ed into the row-rate mode
RNG_MIX32 = 0x100
N_STAGES = 17
N_BINS = 7
STATS_LEN = 32
EXCH_SIZES = 4   # int64 words of a shard's record (include/urcco.h URCCO_EXCH_SIZES)
STAGE_NAMES = ["column_counts", "downsample_flags", "downsample_scan", "downsample_compact", "transpose", "row_work", "binning",
               "entropy", "cco_rows_micro", "cco_rows_wave", "cco_rows_block_small", "cco_rows_block", "cco_rows_cu_half", "cco_rows_cu", "cco_rows_global", "compact_indicators", "exchange"]


class UrccoError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"liburcco: {message} (status {status})")
        self.status = status


class Csr(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_cols", C.c_int64), ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p)]


class Dataset(C.Structure):
    _fields_ = [("matrix", Csr), ("max_elements_per_row", C.c_int32), ("max_interesting_elements", C.c_int32),
                ("min_llr", C.c_double), ("has_min_llr", C.c_int32), ("reserved", C.c_int32)]


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("row_rate_mode", C.c_int32), ("n_gpus", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32 * 4)]


GROUP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)
ALL_GATHER_V_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p)


ALL_TO_ALL_V_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64),
                              C.POINTER(C.c_int64), C.c_void_p)


class Collectives(C.Structure):
    """urcco_collectives; the constructor takes the members after struct_size (which it fills in)."""
    _fields_ = [("struct_size", C.c_size_t), ("user", C.c_void_p), ("group_start", GROUP_FN), ("group_end", GROUP_FN), ("all_reduce_sum", ALL_REDUCE_FN),
                ("all_gather_v", ALL_GATHER_V_FN), ("all_to_all_v", ALL_TO_ALL_V_FN)]

    def __init__(self, user=None, *callbacks):
        super().__init__(C.sizeof(Collectives), user, *callbacks)


class CommConfig(C.Structure):
    _fields_ = [("world_size", C.c_int32), ("first_rank", C.c_int32), ("nccl_unique_id", C.c_void_p), ("collectives", C.POINTER(Collectives))]


class DevShard(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("row_base", C.c_int64), ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p), ("nnz", C.c_int64)]


class DevDataset(C.Structure):
    _fields_ = [("n_cols", C.c_int64), ("max_elements_per_row", C.c_int32), ("max_interesting_elements", C.c_int32), ("min_llr", C.c_double),
                ("has_min_llr", C.c_int32), ("reserved", C.c_int32), ("shards", C.POINTER(DevShard))]


class DevResult(C.Structure):
    _fields_ = [("item_lo", C.c_int32), ("item_hi", C.c_int32), ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p), ("llr", C.c_void_p),
                ("stats", C.c_void_p), ("sampled_row_ptr", C.c_void_p), ("sampled_col_idx", C.c_void_p), ("sampled_rows", C.c_int64),
                ("sampled_nnz_total", C.c_int64), ("sampled_col_mask", C.c_int32)]


class Indicators(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_cols", C.c_int64), ("nnz", C.c_int64), ("row_ptr", C.POINTER(C.c_int64)),
                ("col_idx", C.POINTER(C.c_int32)), ("llr", C.POINTER(C.c_double))]


class DatasetStats(C.Structure):
    _fields_ = [("nnz_raw", C.c_int64), ("nnz_sampled", C.c_int64), ("pairs", C.c_int64), ("nnz_out", C.c_int64),
                ("rows_by_bin", C.c_int64 * 7), ("ms_total", C.c_double)]


# every symbol include/urcco.h declares: (restype, argtypes)
_p = C.c_void_p
SYMBOLS = {
    "urcco_version": (C.c_int, []),
    "urcco_device_count": (C.c_int, []),
    "urcco_last_error": (C.c_char_p, []),
    "urcco_status_string": (C.c_char_p, [C.c_int]),
    "urcco_cooccurrences_idss": (C.c_int, [C.POINTER(Csr), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Options),
                                           C.POINTER(Indicators), C.POINTER(DatasetStats)]),
    "urcco_cross_occurrence_downsampled": (C.c_int, [C.POINTER(Dataset), C.c_int32, C.c_int32, C.POINTER(Options),
                                                     C.POINTER(Indicators), C.POINTER(DatasetStats)]),
    "urcco_free_indicators": (None, [C.POINTER(Indicators), C.c_int32]),
    "urcco_cross_occurrence_stage": (C.c_int, [C.POINTER(Dataset), C.c_int32, C.c_int32, C.POINTER(Options)]),
    "urcco_cross_occurrence_finish": (C.c_int, [C.POINTER(Indicators), C.c_int32, C.POINTER(DatasetStats)]),
    "urcco_cross_occurrence_cancel": (C.c_int, []),
    "urcco_cross_occurrence_cancel_any": (C.c_int, []),
    "urcco_context_stage": (C.c_int, [_p, C.POINTER(Dataset), C.c_int32, C.c_int32]),
    "urcco_context_finish": (C.c_int, [_p, C.POINTER(Indicators), C.POINTER(DatasetStats)]),
    "urcco_shutdown": (C.c_int, []),
    "urcco_comm_unique_id": (C.c_int, [_p]),
    "urcco_context_create": (C.c_int, [C.POINTER(Options), C.POINTER(CommConfig), C.POINTER(_p)]),
    "urcco_context_destroy": (None, [_p]),
    "urcco_context_local_gpus": (C.c_int32, [_p]),
    "urcco_context_cross_occurrence": (C.c_int, [_p, C.POINTER(Dataset), C.c_int32, C.c_int32, C.POINTER(Indicators), C.POINTER(DatasetStats)]),
    "urcco_context_build_device": (C.c_int, [_p, C.POINTER(DevDataset), C.c_int32, C.c_int64, C.c_int32, _p, C.POINTER(DevResult)]),
    "urcco_context_wait_stream": (C.c_int, [_p, _p]),
    "urcco_context_synchronize": (C.c_int, [_p]),
    "urcco_context_set_timing": (C.c_int, [_p, C.c_int32]),
    "urcco_context_get_timings": (C.c_int, [_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "urcco_context_get_timings_gpu": (C.c_int, [_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "urcco_context_set_debug": (C.c_int, [_p, C.c_int32]),
    "urcco_context_set_flags": (C.c_int, [_p, C.c_int32]),
    "urcco_session_create": (C.c_int, [C.c_int32, _p, C.POINTER(_p)]),
    "urcco_session_destroy": (None, [_p]),
    "urcco_session_synchronize": (C.c_int, [_p]),
    "urcco_session_scratch_bytes": (C.c_int64, [_p]),
    "urcco_session_set_timing": (C.c_int, [_p, C.c_int32]),
    "urcco_session_set_debug": (C.c_int, [_p, C.c_int32]),
    "urcco_debug_dump_marks": (None, []),
    "urcco_session_get_timings": (C.c_int, [_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "urcco_dev_column_counts": (C.c_int, [_p, C.c_int64, _p, C.c_int32, _p]),
    "urcco_dev_downsample": (C.c_int, [_p, C.c_int64, _p, _p, C.c_int64, C.c_int32, _p, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                       _p, _p, _p]),
    "urcco_dev_transpose": (C.c_int, [_p, C.c_int64, _p, _p, C.c_int64, C.c_int32, _p, C.c_int32, C.c_int32, _p, _p]),
    "urcco_dev_row_work_csr": (C.c_int, [_p, C.c_int64, _p, _p, C.c_int64, _p, C.c_int32, _p]),
    "urcco_dev_row_work": (C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32, _p, _p, C.c_int64, _p, _p]),
    "urcco_dev_partition": (C.c_int, [_p, C.c_int32, _p, C.c_int32, C.POINTER(C.c_int32)]),
    "urcco_dev_merge_fragments": (C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _p, C.c_int32, _p, C.c_int64, _p, _p, _p, _p]),
    "urcco_dev_cco_rows": (C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32, _p, _p, C.c_int64, _p, _p, C.c_int32, _p, _p, C.c_int64, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_double, _p, _p, _p, _p]),
    "urcco_dev_pack_counts": (C.c_int, [_p, C.c_int64, _p, _p, C.c_int64, _p, C.c_int32, _p, _p]),
    "urcco_dev_cco_rows_packed": (C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32, _p, _p, C.c_int64, _p, _p, C.c_int32, _p, _p, C.c_int64, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_double, _p, _p, _p, _p, _p, _p]),
    "urcco_dev_compact_indicators": (C.c_int, [_p, C.c_int32, C.c_int32, _p, _p, _p, _p, _p, _p]),
    "urcco_dev_pop_counts": (C.c_int, [_p, C.c_int64, _p, _p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), _p]),
    "urcco_dev_llr": (C.c_int, [_p, C.c_int64, _p, _p, _p, _p, _p]),
    "urcco_dev_u01": (C.c_int, [_p, C.c_int64, C.c_int32, _p, _p, _p]),
    "urcco_dev_u01_rng": (C.c_int, [_p, C.c_int64, C.c_int32, _p, _p, C.c_int32, _p]),
    "urcco_dev_dictionary_build": (C.c_int, [_p, C.c_int64, _p, _p, C.c_int32, _p, C.POINTER(_p), C.POINTER(C.c_int64)]),
    "urcco_dev_dictionary_lookup": (C.c_int, [_p, _p, C.c_int64, _p, _p, _p]),
    "urcco_dev_dictionary_verify": (C.c_int, [_p, _p, C.c_int64, _p, _p, _p, _p, C.POINTER(C.c_int64)]),
    "urcco_dev_dictionary_verify_against": (C.c_int, [_p, _p, C.c_int64, _p, _p, _p, _p, _p, C.POINTER(C.c_int64)]),
    "urcco_hash_strings": (C.c_int, [_p, _p, C.c_int64, C.c_uint64, _p]),
    "urcco_key_table_destroy": (None, [_p]),
    "urcco_dev_csr_from_pairs": (C.c_int, [_p, C.c_int64, _p, _p, C.c_int64, _p, _p, C.POINTER(C.c_int64)]),
}

_cache = {}
_lib_path: Optional[str] = None


def _bind(path: str) -> C.CDLL:
    # PyTorch ships its own HIP runtime; when both live in one process the runtime torch initialised must be the one
    # liburcco resolves (same devices, streams and allocations), so torch is imported before the dlopen.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def use_library(path: Optional[str]) -> None:
    """Bind to an explicit shared object (test hook); None returns to the in-tree product library."""
    global _lib_path
    _lib_path = path


def load(path: str) -> C.CDLL:
    """Bind (once) the shared object at `path`."""
    path = os.path.abspath(path)
    if path not in _cache:
        if not os.path.exists(path):
            raise RuntimeError(
                f"liburcco HIP library not found at {path}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _cache[path] = _bind(path)
    return _cache[path]


def library_path() -> str:
    return _lib_path or DEFAULT_PATH


def lib() -> C.CDLL:
    return load(library_path())


def check(status: int, library: Optional[C.CDLL] = None) -> None:
    if status != OK:
        library = library or lib()
        msg = library.urcco_last_error()
        raise UrccoError(status, (msg or b"").decode("utf-8", "replace") or library.urcco_status_string(status).decode())
