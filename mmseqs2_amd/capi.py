"""ctypes binding of include/mmgpu.h.  No compute happens in Python and nothing here falls back to a CPU
implementation: a missing library or a failing HIP call raises MMGpuError."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("MMGPU_LIB") or os.path.join(_HERE, "lib", "libmmgpu.so")

c_p = ctypes.c_void_p


class MMGpuError(RuntimeError):
    pass


def library_path():
    return _LIB


def build_library(force=False):
    """Compile mmseqs2_amd/csrc/*.hip for gfx950 into mmseqs2_amd/lib/libmmgpu.so (hipcc cross-compiles
    without a GPU)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    if not os.path.exists(_LIB):
        raise MMGpuError("build did not produce " + _LIB)
    return _LIB


class SwParams(ctypes.Structure):
    _fields_ = [("mat", c_p), ("alphabet", ctypes.c_int), ("gap_open", ctypes.c_int), ("gap_extend", ctypes.c_int)]


class SwQuery(ctypes.Structure):
    _fields_ = [("q", c_p), ("qlen", ctypes.c_uint32), ("comp_bias", c_p), ("target_ids", c_p),
                ("n_targets", ctypes.c_uint32), ("min_start_score", ctypes.c_int32), ("profile", c_p),
                ("profile_letters", ctypes.c_uint32)]


class NuclParams(ctypes.Structure):
    _fields_ = [("mat", c_p), ("reverse", c_p), ("gap_open", ctypes.c_int), ("gap_extend", ctypes.c_int), ("zdrop", ctypes.c_int),
                ("past_end_query", ctypes.c_int), ("past_end_target", ctypes.c_int), ("wrapped", ctypes.c_int)]


class NuclQuery(ctypes.Structure):
    _fields_ = [("q", c_p), ("qlen", ctypes.c_uint32)]


NUCL_PAIR_DTYPE = np.dtype([("query", np.uint32), ("target", np.uint32), ("diagonal", np.uint16), ("reverse", np.uint8),
                            ("past_end", np.uint8)])      # 0, or 0x80 | query letter | target letter << 3 (MMGPU_NUCL_PAST_END)
NUCL_HIT_DTYPE = np.dtype([("score", np.int32), ("q_start", np.int32), ("q_end", np.int32), ("t_start", np.int32),
                           ("t_end", np.int32), ("ident", np.uint32), ("bt_len", np.uint32), ("status", np.int32),
                           ("bt_off", np.uint64)])

SW_BT_DTYPE = np.dtype([("bt_off", np.uint64), ("bt_len", np.uint32), ("ident", np.uint32), ("status", np.int32),
                        ("reserved", np.int32)])
SW_BLOCK_DTYPE = np.dtype([("q_start", np.int32), ("t_start", np.int32), ("ident", np.uint32), ("bt_len", np.uint32), ("bt_off", np.uint64),
                           ("status", np.int32), ("reserved", np.int32)])
SW_HIT_DTYPE = np.dtype([("score", np.int32), ("q_end", np.int32), ("t_end", np.int32), ("q_start", np.int32),
                         ("t_start", np.int32), ("word", np.int32)])

PF_OK, PF_OVERFLOW, PF_LONG_SEQ, PF_SAT_TIE, PF_SHARD_INEXACT = 0, 1, 2, 3, 4      # mmgpu_pf status codes (include/mmgpu.h)

# every symbol include/mmgpu.h declares (tests check the built library exports all of them)
EXPORTED_SYMBOLS = [
    "mmgpu_init", "mmgpu_warmup", "mmgpu_destroy", "mmgpu_last_error", "mmgpu_set_stream", "mmgpu_synchronize",
    "mmgpu_device_info", "mmgpu_device_memory", "mmgpu_host_comp_bias", "mmgpu_host_round_comp_bias", "mmgpu_host_comp_bias_batch", "mmgpu_load_targets", "mmgpu_sw_batch", "mmgpu_sw_prepare", "mmgpu_sw_run",
    "mmgpu_sw_fetch", "mmgpu_sw_batch_stats", "mmgpu_sw_last_kernel_ms", "mmgpu_sw_kernel_ms_mean", "mmgpu_sw_free",
    "mmgpu_sw_traceback", "mmgpu_sw_prepare_from_pf", "mmgpu_sw_fetch_device", "mmgpu_nucl_align",
    "mmgpu_host_score_matrix", "mmgpu_host_score_matrix_rows", "mmgpu_host_index_build", "mmgpu_pf_load_index", "mmgpu_pf_batch", "mmgpu_pf_prepare",
    "mmgpu_pf_run", "mmgpu_pf_fetch", "mmgpu_pf_stage_ms", "mmgpu_pf_last_cells", "mmgpu_pf_fetch_device", "mmgpu_pf_merge_splits", "mmgpu_pf_build_index", "mmgpu_pf_mask_targets", "mmgpu_pf_debug_masked_targets", "mmgpu_pf_debug_index", "mmgpu_pf_debug_fetch", "mmgpu_pf_free",
    "mmgpu_host_partition_targets", "mmgpu_pf_set_shard", "mmgpu_pf_fetch_exchange", "mmgpu_pf_merge_exchange",
    "mmgpu_pf_localize_lists", "mmgpu_sw_prepare_from_lists",
    "mmgpu_comm_unique_id", "mmgpu_comm_init_rank", "mmgpu_comm_info", "mmgpu_comm_destroy", "mmgpu_pf_exchange_merge",
    "mmgpu_sw_prepare_owned", "mmgpu_sw_gather_owned", "mmgpu_sw_fetch_owned", "mmgpu_sw_block_backtrace", "mmgpu_sw_block_growth", "mmgpu_sw_block_tiers", "mmgpu_sw_reverse_pairs", "mmgpu_sw_block_starts",
    "mmgpu_init_multi", "mmgpu_destroy_multi", "mmgpu_multi_size", "mmgpu_multi_ctx", "mmgpu_multi_synchronize",
    "mmgpu_multi_load_targets", "mmgpu_multi_pf_mask_targets", "mmgpu_multi_pf_build_index", "mmgpu_multi_pf_prepare", "mmgpu_multi_pf_run", "mmgpu_multi_pf_fetch",
    "mmgpu_multi_pf_stride", "mmgpu_multi_pf_free", "mmgpu_multi_sw_from_pf", "mmgpu_multi_has_unsplit", "mmgpu_multi_pf_redone",
    "mmgpu_pf_exchange_redo_unsplit",
    "mmgpu_db_save", "mmgpu_db_probe", "mmgpu_db_load",
]


class DbInfo(ctypes.Structure):      # mmgpu_db_info
    _fields_ = [("source_fingerprint", ctypes.c_uint64), ("index_fingerprint", ctypes.c_uint64), ("n_targets", ctypes.c_uint32),
                ("alphabet", ctypes.c_uint32), ("total_residues", ctypes.c_uint64), ("has_masked_view", ctypes.c_int32),
                ("has_index", ctypes.c_int32), ("kmer_size", ctypes.c_int32), ("spaced", ctypes.c_int32), ("n_entries", ctypes.c_uint64),
                ("file_bytes", ctypes.c_uint64)]


def db_probe(path, lib=None):
    """mmgpu_db_probe: the header of a persisted device layout (no device needed) -> dict, or None if `path` is not one"""
    L = lib or load_library()
    info = DbInfo()
    L.mmgpu_db_probe.argtypes = [ctypes.c_char_p, c_p]
    if L.mmgpu_db_probe(str(path).encode(), ctypes.byref(info)) != 0:
        return None
    return {k: getattr(info, k) for k, _ in DbInfo._fields_}


class PfIndexDesc(ctypes.Structure):
    _fields_ = [("kmer_size", ctypes.c_int), ("alphabet", ctypes.c_int), ("spaced", ctypes.c_int), ("score3", c_p),
                ("index3", c_p), ("row3", ctypes.c_size_t), ("score2", c_p), ("index2", c_p), ("row2", ctypes.c_size_t),
                ("offsets", c_p), ("entry_ids", c_p), ("entry_pos", c_p),
                ("entries6", c_p), ("n_entries", ctypes.c_uint64), ("ungapped_mat", c_p), ("kmer_alphabet", ctypes.c_int)]


class PfParams(ctypes.Structure):
    _fields_ = [("kmer_thr", ctypes.c_int), ("max_hits", ctypes.c_uint32), ("min_diag_score", ctypes.c_uint32),
                ("ref_bins", ctypes.c_uint32), ("exact_kmer", ctypes.c_uint32), ("nucleotide", ctypes.c_uint32),
                ("kmer_score", ctypes.c_uint32)]


class PfQuery(ctypes.Structure):
    _fields_ = [("q", c_p), ("qlen", ctypes.c_uint32), ("comp_bias", c_p), ("identity_id", ctypes.c_uint32),
                ("profile_score", c_p), ("profile_index", c_p), ("profile_row", ctypes.c_uint32), ("profile", c_p)]


class PfShard(ctypes.Structure):
    _fields_ = [("n_shards", ctypes.c_uint32), ("shard", ctypes.c_uint32), ("global_db_size", ctypes.c_uint32),
                ("global_ids", c_p), ("shard_of", c_p), ("local_id", c_p)]


PF_XHIT_DTYPE = np.dtype([("id", np.uint32), ("score", np.uint32), ("diagonal", np.uint16), ("flags", np.uint16),
                          ("order", np.uint32)])
PF_HIT_DTYPE = np.dtype([("id", np.uint32), ("score", np.int32), ("diagonal", np.uint16), ("reserved", np.uint16)])
PF_QSTAT_DTYPE = np.dtype([("db_matches", np.uint64), ("kmer_list_len", np.uint64), ("double_hits", np.uint32),
                           ("diag_thr", np.uint32)])
PF_LIST_DTYPE = np.dtype([("start", np.uint32), ("len", np.uint32), ("lprefix", np.uint32), ("pos", np.uint32)])
PF_CAND_DTYPE = np.dtype([("id", np.uint32), ("arr", np.uint32), ("score", np.uint32), ("diag", np.uint16),
                          ("pad", np.uint16)])
PF_DBG = dict(nsim=(0, np.uint32), list_base=(1, np.uint32), lists=(2, PF_LIST_DTYPE), peb=(3, np.uint32),
              split=(4, np.uint64), bin_off=(5, np.uint16), cand_base=(6, np.uint32), surv=(7, PF_CAND_DTYPE),
              surv_count=(8, np.uint32), bins=(9, np.uint32))


def load_library():
    if not os.path.exists(_LIB):
        raise MMGpuError("libmmgpu.so is not built (%s); run __graft_entry__.build() or make -C mmseqs2_amd/csrc" % _LIB)
    L = ctypes.CDLL(_LIB)
    L.mmgpu_last_error.restype = ctypes.c_char_p
    L.mmgpu_init.argtypes = [ctypes.POINTER(c_p), ctypes.c_int]
    L.mmgpu_destroy.argtypes = [c_p]
    L.mmgpu_destroy.restype = None
    L.mmgpu_set_stream.argtypes = [c_p, c_p]
    L.mmgpu_synchronize.argtypes = [c_p]
    L.mmgpu_device_info.argtypes = [c_p, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_int]
    L.mmgpu_device_memory.argtypes = [c_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.mmgpu_host_comp_bias.argtypes = [c_p, c_p, ctypes.c_int, c_p, ctypes.c_uint32, ctypes.c_float, c_p]
    L.mmgpu_host_round_comp_bias.argtypes = [c_p, ctypes.c_uint32, c_p]
    L.mmgpu_host_comp_bias_batch.argtypes = [c_p, c_p, ctypes.c_int, c_p, c_p, ctypes.c_uint32, ctypes.c_float, c_p, c_p, ctypes.c_int]
    L.mmgpu_load_targets.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, ctypes.c_int]
    L.mmgpu_sw_batch.argtypes = [c_p, ctypes.POINTER(SwParams), c_p, ctypes.c_uint32, ctypes.c_int, c_p]
    L.mmgpu_sw_prepare.argtypes = [c_p, ctypes.POINTER(SwParams), c_p, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(c_p)]
    L.mmgpu_sw_prepare_from_pf.argtypes = [c_p, ctypes.POINTER(SwParams), c_p, ctypes.c_uint32, ctypes.c_int, c_p, ctypes.POINTER(c_p)]
    L.mmgpu_sw_run.argtypes = [c_p, c_p]
    L.mmgpu_nucl_align.argtypes = [c_p, ctypes.POINTER(NuclParams), c_p, ctypes.c_uint32, c_p, ctypes.c_uint32, c_p, c_p,
                                   ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    L.mmgpu_sw_fetch_device.argtypes = [c_p, c_p, c_p]
    L.mmgpu_sw_fetch.argtypes = [c_p, c_p, c_p]
    L.mmgpu_sw_batch_stats.argtypes = [c_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.mmgpu_sw_last_kernel_ms.argtypes = [c_p, c_p, ctypes.POINTER(ctypes.c_float)]
    L.mmgpu_sw_kernel_ms_mean.argtypes = [c_p, c_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32)]
    L.mmgpu_sw_free.argtypes = [c_p, c_p]
    L.mmgpu_sw_free.restype = None
    L.mmgpu_sw_traceback.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, c_p, c_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.mmgpu_host_score_matrix.argtypes = [c_p, ctypes.c_int, ctypes.c_int, c_p, c_p]
    L.mmgpu_host_index_build.argtypes = [c_p, c_p, ctypes.c_uint32, c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, c_p, c_p, c_p, ctypes.POINTER(ctypes.c_uint64)]
    L.mmgpu_pf_load_index.argtypes = [c_p, ctypes.POINTER(PfIndexDesc)]
    L.mmgpu_pf_build_index.argtypes = [c_p, ctypes.POINTER(PfIndexDesc), c_p, ctypes.c_int]
    L.mmgpu_pf_debug_index.argtypes = [c_p, c_p, c_p, c_p, ctypes.POINTER(ctypes.c_uint64)]
    L.mmgpu_pf_batch.argtypes = [c_p, ctypes.POINTER(PfParams), c_p, ctypes.c_uint32, c_p, ctypes.c_uint32, c_p, c_p]
    L.mmgpu_pf_prepare.argtypes = [c_p, ctypes.POINTER(PfParams), c_p, ctypes.c_uint32, ctypes.POINTER(c_p)]
    L.mmgpu_pf_run.argtypes = [c_p, c_p]
    L.mmgpu_pf_fetch.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, c_p, c_p, c_p]
    L.mmgpu_pf_stage_ms.argtypes = [c_p, c_p, ctypes.POINTER(ctypes.c_float)]
    L.mmgpu_pf_fetch_device.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, c_p]
    L.mmgpu_pf_merge_splits.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, c_p, c_p, c_p]
    L.mmgpu_pf_last_cells.argtypes = [c_p, c_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.mmgpu_pf_debug_fetch.argtypes = [c_p, c_p, ctypes.c_int, c_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.mmgpu_pf_free.argtypes = [c_p, c_p]
    L.mmgpu_pf_free.restype = None
    L.mmgpu_host_partition_targets.argtypes = [c_p, ctypes.c_uint32, ctypes.c_uint32, c_p, c_p, c_p, c_p]
    L.mmgpu_pf_set_shard.argtypes = [c_p, ctypes.POINTER(PfShard)]
    L.mmgpu_pf_fetch_exchange.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, c_p]
    L.mmgpu_pf_merge_exchange.argtypes = [c_p, c_p, c_p, c_p, ctypes.c_uint32, ctypes.c_uint32, c_p, c_p, ctypes.c_uint32, c_p, c_p]
    L.mmgpu_pf_localize_lists.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, ctypes.c_uint32, c_p, c_p, c_p]
    L.mmgpu_sw_prepare_from_lists.argtypes = [c_p, ctypes.POINTER(SwParams), c_p, ctypes.c_uint32, ctypes.c_int, c_p, c_p,
                                              ctypes.c_uint32, ctypes.POINTER(c_p)]
    L.mmgpu_sw_block_backtrace.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, c_p, c_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.mmgpu_sw_block_tiers.argtypes = [c_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.mmgpu_sw_reverse_pairs.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, c_p]
    L.mmgpu_sw_block_starts.argtypes = [c_p, c_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.mmgpu_sw_block_growth.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, c_p, c_p, ctypes.c_uint32]
    L.mmgpu_comm_unique_id.argtypes = [c_p]
    L.mmgpu_comm_init_rank.argtypes = [c_p, c_p, ctypes.c_int, ctypes.c_int]
    L.mmgpu_comm_info.argtypes = [c_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_int]
    L.mmgpu_comm_destroy.argtypes = [c_p]
    L.mmgpu_comm_destroy.restype = None
    L.mmgpu_pf_exchange_merge.argtypes = [c_p, c_p, c_p, ctypes.POINTER(c_p), ctypes.POINTER(c_p), ctypes.POINTER(c_p), ctypes.POINTER(ctypes.c_uint32)]
    L.mmgpu_pf_exchange_redo_unsplit.argtypes = [c_p, c_p, c_p, c_p, c_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.mmgpu_multi_has_unsplit.argtypes = [c_p]
    L.mmgpu_multi_pf_redone.argtypes = [c_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.mmgpu_sw_prepare_owned.argtypes = [c_p, ctypes.POINTER(SwParams), c_p, ctypes.c_uint32, ctypes.c_int, c_p, ctypes.POINTER(c_p)]
    L.mmgpu_sw_gather_owned.argtypes = [c_p, c_p, ctypes.POINTER(c_p), ctypes.POINTER(c_p)]
    L.mmgpu_sw_fetch_owned.argtypes = [c_p, c_p, c_p, ctypes.POINTER(ctypes.c_uint32)]
    L.mmgpu_init_multi.argtypes = [ctypes.POINTER(c_p), c_p, ctypes.c_int]
    L.mmgpu_destroy_multi.argtypes = [c_p]
    L.mmgpu_destroy_multi.restype = None
    L.mmgpu_multi_size.argtypes = [c_p]
    L.mmgpu_multi_ctx.argtypes = [c_p, ctypes.c_int]
    L.mmgpu_multi_ctx.restype = c_p
    L.mmgpu_multi_synchronize.argtypes = [c_p]
    L.mmgpu_multi_load_targets.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, ctypes.c_int]
    L.mmgpu_multi_pf_build_index.argtypes = [c_p, ctypes.POINTER(PfIndexDesc), c_p, ctypes.c_int]
    L.mmgpu_multi_pf_prepare.argtypes = [c_p, ctypes.POINTER(PfParams), c_p, ctypes.c_uint32, ctypes.POINTER(c_p)]
    L.mmgpu_multi_pf_run.argtypes = [c_p, c_p]
    L.mmgpu_multi_pf_fetch.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, c_p, c_p]
    L.mmgpu_multi_pf_stride.argtypes = [c_p]
    L.mmgpu_multi_pf_stride.restype = ctypes.c_uint32
    L.mmgpu_multi_pf_free.argtypes = [c_p, c_p]
    L.mmgpu_multi_pf_free.restype = None
    L.mmgpu_multi_sw_from_pf.argtypes = [c_p, ctypes.POINTER(SwParams), c_p, ctypes.c_uint32, ctypes.c_int, c_p, c_p,
                                         ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_float)]
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_p)


def host_comp_bias(submat16, pback, seq, scale=1.0, lib=None):
    """Composition bias exactly as the reference's host code computes it; returns (float32[], int8[])."""
    L = lib or load_library()
    submat16 = np.ascontiguousarray(submat16, np.int16)
    pback = np.ascontiguousarray(pback, np.float64)
    seq = np.ascontiguousarray(seq, np.uint8)
    f = np.zeros(len(seq), np.float32)
    r = np.zeros(len(seq), np.int8)
    if L.mmgpu_host_comp_bias(_ptr(submat16), _ptr(pback), submat16.shape[0], _ptr(seq), len(seq), scale, _ptr(f)) != 0:
        raise MMGpuError(L.mmgpu_last_error().decode())
    if L.mmgpu_host_round_comp_bias(_ptr(f), len(f), _ptr(r)) != 0:
        raise MMGpuError(L.mmgpu_last_error().decode())
    return f, r


def host_comp_bias_batch(submat16, pback, residues, offsets, scale=1.0, want_float=True, want_round=True, threads=0, lib=None):
    """mmgpu_host_comp_bias_batch: the correction of a whole block of sequences (concatenated residues + offsets[n+1]) on
    `threads` host threads (0 = all); returns (float32[] or None, int8[] or None), indexed like residues."""
    L = lib or load_library()
    submat16 = np.ascontiguousarray(submat16, np.int16)
    pback = np.ascontiguousarray(pback, np.float64)
    residues = np.ascontiguousarray(residues, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    f = np.empty(len(residues), np.float32) if want_float else None
    r = np.empty(len(residues), np.int8) if want_round else None
    if L.mmgpu_host_comp_bias_batch(_ptr(submat16), _ptr(pback), submat16.shape[0], _ptr(residues), _ptr(offsets), len(offsets) - 1,
                                    scale, _ptr(f), _ptr(r), int(threads) or (os.cpu_count() or 1)) != 0:
        raise MMGpuError(L.mmgpu_last_error().decode())
    return f, r


# the two query descriptors as numpy records (same layout as the ctypes Structures above / include/mmgpu.h), for callers
# that hold their queries as one residue array + offsets: the descriptor array is filled with vector operations
SW_QUERY_DTYPE = np.dtype([("q", np.uint64), ("qlen", np.uint32), ("_p0", np.uint32), ("comp_bias", np.uint64), ("target_ids", np.uint64),
                           ("n_targets", np.uint32), ("min_start_score", np.int32), ("profile", np.uint64),
                           ("profile_letters", np.uint32), ("_p1", np.uint32)])
PF_QUERY_DTYPE = np.dtype([("q", np.uint64), ("qlen", np.uint32), ("_p0", np.uint32), ("comp_bias", np.uint64), ("identity_id", np.uint32),
                           ("_p1", np.uint32), ("profile_score", np.uint64), ("profile_index", np.uint64), ("profile_row", np.uint32),
                           ("_p2", np.uint32), ("profile", np.uint64)])
assert SW_QUERY_DTYPE.itemsize == ctypes.sizeof(SwQuery) and PF_QUERY_DTYPE.itemsize == ctypes.sizeof(PfQuery)


def host_score_matrix(submat16, span, lib=None):
    """ScoreMatrix of all span-mers (ExtendedSubstitutionMatrix::calcScoreMatrix); returns (score int16[n][n], index)."""
    L = lib or load_library()
    submat16 = np.ascontiguousarray(submat16, np.int16)
    n = (submat16.shape[0] - 1) ** span
    score = np.zeros((n, n), np.int16)
    index = np.zeros((n, n), np.uint32)
    if L.mmgpu_host_score_matrix(_ptr(submat16), submat16.shape[0], span, _ptr(score), _ptr(index)) != 0:
        raise MMGpuError(L.mmgpu_last_error().decode())
    return score, index


def host_index_build(residues, offsets, kmer_submat16, k, spaced, kmer_thr, lib=None):
    """IndexTable over numeric targets (masking off); returns (offsets uint64[kalph^k+1], ids uint32[], pos uint16[])."""
    L = lib or load_library()
    residues = np.ascontiguousarray(residues, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    kmer_submat16 = np.ascontiguousarray(kmer_submat16, np.int16)
    a = kmer_submat16.shape[0]
    table = (a - 1) ** k
    koff = np.zeros(table + 1, np.uint64)
    ne = ctypes.c_uint64()
    args = (_ptr(residues), _ptr(offsets), len(offsets) - 1, _ptr(kmer_submat16), a, k, int(spaced), int(kmer_thr), _ptr(koff))
    if L.mmgpu_host_index_build(*args, None, None, ctypes.byref(ne)) != 0:
        raise MMGpuError(L.mmgpu_last_error().decode())
    ids = np.zeros(max(ne.value, 1), np.uint32)
    pos = np.zeros(max(ne.value, 1), np.uint16)
    if L.mmgpu_host_index_build(*args, _ptr(ids), _ptr(pos), ctypes.byref(ne)) != 0:
        raise MMGpuError(L.mmgpu_last_error().decode())
    return koff, ids[:ne.value], pos[:ne.value]


def partition_targets(offsets, n_shards, lib=None):
    """Length-bucket sharding (mmgpu_host_partition_targets): -> (shard_of[n], local_id[n], shard_sizes[n_shards],
    shard_residues[n_shards])."""
    L = lib or load_library()
    off = np.ascontiguousarray(offsets, np.uint64)
    n = len(off) - 1
    shard_of = np.zeros(max(n, 1), np.uint32)
    local_id = np.zeros(max(n, 1), np.uint32)
    sizes = np.zeros(n_shards, np.uint32)
    res = np.zeros(n_shards, np.uint64)
    if L.mmgpu_host_partition_targets(_ptr(off), n, int(n_shards), _ptr(shard_of), _ptr(local_id), _ptr(sizes), _ptr(res)) != 0:
        raise MMGpuError(L.mmgpu_last_error().decode())
    return shard_of[:n], local_id[:n], sizes, res


def shard_sequences(residues, offsets, shard_of, shard):
    """The sequences of one shard, in ascending global id order: -> (residues, offsets, global_ids)."""
    off = np.asarray(offsets, np.int64)
    gids = np.nonzero(shard_of == shard)[0].astype(np.uint32)
    lens = (off[1:] - off[:-1])[gids]
    soff = np.zeros(len(gids) + 1, np.uint64)
    soff[1:] = np.cumsum(lens)
    idx = np.repeat(off[:-1][gids] - soff[:-1].astype(np.int64), lens) + np.arange(int(soff[-1]), dtype=np.int64)
    return np.ascontiguousarray(residues[idx]), soff, gids


class PfBatch:
    """A prepared prefilter batch (queries resident in HBM)."""

    def __init__(self, gpu, handle, keep, nq, max_hits):
        self.gpu, self.handle, self._keep, self.nq, self.max_hits = gpu, handle, keep, nq, max_hits

    def run(self):
        self.gpu._check(self.gpu.L.mmgpu_pf_run(self.gpu.ctx, self.handle))

    def fetch(self):
        """-> (hits[nq][stride], counts[nq], status[nq], stats[nq])"""
        stride = max(self.max_hits, 1)
        hits = np.zeros((max(self.nq, 1), stride), PF_HIT_DTYPE)
        counts = np.zeros(max(self.nq, 1), np.uint32)
        status = np.zeros(max(self.nq, 1), np.int32)
        stats = np.zeros(max(self.nq, 1), PF_QSTAT_DTYPE)
        self.gpu._check(self.gpu.L.mmgpu_pf_fetch(self.gpu.ctx, self.handle, _ptr(hits), stride, _ptr(counts), _ptr(status),
                                                  _ptr(stats)))
        return hits[:self.nq], counts[:self.nq], status[:self.nq], stats[:self.nq]

    def fetch_device(self, d_hits_ptr, stride, d_counts_ptr):
        """D2D copy of the hit lists into caller-owned device memory (raw pointers, e.g. torch tensor.data_ptr())."""
        self.gpu._check(self.gpu.L.mmgpu_pf_fetch_device(self.gpu.ctx, self.handle, c_p(d_hits_ptr), stride, c_p(d_counts_ptr)))

    def fetch_exchange(self, d_xhits_ptr, stride, d_counts_ptr):
        """sharded run: D2D copy of the exchange records ([nq][stride] PF_XHIT_DTYPE) and their counts"""
        self.gpu._check(self.gpu.L.mmgpu_pf_fetch_exchange(self.gpu.ctx, self.handle, c_p(d_xhits_ptr), stride, c_p(d_counts_ptr)))

    def merge_exchange(self, d_xhits_ptr, d_counts_ptr, n_shards, stride, identity_global, d_out_hits_ptr, out_stride,
                       d_out_counts_ptr, d_out_flags_ptr=None):
        ident = None if identity_global is None else np.ascontiguousarray(identity_global, np.uint32)
        self.gpu._check(self.gpu.L.mmgpu_pf_merge_exchange(self.gpu.ctx, self.handle, c_p(d_xhits_ptr), c_p(d_counts_ptr), n_shards, stride,
                                                           _ptr(ident), c_p(d_out_hits_ptr), out_stride, c_p(d_out_counts_ptr),
                                                           c_p(d_out_flags_ptr) if d_out_flags_ptr else None))

    def redo_unsplit(self, full_gpu):
        """mmgpu_pf_exchange_redo_unsplit: the queries whose merged list is flagged inexact run once more against `full_gpu` (a context
        holding the WHOLE database and its index); their merged lists in this batch are replaced.
        -> (queries re-run, of those left to the host)"""
        par, arr, n = self.prepared_with
        a, b = ctypes.c_uint32(), ctypes.c_uint32()
        self.gpu._check(self.gpu.L.mmgpu_pf_exchange_redo_unsplit(self.gpu.ctx, self.handle, full_gpu.ctx, ctypes.byref(par), ctypes.cast(arr, c_p), n,
                                                                  ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def exchange_merge(self, identity_global=None):
        """mmgpu_pf_exchange_merge: all-gather over the context's communicator + merge, enqueued on its stream.
        -> (d_hits, d_counts, d_flags device pointers owned by the batch, stride)"""
        ig = None if identity_global is None else np.ascontiguousarray(identity_global, np.uint32)
        dh, dc, df, st = c_p(), c_p(), c_p(), ctypes.c_uint32()
        self.gpu._check(self.gpu.L.mmgpu_pf_exchange_merge(self.gpu.ctx, self.handle, _ptr(ig), ctypes.byref(dh), ctypes.byref(dc),
                                                           ctypes.byref(df), ctypes.byref(st)))
        return dh.value, dc.value, df.value, st.value

    def stage_ms(self):
        ms = (ctypes.c_float * 7)()
        self.gpu._check(self.gpu.L.mmgpu_pf_stage_ms(self.gpu.ctx, self.handle, ms))
        return [float(x) for x in ms]

    def last_cells(self):
        """(ungapped cells scored, double-diagonal candidates) of the last run"""
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        self.gpu._check(self.gpu.L.mmgpu_pf_last_cells(self.gpu.ctx, self.handle, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def debug(self, what):
        code, dt = PF_DBG[what]
        n = ctypes.c_size_t()
        self.gpu._check(self.gpu.L.mmgpu_pf_debug_fetch(self.gpu.ctx, self.handle, code, None, 0, ctypes.byref(n)))
        out = np.zeros(max(n.value // np.dtype(dt).itemsize, 1), dt)
        self.gpu._check(self.gpu.L.mmgpu_pf_debug_fetch(self.gpu.ctx, self.handle, code, _ptr(out), n.value, ctypes.byref(n)))
        return out[: n.value // np.dtype(dt).itemsize]

    def free(self):
        if self.handle is not None:
            self.gpu.L.mmgpu_pf_free(self.gpu.ctx, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class SwBatch:
    """A prepared (HBM-resident) Smith-Waterman batch."""

    def __init__(self, gpu, handle, keep):
        self.gpu, self.handle, self._keep = gpu, handle, keep
        cells, pairs = ctypes.c_uint64(), ctypes.c_uint64()
        gpu._check(gpu.L.mmgpu_sw_batch_stats(handle, ctypes.byref(cells), ctypes.byref(pairs)))
        self.cells, self.pairs = cells.value, pairs.value
        self.slots = None   # fused batches: result slots (n_queries * stride), pairs = slots that hold a hit

    def run(self):
        self.gpu._check(self.gpu.L.mmgpu_sw_run(self.gpu.ctx, self.handle))

    def fetch_device(self, d_out_ptr):
        """D2D copy of the results (24 B records, fetch() order) into caller-owned device memory, on the context's stream."""
        self.gpu._check(self.gpu.L.mmgpu_sw_fetch_device(self.gpu.ctx, self.handle, c_p(d_out_ptr)))

    def fetch(self):
        out = np.zeros(self.pairs if self.slots is None else self.slots, SW_HIT_DTYPE)
        self.gpu._check(self.gpu.L.mmgpu_sw_fetch(self.gpu.ctx, self.handle, _ptr(out)))
        return out

    def block_starts(self):
        """mmgpu_sw_block_starts (batches of mode 2): the device selects the int16-range hits that pass the start-score threshold, runs the
        block aligner for their start positions, writes them into the records and scans backwards for what it declined
        -> (selected, declined, too_large)"""
        a, b, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        self.gpu._check(self.gpu.L.mmgpu_sw_block_starts(self.gpu.ctx, self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def reverse_pairs(self, pair_index):
        """mmgpu_sw_reverse_pairs: the reverse scan of exactly these result slots (batches of mode 2: the int16-range hits the block
        aligner declined) -> their SW_HIT_DTYPE records with q_start / t_start filled"""
        idx = np.ascontiguousarray(pair_index, np.uint32)
        out = np.zeros(len(idx), SW_HIT_DTYPE)
        self.gpu._check(self.gpu.L.mmgpu_sw_reverse_pairs(self.gpu.ctx, self.handle, _ptr(idx), len(idx), _ptr(out)))
        return out

    def traceback(self, pair_index):
        """Backtrace strings ('M','I','D') and identity counts of the selected result slots (batch mode >= 1).
        Returns (info SW_BT_DTYPE[n], list of str)."""
        import time
        idx = np.ascontiguousarray(pair_index, np.uint32)
        info = np.zeros(max(len(idx), 1), SW_BT_DTYPE)
        used = ctypes.c_size_t(0)
        t0 = time.perf_counter()
        rc = self.gpu.L.mmgpu_sw_traceback(self.gpu.ctx, self.handle, _ptr(idx), len(idx), _ptr(info), None, 0, ctypes.byref(used))
        t1 = time.perf_counter()
        if rc != 0 and used.value == 0:
            self.gpu._check(rc)
        buf = np.zeros(max(used.value, 1), np.uint8)
        t2 = time.perf_counter()
        self.gpu._check(self.gpu.L.mmgpu_sw_traceback(self.gpu.ctx, self.handle, _ptr(idx), len(idx), _ptr(info), _ptr(buf),
                                                      buf.size, ctypes.byref(used)))
        self.last_traceback_call_s = (t1 - t0) + (time.perf_counter() - t2)     # the two C-ABI calls, no binding work
        info = info[:len(idx)]
        strs = [bytes(buf[int(r["bt_off"]):int(r["bt_off"]) + int(r["bt_len"])]).decode() for r in info]
        return info, strs

    def gather_owned(self):
        """mmgpu_sw_gather_owned -> (d_full, d_status) device pointers owned by the batch"""
        df, ds = c_p(), c_p()
        self.gpu._check(self.gpu.L.mmgpu_sw_gather_owned(self.gpu.ctx, self.handle, ctypes.byref(df), ctypes.byref(ds)))
        return df.value, ds.value

    def fetch_owned(self):
        """mmgpu_sw_fetch_owned -> (SW_HIT_DTYPE [slots] in merged-list order, records gathered)"""
        out = np.zeros(self.slots, SW_HIT_DTYPE)
        n = ctypes.c_uint32()
        self.gpu._check(self.gpu.L.mmgpu_sw_fetch_owned(self.gpu.ctx, self.handle, _ptr(out), ctypes.byref(n)))
        return out, n.value

    def block_backtrace(self, pair_index, mode="strings"):
        """mmgpu_sw_block_backtrace: the block aligner's start positions / identities / backtraces of int16-range hits.
        -> (SW_BLOCK_DTYPE array, list of backtrace strings (None unless status == 0)).  mode "no_strings": start positions,
        identities and lengths (MMGPU_BLOCK_NO_STRINGS); "starts": start positions only (MMGPU_BLOCK_STARTS_ONLY) - both -> (array, None)"""
        pi = np.ascontiguousarray(pair_index, np.uint32)
        out = np.zeros(len(pi), SW_BLOCK_DTYPE)
        used = ctypes.c_size_t()
        import time
        t0 = time.perf_counter()
        if mode != "strings":
            cap = ctypes.c_size_t(-1 if mode == "no_strings" else -2)
            self.gpu._check(self.gpu.L.mmgpu_sw_block_backtrace(self.gpu.ctx, self.handle, _ptr(pi), len(pi), _ptr(out), None, cap, ctypes.byref(used)))
            self.last_block_call_s = time.perf_counter() - t0
            return out, None
        rc = self.gpu.L.mmgpu_sw_block_backtrace(self.gpu.ctx, self.handle, _ptr(pi), len(pi), _ptr(out), None, 0, ctypes.byref(used))
        bt = np.zeros(max(used.value, 1), np.uint8)
        self.gpu._check(self.gpu.L.mmgpu_sw_block_backtrace(self.gpu.ctx, self.handle, _ptr(pi), len(pi), _ptr(out), _ptr(bt), used.value,
                                                            ctypes.byref(used)))
        self.last_block_call_s = time.perf_counter() - t0      # the two C-ABI calls without the string decoding below
        raw = bt.tobytes()
        strs = [raw[int(o["bt_off"]):int(o["bt_off"]) + int(o["bt_len"])].decode() if o["status"] == 0 else None for o in out]
        return out, strs

    def block_growth(self, pair_index, cap=4096):
        """mmgpu_sw_block_growth (test aid): -> (records, lists): per pair the (i, j, height, width, right) rows of its block list"""
        pi = np.ascontiguousarray(pair_index, dtype=np.uint32)
        out = np.zeros(len(pi), dtype=SW_BLOCK_DTYPE)
        g = np.zeros((len(pi), 1 + 4 * cap), dtype=np.uint32)
        self.gpu._check(self.gpu.L.mmgpu_sw_block_growth(self.gpu.ctx, self.handle, _ptr(pi), len(pi), _ptr(out), _ptr(g), cap))
        lists = []
        for r in g:
            n = int(r[0])
            assert n <= cap, "block list longer than the capture buffer"
            b = r[1:1 + 4 * n].reshape(n, 4).astype(np.int64)
            lists.append(np.stack([b[:, 0], b[:, 1], b[:, 2] >> 16, b[:, 2] & 0xFFFF, b[:, 3]], axis=1))
        return out, lists

    def block_tiers(self):
        """mmgpu_sw_block_tiers: (pairs decided with <= 512-row blocks, pairs that needed the 4096-row launch) of the last call"""
        a, b = ctypes.c_uint32(), ctypes.c_uint32()
        self.gpu._check(self.gpu.L.mmgpu_sw_block_tiers(self.handle, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def kernel_ms(self):
        ms = ctypes.c_float()
        self.gpu._check(self.gpu.L.mmgpu_sw_last_kernel_ms(self.gpu.ctx, self.handle, ctypes.byref(ms)))
        return ms.value

    def kernel_ms_mean(self, last_n=0):
        ms, n = ctypes.c_float(), ctypes.c_uint32()
        self.gpu._check(self.gpu.L.mmgpu_sw_kernel_ms_mean(self.gpu.ctx, self.handle, last_n, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def free(self):
        if self.handle is not None:
            self.gpu.L.mmgpu_sw_free(self.gpu.ctx, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MMGpu:
    """One context = one GPU (mmgpu_ctx)."""

    def __init__(self, device=0):
        self.L = load_library()
        ctx = c_p()
        rc = self.L.mmgpu_init(ctypes.byref(ctx), device)
        if rc != 0:
            raise MMGpuError("mmgpu_init failed (%d): %s" % (rc, self.L.mmgpu_last_error().decode()))
        self.ctx = ctx
        self._db_keep = None

    def _check(self, rc):
        if rc != 0:
            raise MMGpuError("libmmgpu error %d: %s" % (rc, self.L.mmgpu_last_error().decode()))

    def close(self):
        if self.ctx is not None:
            self.L.mmgpu_destroy(self.ctx)
            self.ctx = None

    def set_stream(self, stream_handle):
        self._check(self.L.mmgpu_set_stream(self.ctx, c_p(stream_handle)))

    def synchronize(self):
        self._check(self.L.mmgpu_synchronize(self.ctx))

    def device_info(self):
        cus = ctypes.c_int()
        name = ctypes.create_string_buffer(256)
        self._check(self.L.mmgpu_device_info(self.ctx, ctypes.byref(cus), name, 256))
        return cus.value, name.value.decode()

    def device_memory(self):
        """(free, total) bytes of HBM on the context's device"""
        f, t = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self.L.mmgpu_device_memory(self.ctx, ctypes.byref(f), ctypes.byref(t)))
        return f.value, t.value

    def load_targets(self, residues, offsets, alphabet=21):
        residues = np.ascontiguousarray(residues, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        self._check(self.L.mmgpu_load_targets(self.ctx, _ptr(residues), _ptr(offsets), len(offsets) - 1, alphabet))
        self.n_targets = len(offsets) - 1
        self.global_db_size = None      # mmgpu_load_targets resets the shard description
        self._target_lens = np.diff(offsets.astype(np.int64))

    def _marshal(self, mat, gap_open, gap_extend, queries):
        """queries: list of dicts {q: uint8[], comp_bias: int8[]|None, targets: uint32[], min_start_score: int,
        profile: int8[letters][qlen]|None (profile query: Sequence::getAlignmentProfile, q = consensus)}"""
        mat = np.ascontiguousarray(mat, np.int8)
        par = SwParams(_ptr(mat), mat.shape[0], gap_open, gap_extend)
        arr = (SwQuery * max(len(queries), 1))()
        keep = [mat]
        for i, qd in enumerate(queries):
            q = np.ascontiguousarray(qd["q"], np.uint8)
            cb = None if qd.get("comp_bias") is None else np.ascontiguousarray(qd["comp_bias"], np.int8)
            t = np.ascontiguousarray(qd["targets"], np.uint32)
            prof = None if qd.get("profile") is None else np.ascontiguousarray(qd["profile"], np.int8)
            if prof is not None and (prof.ndim != 2 or prof.shape[1] != len(q)):
                raise ValueError("profile must be [letters][qlen]")
            keep += [q, cb, t, prof]
            arr[i] = SwQuery(_ptr(q), len(q), _ptr(cb), _ptr(t), len(t), int(qd.get("min_start_score", 0)), _ptr(prof),
                             0 if prof is None else prof.shape[0])
        return par, arr, keep

    def sw_batch(self, mat, gap_open, gap_extend, queries, mode=0):
        par, arr, keep = self._marshal(mat, gap_open, gap_extend, queries)
        total = sum(len(qd["targets"]) for qd in queries)
        out = np.zeros(total, SW_HIT_DTYPE)
        self._check(self.L.mmgpu_sw_batch(self.ctx, ctypes.byref(par), ctypes.cast(arr, c_p), len(queries), mode, _ptr(out)))
        del keep
        return out

    def sw_prepare(self, mat, gap_open, gap_extend, queries, mode=0):
        par, arr, keep = self._marshal(mat, gap_open, gap_extend, queries)
        h = c_p()
        self._check(self.L.mmgpu_sw_prepare(self.ctx, ctypes.byref(par), ctypes.cast(arr, c_p), len(queries), mode, ctypes.byref(h)))
        return SwBatch(self, h, keep)

    def sw_marshal_queries(self, mat, gap_open, gap_extend, queries):
        """Query descriptors for sw_prepare_from_pf, built once (the ctypes marshalling is binding cost, not path cost)."""
        qd = [dict(q=x["q"], comp_bias=x.get("comp_bias"), targets=np.zeros(0, np.uint32),
                   min_start_score=x.get("min_start_score", 0), profile=x.get("profile")) for x in queries]
        return self._marshal(mat, gap_open, gap_extend, qd) + (len(queries),)

    def sw_marshal_flat(self, mat, gap_open, gap_extend, residues, offsets, comp_bias_round, min_start_score):
        """sw_marshal_queries for sequence queries held as one residue array: residues uint8[], offsets uint64[n+1],
        comp_bias_round int8[] (indexed like residues) or None, min_start_score int32[n]."""
        mat = np.ascontiguousarray(mat, np.int8)
        residues = np.ascontiguousarray(residues, np.uint8)
        off = np.ascontiguousarray(offsets, np.uint64)
        n = len(off) - 1
        arr = np.zeros(max(n, 1), SW_QUERY_DTYPE)
        arr["q"][:n] = residues.ctypes.data + off[:-1]
        arr["qlen"][:n] = (off[1:] - off[:-1]).astype(np.uint32)
        if comp_bias_round is not None:
            comp_bias_round = np.ascontiguousarray(comp_bias_round, np.int8)
            arr["comp_bias"][:n] = comp_bias_round.ctypes.data + off[:-1]
        arr["min_start_score"][:n] = min_start_score
        par = SwParams(_ptr(mat), mat.shape[0], gap_open, gap_extend)
        return par, arr.ctypes.data_as(c_p), [mat, residues, off, comp_bias_round, arr], n

    def pf_prepare_flat(self, residues, offsets, comp_bias_float, kmer_thr, max_hits=300, min_diag_score=15, ref_bins=0):
        """pf_prepare for sequence queries held as one residue array (no identity ids, no profiles)."""
        residues = np.ascontiguousarray(residues, np.uint8)
        off = np.ascontiguousarray(offsets, np.uint64)
        n = len(off) - 1
        arr = np.zeros(max(n, 1), PF_QUERY_DTYPE)
        arr["q"][:n] = residues.ctypes.data + off[:-1]
        arr["qlen"][:n] = (off[1:] - off[:-1]).astype(np.uint32)
        if comp_bias_float is not None:
            comp_bias_float = np.ascontiguousarray(comp_bias_float, np.float32)
            arr["comp_bias"][:n] = comp_bias_float.ctypes.data + 4 * off[:-1]
        arr["identity_id"] = 0xFFFFFFFF
        par = PfParams(int(kmer_thr), int(max_hits), int(min_diag_score), int(ref_bins), 0, 0, 0)
        h = c_p()
        self._check(self.L.mmgpu_pf_prepare(self.ctx, ctypes.byref(par), arr.ctypes.data_as(c_p), n, ctypes.byref(h)))
        db = getattr(self, "global_db_size", None) or self.n_targets
        return PfBatch(self, h, [residues, off, comp_bias_float, arr], n, min(int(max_hits), db))

    def sw_prepare_from_pf(self, mat, gap_open, gap_extend, queries, pf_batch, mode=1, marshalled=None):
        """Alignment batch over the hit lists of a prefilter batch that has been run, lists stay on the device.
        queries: dicts with q, comp_bias (int8), min_start_score.  fetch() returns n_queries * stride slots."""
        par, arr, keep, n = marshalled if marshalled is not None else self.sw_marshal_queries(mat, gap_open, gap_extend, queries)
        h = c_p()
        self._check(self.L.mmgpu_sw_prepare_from_pf(self.ctx, ctypes.byref(par), ctypes.cast(arr, c_p), n, mode,
                                                    pf_batch.handle, ctypes.byref(h)))
        b = SwBatch(self, h, keep)
        b.slots = n * pf_batch.max_hits
        return b

    # ---- nucleotide alignment step ----
    def nucl_align(self, mat, reverse, queries, pairs, gap_open=5, gap_extend=2, zdrop=40, past_end_query=4, past_end_target=4, wrapped=False):
        """queries: list of uint8 arrays (codes 0..4); pairs: structured NUCL_PAIR_DTYPE array (or list of
        (query, target, diagonal, reverse[, past_end])).  Returns (NUCL_HIT_DTYPE array, list of backtrace strings)."""
        mat = np.ascontiguousarray(mat, np.int8).reshape(-1)
        rev = np.ascontiguousarray(reverse, np.uint8)
        if not (isinstance(pairs, np.ndarray) and pairs.dtype == NUCL_PAIR_DTYPE):
            pa = np.zeros(len(pairs), NUCL_PAIR_DTYPE)
            for i, p in enumerate(pairs):
                pa[i] = (p[0], p[1], p[2] & 0xFFFF, p[3], p[4] if len(p) > 4 else 0)
            pairs = pa
        pairs = np.ascontiguousarray(pairs)
        qs = [np.ascontiguousarray(q, np.uint8) for q in queries]
        arr = (NuclQuery * max(len(qs), 1))()
        for i, q in enumerate(qs):
            arr[i] = NuclQuery(_ptr(q), len(q))
        par = NuclParams(_ptr(mat), _ptr(rev), gap_open, gap_extend, zdrop, past_end_query, past_end_target, int(bool(wrapped)))
        out = np.zeros(len(pairs), NUCL_HIT_DTYPE)
        tl = self._target_lens
        # room for every backtrace; indices are clipped here, the library is the one that rejects bad ones
        ql = np.array([len(q) for q in qs] + [0], np.int64)
        cap = int((ql[np.minimum(pairs["query"], len(qs))] + tl[np.minimum(pairs["target"], len(tl) - 1)] + 2).sum()) if len(pairs) else 16
        bt = np.zeros(max(cap, 16), np.uint8)
        used = ctypes.c_uint64()
        import time
        t0 = time.perf_counter()
        self._check(self.L.mmgpu_nucl_align(self.ctx, ctypes.byref(par), ctypes.cast(arr, c_p), len(qs), _ptr(pairs), len(pairs),
                                            _ptr(out), _ptr(bt), cap, ctypes.byref(used)))
        self.last_nucl_call_s = time.perf_counter() - t0     # the C-ABI call alone (upload, kernel, download), no binding work
        raw = bt.tobytes()
        strs = [raw[int(h["bt_off"]):int(h["bt_off"]) + int(h["bt_len"])].decode() if h["status"] == 0 else None for h in out]
        return out, strs

    # ---- prefilter ----
    def pf_load_index(self, k, alphabet, spaced, score3, index3, offsets, entry_ids, entry_pos, ungapped_mat,
                      score2=None, index2=None):
        score3 = None if score3 is None else np.ascontiguousarray(score3, np.int16)     # None: exact k-mer matching only
        index3 = None if index3 is None else np.ascontiguousarray(index3, np.uint32)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        entry_ids = np.ascontiguousarray(entry_ids, np.uint32)
        entry_pos = np.ascontiguousarray(entry_pos, np.uint16)
        ungapped_mat = np.ascontiguousarray(ungapped_mat, np.int8)
        if score2 is not None:
            score2 = np.ascontiguousarray(score2, np.int16)
            index2 = np.ascontiguousarray(index2, np.uint32)
        d = PfIndexDesc(k, alphabet, int(spaced), _ptr(score3), _ptr(index3), 0 if score3 is None else score3.shape[1], _ptr(score2), _ptr(index2),
                        0 if score2 is None else score2.shape[1], _ptr(offsets),
                        _ptr(entry_ids), _ptr(entry_pos), None, len(entry_ids), _ptr(ungapped_mat))
        self._check(self.L.mmgpu_pf_load_index(self.ctx, ctypes.byref(d)))

    def pf_build_index(self, k, alphabet, spaced, score3, index3, kmer_submat16, kmer_thr, ungapped_mat, score2=None,
                       index2=None):
        """Index construction on the device over the loaded targets (masking off).  score3 / index3 None: no similar-k-mer
        tables (exact k-mer matching only; k in 4..15, e.g. nucleotide databases)."""
        score3 = None if score3 is None else np.ascontiguousarray(score3, np.int16)
        index3 = None if index3 is None else np.ascontiguousarray(index3, np.uint32)
        ungapped_mat = np.ascontiguousarray(ungapped_mat, np.int8)
        km = np.ascontiguousarray(kmer_submat16, np.int16)
        if score2 is not None:
            score2 = np.ascontiguousarray(score2, np.int16)
            index2 = np.ascontiguousarray(index2, np.uint32)
        d = PfIndexDesc(k, alphabet, int(spaced), _ptr(score3), _ptr(index3), 0 if score3 is None else score3.shape[1], _ptr(score2), _ptr(index2),
                        0 if score2 is None else score2.shape[1], None, None, None, None, 0, _ptr(ungapped_mat))
        self._check(self.L.mmgpu_pf_build_index(self.ctx, ctypes.byref(d), _ptr(km), int(kmer_thr)))

    def db_save(self, path, source_fp, index_fp=0):
        """mmgpu_db_save: targets (+ masked view, + index when index_fp != 0) of this context into one file in the device layout"""
        self.L.mmgpu_db_save.argtypes = [c_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64]
        self._check(self.L.mmgpu_db_save(self.ctx, str(path).encode(), int(source_fp), int(index_fp)))

    def db_load(self, path, source_fp, index_fp=0, k=6, alphabet=21, spaced=True, score3=None, index3=None, ungapped_mat=None,
                score2=None, index2=None):
        """mmgpu_db_load -> True when the file matched and is resident now, False when the caller has to build (MMGPU_ERR_STATE)"""
        d = None
        if index_fp:
            score3 = None if score3 is None else np.ascontiguousarray(score3, np.int16)
            index3 = None if index3 is None else np.ascontiguousarray(index3, np.uint32)
            ungapped_mat = np.ascontiguousarray(ungapped_mat, np.int8)
            if score2 is not None:
                score2 = np.ascontiguousarray(score2, np.int16)
                index2 = np.ascontiguousarray(index2, np.uint32)
            d = PfIndexDesc(k, alphabet, int(spaced), _ptr(score3), _ptr(index3), 0 if score3 is None else score3.shape[1], _ptr(score2),
                            _ptr(index2), 0 if score2 is None else score2.shape[1], None, None, None, None, 0, _ptr(ungapped_mat))
        self.L.mmgpu_db_load.argtypes = [c_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, c_p]
        rc = self.L.mmgpu_db_load(self.ctx, str(path).encode(), int(source_fp), int(index_fp), ctypes.byref(d) if d is not None else None)
        if rc == -3:      # MMGPU_ERR_STATE: no such file / another database / other index parameters
            return False
        self._check(rc)
        self._db_keep = None
        info = db_probe(path, self.L)
        self.n_targets = int(info["n_targets"])
        return True

    def pf_mask_targets(self, likelihood_ratios, min_mask_prob=0.9, mask_letter=20):
        """mmgpu_pf_mask_targets: tantan masking of the resident targets for the prefilter -> residues masked"""
        lr = np.ascontiguousarray(likelihood_ratios, np.float64)
        n = ctypes.c_uint64()
        self.L.mmgpu_pf_mask_targets.argtypes = [c_p, c_p, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
        self._check(self.L.mmgpu_pf_mask_targets(self.ctx, _ptr(lr), lr.shape[0], float(min_mask_prob), int(mask_letter), ctypes.byref(n)))
        return n.value

    def pf_debug_masked_targets(self, offsets):
        off = np.ascontiguousarray(offsets, np.uint64)
        res = np.zeros(int(off[-1]), np.uint8)
        self.L.mmgpu_pf_debug_masked_targets.argtypes = [c_p, c_p, ctypes.c_uint32, c_p]
        self._check(self.L.mmgpu_pf_debug_masked_targets(self.ctx, _ptr(off), len(off) - 1, _ptr(res)))
        return res

    def pf_debug_index(self, k, alphabet):
        ne = ctypes.c_uint64()
        self._check(self.L.mmgpu_pf_debug_index(self.ctx, None, None, None, ctypes.byref(ne)))
        off = np.zeros((alphabet - 1) ** k + 1, np.uint64)
        ids = np.zeros(max(ne.value, 1), np.uint32)
        pos = np.zeros(max(ne.value, 1), np.uint16)
        self._check(self.L.mmgpu_pf_debug_index(self.ctx, _ptr(off), _ptr(ids), _ptr(pos), ctypes.byref(ne)))
        return off, ids[:ne.value], pos[:ne.value]

    def _pf_marshal(self, queries):
        """queries: list of dicts {q: uint8[], comp_bias: float32[]|None, identity_id: int|None}; a profile query adds
        profile_score int16 [qlen][row], profile_index uint32 [qlen][row] (Sequence::profile_score / profile_index) and
        profile int8 [20][qlen] (Sequence::getAlignmentProfile)"""
        arr = (PfQuery * max(len(queries), 1))()
        keep = []
        for i, qd in enumerate(queries):
            q = np.ascontiguousarray(qd["q"], np.uint8)
            cb = None if qd.get("comp_bias") is None else np.ascontiguousarray(qd["comp_bias"], np.float32)
            ident = qd.get("identity_id")
            ps = pi = pa = None
            row = 0
            if qd.get("profile") is not None:
                ps = np.ascontiguousarray(qd["profile_score"], np.int16)
                pi = np.ascontiguousarray(qd["profile_index"], np.uint32)
                pa = np.ascontiguousarray(qd["profile"], np.int8)
                if ps.shape != pi.shape or ps.shape[0] != len(q) or pa.shape != (20, len(q)):
                    raise ValueError("profile query: profile_score / profile_index must be [qlen][row], profile [20][qlen]")
                row = ps.shape[1]
            keep += [q, cb, ps, pi, pa]
            arr[i] = PfQuery(_ptr(q), len(q), _ptr(cb), 0xFFFFFFFF if ident is None else int(ident), _ptr(ps), _ptr(pi), row, _ptr(pa))
        return arr, keep

    def pf_prepare(self, queries, kmer_thr, max_hits=300, min_diag_score=15, ref_bins=0, exact=False, nucleotide=False, kmer_score=False):
        arr, keep = self._pf_marshal(queries)
        par = PfParams(int(kmer_thr), int(max_hits), int(min_diag_score), int(ref_bins), int(exact), int(nucleotide), int(kmer_score))
        h = c_p()
        self._check(self.L.mmgpu_pf_prepare(self.ctx, ctypes.byref(par), ctypes.cast(arr, c_p), len(queries), ctypes.byref(h)))
        db = getattr(self, "global_db_size", None) or self.n_targets
        b = PfBatch(self, h, keep, len(queries), min(int(max_hits), db))
        b.prepared_with = (par, arr, len(queries))      # what mmgpu_pf_exchange_redo_unsplit is given again
        return b

    def pf_batch(self, queries, kmer_thr, max_hits=300, min_diag_score=15, ref_bins=0, exact=False, nucleotide=False, kmer_score=False):
        b = self.pf_prepare(queries, kmer_thr, max_hits, min_diag_score, ref_bins, exact, nucleotide, kmer_score)
        b.run()
        out = b.fetch()
        b.free()
        return out

    def pf_set_shard(self, n_shards, shard, global_db_size, global_ids, shard_of, local_id):
        """this context holds shard `shard` of a database of global_db_size targets (None-like: pf_clear_shard)"""
        g = np.ascontiguousarray(global_ids, np.uint32)
        so = np.ascontiguousarray(shard_of, np.uint32)
        li = np.ascontiguousarray(local_id, np.uint32)
        sh = PfShard(int(n_shards), int(shard), int(global_db_size), _ptr(g), _ptr(so), _ptr(li))
        self._check(self.L.mmgpu_pf_set_shard(self.ctx, ctypes.byref(sh)))
        self.global_db_size = int(global_db_size)

    def pf_clear_shard(self):
        self._check(self.L.mmgpu_pf_set_shard(self.ctx, None))
        self.global_db_size = None

    def pf_localize_lists(self, d_hits_ptr, d_counts_ptr, nq, stride, d_local_hits_ptr, d_local_counts_ptr, d_local_slot_ptr):
        self._check(self.L.mmgpu_pf_localize_lists(self.ctx, c_p(d_hits_ptr), c_p(d_counts_ptr), nq, stride, c_p(d_local_hits_ptr),
                                                   c_p(d_local_counts_ptr), c_p(d_local_slot_ptr)))

    def sw_prepare_from_lists(self, mat, gap_open, gap_extend, queries, d_hits_ptr, d_counts_ptr, stride, mode=1, marshalled=None):
        """Alignment batch over device-resident lists of LOCAL target ids ([nq][stride] PF_HIT_DTYPE + counts)."""
        par, arr, keep, n = marshalled if marshalled is not None else self.sw_marshal_queries(mat, gap_open, gap_extend, queries)
        h = c_p()
        self._check(self.L.mmgpu_sw_prepare_from_lists(self.ctx, ctypes.byref(par), ctypes.cast(arr, c_p), n, mode, c_p(d_hits_ptr),
                                                       c_p(d_counts_ptr), stride, ctypes.byref(h)))
        b = SwBatch(self, h, keep)
        b.slots = n * stride
        return b

    # ---- communicator owned by the library (RCCL) ----
    def comm_unique_id(self):
        b = np.zeros(128, np.uint8)
        self._check(self.L.mmgpu_comm_unique_id(_ptr(b)))
        return b

    def comm_init_rank(self, comm_id, rank, n_ranks):
        b = np.ascontiguousarray(comm_id, np.uint8)
        assert b.size == 128
        self._check(self.L.mmgpu_comm_init_rank(self.ctx, _ptr(b), int(rank), int(n_ranks)))

    def comm_info(self):
        r, n = ctypes.c_int(), ctypes.c_int()
        t = ctypes.create_string_buffer(32)
        self._check(self.L.mmgpu_comm_info(self.ctx, ctypes.byref(r), ctypes.byref(n), t, 32))
        return r.value, n.value, t.value.decode()

    def sw_prepare_owned(self, mat, gap_open, gap_extend, queries, pf_batch, mode=1, marshalled=None):
        """alignment batch of the pairs of pf_batch's merged lists whose target this context's shard holds"""
        par, arr, keep, n = marshalled if marshalled is not None else self.sw_marshal_queries(mat, gap_open, gap_extend, queries)
        h = c_p()
        self._check(self.L.mmgpu_sw_prepare_owned(self.ctx, ctypes.byref(par), ctypes.cast(arr, c_p), n, mode, pf_batch.handle, ctypes.byref(h)))
        b = SwBatch(self, h, keep)
        b.slots = n * pf_batch.max_hits
        return b

    def pf_merge_splits(self, d_hits_ptr, d_counts_ptr, n_splits, nq, stride, id_offsets, d_out_hits_ptr, d_out_counts_ptr):
        off = np.ascontiguousarray(id_offsets, np.uint32)
        self._check(self.L.mmgpu_pf_merge_splits(self.ctx, c_p(d_hits_ptr), c_p(d_counts_ptr), n_splits, nq, stride, _ptr(off),
                                                 c_p(d_out_hits_ptr), c_p(d_out_counts_ptr)))


def split_max_hits(max_hits, n_splits):
    """maxResListLen of one target split (Prefiltering.cpp:391-394)."""
    if n_splits <= 1:
        return int(max_hits)
    import math
    return max(1, int(max_hits) // n_splits + int(4 * math.sqrt(float(max_hits) / float(n_splits))))


def merge_hit_lists_host(lists, id_offsets):
    """Host mirror of mmgpu_pf_merge_splits for one query (used by the CPU tests of the multi-GPU path):
    lists = [structured PF_HIT_DTYPE arrays per split] -> concatenated, global ids, sorted (|score| desc, id asc)."""
    parts = []
    for h, off in zip(lists, id_offsets):
        h = h.copy()
        h["id"] = h["id"] + np.uint32(off)
        parts.append(h)
    allh = np.concatenate(parts) if parts else np.zeros(0, PF_HIT_DTYPE)
    order = np.lexsort((allh["id"], -np.abs(allh["score"].astype(np.int64))))
    return allh[order]


def merge_exchange_host(records, max_hits, min_diag_score, ref_bins, self_score, identity_global=None):
    """Host mirror of mmgpu_pf_merge_exchange for one query (CPU tests of the multi-GPU path, never a fallback):
    records = PF_XHIT_DTYPE array, the concatenation of every shard's exchange list -> PF_HIT_DTYPE list (global ids) in
    the reference's final order.  Follows pf_xmerge_kernel step by step (QueryMatcher.cpp:161-241, 401-458, 563-586)."""
    r = np.asarray(records)
    cnt = np.minimum(r["score"], 255).astype(np.int64)
    hist = np.bincount(cnt, minlength=256)
    found, thr = 0, 0
    for thr in range(255, 0, -1):
        found += int(hist[thr])
        if found >= max_hits:
            break
    else:
        thr = 0
    dthr = max(int(min_diag_score), thr)
    trunc = dthr >= 255
    ms = min(max(int(self_score) - 255, 1), 65535)
    ident = 0xFFFFFFFF if identity_global is None else int(identity_global)
    if trunc:
        ns = np.minimum(r["score"].astype(np.int64) - 255, 65535).astype(np.float32)
        resc = ((ns / np.float32(ms)) * np.float32(255.0)).astype(np.float32).astype(np.float64) + 0.5
        kc = resc.astype(np.int64) & 0xFF
        elig = (cnt >= 255) & (r["id"] != ident)
    else:
        kc = cnt
        elig = (cnt >= dthr) & (r["id"] != ident)
    idx = np.nonzero(elig)[0]
    key_a = ((255 - kc[idx]) << 11) | (r["id"][idx].astype(np.int64) & (ref_bins - 1))
    order = np.lexsort((r["id"][idx], r["order"][idx], key_a))
    has_ident = 1 if ident != 0xFFFFFFFF else 0
    want = max(max_hits - has_ident, 0)
    sel = idx[order][:want]
    if trunc:
        pref = 255 + (kc[sel] * ms) // 255
    else:
        pref = np.where(cnt[sel] >= 255, r["score"][sel].astype(np.int64), cnt[sel])
    out = np.zeros(len(sel) + (has_ident if max_hits > 0 else 0), PF_HIT_DTYPE)
    fin = np.lexsort((r["id"][sel], -pref))
    o = has_ident if max_hits > 0 else 0
    out["id"][o:] = r["id"][sel][fin]
    out["score"][o:] = pref[fin]
    out["diagonal"][o:] = r["diagonal"][sel][fin]
    if has_ident and max_hits > 0:
        out[0] = (ident, 65535, 0, 0)
    return out


def select_exchange_host(records, max_hits, min_diag_score, ref_bins, self_score):
    """Host mirror of the shard-side selection (pf_select_kernel<true>) for one query: records = PF_XHIT_DTYPE array of
    ALL surviving elements of the shard (one per target) -> the shard's exchange list: its top max_hits by the unsplit
    run's order, the mode (truncated threshold or not) decided from the shard's own histogram."""
    r = np.asarray(records)
    cnt = np.minimum(r["score"], 255).astype(np.int64)
    hist = np.bincount(cnt, minlength=256)
    found, thr = 0, 0
    for thr in range(255, 0, -1):
        found += int(hist[thr])
        if found >= max_hits:
            break
    else:
        thr = 0
    dthr = max(int(min_diag_score), thr)
    ms = min(max(int(self_score) - 255, 1), 65535)
    if dthr >= 255:
        ns = np.minimum(r["score"].astype(np.int64) - 255, 65535).astype(np.float32)
        resc = ((ns / np.float32(ms)) * np.float32(255.0)).astype(np.float32).astype(np.float64) + 0.5
        kc = resc.astype(np.int64) & 0xFF
        elig = cnt >= 255
    else:
        kc = cnt
        elig = cnt >= dthr
    idx = np.nonzero(elig)[0]
    key_a = ((255 - kc[idx]) << 11) | (r["id"][idx].astype(np.int64) & (ref_bins - 1))
    order = np.lexsort((r["id"][idx], r["order"][idx], key_a))
    return r[idx[order][:max_hits]]


class MMGpuMulti:
    """mmgpu_init_multi: one process, several contexts (one per shard of the target database) + their communicator."""

    def __init__(self, device_ids):
        self.L = load_library()
        ids = np.ascontiguousarray(device_ids, np.int32)
        h = c_p()
        rc = self.L.mmgpu_init_multi(ctypes.byref(h), _ptr(ids), len(ids))
        if rc != 0:
            raise MMGpuError("mmgpu_init_multi failed (%d): %s" % (rc, self.L.mmgpu_last_error().decode()))
        self.h = h
        self.n = len(ids)
        self._keep = None

    def _check(self, rc):
        if rc != 0:
            raise MMGpuError("libmmgpu error %d: %s" % (rc, self.L.mmgpu_last_error().decode()))

    def close(self):
        if self.h is not None:
            self.L.mmgpu_destroy_multi(self.h)
            self.h = None

    def transport(self):
        t = ctypes.create_string_buffer(32)
        self._check(self.L.mmgpu_comm_info(c_p(self.L.mmgpu_multi_ctx(self.h, 0)), None, None, t, 32))
        return t.value.decode()

    def synchronize(self):
        self._check(self.L.mmgpu_multi_synchronize(self.h))

    def load_targets(self, residues, offsets, alphabet=21):
        residues = np.ascontiguousarray(residues, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        self._check(self.L.mmgpu_multi_load_targets(self.h, _ptr(residues), _ptr(offsets), len(offsets) - 1, alphabet))
        self.n_targets = len(offsets) - 1

    def pf_mask_targets(self, likelihood_ratios, min_mask_prob=0.9, mask_letter=20):
        """mmgpu_multi_pf_mask_targets: tantan masking of every shard -> residues masked over all shards"""
        lr = np.ascontiguousarray(likelihood_ratios, np.float64)
        n = ctypes.c_uint64()
        self.L.mmgpu_multi_pf_mask_targets.argtypes = [c_p, c_p, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
        self._check(self.L.mmgpu_multi_pf_mask_targets(self.h, _ptr(lr), lr.shape[0], float(min_mask_prob), int(mask_letter), ctypes.byref(n)))
        return n.value

    def pf_build_index(self, k, alphabet, spaced, score3, index3, kmer_submat16, kmer_thr, ungapped_mat):
        score3 = np.ascontiguousarray(score3, np.int16)
        index3 = np.ascontiguousarray(index3, np.uint32)
        um = np.ascontiguousarray(ungapped_mat, np.int8)
        km = np.ascontiguousarray(kmer_submat16, np.int16)
        d = PfIndexDesc(k, alphabet, int(spaced), _ptr(score3), _ptr(index3), score3.shape[1], None, None, 0, None, None, None, None, 0, _ptr(um))
        self._check(self.L.mmgpu_multi_pf_build_index(self.h, ctypes.byref(d), _ptr(km), int(kmer_thr)))

    def pf_prepare(self, queries, kmer_thr, max_hits=300, min_diag_score=15, ref_bins=0):
        """queries as for MMGpu.pf_prepare; identity_id is the GLOBAL id.  -> opaque batch handle (c_p)"""
        arr, keep = MMGpu._pf_marshal(None, queries)
        par = PfParams(int(kmer_thr), int(max_hits), int(min_diag_score), int(ref_bins), 0, 0, 0)
        h = c_p()
        self._check(self.L.mmgpu_multi_pf_prepare(self.h, ctypes.byref(par), ctypes.cast(arr, c_p), len(queries), ctypes.byref(h)))
        self._keep = keep
        return h

    def pf_run(self, batch):
        self._check(self.L.mmgpu_multi_pf_run(self.h, batch))

    def pf_fetch(self, batch, nq):
        stride = int(self.L.mmgpu_multi_pf_stride(batch))
        hits = np.zeros((nq, max(stride, 1)), PF_HIT_DTYPE)
        counts = np.zeros(nq, np.uint32)
        status = np.zeros(nq, np.int32)
        self._check(self.L.mmgpu_multi_pf_fetch(self.h, batch, _ptr(hits), max(stride, 1), _ptr(counts), _ptr(status)))
        return hits, counts, status

    def has_unsplit(self):
        """the first device also holds the whole database in a context of its own (queries flagged inexact are re-run there)"""
        return bool(self.L.mmgpu_multi_has_unsplit(self.h))

    def pf_redone(self, batch):
        """(queries of the batch's last run that were re-run against the whole database, of those left to the host)"""
        a, b = ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self.L.mmgpu_multi_pf_redone(batch, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def pf_free(self, batch):
        self.L.mmgpu_multi_pf_free(self.h, batch)

    def sw_from_pf(self, mat, gap_open, gap_extend, queries, batch, nq, mode=1):
        """-> (SW_HIT_DTYPE [nq, stride] in merged-list order, forward cells, slowest context's kernel ms)"""
        qd = [dict(q=x["q"], comp_bias=x.get("comp_bias"), targets=np.zeros(0, np.uint32), min_start_score=x.get("min_start_score", 0))
              for x in queries]
        par, arr, keep = MMGpu._marshal(None, mat, gap_open, gap_extend, qd)
        stride = int(self.L.mmgpu_multi_pf_stride(batch))
        out = np.zeros((nq, max(stride, 1)), SW_HIT_DTYPE)
        cells, ms = ctypes.c_uint64(), ctypes.c_float()
        self._check(self.L.mmgpu_multi_sw_from_pf(self.h, ctypes.byref(par), ctypes.cast(arr, c_p), nq, mode, batch, _ptr(out),
                                                  ctypes.byref(cells), ctypes.byref(ms)))
        del keep
        return out, cells.value, ms.value
