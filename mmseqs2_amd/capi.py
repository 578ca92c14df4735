"""ctypes binding of include/mmgpu.h.  No compute happens in Python and nothing here falls back to a CPU
implementation: a missing library or a failing HIP call raises MMGpuError."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "lib", "libmmgpu.so")

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
                ("n_targets", ctypes.c_uint32), ("min_start_score", ctypes.c_int32)]


SW_HIT_DTYPE = np.dtype([("score", np.int32), ("q_end", np.int32), ("t_end", np.int32), ("q_start", np.int32),
                         ("t_start", np.int32), ("word", np.int32)])

# every symbol include/mmgpu.h declares (tests check the built library exports all of them)
EXPORTED_SYMBOLS = [
    "mmgpu_init", "mmgpu_destroy", "mmgpu_last_error", "mmgpu_set_stream", "mmgpu_synchronize",
    "mmgpu_device_info", "mmgpu_host_comp_bias", "mmgpu_host_round_comp_bias", "mmgpu_load_targets", "mmgpu_sw_batch", "mmgpu_sw_prepare", "mmgpu_sw_run",
    "mmgpu_sw_fetch", "mmgpu_sw_batch_stats", "mmgpu_sw_last_kernel_ms", "mmgpu_sw_kernel_ms_mean", "mmgpu_sw_free",
]


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
    L.mmgpu_host_comp_bias.argtypes = [c_p, c_p, ctypes.c_int, c_p, ctypes.c_uint32, ctypes.c_float, c_p]
    L.mmgpu_host_round_comp_bias.argtypes = [c_p, ctypes.c_uint32, c_p]
    L.mmgpu_load_targets.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, ctypes.c_int]
    L.mmgpu_sw_batch.argtypes = [c_p, ctypes.POINTER(SwParams), c_p, ctypes.c_uint32, ctypes.c_int, c_p]
    L.mmgpu_sw_prepare.argtypes = [c_p, ctypes.POINTER(SwParams), c_p, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(c_p)]
    L.mmgpu_sw_run.argtypes = [c_p, c_p]
    L.mmgpu_sw_fetch.argtypes = [c_p, c_p, c_p]
    L.mmgpu_sw_batch_stats.argtypes = [c_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.mmgpu_sw_last_kernel_ms.argtypes = [c_p, c_p, ctypes.POINTER(ctypes.c_float)]
    L.mmgpu_sw_kernel_ms_mean.argtypes = [c_p, c_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32)]
    L.mmgpu_sw_free.argtypes = [c_p, c_p]
    L.mmgpu_sw_free.restype = None
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


class SwBatch:
    """A prepared (HBM-resident) Smith-Waterman batch."""

    def __init__(self, gpu, handle, keep):
        self.gpu, self.handle, self._keep = gpu, handle, keep
        cells, pairs = ctypes.c_uint64(), ctypes.c_uint64()
        gpu._check(gpu.L.mmgpu_sw_batch_stats(handle, ctypes.byref(cells), ctypes.byref(pairs)))
        self.cells, self.pairs = cells.value, pairs.value

    def run(self):
        self.gpu._check(self.gpu.L.mmgpu_sw_run(self.gpu.ctx, self.handle))

    def fetch(self):
        out = np.zeros(self.pairs, SW_HIT_DTYPE)
        self.gpu._check(self.gpu.L.mmgpu_sw_fetch(self.gpu.ctx, self.handle, _ptr(out)))
        return out

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

    def load_targets(self, residues, offsets, alphabet=21):
        residues = np.ascontiguousarray(residues, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        self._check(self.L.mmgpu_load_targets(self.ctx, _ptr(residues), _ptr(offsets), len(offsets) - 1, alphabet))
        self.n_targets = len(offsets) - 1

    def _marshal(self, mat, gap_open, gap_extend, queries):
        """queries: list of dicts {q: uint8[], comp_bias: int8[]|None, targets: uint32[], min_start_score: int}"""
        mat = np.ascontiguousarray(mat, np.int8)
        par = SwParams(_ptr(mat), mat.shape[0], gap_open, gap_extend)
        arr = (SwQuery * max(len(queries), 1))()
        keep = [mat]
        for i, qd in enumerate(queries):
            q = np.ascontiguousarray(qd["q"], np.uint8)
            cb = None if qd.get("comp_bias") is None else np.ascontiguousarray(qd["comp_bias"], np.int8)
            t = np.ascontiguousarray(qd["targets"], np.uint32)
            keep += [q, cb, t]
            arr[i] = SwQuery(_ptr(q), len(q), _ptr(cb), _ptr(t), len(t), int(qd.get("min_start_score", 0)))
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
