"""TEST INFRASTRUCTURE - ctypes bindings for the checkers.

  Oracle  : oracle/_build/libmmoracle.so  (plain-C restatement, oracle/*.c)
  RefLib  : oracle/_ref/libmmref.so       (the REAL reference classes; exists only where
                                           oracle/Makefile `ref` target was run, i.e. where
                                           /root/reference is present, or where the prebuilt
                                           .so travelled with the snapshot)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libmmoracle.so")
REF_SO = os.path.join(HERE, "_ref", "libmmref.so")
REFERENCE_ROOT = "/root/reference"

c_p = ctypes.c_void_p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_p)


class SwRes(ctypes.Structure):
    _fields_ = [("score", ctypes.c_int32), ("q_start", ctypes.c_int32), ("q_end", ctypes.c_int32),
                ("t_start", ctypes.c_int32), ("t_end", ctypes.c_int32), ("word", ctypes.c_int32),
                ("ident", ctypes.c_uint32), ("bt_len", ctypes.c_int32)]


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


class Oracle:
    """Plain-C restatement."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = self.L = ctypes.CDLL(ORACLE_SO)
        L.mmo_sw_align.restype = ctypes.c_int
        L.mmo_sw_bias.restype = ctypes.c_int
        L.mmo_sw_check_params.restype = ctypes.c_int
        L.mmo_sw_score_identical.restype = ctypes.c_int
        L.mmo_sw_cov.restype = ctypes.c_float

    def round_comp_bias(self, bias_f):
        bias_f = np.ascontiguousarray(bias_f, np.float32)
        out = np.zeros(len(bias_f), np.int8)
        self.L.mmo_round_comp_bias(_ptr(bias_f), len(bias_f), _ptr(out))
        return out

    def comp_bias(self, submat16, pback, seq, scale=1.0):
        submat16 = np.ascontiguousarray(submat16, np.int16)
        pback = np.ascontiguousarray(pback, np.float64)
        seq = np.ascontiguousarray(seq, np.uint8)
        out = np.zeros(len(seq), np.float32)
        self.L.mmo_comp_bias(_ptr(submat16), _ptr(pback), submat16.shape[0], _ptr(seq), len(seq),
                             ctypes.c_float(scale), _ptr(out))
        return out

    def sw_bias(self, mat, cb, qlen):
        return self.L.mmo_sw_bias(_ptr(mat), mat.shape[0], _ptr(cb), qlen)

    def sw_align(self, q, cb, t, mat, go, ge, need_start=False, need_bt=False):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        mat = np.ascontiguousarray(mat, np.int8)
        cbp = None if cb is None else np.ascontiguousarray(cb, np.int8)
        r = SwRes()
        cap = len(q) + len(t) + 8
        bt = ctypes.create_string_buffer(cap)
        rc = self.L.mmo_sw_align(_ptr(q), len(q), _ptr(cbp), _ptr(t), len(t), _ptr(mat), mat.shape[0], go, ge,
                                 int(need_start), int(need_bt), ctypes.byref(r), bt, cap)
        if rc != 0:
            raise RuntimeError("mmo_sw_align rc=%d" % rc)
        return dict(score=r.score, q_start=r.q_start, q_end=r.q_end, t_start=r.t_start, t_end=r.t_end,
                    word=r.word, ident=r.ident, bt=bt.value.decode() if r.bt_len else "")

    def sw_batch_score(self, q, cb, tdata, toff, ids, mat, go, ge):
        q = np.ascontiguousarray(q, np.uint8)
        mat = np.ascontiguousarray(mat, np.int8)
        cbp = None if cb is None else np.ascontiguousarray(cb, np.int8)
        tdata = np.ascontiguousarray(tdata, np.uint8)
        toff = np.ascontiguousarray(toff, np.uint64)
        ids = np.ascontiguousarray(ids, np.uint32)
        n = len(ids)
        score = np.zeros(n, np.int32)
        qe = np.zeros(n, np.int32)
        te = np.zeros(n, np.int32)
        word = np.zeros(n, np.int32)
        self.L.mmo_sw_batch_score(_ptr(q), len(q), _ptr(cbp), _ptr(tdata), _ptr(toff), _ptr(ids), n, _ptr(mat),
                                  mat.shape[0], go, ge, _ptr(score), _ptr(qe), _ptr(te), _ptr(word))
        return score, qe, te, word

    def sw_score_identical(self, q, cb, t, mat):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        cbp = None if cb is None else np.ascontiguousarray(cb, np.int8)
        return self.L.mmo_sw_score_identical(_ptr(q), len(q), _ptr(cbp), _ptr(t), _ptr(mat), mat.shape[0])


class RefSwRes(ctypes.Structure):
    _fields_ = [("score", ctypes.c_uint32), ("q_start", ctypes.c_int32), ("q_end", ctypes.c_int32),
                ("t_start", ctypes.c_int32), ("t_end", ctypes.c_int32), ("word", ctypes.c_int32),
                ("q_cov", ctypes.c_float), ("t_cov", ctypes.c_float), ("evalue", ctypes.c_double),
                ("ident", ctypes.c_uint32), ("bt_len", ctypes.c_int32)]


def ref_available():
    return os.path.exists(REF_SO)


def ref_matrix_available():
    return os.path.exists(os.path.join(REFERENCE_ROOT, "data", "blosum62.out"))


class RefLib:
    """The real reference (needs /root/reference/data/*.out at run time for the matrices)."""

    def __init__(self, matrix="blosum62.out", bit_factor=2.0, score_bias=0.0, max_len=70000, gap_open=11,
                 gap_extend=1, comp_bias=True, db_residues=1000000, serialized=None):
        """matrix: file name under /root/reference/data, or - where that tree is absent (GPU box) -
        `serialized` = the "name.out:DATA" bytes stored in tests/golden/matrices.npz."""
        L = self.L = ctypes.CDLL(REF_SO)
        L.mmref_new.restype = c_p
        L.mmref_new.argtypes = [ctypes.c_char_p, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
        L.mmref_evalue.restype = ctypes.c_double
        L.mmref_evalue.argtypes = [c_p, ctypes.c_double, ctypes.c_double]
        L.mmref_bitscore.restype = ctypes.c_double
        L.mmref_bitscore.argtypes = [c_p, ctypes.c_double]
        L.mmref_alphabet_size.argtypes = [c_p]
        path = bytes(serialized) if serialized is not None else os.path.join(REFERENCE_ROOT, "data", matrix).encode()
        self.c = c_p(L.mmref_new(path, bit_factor, score_bias, max_len, gap_open, gap_extend, int(comp_bias),
                                 db_residues))
        self.alphabet = L.mmref_alphabet_size(self.c)

    def serialized_matrix(self):
        b = ctypes.create_string_buffer(1 << 16)
        self.L.mmref_serialized_matrix.argtypes = [c_p, ctypes.c_char_p, ctypes.c_int]
        n = self.L.mmref_serialized_matrix(self.c, b, 1 << 16)
        assert n > 0
        return b.raw[:n]

    def matrix(self):
        m = np.zeros((self.alphabet, self.alphabet), np.int8)
        self.L.mmref_get_matrix(self.c, _ptr(m))
        return m

    def num2aa(self):
        b = ctypes.create_string_buffer(self.alphabet + 1)
        self.L.mmref_num2aa(self.c, b)
        return b.raw[: self.alphabet].decode()

    def aa2num(self, s):
        out = np.zeros(len(s), np.uint8)
        self.L.mmref_aa2num(self.c, s.encode(), len(s), _ptr(out))
        return out

    def comp_bias(self, num, scale=1.0):
        num = np.ascontiguousarray(num, np.uint8)
        out = np.zeros(len(num), np.float32)
        self.L.mmref_comp_bias(self.c, _ptr(num), len(num), ctypes.c_float(scale), _ptr(out))
        return out

    def sw_set_query(self, q):
        self._q = np.ascontiguousarray(q, np.uint8)
        self.L.mmref_sw_set_query(self.c, _ptr(self._q), len(self._q))

    def sw_align(self, t, mode=0, evalue_thr=1e300, cov_mode=0, cov_thr=0.0):
        t = np.ascontiguousarray(t, np.uint8)
        r = RefSwRes()
        cap = len(self._q) + len(t) + 8
        bt = ctypes.create_string_buffer(cap)
        self.L.mmref_sw_align(self.c, _ptr(t), len(t), mode, ctypes.c_double(evalue_thr), cov_mode,
                              ctypes.c_float(cov_thr), ctypes.byref(r), bt, cap)
        return dict(score=int(r.score), q_start=r.q_start, q_end=r.q_end, t_start=r.t_start, t_end=r.t_end,
                    word=r.word, q_cov=r.q_cov, t_cov=r.t_cov, evalue=r.evalue, ident=r.ident,
                    bt=bt.value.decode() if r.bt_len else "")

    def sw_batch_score(self, tdata, toff, ids):
        tdata = np.ascontiguousarray(tdata, np.uint8)
        toff = np.ascontiguousarray(toff, np.uint64)
        ids = np.ascontiguousarray(ids, np.uint32)
        n = len(ids)
        score = np.zeros(n, np.uint32)
        qe = np.zeros(n, np.int32)
        te = np.zeros(n, np.int32)
        self.L.mmref_sw_batch_score(self.c, _ptr(tdata), _ptr(toff), _ptr(ids), n, _ptr(score), _ptr(qe), _ptr(te))
        return score.astype(np.int32), qe, te

    def evalue(self, score, qlen):
        return self.L.mmref_evalue(self.c, float(score), float(qlen))

    def bitscore(self, score):
        return self.L.mmref_bitscore(self.c, float(score))
