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
# the same reference with its block-aligner branch alive (the restated block aligner behind the crate's C API, oracle/Makefile refblock)
REF_BLOCK_SO = os.path.join(HERE, "_ref", "libmmref_block.so")
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

    def block_growth(self, call, cap=4096):
        """runs call() (block_backtrace / sw_block_backtrace_profile / block_align ...) with the block-list capture armed and returns
        (its result, int64 [n, 5] rows (i, j, height, width, right) of the last alignment run's block list - Trace::block_start /
        block_size / right when align_core returned)"""
        buf = np.zeros(4 * cap, np.uint32)
        self.L.mmo_block_growth_capture.argtypes = [c_p, ctypes.c_uint32]
        self.L.mmo_block_growth_capture.restype = None
        self.L.mmo_block_growth_count.restype = ctypes.c_uint32
        self.L.mmo_block_growth_capture(_ptr(buf), cap)
        try:
            res = call()
            n = int(self.L.mmo_block_growth_count())
        finally:
            self.L.mmo_block_growth_capture(None, 0)
        assert n <= cap, "block list longer than the capture buffer"
        b = buf[:4 * n].reshape(n, 4).astype(np.int64)
        return res, np.stack([b[:, 0], b[:, 1], b[:, 2] >> 16, b[:, 2] & 0xFFFF, b[:, 3]], axis=1)

    def block_backtrace(self, q, cb, t, mat, go, ge, score, q_end, t_end):
        """alignStartPosBacktraceBlock<SEQ_SEQ> (block_oracle.c: the restated block aligner).  -> dict(ok, q_start, t_start,
        ident, bt, block_size) ; ok False = "Block alignment failed" (the reference then falls back to its SW traceback)."""
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        mat = np.ascontiguousarray(mat, np.int8)
        cbp = None if cb is None else np.ascontiguousarray(cb, np.int8)
        cap = int(q_end) + int(t_end) + 16
        bt = ctypes.create_string_buffer(cap)
        qs, ts, bl, bs = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(0), ctypes.c_int(0)
        ident = ctypes.c_uint32(0)
        f = self.L.mmo_sw_block_backtrace
        f.argtypes = [c_p, c_p, ctypes.c_int, c_p, ctypes.c_int, c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                      ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint32),
                      ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        ok = f(_ptr(q), _ptr(cbp), len(q), _ptr(t), len(t), _ptr(mat), mat.shape[0], go, ge, int(score), int(q_end), int(t_end),
               ctypes.byref(qs), ctypes.byref(ts), ctypes.byref(ident), bt, cap, ctypes.byref(bl), ctypes.byref(bs))
        return dict(ok=bool(ok), q_start=qs.value, t_start=ts.value, ident=ident.value, bt=bt.raw[:bl.value].decode() if ok else "",
                    block_size=bs.value)

    def block_align(self, q, r, mat, gap_open, gap_extend, min_size, max_size, x_drop, qbias=None, rbias=None):
        """one Block<true, true>::align_aa call: -> (score, query_idx, reference_idx, ops origin -> end as a string of M / I / D)"""
        q = np.ascontiguousarray(q, np.uint8)
        r = np.ascontiguousarray(r, np.uint8)
        mat = np.ascontiguousarray(mat, np.int8)
        qb = None if qbias is None else np.ascontiguousarray(qbias, np.int16)
        rb = None if rbias is None else np.ascontiguousarray(rbias, np.int16)

        class Res(ctypes.Structure):
            _fields_ = [("score", ctypes.c_int32), ("qi", ctypes.c_uint32), ("ri", ctypes.c_uint32)]
        res = Res()
        ops = np.zeros(len(q) + len(r) + 8, np.uint8)
        n = ctypes.c_uint32()
        f = self.L.mmo_block_align
        f.argtypes = [c_p, c_p, ctypes.c_int, c_p, c_p, ctypes.c_int, c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                      ctypes.c_int, ctypes.POINTER(Res), c_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        rc = f(_ptr(q), _ptr(qb), len(q), _ptr(r), _ptr(rb), len(r), _ptr(mat), mat.shape[0], gap_open, gap_extend, min_size, max_size, x_drop,
               ctypes.byref(res), _ptr(ops), len(ops), ctypes.byref(n))
        if rc != 0:
            raise RuntimeError("mmo_block_align rc=%d" % rc)
        letters = {1: "M", 4: "I", 5: "D"}
        return res.score, res.qi, res.ri, "".join(letters[int(x)] for x in ops[:n.value][::-1])

    def block_align_profile(self, q, pos_aa_rows, gap_open_C, gap_close_C, gap_open_R, gap_extend, pad_block, min_size, max_size, x_drop,
                            trace, xdrop, upstream_gaps=False):
        """Block<trace, xdrop>::align_profile: q = query bytes (letter - 'A'), pos_aa_rows int8 [plen, 32], gap arrays int16 for the
        indices 0 .. len - 1 of the profile's arrays -> (score, query_idx, reference_idx, CIGAR in the crate's Display format or None)"""
        q = np.ascontiguousarray(q, np.uint8)
        rows = np.ascontiguousarray(pos_aa_rows, np.int8).reshape(-1, 32)
        goc, gcc, gor = (np.ascontiguousarray(x, np.int16) for x in (gap_open_C, gap_close_C, gap_open_R))
        assert len(goc) == len(gcc) == len(gor)

        class Res(ctypes.Structure):
            _fields_ = [("score", ctypes.c_int32), ("qi", ctypes.c_uint32), ("ri", ctypes.c_uint32)]
        res = Res()
        ops = np.zeros(len(q) + len(rows) + 8, np.uint8)
        n = ctypes.c_uint32()
        f = self.L.mmo_block_align_profile
        f.argtypes = [c_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_p, c_p, c_p, c_p] + [ctypes.c_int] * 7 + \
                     [ctypes.POINTER(Res), c_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        rc = f(_ptr(q), len(q), len(rows), int(pad_block), _ptr(rows), _ptr(goc), _ptr(gcc), _ptr(gor), len(goc), int(gap_extend), int(min_size),
               int(max_size), int(x_drop), int(bool(trace)) | (2 if upstream_gaps else 0), int(xdrop), ctypes.byref(res), _ptr(ops), len(ops),
               ctypes.byref(n))
        if rc != 0:
            raise RuntimeError("mmo_block_align_profile rc=%d" % rc)
        cigar = None
        if trace:
            cigar, run, last = "", 0, None
            for x in ops[:n.value][::-1]:
                c = {1: "M", 2: "=", 3: "X", 4: "I", 5: "D"}[int(x)]
                if c != last and last is not None:
                    cigar += "%d%s" % (run, last)
                    run = 0
                last = c
                run += 1
            if last is not None:
                cigar += "%d%s" % (run, last)
        return res.score, res.qi, res.ri, cigar

    def sw_block_backtrace_profile(self, prof, q, t, gap_open, gap_extend, score, q_end, t_end):
        """alignStartPosBacktraceBlock<PROFILE_SEQ>: prof int8 [alphabet, qlen], q = consensus -> None ("Block alignment failed") or
        dict(q_start, t_start, ident, bt, block_size)"""
        prof = np.ascontiguousarray(prof, np.int8)
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        qs, ts, bl, bs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        ident = ctypes.c_uint32()
        cap = len(q) + len(t) + 8
        bt = np.zeros(cap, np.uint8)
        f = self.L.mmo_sw_block_backtrace_profile
        f.argtypes = [c_p, c_p, ctypes.c_int, c_p, ctypes.c_int] + [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_int)] * 2 + \
                     [ctypes.POINTER(ctypes.c_uint32), c_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        ok = f(_ptr(prof), _ptr(q), len(q), _ptr(t), len(t), prof.shape[0], int(gap_open), int(gap_extend), int(score), int(q_end), int(t_end),
               ctypes.byref(qs), ctypes.byref(ts), ctypes.byref(ident), _ptr(bt), cap, ctypes.byref(bl), ctypes.byref(bs))
        if not ok:
            return None
        return dict(q_start=qs.value, t_start=ts.value, ident=ident.value, bt=bt[:bl.value].tobytes().decode(), block_size=bs.value)

    def block_align_generic(self, q, r, kind, table, gap_open, gap_extend, min_size, max_size, x_drop, trace, xdrop, eq=False):
        """Block<trace, xdrop>::align over an AAMatrix (kind 0) / NucMatrix (1) / ByteMatrix (2) table, q / r = bytes after
        Matrix::convert_char -> (score, query_idx, reference_idx, CIGAR in the crate's Display format or None without trace)"""
        q = np.ascontiguousarray(q, np.uint8)
        r = np.ascontiguousarray(r, np.uint8)
        table = np.ascontiguousarray(table, np.int8)

        class Res(ctypes.Structure):
            _fields_ = [("score", ctypes.c_int32), ("qi", ctypes.c_uint32), ("ri", ctypes.c_uint32)]
        res = Res()
        ops = np.zeros(len(q) + len(r) + 8, np.uint8)
        n = ctypes.c_uint32()
        f = self.L.mmo_block_align_generic
        f.argtypes = [c_p, ctypes.c_int, c_p, ctypes.c_int, ctypes.c_int, c_p] + [ctypes.c_int] * 8 + \
                     [ctypes.POINTER(Res), c_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        rc = f(_ptr(q), len(q), _ptr(r), len(r), int(kind), _ptr(table), gap_open, gap_extend, min_size, max_size, x_drop, int(trace),
               int(xdrop), int(eq), ctypes.byref(res), _ptr(ops), len(ops), ctypes.byref(n))
        if rc != 0:
            raise RuntimeError("mmo_block_align_generic rc=%d" % rc)
        cigar = None
        if trace:       # Display for Cigar (cigar.rs:139-155): run-length encoded, origin -> end
            cigar, run, last = "", 0, None
            for x in ops[:n.value][::-1]:
                c = {1: "M", 2: "=", 3: "X", 4: "I", 5: "D"}[int(x)]
                if c != last and last is not None:
                    cigar += "%d%s" % (run, last)
                    run = 0
                last = c
                run += 1
            if last is not None:
                cigar += "%d%s" % (run, last)
        return res.score, res.qi, res.ri, cigar

    def tantan_probs(self, seq, lr):
        """repeat probabilities of one sequence (tantan_oracle.c, the Masker's constants)"""
        seq = np.ascontiguousarray(seq, np.uint8)
        lr = np.ascontiguousarray(lr, np.float64)
        probs = np.zeros(max(len(seq), 1), np.float32)
        f = self.L.mmo_tantan_probs
        f.argtypes = [c_p, ctypes.c_int, c_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, c_p]
        f.restype = None
        f(_ptr(seq), len(seq), _ptr(lr), lr.shape[0], 0.005, 0.05, 0.9, 50, _ptr(probs))
        return probs[:len(seq)]

    def tantan_mask(self, tres, toff, lr, mask_prob=0.9, mask_letter=20):
        """every sequence masked by the restatement -> (masked copy, residues masked)"""
        res = np.ascontiguousarray(tres, np.uint8).copy()
        lr = np.ascontiguousarray(lr, np.float64)
        f = self.L.mmo_tantan_mask
        f.argtypes = [c_p, ctypes.c_int, c_p, ctypes.c_int, ctypes.c_double, ctypes.c_uint8, c_p]
        f.restype = ctypes.c_int
        masked = 0
        base = res.ctypes.data
        for i in range(len(toff) - 1):
            a, b = int(toff[i]), int(toff[i + 1])
            masked += f(c_p(base + a), b - a, _ptr(lr), lr.shape[0], float(mask_prob), mask_letter, None)
        return res, masked

    def block_prefix_scan(self, v, gap):
        v = np.ascontiguousarray(v, np.int16)
        out = np.zeros(16, np.int16)
        self.L.mmo_block_prefix_scan.argtypes = [c_p, ctypes.c_int, c_p]
        self.L.mmo_block_prefix_scan(_ptr(v), int(gap), _ptr(out))
        return out

    def sw_align_profile(self, profile, cons, t, alphabet, go, ge, need_start=False, need_bt=False):
        """profile: int8 [letters][qlen] (Sequence::getAlignmentProfile), cons: consensus sequence (numSequence)."""
        profile = np.ascontiguousarray(profile, np.int8)
        cons = np.ascontiguousarray(cons, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        r = SwRes()
        cap = len(cons) + len(t) + 8
        bt = ctypes.create_string_buffer(cap)
        rc = self.L.mmo_sw_align_profile(_ptr(profile), profile.shape[0], _ptr(cons), len(cons), _ptr(t), len(t), alphabet, go, ge,
                                         int(need_start), int(need_bt), ctypes.byref(r), bt, cap)
        if rc != 0:
            raise RuntimeError("mmo_sw_align_profile rc=%d" % rc)
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
                 gap_extend=1, comp_bias=True, db_residues=1000000, serialized=None, lib_path=None):
        """matrix: file name under /root/reference/data, or - where that tree is absent (GPU box) -
        `serialized` = the "name.out:DATA" bytes stored in tests/golden/matrices.npz."""
        L = self.L = ctypes.CDLL(lib_path or REF_SO)
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

    def sw_set_profile_query(self, entry):
        """entry: uint8/int8 [qlen][25] = one profile-database entry (Sequence::PROFILE_READIN_SIZE bytes per position).
        Returns (alignment profile int8 [20][qlen], consensus/query sequence uint8 [qlen]) as the reference derives them."""
        e = np.ascontiguousarray(entry).view(np.int8).reshape(-1, 25)
        n = e.shape[0]
        prof = np.zeros((20, n), np.int8)
        cons = np.zeros(n, np.uint8)
        self._pq = e
        self._q = cons
        self.L.mmref_sw_set_profile_query(self.c, _ptr(e), n, _ptr(prof), _ptr(cons))
        return prof, cons

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


def ref_sw_lists_omp(serialized, qres, qoff, lists_ids, lists_off, tres, toff, q_from, q_to, threads, mode=1,
                     evalue_thr=1e-3, db_residues=None, max_len=70000, gap_open=11, gap_extend=1, comp_bias=True,
                     want_starts=True):
    """Alignment::run's inner loop as ONE native OpenMP call of the real reference (oracle/ref_shim.cpp
    mmref_sw_lists_omp): queries [q_from, q_to) against their lists.  Returns (seconds inside the call's timed
    region, threads used, dict of result arrays indexed like lists_ids)."""
    L = ctypes.CDLL(REF_SO)
    f = L.mmref_sw_lists_omp
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_int,
                  ctypes.c_int, ctypes.c_double, c_p, c_p, ctypes.c_uint32, ctypes.c_uint32, c_p, c_p, c_p, c_p,
                  c_p, c_p, c_p, c_p, c_p, ctypes.POINTER(ctypes.c_int)]
    qres = np.ascontiguousarray(qres, np.uint8)
    qoff = np.ascontiguousarray(qoff, np.uint64)
    lists_ids = np.ascontiguousarray(lists_ids, np.uint32)
    lists_off = np.ascontiguousarray(lists_off, np.uint64)
    tres = np.ascontiguousarray(tres, np.uint8)
    toff = np.ascontiguousarray(toff, np.uint64)
    n = len(lists_ids)
    out = {"score": np.zeros(n, np.uint32), "q_end": np.full(n, -1, np.int32), "t_end": np.full(n, -1, np.int32),
           "q_start": np.full(n, -1, np.int32), "t_start": np.full(n, -1, np.int32)}
    used = ctypes.c_int(0)
    sec = f(bytes(serialized), max_len, gap_open, gap_extend, int(comp_bias),
            int(db_residues if db_residues is not None else toff[-1]), int(threads), int(mode), float(evalue_thr),
            _ptr(qres), _ptr(qoff), int(q_from), int(q_to), _ptr(lists_ids), _ptr(lists_off), _ptr(tres), _ptr(toff),
            _ptr(out["score"]), _ptr(out["q_end"]), _ptr(out["t_end"]),
            _ptr(out["q_start"]) if want_starts else None, _ptr(out["t_start"]) if want_starts else None,
            ctypes.byref(used))
    return sec, used.value, out


# ---------------------------------------------------------------------------------------------------------
# prefilter (prefilter_oracle.c / ref_shim_pref.cpp)
class PfGen(ctypes.Structure):
    _fields_ = [("k", ctypes.c_int), ("kalph", ctypes.c_int), ("s3", c_p), ("i3", c_p), ("s2", c_p), ("i2", c_p)]


class PfParams(ctypes.Structure):
    _fields_ = [("gen", ctypes.POINTER(PfGen)), ("alphabet", ctypes.c_int), ("spaced", ctypes.c_int),
                ("kmer_thr", ctypes.c_int), ("offsets", c_p), ("ids", c_p), ("pos", c_p), ("tdata", c_p),
                ("toff", c_p), ("n_targets", ctypes.c_uint32), ("ungapped_mat", c_p), ("bins", ctypes.c_uint32),
                ("max_hits", ctypes.c_uint64), ("min_diag_score", ctypes.c_uint32), ("exact_kmer", ctypes.c_int),
                ("nucleotide", ctypes.c_int), ("kmer_score", ctypes.c_int), ("index_base", ctypes.c_int)]


class PfStats(ctypes.Structure):
    _fields_ = [("db_matches", ctypes.c_uint64), ("kmer_list_len", ctypes.c_uint64),
                ("double_hits", ctypes.c_uint64), ("after_keepmax", ctypes.c_uint64),
                ("diag_thr", ctypes.c_uint32), ("truncated", ctypes.c_int), ("overflow", ctypes.c_int), ("big_list", ctypes.c_int),
                ("sat_tie", ctypes.c_int), ("sat_len", ctypes.c_int)]


class PfDump(ctypes.Structure):
    _fields_ = [("thr_out", c_p), ("nsim_out", c_p), ("arr_id", c_p), ("arr_diag", c_p), ("arr_cap", ctypes.c_uint64),
                ("dd_id", c_p), ("dd_diag", c_p), ("dd_count", c_p), ("dd_cap", ctypes.c_uint64)]


class PfHit(ctypes.Structure):
    _fields_ = [("id", ctypes.c_uint32), ("score", ctypes.c_int32), ("diagonal", ctypes.c_uint16)]


PF_HIT_DTYPE = np.dtype([("id", np.uint32), ("score", np.int32), ("diagonal", np.uint16), ("_pad", np.uint16)])


def kmer_threshold(sens, k):
    """Prefiltering::getKmerThreshold (Prefiltering.cpp:1080-1095), sequence queries."""
    if k == 5:
        return int(np.float32(160.75) - np.float32(sens) * np.float32(12.75))
    if k == 6:
        return int(163.2 - 8.917 * sens)
    if k == 7:
        return int(186.15 - 11.22 * sens)
    raise ValueError(k)


class PfOracle:
    """Plain-C restatement of the prefilter: score matrices, similar k-mers, index, matchQuery."""

    def __init__(self, kmer_mat16, ungapped_mat8, k=6, spaced=True):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = self.L = ctypes.CDLL(ORACLE_SO)
        L.mmo_pf_kmer_list.restype = ctypes.c_size_t
        L.mmo_pf_index_build.restype = ctypes.c_uint64
        self.kmer_mat16 = np.ascontiguousarray(kmer_mat16, np.int16)
        self.ungapped_mat = np.ascontiguousarray(ungapped_mat8, np.int8)
        self.alphabet = self.kmer_mat16.shape[0]
        self.kalph = self.alphabet - 1
        self.k = k
        self.spaced = int(spaced)
        n3, n2 = self.kalph ** 3, self.kalph ** 2
        self.s3 = np.zeros((n3, n3), np.int16)
        self.i3 = np.zeros((n3, n3), np.uint32)
        self.s2 = np.zeros((n2, n2), np.int16)
        self.i2 = np.zeros((n2, n2), np.uint32)
        L.mmo_pf_score_matrix(_ptr(self.kmer_mat16), self.alphabet, self.kalph, 3, _ptr(self.s3), _ptr(self.i3))
        L.mmo_pf_score_matrix(_ptr(self.kmer_mat16), self.alphabet, self.kalph, 2, _ptr(self.s2), _ptr(self.i2))
        self.gen = PfGen(k, self.kalph, self.s3.ctypes.data, self.i3.ctypes.data, self.s2.ctypes.data,
                         self.i2.ctypes.data)

    def kmer_list(self, kmer, thr, cap=1 << 20):
        kmer = np.ascontiguousarray(kmer, np.uint8)
        out = np.zeros(cap, np.uint64)
        n = self.L.mmo_pf_kmer_list(ctypes.byref(self.gen), _ptr(kmer), int(thr), _ptr(out), ctypes.c_size_t(cap))
        return out[:min(n, cap)].copy(), n

    def build_index(self, tdata, toff, kmer_thr):
        self.tdata = np.ascontiguousarray(tdata, np.uint8)
        self.toff = np.ascontiguousarray(toff, np.uint64)
        n = len(self.toff) - 1
        table = self.kalph ** self.k
        self.offsets = np.zeros(table + 1, np.uint64)
        args = (_ptr(self.tdata), _ptr(self.toff), n, _ptr(self.kmer_mat16), self.alphabet, self.k, self.spaced,
                int(kmer_thr), _ptr(self.offsets))
        total = self.L.mmo_pf_index_build(*args, None, None)
        self.ids = np.zeros(max(total, 1), np.uint32)
        self.pos = np.zeros(max(total, 1), np.uint16)
        self.L.mmo_pf_index_build(*args, _ptr(self.ids), _ptr(self.pos))
        self.n_entries = int(total)
        self.n_targets = n
        self.kmer_thr = int(kmer_thr)
        return self.offsets, self.ids[:total], self.pos[:total]

    def ungapped_corr(self, bias, qlen):
        out = np.zeros(qlen, np.int8)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        self.L.mmo_pf_ungapped_corr(_ptr(b), qlen, _ptr(out))
        return out

    def ungapped_score(self, q, corr, t, diagonal):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        corr = np.ascontiguousarray(corr, np.int8)
        return self.L.mmo_pf_ungapped_score(_ptr(q), _ptr(corr), len(q), _ptr(self.ungapped_mat), self.alphabet,
                                            _ptr(t), len(t), ctypes.c_uint16(int(diagonal) & 0xFFFF))

    def match(self, q, comp_bias, bins, max_hits=300, min_diag_score=15, identity_id=None, dump=False, exact=False, nucleotide=False,
              kmer_score=False):
        q = np.ascontiguousarray(q, np.uint8)
        cb = None if comp_bias is None else np.ascontiguousarray(comp_bias, np.float32)
        P = PfParams(ctypes.pointer(self.gen), self.alphabet, self.spaced, self.kmer_thr, self.offsets.ctypes.data,
                     self.ids.ctypes.data, self.pos.ctypes.data, self.tdata.ctypes.data, self.toff.ctypes.data,
                     self.n_targets, self.ungapped_mat.ctypes.data, bins, max_hits, min_diag_score, int(exact), int(nucleotide),
                     int(kmer_score))
        cap = int(min(max_hits, self.n_targets)) + 2
        hits = np.zeros(cap, PF_HIT_DTYPE)
        nh = ctypes.c_uint64(0)
        st = PfStats()
        D = None
        keep = {}
        if dump:
            arr_cap = 2 * max(1000000, self.n_targets)
            keep = dict(thr=np.zeros(len(q), np.int32), nsim=np.zeros(len(q), np.uint32),
                        arr_id=np.zeros(arr_cap, np.uint32), arr_diag=np.zeros(arr_cap, np.uint16),
                        dd_id=np.zeros(arr_cap // 2, np.uint32), dd_diag=np.zeros(arr_cap // 2, np.uint16),
                        dd_count=np.zeros(arr_cap // 2, np.uint8))
            D = PfDump(keep["thr"].ctypes.data, keep["nsim"].ctypes.data, keep["arr_id"].ctypes.data,
                       keep["arr_diag"].ctypes.data, arr_cap, keep["dd_id"].ctypes.data, keep["dd_diag"].ctypes.data,
                       keep["dd_count"].ctypes.data, arr_cap // 2)
        ident = 0xFFFFFFFF if identity_id is None else int(identity_id)
        rc = self.L.mmo_pf_match_query(ctypes.byref(P), _ptr(q), len(q), _ptr(cb), ctypes.c_uint32(ident), _ptr(hits),
                                       ctypes.c_uint64(cap), ctypes.byref(nh), ctypes.byref(st),
                                       ctypes.byref(D) if D is not None else None)
        stats = {f: getattr(st, f) for f, _ in PfStats._fields_}
        stats["rc"] = rc
        res = hits[: nh.value]
        out = dict(id=res["id"].copy(), score=res["score"].copy(), diagonal=res["diagonal"].copy(), stats=stats)
        if dump:
            out["thr"] = keep["thr"]
            out["nsim"] = keep["nsim"]
            n = stats["db_matches"]
            out["arr_id"] = keep["arr_id"][:n].copy()
            out["arr_diag"] = keep["arr_diag"][:n].copy()
            d = stats["double_hits"]
            out["dd_id"] = keep["dd_id"][:d].copy()
            out["dd_diag"] = keep["dd_diag"][:d].copy()
            out["dd_count"] = keep["dd_count"][:d].copy()
        return out


class PfProfile(ctypes.Structure):
    _fields_ = [("score", c_p), ("index", c_p), ("row", ctypes.c_int), ("aln", c_p)]


def _pf_match_profile(self, letters, pscore, pindex, aln, bins, kmer_thr, max_hits=300, min_diag_score=15, identity_id=None):
    """Profile query: letters = Sequence::numSequence, pscore / pindex = Sequence::profile_score / profile_index
    ([qlen][row], rows sorted descending), aln = Sequence::getAlignmentProfile() [20][qlen]."""
    q = np.ascontiguousarray(letters, np.uint8)
    pscore = np.ascontiguousarray(pscore, np.int16)
    pindex = np.ascontiguousarray(pindex, np.uint32)
    aln = np.ascontiguousarray(aln, np.int8)
    assert pscore.shape == pindex.shape and pscore.shape[0] == len(q) and aln.shape == (20, len(q))
    prof = PfProfile(pscore.ctypes.data, pindex.ctypes.data, pscore.shape[1], aln.ctypes.data)
    # (the index of a profile search is built with threshold 0, the matcher still runs with the profile k-mer threshold)
    P = PfParams(ctypes.pointer(self.gen), self.alphabet, self.spaced, int(kmer_thr), self.offsets.ctypes.data,
                 self.ids.ctypes.data, self.pos.ctypes.data, self.tdata.ctypes.data, self.toff.ctypes.data,
                 self.n_targets, self.ungapped_mat.ctypes.data, bins, max_hits, min_diag_score, 0, 0)
    cap = int(min(max_hits, self.n_targets)) + 2
    hits = np.zeros(cap, PF_HIT_DTYPE)
    nh = ctypes.c_uint64(0)
    st = PfStats()
    ident = 0xFFFFFFFF if identity_id is None else int(identity_id)
    rc = self.L.mmo_pf_match_query_profile(ctypes.byref(P), _ptr(q), len(q), ctypes.byref(prof), ctypes.c_uint32(ident), _ptr(hits),
                                           ctypes.c_uint64(cap), ctypes.byref(nh), ctypes.byref(st), None)
    stats = {f: getattr(st, f) for f, _ in PfStats._fields_}
    stats["rc"] = rc
    res = hits[: nh.value]
    return dict(id=res["id"].copy(), score=res["score"].copy(), diagonal=res["diagonal"].copy(), stats=stats)


PfOracle.match_profile = _pf_match_profile


class RefPrefilter:
    """The real reference prefilter classes (needs /root/reference/data at run time)."""

    def __init__(self, k=6, kmer_matrix="VTML80.out", ungapped_matrix="blosum62.out", serialized=None):
        """serialized = (kmer "name.out:DATA" bytes, ungapped bytes) from tests/golden/matrices.npz where
        /root/reference/data is absent (GPU box)."""
        L = self.L = ctypes.CDLL(REF_SO)
        L.mmref_pref_new.restype = c_p
        L.mmref_pref_new.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        for f in ("mmref_pref_index_entries", "mmref_pref_index_table_size", "mmref_pref_kmer_list",
                  "mmref_pref_match", "mmref_pref_score_matrix"):
            getattr(L, f).restype = ctypes.c_uint64
        L.mmref_pref_make_matcher.restype = ctypes.c_uint
        L.mmref_pref_match_batch.restype = ctypes.c_double
        d = os.path.join(REFERENCE_ROOT, "data")
        if serialized is not None:
            a, b = bytes(serialized[0]), bytes(serialized[1])
        else:
            a, b = os.path.join(d, kmer_matrix).encode(), os.path.join(d, ungapped_matrix).encode()
        self.c = c_p(L.mmref_pref_new(a, b, k))
        self.k = k
        self.alphabet = L.mmref_pref_alphabet(self.c)

    def matrices(self):
        a = self.alphabet
        km = np.zeros((a, a), np.int8)
        um = np.zeros((a, a), np.int8)
        km16 = np.zeros((a, a), np.int16)
        pb = np.zeros(a, np.float64)
        self.L.mmref_pref_get_matrices(self.c, _ptr(km), _ptr(um), _ptr(km16), _ptr(pb))
        return km, um, km16, pb

    def tantan_mask(self, tres, toff, mask_prob=0.9):
        """the reference's own tantan masking of numeric targets (Masker::maskSequence, maskTantan only) -> (masked copy,
        residues masked, likelihood-ratio table [alphabet, alphabet], repeat probabilities of sequence 0)"""
        res = np.ascontiguousarray(tres, np.uint8).copy()
        off = np.ascontiguousarray(toff, np.uint64)
        a = self.alphabet
        lr = np.zeros((a, a), np.float64)
        n = len(off) - 1
        probs = np.zeros(int(off[1] - off[0]) if n else 1, np.float32)
        self.L.mmref_tantan_mask.restype = ctypes.c_uint64
        self.L.mmref_tantan_mask.argtypes = [c_p, c_p, c_p, ctypes.c_uint32, ctypes.c_double, c_p, c_p]
        masked = self.L.mmref_tantan_mask(self.c, _ptr(res), _ptr(off), n, float(mask_prob), _ptr(lr), _ptr(probs))
        return res, int(masked), lr, probs

    def score_matrix(self, which):
        rs = ctypes.c_uint64(0)
        n = self.L.mmref_pref_score_matrix(self.c, which, ctypes.byref(rs), None, None)
        s = np.zeros((n, n), np.int16)
        i = np.zeros((n, n), np.uint32)
        self.L.mmref_pref_score_matrix(self.c, which, ctypes.byref(rs), _ptr(s), _ptr(i))
        return s, i

    def build_index(self, tdata, toff, kmer_thr, spaced=True):
        self._tdata = np.ascontiguousarray(tdata, np.uint8)
        self._toff = np.ascontiguousarray(toff, np.uint64)
        self.L.mmref_pref_build_index(self.c, _ptr(self._tdata), _ptr(self._toff), len(self._toff) - 1, int(kmer_thr),
                                      int(spaced))
        self.kmer_thr = int(kmer_thr)
        self.spaced = spaced

    def index_dump(self):
        ne = self.L.mmref_pref_index_entries(self.c)
        ts = self.L.mmref_pref_index_table_size(self.c)
        off = np.zeros(ts + 1, np.uint64)
        ids = np.zeros(max(ne, 1), np.uint32)
        pos = np.zeros(max(ne, 1), np.uint16)
        self.L.mmref_pref_index_dump(self.c, _ptr(off), _ptr(ids), _ptr(pos))
        return off, ids[:ne], pos[:ne]

    def kmer_list(self, kmer, thr, cap=1 << 20):
        kmer = np.ascontiguousarray(kmer, np.uint8)
        out = np.zeros(cap, np.uint64)
        n = self.L.mmref_pref_kmer_list(self.c, _ptr(kmer), int(thr), _ptr(out), ctypes.c_uint64(cap))
        return out[:min(n, cap)].copy(), n

    def make_matcher(self, max_seq_len=32000, max_hits=300, comp_bias=True, comp_bias_scale=1.0, min_diag_score=15,
                     force_bins=0, diag_score=True):
        self.max_hits = max_hits
        return self.L.mmref_pref_make_matcher(self.c, self.kmer_thr, max_seq_len, ctypes.c_uint64(max_hits),
                                              int(comp_bias), ctypes.c_float(comp_bias_scale), int(diag_score), min_diag_score,
                                              int(self.spaced), int(force_bins))

    def match(self, q, identity_id=None):
        q = np.ascontiguousarray(q, np.uint8)
        cap = self.max_hits + 2
        ids = np.zeros(cap, np.uint32)
        sc = np.zeros(cap, np.int32)
        dg = np.zeros(cap, np.uint16)
        dbm = ctypes.c_uint64(0)
        kpp = ctypes.c_double(0)
        ident = 0xFFFFFFFF if identity_id is None else int(identity_id)
        n = self.L.mmref_pref_match(self.c, _ptr(q), len(q), ctypes.c_uint32(ident), _ptr(ids), _ptr(sc), _ptr(dg),
                                    ctypes.c_uint64(cap), ctypes.byref(dbm), ctypes.byref(kpp))
        return dict(id=ids[:n].copy(), score=sc[:n].copy(), diagonal=dg[:n].copy(), db_matches=dbm.value,
                    kmers_per_pos=kpp.value)

    def match_profile(self, entry, kmer_thr, max_hits=300, min_diag_score=15, max_seq_len=32000, force_bins=0, identity_id=None):
        """entry: [qlen][25] profile-database entry.  Returns the hit list and what the reference derived from the entry
        (sorted score rows, their letters, alignment profile, query letters)."""
        e = np.ascontiguousarray(entry).view(np.int8).reshape(-1, 25)
        n = e.shape[0]
        cap = max_hits + 2
        ids = np.zeros(cap, np.uint32)
        sc = np.zeros(cap, np.int32)
        dg = np.zeros(cap, np.uint16)
        dbm = ctypes.c_uint64(0)
        row = ctypes.c_uint32(0)
        ps = np.zeros((n, 64), np.int16)
        pi = np.zeros((n, 64), np.uint32)
        aln = np.zeros((20, n), np.int8)
        letters = np.zeros(n, np.uint8)
        ident = 0xFFFFFFFF if identity_id is None else int(identity_id)
        # first call only asks for the row size (the arrays above are sized for up to 64)
        self.L.mmref_pref_match_profile.restype = ctypes.c_uint64
        k = self.L.mmref_pref_match_profile(self.c, _ptr(e), n, int(kmer_thr), max_seq_len, ctypes.c_uint64(max_hits),
                                            min_diag_score, int(self.spaced), int(force_bins), ctypes.c_uint32(ident),
                                            _ptr(ids), _ptr(sc), _ptr(dg), ctypes.c_uint64(cap), ctypes.byref(dbm),
                                            None, None, ctypes.byref(row), _ptr(aln), _ptr(letters))
        r = int(row.value)
        ps = np.zeros((n, r), np.int16)
        pi = np.zeros((n, r), np.uint32)
        self.L.mmref_pref_match_profile(self.c, _ptr(e), n, int(kmer_thr), max_seq_len, ctypes.c_uint64(max_hits),
                                        min_diag_score, int(self.spaced), int(force_bins), ctypes.c_uint32(ident),
                                        _ptr(ids), _ptr(sc), _ptr(dg), ctypes.c_uint64(cap), ctypes.byref(dbm),
                                        _ptr(ps), _ptr(pi), ctypes.byref(row), _ptr(aln), _ptr(letters))
        return dict(id=ids[:k].copy(), score=sc[:k].copy(), diagonal=dg[:k].copy(), db_matches=dbm.value,
                    pscore=ps, pindex=pi, aln=aln, letters=letters)

    def match_batch(self, qres, qoff, n_threads, max_hits=300, comp_bias=True, min_diag_score=15, max_seq_len=32000,
                    want_lists=False):
        """Multi-threaded query loop (CPU baseline). Returns (seconds, total hits, total db matches, hit counts[,
        lists dict with ids/scores/diags [nq][max_hits] and the CacheFriendlyOperations bin count used])."""
        qres = np.ascontiguousarray(qres, np.uint8)
        qoff = np.ascontiguousarray(qoff, np.uint64)
        nq = len(qoff) - 1
        th = ctypes.c_uint64(0)
        dbm = ctypes.c_uint64(0)
        counts = np.zeros(max(nq, 1), np.uint32)
        ids = sc = dg = None
        if want_lists:
            ids = np.zeros((max(nq, 1), max_hits), np.uint32)
            sc = np.zeros((max(nq, 1), max_hits), np.int32)
            dg = np.zeros((max(nq, 1), max_hits), np.uint16)
        bins = ctypes.c_uint(0)
        sec = self.L.mmref_pref_match_batch(self.c, _ptr(qres), _ptr(qoff), nq, int(n_threads), self.kmer_thr,
                                            int(max_seq_len), ctypes.c_uint64(max_hits), int(comp_bias),
                                            int(min_diag_score), int(self.spaced), ctypes.byref(th), ctypes.byref(dbm),
                                            _ptr(counts), _ptr(ids), _ptr(sc), _ptr(dg), ctypes.byref(bins))
        if want_lists:
            return sec, th.value, dbm.value, counts[:nq], dict(ids=ids[:nq], scores=sc[:nq], diags=dg[:nq], bins=bins.value)
        return sec, th.value, dbm.value, counts[:nq]


class RefNuclPrefilter:
    """The real reference classes driven as a nucleotide prefilter (exact k-mers, isNucleotide branch of matchQuery)."""

    def __init__(self, k=15, spaced=True, serialized=None):
        L = self.L = ctypes.CDLL(REF_SO)
        L.mmref_npref_new.restype = c_p
        L.mmref_npref_new.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.mmref_npref_match.restype = ctypes.c_uint64
        L.mmref_npref_index_entries.restype = ctypes.c_uint64
        L.mmref_npref_index_entries.argtypes = [c_p]
        path = bytes(serialized) if serialized is not None else os.path.join(REFERENCE_ROOT, "data", "nucleotide.out").encode()
        self.c = c_p(L.mmref_npref_new(path, int(k)))
        self.k = k
        self.spaced = bool(spaced)

    def matrix(self):
        out = np.zeros((5, 5), np.int8)
        self.L.mmref_npref_matrix(self.c, _ptr(out))
        return out

    def build_index(self, tdata, toff):
        self._t = (np.ascontiguousarray(tdata, np.uint8), np.ascontiguousarray(toff, np.uint64))
        self.L.mmref_npref_build_index(self.c, _ptr(self._t[0]), _ptr(self._t[1]), len(self._t[1]) - 1, int(self.spaced))

    def index_dump(self):
        n = int(self.L.mmref_npref_index_entries(self.c))
        off = np.zeros(4 ** self.k + 1, np.uint64)
        ids = np.zeros(max(n, 1), np.uint32)
        pos = np.zeros(max(n, 1), np.uint16)
        self.L.mmref_npref_index_dump(self.c, _ptr(off), _ptr(ids), _ptr(pos))
        return off, ids[:n], pos[:n]

    def match(self, q, max_hits=300, min_diag_score=15, max_seq_len=32000, force_bins=0, identity_id=None):
        q = np.ascontiguousarray(q, np.uint8)
        cap = max_hits + 2
        ids = np.zeros(cap, np.uint32)
        sc = np.zeros(cap, np.int32)
        dg = np.zeros(cap, np.uint16)
        dbm = ctypes.c_uint64(0)
        ident = 0xFFFFFFFF if identity_id is None else int(identity_id)
        n = self.L.mmref_npref_match(self.c, _ptr(q), len(q), int(max_seq_len), ctypes.c_uint64(max_hits), int(min_diag_score),
                                     int(self.spaced), int(force_bins), ctypes.c_uint32(ident), _ptr(ids), _ptr(sc), _ptr(dg),
                                     ctypes.c_uint64(cap), ctypes.byref(dbm))
        return dict(id=ids[:n].copy(), score=sc[:n].copy(), diagonal=dg[:n].copy(), db_matches=dbm.value)


# ---------------------------------------------------------------------------------------------------------
# nucleotide alignment step: oracle/nucl_oracle.c and the real reference classes
class KswEz(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "n_cigar")]


class NuclRes(ctypes.Structure):
    _fields_ = [("score", ctypes.c_int32), ("q_start", ctypes.c_int32), ("q_end", ctypes.c_int32), ("t_start", ctypes.c_int32),
                ("t_end", ctypes.c_int32), ("ident", ctypes.c_uint32), ("bt_len", ctypes.c_int32), ("cigar_len", ctypes.c_int32)]

    def as_tuple(self):
        return (self.score, self.q_start, self.q_end, self.t_start, self.t_end, self.ident, self.cigar_len)


NUCL_LETTERS = "ACTGN"          # numeric codes 0..4 of NucleotideMatrix (nucleotide.out column order, N -> X)
KSW_SCORE_ONLY, KSW_EXTZ_ONLY = 0x01, 0x40


class NuclOracle:
    """Plain-C restatement (oracle/nucl_oracle.c)."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        self.L = ctypes.CDLL(ORACLE_SO)

    def ksw_extz2(self, q, t, mat, gapo, gape, w, zdrop, flag):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        mat = np.ascontiguousarray(mat, np.int8)
        ez = KswEz()
        cap = len(q) + len(t) + 4
        cg = np.zeros(cap, np.uint32)
        n = self.L.mmo_ksw_extz2(len(q), _ptr(q), len(t), _ptr(t), 5, _ptr(mat), gapo, gape, w, zdrop, flag, ctypes.byref(ez),
                                 _ptr(cg), cap)
        return [getattr(ez, f) for f in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score")], cg[:max(n, 0)].copy()

    def align(self, q, t, mat, rev_lookup, gapo, gape, zdrop, diagonal, reverse, past_end_q=4, past_end_t=4, wrapped=False):
        """wrapped: --wrapped-scoring, q is the query written twice"""
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        mat = np.ascontiguousarray(mat, np.int8)
        rl = np.ascontiguousarray(rev_lookup, np.uint8)
        res = NuclRes()
        cap = len(q) + len(t) + 8
        bt = ctypes.create_string_buffer(cap)
        rc = self.L.mmo_nucl_align_wrapped(_ptr(q), len(q), _ptr(t), len(t), _ptr(mat), 5, _ptr(rl), gapo, gape, zdrop,
                                           ctypes.c_uint(diagonal & 0xFFFF), int(reverse), int(past_end_q), int(past_end_t), int(bool(wrapped)),
                                           ctypes.byref(res), bt, cap)
        assert rc == 0
        return res.as_tuple(), bt.value.decode()


class RefNucl:
    """The real BandedNucleotideAligner / ksw_extz2_sse (oracle/ref_shim_nucl.cpp); needs /root/reference/data."""

    def __init__(self, max_len=70000, gap_open=5, gap_extend=2, zdrop=40, db_residues=100000000, serialized=None):
        """serialized: the "nucleotide.out:DATA" bytes of tests/golden/matrices.npz where /root/reference is absent"""
        L = self.L = ctypes.CDLL(REF_SO)
        L.mmref_nucl_new.restype = c_p
        L.mmref_nucl_new.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
        L.mmref_nucl_free.argtypes = [c_p]
        L.mmref_nucl_matrix.argtypes = [c_p, c_p]
        L.mmref_nucl_aa2num.argtypes = [c_p, ctypes.c_char_p, ctypes.c_int, c_p]
        L.mmref_nucl_reverse_lookup.argtypes = [c_p, c_p]
        L.mmref_nucl_set_query.argtypes = [c_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
        L.mmref_nucl_align.argtypes = [c_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(NuclRes), ctypes.c_char_p, ctypes.c_int]
        L.mmref_ksw_extz2.argtypes = [ctypes.c_int, c_p, ctypes.c_int, c_p, ctypes.c_int, c_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, c_p, c_p, ctypes.c_int]
        self.path = bytes(serialized) if serialized is not None else os.path.join(REFERENCE_ROOT, "data", "nucleotide.out").encode()
        self.args = (max_len, gap_open, gap_extend, zdrop)
        self.ctx = L.mmref_nucl_new(self.path, max_len, gap_open, gap_extend, zdrop, db_residues)
        self._q = None

    def serialized_matrix(self):
        b = ctypes.create_string_buffer(1 << 16)
        self.L.mmref_nucl_serialized_matrix.argtypes = [c_p, ctypes.c_char_p, ctypes.c_int]
        n = self.L.mmref_nucl_serialized_matrix(self.ctx, b, 1 << 16)
        assert n > 0
        return b.raw[:n]

    def batch(self, q_chars, q_off, t_chars, t_off, pair_q, pair_t, pair_diag, pair_rev, n_threads, past_end=4):
        """BandedNucleotideAligner::align for every pair on n_threads host threads -> (seconds, [n, 6] int32 results
        (score, q_start, q_end, t_start, t_end, ident), backtrace lengths)."""
        L = self.L
        L.mmref_nucl_batch.restype = ctypes.c_double
        L.mmref_nucl_batch.argtypes = [ctypes.c_char_p] + [ctypes.c_int] * 6 + [c_p] * 8 + [ctypes.c_uint32, c_p, c_p]
        q_off = np.ascontiguousarray(q_off, np.uint64)
        t_off = np.ascontiguousarray(t_off, np.uint64)
        pq = np.ascontiguousarray(pair_q, np.uint32)
        pt = np.ascontiguousarray(pair_t, np.uint32)
        pd = np.ascontiguousarray(pair_diag, np.uint16)
        pr = np.ascontiguousarray(pair_rev, np.uint8)
        out = np.zeros((len(pq), 6), np.int32)
        bl = np.zeros(len(pq), np.uint32)
        qc = np.ascontiguousarray(q_chars, np.uint8)
        tc = np.ascontiguousarray(t_chars, np.uint8)
        sec = L.mmref_nucl_batch(self.path, self.args[0], self.args[1], self.args[2], self.args[3], int(past_end), int(n_threads),
                                 _ptr(qc), _ptr(q_off), _ptr(tc), _ptr(t_off), _ptr(pq), _ptr(pt), _ptr(pd), _ptr(pr), len(pq),
                                 _ptr(out), _ptr(bl))
        return sec, out, bl

    def matrix(self):
        out = np.zeros((5, 5), np.int8)
        self.L.mmref_nucl_matrix(self.ctx, _ptr(out))
        return out

    def reverse_lookup(self):
        out = np.zeros(5, np.uint8)
        self.L.mmref_nucl_reverse_lookup(self.ctx, _ptr(out))
        return out

    def aa2num(self, s):
        out = np.zeros(len(s), np.uint8)
        self.L.mmref_nucl_aa2num(self.ctx, s.encode(), len(s), _ptr(out))
        return out

    def set_query(self, s, past_end=4):
        self._q = s.encode()           # the reference keeps the pointer
        self.L.mmref_nucl_set_query(self.ctx, self._q, len(s), int(past_end))

    def align(self, tseq, diagonal, reverse, past_end=4, wrapped=False):
        res = NuclRes()
        cap = len(self._q) + len(tseq) + 8
        bt = ctypes.create_string_buffer(cap)
        tb = tseq.encode()
        self.L.mmref_nucl_align(self.ctx, tb, len(tseq), int(past_end), int(diagonal), int(reverse), int(bool(wrapped)), ctypes.byref(res), bt, cap)
        return res.as_tuple(), bt.value.decode()

    def ksw_extz2(self, q, t, mat, gapo, gape, w, zdrop, flag):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        mat = np.ascontiguousarray(mat, np.int8)
        out = np.zeros(9, np.int32)
        cap = len(q) + len(t) + 4
        cg = np.zeros(cap, np.uint32)
        n = self.L.mmref_ksw_extz2(len(q), _ptr(q), len(t), _ptr(t), 5, _ptr(mat), gapo, gape, w, zdrop, flag, _ptr(out), _ptr(cg), cap)
        return out.tolist(), cg[:n].copy()

    def close(self):
        if self.ctx:
            self.L.mmref_nucl_free(self.ctx)
            self.ctx = None
