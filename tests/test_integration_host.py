"""CPU check of the host side of the alignment seam (integration/MMGpuMatcher.cpp, SURVEY.md section 8 rows a16 / b): the
E-value and coverage gates of ssw_align_private, the start-score threshold handed to the device, sequence identity, bit
score and the result_t record must equal what the real Matcher::initQuery + Matcher::getSWResult produce, pair by pair,
for every alignment mode.  The device is replaced by a backend built on the reference's own SmithWaterman
(oracle/ref_shim_matcher.cpp); needs oracle/_ref and /root/reference/data (skipped elsewhere)."""
import ctypes
import os

import numpy as np
import pytest

from mmseqs2_amd import workloads as wl
from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not (po.ref_available() and po.ref_matrix_available()),
                                reason="needs oracle/_ref/libmmref.so and /root/reference/data")


def _check(qs, ts, lists, identity, cov_mode, cov_thr, eval_thr, aln_mode, seqid_mode=0, comp_bias=1):
    L = ctypes.CDLL(po.REF_SO)
    c_p = ctypes.c_void_p
    L.mmref_matcher_check.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, c_p, c_p,
                                      ctypes.c_uint32, c_p, c_p, ctypes.c_uint32, c_p, c_p, c_p, ctypes.c_int, ctypes.c_float,
                                      ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p,
                                      ctypes.c_int]
    qres, qoff = wl.seqs_from_list(qs)
    tres, toff = wl.seqs_from_list(ts)
    loff = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint32)
    lids = np.concatenate(lists).astype(np.uint32)
    lid = np.concatenate(identity).astype(np.uint8)
    n = ctypes.c_int()
    msg = ctypes.create_string_buffer(1024)
    path = os.path.join(po.REFERENCE_ROOT, "data", "blosum62.out").encode()
    bad = L.mmref_matcher_check(path, 11, 1, comp_bias, 300000000, qres.ctypes.data, qoff.ctypes.data, len(qs), tres.ctypes.data,
                                toff.ctypes.data, len(ts), loff.ctypes.data, lids.ctypes.data, lid.ctypes.data, cov_mode,
                                ctypes.c_float(cov_thr), ctypes.c_double(eval_thr), aln_mode, seqid_mode, ctypes.byref(n), msg, 1024)
    return bad, n.value, msg.value.decode()


def _workload(seed, nq=6, nt=60):
    rng = np.random.default_rng(seed)
    qs, ts, lists, ident = [], [], [], []
    for _ in range(nq):
        L = int(rng.integers(20, 400))
        qs.append(rng.choice(20, size=L, p=wl.BACKGROUND).astype(np.uint8))
    for k in range(nt):
        if k % 3 == 0:
            ts.append(wl.mutate(rng, qs[k % nq], float(rng.uniform(0.3, 0.95))))
        else:
            ts.append(rng.choice(21, size=int(rng.integers(5, 500)), p=np.append(wl.BACKGROUND * 0.98, 0.02)).astype(np.uint8))
    for q in range(nq):
        ids = rng.permutation(nt)[:40].astype(np.uint32)
        lists.append(ids)
        ident.append(np.zeros(40, np.uint8))
    return qs, ts, lists, ident


@pytest.mark.parametrize("aln_mode", [0, 1, 2])
@pytest.mark.parametrize("cov", [(0, 0.0), (0, 0.8), (2, 0.5)])
@pytest.mark.parametrize("eval_thr", [1e-3, 10.0])
def test_host_matcher_equals_reference_matcher(aln_mode, cov, eval_thr):
    qs, ts, lists, ident = _workload(11 + aln_mode)
    bad, n, msg = _check(qs, ts, lists, ident, cov[0], cov[1], eval_thr, aln_mode)
    assert n == 240 and bad == 0, msg


def test_identity_hits_and_seqid_modes():
    """self hits (Matcher's scoreIdentical path) and the three sequence-identity definitions"""
    qs, ts, lists, ident = _workload(5)
    ts = ts + [q.copy() for q in qs]                 # target nt + q is query q itself
    for q in range(len(qs)):
        lists[q] = np.append(lists[q], np.uint32(60 + q))
        ident[q] = np.append(ident[q], np.uint8(1))
    for seqid_mode in (0, 1, 2):
        for aln_mode in (0, 2):
            bad, n, msg = _check(qs, ts, lists, ident, 0, 0.0, 1e-3, aln_mode, seqid_mode)
            assert n == 246 and bad == 0, (seqid_mode, aln_mode, msg)


def test_without_composition_bias():
    qs, ts, lists, ident = _workload(8)
    bad, n, msg = _check(qs, ts, lists, ident, 0, 0.0, 1e-3, 2, comp_bias=0)
    assert bad == 0, msg


def test_device_backend_and_prefilter_host_side_compile_against_both_header_sets():
    """integration/MMGpuDeviceBackend.cpp (the C-ABI calls behind MMGpuMatcher) and integration/MMGpuPrefilter.cpp (IndexTable /
    SequenceLookup / ScoreMatrix -> mmgpu_pf_index, QueryMatcher::matchQuery's batch form) are compiled by oracle/Makefile with
    the reference's headers and include/mmgpu.h: the reference-side binding of INTEGRATION.md is real code that type-checks."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "ref"])
    for o in ("MMGpuDeviceBackend.o", "MMGpuPrefilter.o", "MMGpuMatcher.o"):
        assert os.path.getsize(os.path.join(root, "oracle", "_build", o)) > 1000
