"""CPU test of the GPU kernel's own source (mmseqs2_amd/csrc/nucl_core.h) for the nucleotide alignment step: the
kernel body is compiled for the host with 16 cooperative contexts standing in for the 16 lanes of an alignment group
(tests/nucl_emu.cpp) and run against the vectors recorded from the real reference.  It does not replace the GPU parity
test (tests/test_nucl_gpu.py); it catches logic errors of the lane-parallel formulation where no GPU is at hand."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from mmseqs2_amd import capi
from tests import nucl_common as nc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def _lib(lanes=16, wave=False):
    """the kernel source compiled for the host with `lanes` emulated lanes per alignment: 16 = the LDS formulation of
    nucl_core.h, 64 + wave = the default GPU kernel (one wavefront per alignment, state in registers: nucl_wave.h)"""
    so = os.path.join(ROOT, "tests", "_build", "libnuclemu%d%s.so" % (lanes, "w" if wave else ""))
    src = os.path.join(ROOT, "tests", "nucl_emu.cpp")
    deps = [src] + [os.path.join(ROOT, "mmseqs2_amd", "csrc", f) for f in ("nucl_core.h", "nucl_wave.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-DEMU_LANES=%d" % lanes] + (["-DEMU_WAVE"] if wave else []) +
                              ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "mmseqs2_amd", "csrc"), src, "-o", so])
    return ctypes.CDLL(so)


def emu_align(L, mat, reverse, queries, targets, pairs, past_q, past_t, gapo=5, gape=2, zdrop=40, wrapped=False):
    c_p = ctypes.c_void_p
    mat = np.ascontiguousarray(mat, np.int8).reshape(-1)
    rev = np.ascontiguousarray(reverse, np.uint8)
    qs = [np.ascontiguousarray(q, np.uint8) for q in queries]
    arr = (capi.NuclQuery * len(qs))()
    for i, q in enumerate(qs):
        arr[i] = capi.NuclQuery(q.ctypes.data_as(c_p), len(q))
    tres = np.ascontiguousarray(np.concatenate(targets), np.uint8)
    toff = np.concatenate([[0], np.cumsum([len(t) for t in targets])]).astype(np.uint64)
    pa = np.zeros(len(pairs), capi.NUCL_PAIR_DTYPE)
    for i, p in enumerate(pairs):
        pa[i] = (p[0], p[1], p[2] & 0xFFFF, p[3], p[4] if len(p) > 4 else 0)
    par = capi.NuclParams(mat.ctypes.data_as(c_p), rev.ctypes.data_as(c_p), gapo, gape, zdrop, past_q, past_t, int(bool(wrapped)))
    out = np.zeros(len(pairs), capi.NUCL_HIT_DTYPE)
    cap = int(sum(len(qs[p[0]]) + len(targets[p[1]]) + 2 for p in pairs))
    bt = np.zeros(cap, np.uint8)
    used = ctypes.c_uint64()
    L.nucl_emu_align(ctypes.byref(par), ctypes.cast(arr, c_p), len(qs), tres.ctypes.data_as(c_p), toff.ctypes.data_as(c_p),
                     len(targets), pa.ctypes.data_as(c_p), len(pairs), out.ctypes.data_as(c_p), bt.ctypes.data_as(c_p),
                     ctypes.c_uint64(cap), ctypes.byref(used))
    raw = bt.tobytes()
    return out, [raw[int(h["bt_off"]):int(h["bt_off"]) + int(h["bt_len"])].decode() for h in out]


@pytest.mark.parametrize("lanes,wave", [(16, False), (64, True)])
def test_kernel_source_on_emulated_lanes_matches_golden(lanes, wave):
    L = _lib(lanes, wave)
    g = nc.golden()
    cases = [c for c in nc.golden_cases(g) if len(c[0]) + len(c[1]) <= 1800]
    assert len(cases) > 250
    n = 0
    for pq in range(5):
        for pt in range(5):
            sub = [c for c in cases if c[4] == pq and c[5] == pt]
            if not sub:
                continue
            hits, strs = emu_align(L, g["mat"], g["reverse"], [c[0] for c in sub], [c[1] for c in sub],
                                   [(i, i, c[2], c[3]) for i, c in enumerate(sub)], pq, pt)
            for c, h, s in zip(sub, hits, strs):
                got = (int(h["score"]), int(h["q_start"]), int(h["q_end"]), int(h["t_start"]), int(h["t_end"]), int(h["ident"]))
                assert got == c[6][:6] and s == c[7], (len(c[0]), len(c[1]), c[2], c[3], got, c[6], s[:40], c[7][:40])
                n += 1
    assert n == len(cases)


def test_past_end_letters_per_pair_equal_the_per_call_ones():
    """mmgpu_nucl_pair::past_end (MMGPU_NUCL_PAST_END): every golden case in ONE call, each pair carrying the letters the
    reference found past the ends when the vector was recorded; the call's own letters are set to something else."""
    L = _lib(64, True)
    g = nc.golden()
    cases = [c for c in nc.golden_cases(g) if len(c[0]) + len(c[1]) <= 1800][:120]
    pairs = [(i, i, c[2], c[3], 0x80 | (c[4] & 7) | ((c[5] & 7) << 3)) for i, c in enumerate(cases)]
    hits, strs = emu_align(L, g["mat"], g["reverse"], [c[0] for c in cases], [c[1] for c in cases], pairs, 2, 1)
    assert len({(c[4], c[5]) for c in cases}) > 3
    for c, h, s in zip(cases, hits, strs):
        got = (int(h["score"]), int(h["q_start"]), int(h["q_end"]), int(h["t_start"]), int(h["t_end"]), int(h["ident"]))
        assert got == c[6][:6] and s == c[7], (c[4], c[5], got, c[6])


@pytest.mark.parametrize("lanes,wave", [(16, False), (64, True)])
def test_wrapped_scoring_on_emulated_lanes_equals_the_oracle(lanes, wave):
    """--wrapped-scoring: doubled queries, the wrap-around seed where the doubled query is at least twice the target, extensions
    capped at the original length - the kernel source against the restatement (which tests/test_nucl_oracle.py pins against
    the reference's BandedNucleotideAligner::align(..., wrappedScoring = true))."""
    from oracle import pyoracle as po
    L = _lib(lanes, wave)
    g = nc.golden()
    mat, rl = np.asarray(g["mat"]).reshape(5, 5), np.asarray(g["reverse"])
    orc = po.NuclOracle()
    rng = np.random.default_rng(31)
    queries, targets, pairs, expected = [], [], [], []
    for it in range(10):
        n = int(rng.choice([40, 64, 130, 300, 500]))
        circle = rng.integers(0, 4, size=n).astype(np.uint8)
        rot = int(rng.integers(0, n))
        q1 = nc.mutate(rng, np.roll(circle, -rot), rng.choice([0.0, 0.04]), rng.choice([0.0, 0.02]))
        queries.append(np.concatenate([q1, q1]))
        for t in (nc.mutate(rng, circle, 0.03, 0.01), circle[: max(8, n // 3)].copy(), np.concatenate([circle, circle[: n // 2]])):
            for rev in (0, 1):
                tt = np.array([rl[x] for x in t[::-1]], np.uint8) if rev else t
                targets.append(tt)
                for diag in (0, rot & 0xFFFF, (-rot) & 0xFFFF, int(rng.integers(0, 65536))):
                    pq, pt = int(rng.integers(0, 5)), int(rng.integers(0, 5))
                    pairs.append((len(queries) - 1, len(targets) - 1, diag, rev, 0x80 | pq | (pt << 3)))
                    expected.append(orc.align(queries[-1], tt, mat.reshape(-1), rl, 5, 2, 40, diag, rev, pq, pt, wrapped=True))
    hits, strs = emu_align(L, mat, rl, queries, targets, pairs, 4, 4, wrapped=True)
    for k, (e, h, s) in enumerate(zip(expected, hits, strs)):
        got = (int(h["score"]), int(h["q_start"]), int(h["q_end"]), int(h["t_start"]), int(h["t_end"]), int(h["ident"]))
        assert got == e[0][:6] and s == e[1], (k, pairs[k], got, e[0])
