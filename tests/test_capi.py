"""CPU tests of the boundary: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/mmgpu.h declares (no compute without a GPU)."""
import os
import re

import pytest

import mmseqs2_amd
from mmseqs2_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    mmseqs2_amd.build_library()
    L = capi.load_library()
    hdr = open(os.path.join(ROOT, "include", "mmgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mmgpu_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(capi.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(L, sym), sym


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(mmseqs2_amd.MMGpuError):
        mmseqs2_amd.MMGpu(0)


def test_product_does_not_import_oracle():
    """The product path (mmseqs2_amd/, include/) must never reach into oracle/."""
    bad = []
    for base in ("mmseqs2_amd", "include"):
        for dp, _dn, fn in os.walk(os.path.join(ROOT, base)):
            for f in fn:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                    txt = open(os.path.join(dp, f)).read()
                    if re.search(r"(import\s+oracle|from\s+oracle|mm_oracle\.h|libmmoracle|libmmref|pyoracle)", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_host_comp_bias_matches_oracle(oracle, matrices):
    """Host helper of the product (runs without a GPU) against the oracle restatement, bit for bit."""
    import numpy as np
    rng = np.random.default_rng(2)
    sub = matrices["blosum62_sw"].astype(np.int16)
    pb = matrices["blosum62_pback"]
    for L in [1, 19, 20, 21, 41, 350, 3000]:
        s = rng.integers(0, 21, L).astype(np.uint8)
        f, r = capi.host_comp_bias(sub, pb, s, 1.0)
        fo = oracle.comp_bias(sub, pb, s, 1.0)
        assert np.array_equal(f.view(np.uint32), fo.view(np.uint32))
        assert np.array_equal(r, oracle.round_comp_bias(fo))


def test_host_comp_bias_batch_equals_per_sequence_calls(matrices):
    """mmgpu_host_comp_bias_batch (threads) = mmgpu_host_comp_bias + mmgpu_host_round_comp_bias per sequence, incl. empty
    sequences and a single thread."""
    import numpy as np
    rng = np.random.default_rng(5)
    sub = matrices["vtml80_kmer"].astype(np.int16)
    pb = matrices["vtml80_pback"]
    lens = [0, 1, 19, 40, 41, 350, 0, 1200] + [int(x) for x in rng.integers(1, 400, 300)]
    res = rng.integers(0, 21, sum(lens)).astype(np.uint8)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for threads in (1, 3, 16):
        f, r = capi.host_comp_bias_batch(sub, pb, res, off, threads=threads)
        for i in range(len(lens)):
            a, b = int(off[i]), int(off[i + 1])
            f1, r1 = capi.host_comp_bias(sub, pb, res[a:b])
            assert np.array_equal(f[a:b].view(np.uint32), f1.view(np.uint32)) and np.array_equal(r[a:b], r1)
    bad = res.copy()
    bad[5] = 77
    with pytest.raises(capi.MMGpuError):
        capi.host_comp_bias_batch(sub, pb, bad, off)


def test_flat_query_descriptors_match_the_struct_layout():
    import ctypes
    import numpy as np
    for dt, st in ((capi.SW_QUERY_DTYPE, capi.SwQuery), (capi.PF_QUERY_DTYPE, capi.PfQuery)):
        assert dt.itemsize == ctypes.sizeof(st)
        for name, _ in st._fields_:
            assert dt.fields[name][1] == getattr(st, name).offset, name
