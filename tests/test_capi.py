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
