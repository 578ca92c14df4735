"""GPU parity tests of the prefilter (through the C-ABI): every stage of the device pipeline and the final hit_t
lists against the oracle, and the final lists against the golden vectors recorded from the real reference."""
import numpy as np
import pytest

from mmseqs2_amd import workloads as wl
from tests import pf_common as pc
from tests import pf_gpu_check as chk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden_case(gpu):
    g = pc.golden()
    orc = pc.pf_oracle()
    orc.build_index(g["tres"], g["toff"], int(g["kmer_thr"]))
    chk.load_case(gpu, g, g["tres"], g["toff"], int(g["kmer_thr"]))
    return g, orc


def test_prefilter_stages_match_oracle(gpu, golden_case):
    g, orc = golden_case
    ok, rep = chk.check(gpu, orc, pc.golden_queries(g), 300, 2, stages=True, label="golden/stages")
    assert ok, "\n".join(rep)


def test_prefilter_hits_match_golden(gpu, golden_case):
    """Final hit lists against the lists the REAL reference produced (tests/golden/prefilter_vectors.npz)."""
    g, orc = golden_case
    qs = pc.golden_queries(g)
    for si, (mh, bins) in enumerate(g["settings"].tolist()):
        hits, counts, status, stats = gpu.pf_batch(qs, int(g["kmer_thr"]), max_hits=mh, min_diag_score=15, ref_bins=bins)
        exp = pc.expected_hits(g, si)
        for qi in range(len(qs)):
            n = int(counts[qi])
            assert status[qi] == 0
            assert int(stats[qi]["db_matches"]) == int(g["db_matches"][qi])
            assert np.array_equal(hits[qi]["id"][:n], exp[qi][0]), (si, qi)
            assert np.array_equal(hits[qi]["score"][:n], exp[qi][1]), (si, qi)
            assert np.array_equal(hits[qi]["diagonal"][:n], exp[qi][2]), (si, qi)


def test_prefilter_larger_db_and_bins(gpu):
    """20 000 targets (5 device bins), ragged query lengths, several max_hits / reference-bin settings."""
    g = pc.golden()
    (qres, qoff), (tres, toff) = pc.synthetic_case(48, 20000, seed=5, planted=0.1)
    orc = pc.pf_oracle()
    thr = int(g["kmer_thr"])
    orc.build_index(tres, toff, thr)
    chk.load_case(gpu, g, tres, toff, thr)
    from oracle.pyoracle import Oracle
    swo = Oracle()
    qs = []
    for i, q in enumerate(wl.split(qres, qoff)):
        if i % 7 == 3:
            q = q[: 30 + i]
        qs.append(dict(q=q, comp_bias=swo.comp_bias(g["vtml80_kmer16"], g["vtml80_pback"], q),
                       identity_id=(i * 131) % 20000 if i % 4 == 0 else None))
    for mh, rb, st in ((300, 2, True), (40, 2, False), (40, 64, False), (10, 8, False)):
        ok, rep = chk.check(gpu, orc, qs, mh, rb, stages=st, label="20k/%d/%d" % (mh, rb))
        assert ok, "\n".join(rep)
    # restore the golden case for tests that run after this one in the same module
    orc.build_index(g["tres"], g["toff"], thr)
    chk.load_case(gpu, g, g["tres"], g["toff"], thr)


def test_prefilter_empty_and_tiny(gpu, golden_case):
    g, orc = golden_case
    hits, counts, status, stats = gpu.pf_batch([], int(g["kmer_thr"]))
    assert len(counts) == 0
    q = np.array([0, 1, 2], np.uint8)
    hits, counts, status, stats = gpu.pf_batch([dict(q=q, comp_bias=None, identity_id=7)], int(g["kmer_thr"]))
    assert counts[0] == 1 and hits[0]["id"][0] == 7 and hits[0]["score"][0] == 65535
