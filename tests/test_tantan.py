"""tantan repeat masking (the masking step of IndexBuilder::fillDatabase, IndexBuilder.cpp:148 ->
Masker.cpp:14-57 -> lib/tantan/tantan.cpp) - round 4: on the device (mmgpu_pf_mask_targets, tantan_kernel.hip).

The repeat probabilities are doubles, the decision compares a float with a threshold: results depend on the order and the rounding
of every operation, and the reference's AVX2 build is compiled with -mfma, so gcc fuses some multiply-adds.  What is pinned:
  * oracle/tantan_oracle.c (plain C, fused operations written out) == the reference's compiled tantan, probabilities bit for
    bit and masks byte for byte: against vectors recorded from oracle/_ref/libmmref.so (tests/golden/make_tantan_golden.py) and,
    where that library and /root/reference/data are present, against the library itself on fresh sequences;
  * the device == the recorded vectors and == the restatement on a larger set (gpu tests);
  * `mmseqs prefilter` with default masking through the patched binary == stock (tests/test_mmseqs_dropin.py)."""
import os

import numpy as np
import pytest

from mmseqs2_amd import workloads as wl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vec():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "tantan_vectors.npz")))


def test_restatement_equals_the_recorded_reference_probabilities_and_masks(oracle, vec):
    tres, toff, lr = vec["tres"], vec["toff"], vec["vtml80_likelihood_ratios"]
    n_bits = 0
    for i in range(len(toff) - 1):
        a, b = int(toff[i]), int(toff[i + 1])
        if b == a:
            continue
        p = oracle.tantan_probs(tres[a:b], lr)
        assert np.array_equal(p.view(np.uint32), vec["probs"][a:b].view(np.uint32)), i
        n_bits += b - a
    masked, n = oracle.tantan_mask(tres, toff, lr, float(vec["mask_prob"]))
    assert n == int(vec["n_masked"]) and n > 1000 and np.array_equal(masked, vec["masked"])
    assert n_bits == len(tres)


def test_restatement_equals_the_compiled_reference_on_fresh_sequences(oracle):
    from oracle import pyoracle
    if not (pyoracle.ref_available() and pyoracle.ref_matrix_available()):
        pytest.skip("real reference (oracle/_ref + /root/reference/data) not available here")
    ref = pyoracle.RefPrefilter()
    rng = np.random.default_rng(11)
    (_, _), (tres, toff), _, _ = wl.config3_prefilter(200, 5, 5, seed=21)
    ts = wl.split(tres, toff)
    for i in range(0, len(ts), 2):      # periodic and low-complexity inserts of many shapes
        t = ts[i].copy()
        unit = rng.integers(0, 20, size=rng.integers(1, 30)).astype(np.uint8)
        a = int(rng.integers(0, max(1, len(t) - 10)))
        n = min(len(t) - a, int(rng.integers(5, 200)))
        t[a:a + n] = np.resize(unit, n)
        ts[i] = t
    tres, toff = wl.seqs_from_list(ts)
    mp = float(np.float32(0.9))
    want, n_want, lr, _ = ref.tantan_mask(tres, toff, mp)
    got, n_got = oracle.tantan_mask(tres, toff, lr, mp)
    assert n_got == n_want and np.array_equal(got, want)


@pytest.mark.gpu
def test_device_masking_equals_the_recorded_reference(gpu, vec):
    gpu.load_targets(vec["tres"], vec["toff"], 21)
    n = gpu.pf_mask_targets(vec["vtml80_likelihood_ratios"], float(vec["mask_prob"]), 20)
    got = gpu.pf_debug_masked_targets(vec["toff"])
    assert n == int(vec["n_masked"])
    assert np.array_equal(got, vec["masked"])


@pytest.mark.gpu
@pytest.mark.parametrize("scratch_mb", [None, "1"])
def test_device_masking_equals_the_restatement_on_a_database(gpu, oracle, vec, scratch_mb, monkeypatch):
    """20 000 family-structured targets with planted repeats (lengths 30 .. 5000: wavefronts of sequences of about equal length,
    the partial first 50 positions, the rescaling every 16 letters, sequences shorter than one SIMD group of states).  The second
    case bounds the scratch (forward probabilities + scale factors) at 1 MB: the wavefronts then run in some thirty chunks, the
    longest ones alone."""
    if scratch_mb is not None:
        monkeypatch.setenv("MMGPU_TANTAN_SCRATCH_MB", scratch_mb)
    rng = np.random.default_rng(5)
    (_, _), (tres, toff), _, _ = wl.config3_prefilter(400, 50, 10, seed=31)
    ts = wl.split(tres, toff)
    for i in range(0, len(ts), 4):
        t = ts[i].copy()
        unit = rng.integers(0, 20, size=rng.integers(1, 12)).astype(np.uint8)
        a = int(rng.integers(0, max(1, len(t) - 10)))
        n = min(len(t) - a, int(rng.integers(10, 150)))
        t[a:a + n] = np.resize(unit, n)
        ts[i] = t
    tres, toff = wl.seqs_from_list(ts)
    lr, mp = vec["vtml80_likelihood_ratios"], float(vec["mask_prob"])
    gpu.load_targets(tres, toff, 21)
    n = gpu.pf_mask_targets(lr, mp, 20)
    got = gpu.pf_debug_masked_targets(toff)
    want, n_want = oracle.tantan_mask(tres, toff, lr, mp)
    assert n == n_want and n_want > 10000
    assert np.array_equal(got, want)
