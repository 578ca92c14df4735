"""GPU parity tests of the nucleotide alignment step (mmgpu_nucl_align) through the C-ABI:
against the vectors recorded from the real reference and against the oracle on seeded random reads."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import nucl_common as nc

pytestmark = pytest.mark.gpu


def _run(gpu, g, cases, past_q, past_t):
    """cases sharing one (past_q, past_t): one device call; returns (hits, strings)"""
    tres = np.concatenate([c[1] for c in cases])
    toff = np.concatenate([[0], np.cumsum([len(c[1]) for c in cases])]).astype(np.uint64)
    gpu.load_targets(tres, toff, 5)
    pairs = [(i, i, c[2], c[3]) for i, c in enumerate(cases)]
    return gpu.nucl_align(g["mat"], g["reverse"], [c[0] for c in cases], pairs, 5, 2, 40, past_q, past_t)


def test_golden_vectors(gpu):
    g = nc.golden()
    cases = nc.golden_cases(g)
    n = 0
    for pq in range(5):
        for pt in range(5):
            sub = [c for c in cases if c[4] == pq and c[5] == pt]
            if not sub:
                continue
            hits, strs = _run(gpu, g, sub, pq, pt)
            for c, h, s in zip(sub, hits, strs):
                got = (int(h["score"]), int(h["q_start"]), int(h["q_end"]), int(h["t_start"]), int(h["t_end"]), int(h["ident"]))
                assert h["status"] == 0 and got == c[6][:6] and s == c[7], (len(c[0]), len(c[1]), c[2], c[3], got, c[6])
                n += 1
    assert n == len(cases)


def test_past_end_letters_per_pair(gpu):
    """mmgpu_nucl_pair::past_end (MMGPU_NUCL_PAST_END): all golden cases in ONE call, every pair carrying the letters the
    reference found past the ends when the vector was recorded (the call's own letters are set to something else)"""
    g = nc.golden()
    cases = nc.golden_cases(g)
    tres = np.concatenate([c[1] for c in cases])
    toff = np.concatenate([[0], np.cumsum([len(c[1]) for c in cases])]).astype(np.uint64)
    gpu.load_targets(tres, toff, 5)
    pairs = [(i, i, c[2], c[3], 0x80 | (c[4] & 7) | ((c[5] & 7) << 3)) for i, c in enumerate(cases)]
    hits, strs = gpu.nucl_align(g["mat"], g["reverse"], [c[0] for c in cases], pairs, 5, 2, 40, 2, 1)
    for c, h, s in zip(cases, hits, strs):
        got = (int(h["score"]), int(h["q_start"]), int(h["q_end"]), int(h["t_start"]), int(h["t_end"]), int(h["ident"]))
        assert h["status"] == 0 and got == c[6][:6] and s == c[7], (c[4], c[5], got, c[6])


def test_random_reads_vs_oracle(gpu):
    """config-5-like reads (10 % substitutions, 2 % indels) up to 10 kb against their source contig window, both
    strands, plus unrelated pairs; many queries against one database in a single call"""
    g = nc.golden()
    mat, rl = g["mat"], g["reverse"]
    rng = np.random.default_rng(77)
    orc = po.NuclOracle()
    queries, targets, pairs = [], [], []
    for L in [40, 300, 1200, 5000, 10000, 10000]:
        contig = rng.integers(0, 4, size=L + 400).astype(np.uint8)
        start = int(rng.integers(0, 400))
        read = nc.mutate(rng, contig[start:start + L], 0.10, 0.02)
        qi = len(queries)
        queries.append(read)
        targets.append(contig)
        pairs.append((qi, len(targets) - 1, (-start) & 0xFFFF, 0))                 # true diagonal (i - j = -start)
        pairs.append((qi, len(targets) - 1, int(rng.integers(0, 65536)), 0))        # a wrong one
        rc = np.array([rl[x] for x in read[::-1]], np.uint8)
        queries.append(rc)
        pairs.append((qi + 1, len(targets) - 1, (-start) & 0xFFFF, 1))             # reverse strand restores the read
        other = rng.integers(0, 4, size=int(rng.integers(30, 900))).astype(np.uint8)
        targets.append(other)
        pairs.append((qi, len(targets) - 1, 0, 0))                                 # unrelated
    tres = np.concatenate(targets)
    toff = np.concatenate([[0], np.cumsum([len(t) for t in targets])]).astype(np.uint64)
    gpu.load_targets(tres, toff, 5)
    hits, strs = gpu.nucl_align(mat, rl, queries, pairs, 5, 2, 40, 4, 4)
    longest = 0
    for (qi, ti, diag, rev), h, s in zip(pairs, hits, strs):
        exp, bt = orc.align(queries[qi], targets[ti], mat.reshape(-1), rl, 5, 2, 40, diag, rev, 4, 4)
        got = (int(h["score"]), int(h["q_start"]), int(h["q_end"]), int(h["t_start"]), int(h["t_end"]), int(h["ident"]))
        assert got == exp[:6] and s == bt, (qi, ti, diag, rev, got, exp)
        longest = max(longest, len(bt))
    assert longest > 5000


def test_wrapped_scoring_vs_oracle(gpu):
    """--wrapped-scoring (mmgpu_nucl_params::wrapped): doubled queries of circular sequences read from another origin, against the
    circle, a piece of it and the circle with a repeated part, both strands, true / shifted / arbitrary diagonals - the
    restatement is pinned against the reference's BandedNucleotideAligner::align(..., true) in tests/test_nucl_oracle.py"""
    g = nc.golden()
    mat, rl = g["mat"], g["reverse"]
    rng = np.random.default_rng(41)
    orc = po.NuclOracle()
    queries, targets, pairs = [], [], []
    for n in [40, 64, 130, 400, 900, 3000, 9000]:
        for _ in range(3):
            circle = rng.integers(0, 4, size=n).astype(np.uint8)
            rot = int(rng.integers(0, n))
            q1 = nc.mutate(rng, np.roll(circle, -rot), rng.choice([0.0, 0.04]), rng.choice([0.0, 0.02]))
            queries.append(np.concatenate([q1, q1]))
            for t in (nc.mutate(rng, circle, 0.03, 0.01), circle[: max(8, n // 3)].copy(), np.concatenate([circle, circle[: n // 2]])):
                for rev in (0, 1):
                    targets.append(np.array([rl[x] for x in t[::-1]], np.uint8) if rev else t)
                    for diag in (0, rot & 0xFFFF, (-rot) & 0xFFFF, int(rng.integers(0, 65536))):
                        pairs.append((len(queries) - 1, len(targets) - 1, diag, rev))
    tres = np.concatenate(targets)
    toff = np.concatenate([[0], np.cumsum([len(t) for t in targets])]).astype(np.uint64)
    gpu.load_targets(tres, toff, 5)
    hits, strs = gpu.nucl_align(mat, rl, queries, pairs, 5, 2, 40, 4, 4, wrapped=True)
    wrapped_seed = 0
    for (qi, ti, diag, rev), h, s in zip(pairs, hits, strs):
        exp, bt = orc.align(queries[qi], targets[ti], mat.reshape(-1), rl, 5, 2, 40, diag, rev, 4, 4, wrapped=True)
        got = (int(h["score"]), int(h["q_start"]), int(h["q_end"]), int(h["t_start"]), int(h["t_end"]), int(h["ident"]))
        assert h["status"] == 0 and got == exp[:6] and s == bt, (qi, ti, diag, rev, got, exp)
        wrapped_seed += len(queries[qi]) >= 2 * len(targets[ti])
    assert len(pairs) == 7 * 3 * 3 * 2 * 4 and wrapped_seed > 100


def test_errors_and_empty(gpu):
    from mmseqs2_amd import capi
    g = nc.golden()
    t = np.array([0, 1, 2, 3], np.uint8)
    gpu.load_targets(t, np.array([0, 4], np.uint64), 5)
    hits, strs = gpu.nucl_align(g["mat"], g["reverse"], [t], [], 5, 2, 40)
    assert len(hits) == 0
    with pytest.raises(capi.MMGpuError):
        gpu.nucl_align(g["mat"], g["reverse"], [t], [(0, 7, 0, 0)])            # target id out of range
    with pytest.raises(capi.MMGpuError):
        gpu.nucl_align(g["mat"], g["reverse"], [np.array([9], np.uint8)], [(0, 0, 0, 0)])   # residue code > 4
    # identical sequences: the seed spans both, one run of M
    hits, strs = gpu.nucl_align(g["mat"], g["reverse"], [t], [(0, 0, 0, 0)])
    assert strs[0] == "MMMM" and int(hits[0]["score"]) == 8 and int(hits[0]["ident"]) == 4
