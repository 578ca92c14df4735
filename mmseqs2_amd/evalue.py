"""E-value of a raw Smith-Waterman score as the reference's host code computes it for its default scoring
(EvalueComputation::computeEvalue, src/alignment/EvalueComputation.h:37-41, BLOSUM62 with gap open 11 / extend 1):
evaluePerArea(score) * area(score, queryLength, dbResidues) of the ALP library (lib/alp/sls_alignment_evaluer.hpp:154-157,
lib/alp/sls_pvalues.cpp:366-510 - Gumbel parameters with finite-size correction; the parameter set is the one the
reference ships for this matrix, EvalueComputation.h:69-74, so ALP's own estimation never runs).

In a drop-in build the host's EvalueComputation supplies these numbers (integration/MMGpuMatcher.cpp); this module exists
for callers without the reference (bench.py, tests), which need the same `min_start_score` the host would hand to the
device: the smallest raw score whose E-value passes -e (ssw_align_private's gate, StripedSmithWaterman.cpp:857-863).
The normal distribution function is math.erfc here and a tabulated approximation in ALP (sls_basic.cpp:107-184); the
thresholds agree (tests/test_evalue.py compares them with the real reference over a grid)."""
import math

# lambda, K, a_I, b_I, a_J, b_J, alpha_I, beta_I, alpha_J, beta_J, sigma, tau   ("blosum62.out", 11, 1, gapped)
BLOSUM62_11_1 = (0.27359865037097330642, 0.044620920658722244834,
                 1.5938724404943873658, -19.959867650284412122, 1.5938724404943873658, -19.959867650284412122,
                 30.455610143099914211, -622.28684628915891608, 30.455610143099914211, -622.28684628915891608,
                 29.602444874818868215, -601.81087985041381216)
_NAT_CUT_OFF_IN_MAX = 2.0          # sls_pvalues.cpp:46
_CONST = 1.0 / math.sqrt(2.0 * math.pi)


def _phi(x):
    return 0.5 * math.erfc(-x / math.sqrt(2.0))


def area(score, qlen, db_residues, par=BLOSUM62_11_1):
    lam, _K, a_i, b_i, a_j, b_j, al_i, be_i, al_j, be_j, sigma, tau = par
    vi_thr = max(_NAT_CUT_OFF_IN_MAX * al_i / lam, 0.0)
    vj_thr = max(_NAT_CUT_OFF_IN_MAX * al_j / lam, 0.0)
    c_thr = max(_NAT_CUT_OFF_IN_MAX * sigma / lam, 0.0)
    m, n, y = float(db_residues), float(qlen), float(score)
    m_li = m - (a_i * y + b_i)
    sv = math.sqrt(max(vi_thr, al_i * y + be_i))
    m_f = 1e100 if sv == 0.0 else m_li / sv
    p_m = _phi(m_f)
    p1 = m_li * p_m + sv * _CONST * math.exp(-0.5 * m_f * m_f)
    n_lj = n - (a_j * y + b_j)
    sw = math.sqrt(max(vj_thr, al_j * y + be_j))
    n_f = 1e100 if sw == 0.0 else n_lj / sw
    p_n = _phi(n_f)
    p2 = n_lj * p_n + sw * _CONST * math.exp(-0.5 * n_f * n_f)
    c_y = max(c_thr, sigma * y + tau)
    return p1 * p2 + c_y * p_m * p_n


def evalue(score, qlen, db_residues, par=BLOSUM62_11_1):
    return par[1] * math.exp(-par[0] * float(score)) * area(score, qlen, db_residues, par)


def bit_score(score, par=BLOSUM62_11_1):
    return (par[0] * float(score) - math.log(par[1])) / math.log(2.0)


def min_score_for_evalue(evalue_thr, qlen, db_residues, par=BLOSUM62_11_1):
    """smallest raw score in [1, 32767] whose E-value is <= evalue_thr, 32768 if none (bisection, like
    MMGpuMatcher::minScoreForEvalue in integration/MMGpuMatcher.cpp)"""
    if evalue(32767, qlen, db_residues, par) > evalue_thr:
        return 32768
    lo, hi = 1, 32767
    while lo < hi:
        mid = (lo + hi) // 2
        if evalue(mid, qlen, db_residues, par) > evalue_thr:
            lo = mid + 1
        else:
            hi = mid
    return lo


def min_scores_for_evalue(evalue_thr, qlens, db_residues, par=BLOSUM62_11_1):
    """min_score_for_evalue for an array of query lengths at once (numpy; the same bisection on the same expression,
    scipy's erfc in place of math.erfc - both are the C library's): int32[len(qlens)]."""
    import numpy as np
    from scipy.special import erfc
    lam, K, a_i, b_i, a_j, b_j, al_i, be_i, al_j, be_j, sigma, tau = par
    vi_thr = max(_NAT_CUT_OFF_IN_MAX * al_i / lam, 0.0)
    vj_thr = max(_NAT_CUT_OFF_IN_MAX * al_j / lam, 0.0)
    c_thr = max(_NAT_CUT_OFF_IN_MAX * sigma / lam, 0.0)
    uniq, inv = np.unique(np.asarray(qlens, np.int64), return_inverse=True)
    n = uniq.astype(np.float64)
    m = float(db_residues)

    def ev(y):
        y = y.astype(np.float64)
        m_li = m - (a_i * y + b_i)
        sv = np.sqrt(np.maximum(vi_thr, al_i * y + be_i))
        m_f = np.where(sv == 0.0, 1e100, m_li / np.where(sv == 0.0, 1.0, sv))
        p_m = 0.5 * erfc(-m_f / math.sqrt(2.0))
        p1 = m_li * p_m + sv * _CONST * np.exp(-0.5 * m_f * m_f)
        n_lj = n - (a_j * y + b_j)
        sw = np.sqrt(np.maximum(vj_thr, al_j * y + be_j))
        n_f = np.where(sw == 0.0, 1e100, n_lj / np.where(sw == 0.0, 1.0, sw))
        p_n = 0.5 * erfc(-n_f / math.sqrt(2.0))
        p2 = n_lj * p_n + sw * _CONST * np.exp(-0.5 * n_f * n_f)
        c_y = np.maximum(c_thr, sigma * y + tau)
        return K * np.exp(-lam * y) * (p1 * p2 + c_y * p_m * p_n)

    lo = np.ones(len(uniq), np.int64)
    hi = np.full(len(uniq), 32767, np.int64)
    none = ev(hi) > evalue_thr
    while np.any(lo < hi):
        mid = (lo + hi) // 2
        worse = ev(mid) > evalue_thr
        act = lo < hi
        lo = np.where(act & worse, mid + 1, lo)
        hi = np.where(act & ~worse, mid, hi)
    out = np.where(none, 32768, lo).astype(np.int32)
    return out[inv]
