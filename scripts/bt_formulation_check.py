"""Check (CPU, against the oracle) of the row-parallel formulation of banded_sw used by the wave traceback kernel:
 * previous-row H / E by COLUMN with explicit rules instead of the reference's band-frame arrays and their zeroed slots
   (`zero_last`: what row i sees at its last column is 0 when i <= w + 1 or the band is not clipped by the target end);
 * F from H-without-F (Hnf = max(e1, diag)): f[j] = max(Hnf[j-1] - go, f[j-1] - ge), same value and same tie flag;
so that every cell of a row depends on the previous row only, plus a max-plus prefix scan along the row."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle
from mmseqs2_amd import workloads as wl


def banded_rows(q, cb, t, mat, go, ge, score):
    ql, tl = len(q), len(t)
    bw = abs(tl - ql) + 1
    while True:
        Hprev, Eprev = {}, {}
        dirs = []
        mx = 0
        for i in range(ql):
            beg, end = max(0, i - bw), min(tl - 1, i + bw)
            pbeg, pend = max(0, i - 1 - bw), min(tl - 1, i - 1 + bw)
            zero_last = i <= bw + 1 or i + bw <= tl - 1
            Hc, Ec, row = {}, {}, {}
            f, hnf_left = 0, 0          # virtual predecessor of the first column: H = 0, f = 0
            for j in range(beg, end + 1):
                if i == 0:
                    t1, t2 = -go, -ge
                else:
                    hp = Hprev.get(j, 0) if pbeg <= j <= pend else 0
                    ep = Eprev.get(j, 0) if pbeg <= j <= pend else 0
                    if j == end and zero_last:
                        hp = ep = 0
                    t1, t2 = hp - go, ep - ge
                ev = max(t1, t2)
                de = 3 if t1 > t2 else 2
                t1, t2 = hnf_left - go, f - ge
                f = max(t1, t2)
                df = 5 if t1 > t2 else 4
                f1, e1 = max(f, 0), max(ev, 0)
                hd = Hprev.get(j - 1, 0) if (i > 0 and pbeg <= j - 1 <= pend) else 0
                diag = hd + int(mat[q[i], t[j]]) + (int(cb[i]) if cb is not None else 0)
                a = max(e1, f1)
                h = max(a, diag)
                mx = max(mx, h)
                dh = 1 if a <= diag else (de if e1 > f1 else df)
                row[j] = (de, df, dh)
                Hc[j], Ec[j] = h, ev
                hnf_left = max(e1, diag)      # H without its F term
            Hprev, Eprev = Hc, Ec
            dirs.append(row)
        if mx >= score:
            break
        bw *= 2
    i, j, state, out = ql - 1, tl - 1, 2, []
    while i > 0 or j > 0:
        d = dirs[i][j][state]
        if d == 1: i -= 1; j -= 1; state = 2; out.append("M")
        elif d == 2: i -= 1; state = 0; out.append("I")
        elif d == 3: i -= 1; state = 2; out.append("I")
        elif d == 4: j -= 1; state = 1; out.append("D")
        else: j -= 1; state = 2; out.append("D")
    out.append("M")
    return "".join(reversed(out))


def main():
    orc = Oracle()
    m = dict(np.load(os.path.join(ROOT, "tests", "golden", "matrices.npz")))
    mat = m["blosum62_sw"]
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    n = bad = 0
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 400):
        L = int(rng.integers(8, 260))
        q = rng.choice(20, size=L, p=wl.BACKGROUND).astype(np.uint8)
        kind = it % 4
        if kind == 0:
            t = wl.mutate(rng, q, float(rng.uniform(0.4, 0.95)), max_indels=6, max_indel_len=25)
        elif kind == 1:     # a long insertion: big |tlen - qlen|, wide first band, band clipped by the target end
            p = int(rng.integers(1, L))
            ins = rng.choice(20, size=int(rng.integers(20, 120)), p=wl.BACKGROUND).astype(np.uint8)
            t = np.concatenate([q[:p], ins, q[p:]])
        elif kind == 2:     # a long deletion
            a = int(rng.integers(1, max(2, L // 2)))
            t = np.concatenate([q[:a], q[min(L - 1, a + int(rng.integers(5, 90))):]])
        else:               # offsetting indels: band 1 must double several times
            t = wl.mutate(rng, q, 0.9, max_indels=4, max_indel_len=15)
        cb = rng.integers(-3, 4, size=L).astype(np.int8) if it % 3 == 0 else None
        r = orc.sw_align(q, cb, t, mat, 11, 1, need_start=True, need_bt=True)
        if r["score"] <= 0 or not r["bt"]:
            continue
        qs, qe, ts, te = r["q_start"], r["q_end"], r["t_start"], r["t_end"]
        got = banded_rows(q[qs:qe + 1], None if cb is None else cb[qs:qe + 1], t[ts:te + 1], mat, 11, 1, r["score"])
        n += 1
        if got != r["bt"]:
            bad += 1
            if bad < 4:
                print("MISMATCH", it, kind, L, len(t), r["score"], got[:60], r["bt"][:60])
    print("compared %d backtraces, %d differ" % (n, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
