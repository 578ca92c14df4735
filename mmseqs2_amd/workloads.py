"""Seeded synthetic workloads of SURVEY.md section 8(d) / BASELINE.json configs, as numeric residues.

Residue codes follow the reference's aa2num for blosum62.out / VTML80.out (SubstitutionMatrix.cpp:257-298):
A C D E F G H I K L M N P Q R S T V W Y X -> 0..20.  Everything is deterministic in the seed
(numpy Generator(PCG64)); nothing here reads /root/reference.
"""
import numpy as np

NUM2AA = "ACDEFGHIKLMNPQRSTVWYX"
# Robinson & Robinson background, keyed by letter
_RR = dict(A=.078, R=.051, N=.045, D=.054, C=.019, Q=.043, E=.063, G=.074, H=.022, I=.051, L=.091, K=.057,
           M=.022, F=.039, P=.052, S=.071, T=.058, W=.013, Y=.032, V=.064)
BACKGROUND = np.array([_RR[a] for a in NUM2AA[:20]], np.float64)
BACKGROUND /= BACKGROUND.sum()


def random_seqs(rng, n, mean_len, sd_len, min_len=30, max_len=65535):
    """n i.i.d. sequences, L = clamp(round(N(mean, sd))); returns (residues uint8, offsets uint64[n+1])."""
    lens = np.clip(np.rint(rng.normal(mean_len, sd_len, n)), min_len, max_len).astype(np.int64)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    res = rng.choice(20, size=int(off[-1]), p=BACKGROUND).astype(np.uint8)
    return res, off


def lognormal_seqs(rng, n, mu=5.45, sigma=0.6, min_len=30, max_len=5000):
    """UniRef50-like length law of config 3 (median 233)."""
    lens = np.clip(np.rint(rng.lognormal(mu, sigma, n)), min_len, max_len).astype(np.int64)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    res = rng.choice(20, size=int(off[-1]), p=BACKGROUND).astype(np.uint8)
    return res, off


def mutate(rng, seq, identity, max_indels=3, max_indel_len=10):
    """A homolog of seq: substitutions down to `identity`, up to max_indels indels of length 1..max_indel_len."""
    s = seq.copy()
    mut = rng.random(len(s)) > identity
    s[mut] = rng.choice(20, size=int(mut.sum()), p=BACKGROUND).astype(np.uint8)
    for _ in range(int(rng.integers(0, max_indels + 1))):
        if len(s) < 2 * max_indel_len + 4:
            break
        p = int(rng.integers(1, len(s) - max_indel_len - 1))
        l = int(rng.integers(1, max_indel_len + 1))
        if rng.random() < 0.5:
            s = np.delete(s, slice(p, p + l))
        else:
            s = np.insert(s, p, rng.choice(20, size=l, p=BACKGROUND).astype(np.uint8))
    return s


def seqs_from_list(lst):
    off = np.zeros(len(lst) + 1, np.uint64)
    off[1:] = np.cumsum([len(x) for x in lst])
    res = np.concatenate(lst).astype(np.uint8) if lst else np.zeros(0, np.uint8)
    return res, off


def config2_align_only(n_queries=1000, n_targets=100000, planted_frac=0.10, seed=1):
    """BASELINE.json configs[1]: random L~N(350,35) queries vs targets of the same law, 10 % of the targets
    carry a planted homolog of a random query (identity U(0.3,0.9), <=3 indels)."""
    rq = np.random.default_rng(seed)
    rt = np.random.default_rng(seed + 1)
    rp = np.random.default_rng(seed + 2)
    qres, qoff = random_seqs(rq, n_queries, 350, 35)
    tres, toff = random_seqs(rt, n_targets, 350, 35)
    n_pl = int(n_targets * planted_frac)
    if n_pl:
        ids = rp.choice(n_targets, n_pl, replace=False)
        tl = [tres[int(toff[i]):int(toff[i + 1])] for i in range(n_targets)]
        for i in ids:
            qi = int(rp.integers(0, n_queries))
            qs = qres[int(qoff[qi]):int(qoff[qi + 1])]
            h = mutate(rp, qs, float(rp.uniform(0.3, 0.9)))
            tl[i] = h[:65535]
        tres, toff = seqs_from_list(tl)
    return (qres, qoff), (tres, toff)


def split(res, off):
    return [res[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
