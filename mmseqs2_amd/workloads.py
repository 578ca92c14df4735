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


def write_fasta(path, res, off, prefix="s"):
    """numeric sequences -> FASTA with the letters of NUM2AA (input of `mmseqs createdb`)"""
    lut = np.frombuffer(NUM2AA.encode(), np.uint8)
    letters = lut[res]
    off = off.astype(np.int64)
    with open(path, "wb") as fh:
        for i in range(len(off) - 1):
            fh.write(b">%s%d\n" % (prefix.encode(), i))
            fh.write(letters[off[i]:off[i + 1]].tobytes())
            fh.write(b"\n")


def split(res, off):
    return [res[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]


def mutate_many(rng, seeds_res, seeds_off, parents, id_lo=0.3, id_hi=0.95, max_indels=3, max_indel_len=10):
    """Vectorised family generator: child c is a copy of seed parents[c] with substitutions down to an identity
    drawn from U(id_lo, id_hi) and up to max_indels deletions and insertions of 1..max_indel_len residues.
    Returns (residues, offsets)."""
    seeds_off = seeds_off.astype(np.int64)
    parents = np.asarray(parents, np.int64)
    lens = (seeds_off[parents + 1] - seeds_off[parents])
    off = np.zeros(len(parents) + 1, np.int64)
    off[1:] = np.cumsum(lens)
    total = int(off[-1])
    child_of = np.repeat(np.arange(len(parents)), lens)
    within = np.arange(total, dtype=np.int64) - off[child_of]
    res = seeds_res[seeds_off[parents][child_of] + within].astype(np.uint8)
    ident = rng.uniform(id_lo, id_hi, len(parents))
    mut = rng.random(total) > ident[child_of]
    res[mut] = rng.choice(20, size=int(mut.sum()), p=BACKGROUND).astype(np.uint8)
    # deletions
    ok = lens >= 2 * max_indel_len + 4
    nev = np.where(ok, rng.integers(0, max_indels + 1, len(parents)), 0)
    ev_child = np.repeat(np.arange(len(parents)), nev)
    ev_len = rng.integers(1, max_indel_len + 1, len(ev_child))
    ev_pos = (rng.random(len(ev_child)) * (lens[ev_child] - max_indel_len - 2)).astype(np.int64) + 1
    is_del = rng.random(len(ev_child)) < 0.5
    ds = off[ev_child[is_del]] + ev_pos[is_del]
    d = (np.bincount(ds, minlength=total + 1) - np.bincount(ds + ev_len[is_del], minlength=total + 1))[:total + 1]
    keep = np.cumsum(d[:-1]) == 0
    kept_per_child = np.add.reduceat(keep.astype(np.int64), off[:-1]) if total else np.zeros(0, np.int64)
    kept_per_child[lens == 0] = 0
    res = res[keep]
    lens2 = kept_per_child
    off2 = np.zeros(len(parents) + 1, np.int64)
    off2[1:] = np.cumsum(lens2)
    # insertions (positions relative to the sequence after deletions)
    ic, il = ev_child[~is_del], ev_len[~is_del]
    ipos = off2[ic] + np.minimum(ev_pos[~is_del], np.maximum(lens2[ic] - 1, 0))
    where = np.repeat(ipos, il)
    res = np.insert(res, where, rng.choice(20, size=len(where), p=BACKGROUND).astype(np.uint8))
    lens3 = lens2 + np.bincount(ic, weights=il, minlength=len(parents)).astype(np.int64)
    off3 = np.zeros(len(parents) + 1, np.uint64)
    off3[1:] = np.cumsum(lens3)
    return res, off3


def config3_prefilter(n_families=20000, members=50, n_queries=10000, seed=10, chunk=100000, target_seed=None):
    """BASELINE.json configs[2]: family seeds with L = clamp(LogNormal(5.45, 0.6), 30, 5000);
    targets = `members` mutated members per family, shuffled; queries = one new mutated member of n_queries distinct
    families.  Returns ((qres, qoff), (tres, toff), family_of_target, family_of_query)."""
    rs = np.random.default_rng(seed)
    # target_seed: a different set of family members over the SAME families and queries (one shard per GPU)
    rt = np.random.default_rng(seed + 1 if target_seed is None else target_seed)
    rq = np.random.default_rng(seed + 2)
    sres, soff = lognormal_seqs(rs, n_families)
    fam = np.repeat(np.arange(n_families), members)
    rt.shuffle(fam)
    parts, lens = [], []
    for c in range(0, len(fam), chunk):
        r, o = mutate_many(rt, sres, soff, fam[c:c + chunk])
        parts.append(r)
        lens.append(np.diff(o.astype(np.int64)))
    tres = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    lens = np.concatenate(lens) if lens else np.zeros(0, np.int64)
    toff = np.zeros(len(fam) + 1, np.uint64)
    toff[1:] = np.cumsum(lens)
    qfam = rq.choice(n_families, n_queries, replace=False)
    qres, qoff = mutate_many(rq, sres, soff, qfam)
    return (qres, qoff), (tres, toff), fam, qfam


# ---------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4] (nucleotide): reads sampled from contigs with substitutions and indels
NUCL_LETTERS = "ACTGN"                       # numeric codes of NucleotideMatrix (nucleotide.out column order, N -> X = 4)
NUCL_REVERSE = np.array([2, 3, 0, 1, 4], np.uint8)


def mutate_nucl(rng, seq, sub=0.10, indel=0.02):
    """substitutions at rate `sub`, insertions and deletions (1..3 bases) at rate `indel` / 2 each, vectorised"""
    s = seq.copy()
    m = rng.random(len(s)) < sub
    s[m] = rng.integers(0, 4, size=int(m.sum())).astype(np.uint8)
    keep = np.ones(len(s), bool)
    for p in np.nonzero(rng.random(len(s)) < indel / 2)[0]:
        keep[p:p + int(rng.integers(1, 4))] = False
    s = s[keep]
    ins = np.nonzero(rng.random(len(s)) < indel / 2)[0]
    if len(ins):
        s = np.insert(s, ins, rng.integers(0, 4, size=len(ins)).astype(np.uint8))
    return s if len(s) else np.zeros(1, np.uint8)


def write_nucl_fasta(path, seqs, prefix="s"):
    """numeric nucleotide sequences (codes of NUCL_LETTERS) -> FASTA (input of `mmseqs createdb`)"""
    lut = np.frombuffer(NUCL_LETTERS.encode(), np.uint8)
    with open(path, "wb") as fh:
        for i, s in enumerate(seqs):
            fh.write(b">%s%d\n" % (prefix.encode(), i))
            fh.write(lut[s].tobytes())
            fh.write(b"\n")


def config5_nucleotide(n_contigs=4000, n_reads=1000, read_len=10000, false_hits=4, seed=20):
    """-> (queries list, (tres, toff), pairs [(query, target, diagonal16, reverse)]).  Contig lengths ~ LogNormal with
    median 20 kb, clipped to [read_len + 500, 60000] (the reference splits longer sequences, Parameters.h:271).  Every
    read has the pair a prefilter would report (its source contig, the true diagonal; every second read is stored as its
    reverse complement and flagged `reverse`) and `false_hits` unrelated contigs with arbitrary diagonals."""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.round(rng.lognormal(np.log(20000.0), 0.5, size=n_contigs)), read_len + 500, 60000).astype(np.int64)
    toff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    tres = rng.integers(0, 4, size=int(toff[-1]), dtype=np.uint8)      # (1.1e9 letters for the 50 000 contigs of configs[4])
    queries, pairs = [], []
    for r in range(n_reads):
        c = int(rng.integers(0, n_contigs))
        start = int(rng.integers(0, lens[c] - read_len))
        read = mutate_nucl(rng, tres[int(toff[c]) + start:int(toff[c]) + start + read_len])
        rev = r & 1
        queries.append(NUCL_REVERSE[read[::-1]] if rev else read)
        pairs.append((r, c, (-start) & 0xFFFF, rev))
        for _ in range(false_hits):
            pairs.append((r, int(rng.integers(0, n_contigs)), int(rng.integers(0, 65536)), int(rng.integers(0, 2))))
    return queries, (tres, toff), pairs
