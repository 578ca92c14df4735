"""Records tests/golden/block_vectors.npz from a reference library whose block-aligner branch runs:
    python tests/golden/make_block_golden.py <libmmref_*.so> <out.npz>
With oracle/_ref/libmmref_rust.so (scripts/make_block_goldens.sh, a Rust-equipped box) the file pins the restatement against
the real crate.  With oracle/_ref/libmmref_block.so (the restatement behind the crate's C API) it only exercises this script.
Pairs: family members of the headline workload generator with int16-range scores (s_align::word == 1), lengths 30..2000+,
BLOSUM62 11/1 with composition bias - the configuration `mmseqs search` runs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mmseqs2_amd import workloads as wl      # noqa: E402
from oracle import pyoracle as po            # noqa: E402


def main():
    lib, out = sys.argv[1], sys.argv[2]
    m = dict(np.load(os.path.join(ROOT, "tests", "golden", "matrices.npz")))
    mat = m["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    ref = po.RefLib(serialized=m["blosum62_serialized"], db_residues=300000000, lib_path=lib)
    orc = po.Oracle()
    rec = {k: [] for k in ("q", "cb", "t", "score", "q_end", "t_end", "q_start", "t_start", "ident", "bt")}
    for seed, n_fam, members in ((101, 260, 6), (102, 40, 4)):
        (qres, qoff), (tres, toff), fam_t, fam_q = wl.config3_prefilter(n_fam, members, n_fam, seed=seed)
        qs, ts = wl.split(qres, qoff), wl.split(tres, toff)
        for qi, q in enumerate(qs):
            ref.sw_set_query(q)
            cb = orc.round_comp_bias(orc.comp_bias(sub16, m["blosum62_pback"], q, 1.0))
            for ti in np.nonzero(fam_t == fam_q[qi])[0][:3]:
                t = ts[ti]
                r = ref.sw_align(t, mode=2, evalue_thr=1e300)
                if r["word"] != 1:
                    continue
                # (whether the block aligner answered or the reference fell back to its Smith-Waterman traceback, :873-882, is not
                # visible in s_align: the test accepts the fallback's result only where the restatement declines as well)
                rec["q"].append(q); rec["cb"].append(cb); rec["t"].append(t)
                for k in ("score", "q_end", "t_end", "q_start", "t_start", "ident"):
                    rec[k].append(r[k])
                rec["bt"].append(np.frombuffer(r["bt"].encode(), np.uint8))
    off = lambda xs: np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.int64)
    np.savez_compressed(out, matrix=mat, q_res=np.concatenate(rec["q"]), q_cb=np.concatenate(rec["cb"]).astype(np.int8), q_off=off(rec["q"]),
                        t_res=np.concatenate(rec["t"]), t_off=off(rec["t"]), bt=np.concatenate(rec["bt"]), bt_off=off(rec["bt"]),
                        **{k: np.array(rec[k]) for k in ("score", "q_end", "t_end", "q_start", "t_start", "ident")})
    print("make_block_golden: %d int16-range pairs from %s -> %s" % (len(rec["score"]), os.path.basename(lib), out))


if __name__ == "__main__":
    main()
