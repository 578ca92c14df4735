"""Records tests/golden/profile_sw.npz from the REAL reference (oracle/_ref/libmmref.so): profile-database entries,
targets and what SmithWaterman::ssw_align returns for them with a profile query (PROFILE_SEQ).
Run in the build container (needs /root/reference/data): python tests/golden/make_profile_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.pyoracle import RefLib          # noqa: E402
from tests.test_profile_query import cases        # noqa: E402

ref = RefLib(comp_bias=False)
mat = np.load(os.path.join(HERE, "matrices.npz"))["blosum62_sw"]
rng = np.random.default_rng(2024)
out = {}
cs = cases(rng, mat, n_queries=7)
out["n_queries"] = np.int64(len(cs))
out["n_targets"] = np.array([len(ts) for _, ts in cs], np.int64)
for qi, (e, ts) in enumerate(cs):
    ref.sw_set_profile_query(e)
    out["entry_%d" % qi] = e
    exp = np.zeros((len(ts), 7), np.int64)
    bts = []
    for k, t in enumerate(ts):
        out["t_%d_%d" % (qi, k)] = t
        r = ref.sw_align(t, mode=2)
        exp[k] = (r["score"], r["q_end"], r["t_end"], r["word"], r["q_start"], r["t_start"], r["ident"])
        bts.append(r["bt"])
    out["exp_%d" % qi] = exp
    out["bt_%d" % qi] = np.array(bts)
np.savez_compressed(os.path.join(HERE, "profile_sw.npz"), **out)
print("wrote profile_sw.npz:", len(cs), "profile queries")
