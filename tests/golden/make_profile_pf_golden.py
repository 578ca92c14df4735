"""Records tests/golden/profile_pf.npz from the REAL reference (oracle/_ref/libmmref.so): profile-database entries run through
QueryMatcher::matchQuery with Sequence::profile_matrix (Prefiltering.cpp:832-834) against an index built with k-mer
threshold 0, for several (max_hits, CacheFriendlyOperations bins) settings; plus what Sequence::mapProfile derived from every
entry (the inputs of mmgpu_pf_query's profile fields).  Run in the build container: python tests/golden/make_profile_pf_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle                      # noqa: E402
from tests.test_profile_query import pf_profile_case, PF_PROFILE_SETTINGS, PF_PROFILE_THR   # noqa: E402

ref = pyoracle.RefPrefilter(6)
entries, tres, toff = pf_profile_case(2025, n_queries=8, n_targets=3000)
ref.build_index(tres, toff, 0)
out = dict(tres=tres, toff=toff, n_queries=np.int64(len(entries)), thr=np.int64(PF_PROFILE_THR),
           settings=np.array(PF_PROFILE_SETTINGS, np.int64))
for qi, e in enumerate(entries):
    out["entry_%d" % qi] = e
    for si, (mh, fb) in enumerate(PF_PROFILE_SETTINGS):
        ident = None if qi % 2 else qi
        r = ref.match_profile(e, PF_PROFILE_THR, max_hits=mh, force_bins=fb, identity_id=ident)
        out["hits_%d_%d" % (si, qi)] = np.stack([r["id"].astype(np.int64), r["score"].astype(np.int64), r["diagonal"].astype(np.int64)])
        if si == 0:
            out["pscore_%d" % qi] = r["pscore"]
            out["pindex_%d" % qi] = r["pindex"]
            out["aln_%d" % qi] = r["aln"]
            out["letters_%d" % qi] = r["letters"]
np.savez_compressed(os.path.join(HERE, "profile_pf.npz"), **out)
print("wrote profile_pf.npz")
