"""Records tests/golden/profile_long_pf.npz from the REAL reference (oracle/_ref/libmmref.so): profile-database entries run through
QueryMatcher::matchQuery against targets of 32768 residues or more (tests/test_profile_query.py pf_profile_long_case: computeLongScore
and the batches of scoreDiagonalAndUpdateHits with the profile's own score rows), and what Sequence::mapProfile derived from every
entry (the inputs of mmgpu_pf_query's profile fields).  The targets are regenerated from the seed by the test; their CRC is kept.
Run in the build container: python tests/golden/make_profile_long_pf_golden.py"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle                      # noqa: E402
from tests.test_profile_query import pf_profile_long_case, PF_LONG_SEEDS, PF_LONG_MAX_HITS   # noqa: E402

ref = pyoracle.RefPrefilter(6)
out = dict(seeds=np.array(PF_LONG_SEEDS, np.int64), max_hits=np.array(PF_LONG_MAX_HITS, np.int64))
for seed in PF_LONG_SEEDS:
    entries, tres, toff = pf_profile_long_case(seed)
    ref.build_index(tres, toff, 0)
    out["tres_crc_%d" % seed] = np.int64(zlib.crc32(tres.tobytes()))
    out["n_queries_%d" % seed] = np.int64(len(entries))
    for qi, e in enumerate(entries):
        for mh in PF_LONG_MAX_HITS:
            r = ref.match_profile(e, 99, max_hits=mh, max_seq_len=65535, identity_id=None)
            out["hits_%d_%d_%d" % (seed, mh, qi)] = np.stack([r["id"].astype(np.int64), r["score"].astype(np.int64), r["diagonal"].astype(np.int64)])
        out["pscore_%d_%d" % (seed, qi)] = r["pscore"]
        out["pindex_%d_%d" % (seed, qi)] = r["pindex"].astype(np.uint8) if r["pindex"].max() < 256 else r["pindex"]
        out["aln_%d_%d" % (seed, qi)] = r["aln"]
        out["letters_%d_%d" % (seed, qi)] = r["letters"]
np.savez_compressed(os.path.join(HERE, "profile_long_pf.npz"), **out)
print("wrote profile_long_pf.npz")
