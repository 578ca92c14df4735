"""Records tests/golden/nucl_pf.npz from the REAL reference (oracle/_ref/libmmref.so): nucleotide queries through
QueryMatcher::matchQuery(.., isNucleotide = true) with exact k-mer matching (k = 13 spaced; 4^13 offsets keep the CPU test
small) for several (max_hits, bins) settings.  Queries whose result depends on the reference's unstable sort (two saturated
diagonals of one target with the same exact score) are marked.  Run in the build container."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle                                        # noqa: E402
from tests.test_nucl_prefilter import nucl_case, nucl_oracle, NUCL_SETTINGS   # noqa: E402

K = 13
ref = pyoracle.RefNuclPrefilter(K, True)
o, _ = nucl_oracle(K, True)
qs, tres, toff = nucl_case(4242, n_targets=500, n_queries=10)
ref.build_index(tres, toff)
o.build_index(tres, toff, 0)
out = dict(tres=tres, toff=toff, n_queries=np.int64(len(qs)), k=np.int64(K))
for qi, q in enumerate(qs):
    out["q_%d" % qi] = q
    for si, (mh, bins) in enumerate(NUCL_SETTINGS):
        r = ref.match(q, max_hits=mh, force_bins=bins)
        x = o.match(q, None, bins, max_hits=mh, exact=True, nucleotide=True)
        out["tie_%d_%d" % (si, qi)] = np.int64(x["stats"]["sat_tie"])
        out["hits_%d_%d" % (si, qi)] = np.stack([r["id"].astype(np.int64), r["score"].astype(np.int64), r["diagonal"].astype(np.int64)])
np.savez_compressed(os.path.join(HERE, "nucl_pf.npz"), **out)
print("wrote nucl_pf.npz; ties:", sum(int(v) for k, v in out.items() if k.startswith("tie_")))
