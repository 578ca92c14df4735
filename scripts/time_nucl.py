"""GPU box helper: the nucleotide alignment section of bench.py a few times, with the library's own time breakdown
(MMGPU_TRACE=1) - where mmgpu_nucl_align's time goes."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

torch.cuda.init()
import bench
import mmseqs2_amd

m = dict(np.load(os.path.join(ROOT, "tests", "golden", "matrices.npz")))
gpu = mmseqs2_amd.MMGpu(0)
a = argparse.Namespace(nucl_contigs=4000, nucl_reads=1000, nucl_read_len=10000, no_cpu_baseline=len(sys.argv) > 1)
for rep in range(3):
    r = bench.nucl_section(a, gpu, m, 0)
    print(json.dumps({k: r[k] for k in ("pairs_per_s", "s_incl_upload_and_download")}), r.get("cpu_baseline", {}).get("value"),
          r.get("cpu_baseline", {}).get("parity_vs_reference"))
