"""GPU box: the nucleotide search section of bench.py alone."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()
import mmseqs2_amd, bench
m = dict(np.load(os.path.join(ROOT, "tests", "golden", "matrices.npz")))
gpu = mmseqs2_amd.MMGpu(0)
a = argparse.Namespace(nucl_contigs=int(sys.argv[1]) if len(sys.argv) > 1 else 4000, nucl_reads=int(sys.argv[2]) if len(sys.argv) > 2 else 1000,
                       nucl_read_len=10000, no_cpu_baseline=False)
print(json.dumps(bench.nucl_search_section(a, gpu, m, a.nucl_contigs, False), indent=1))
