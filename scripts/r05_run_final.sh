# round-5 final package: the bench line (reads the digest-matched PMC passes under profiles/), the whole GPU suite, smoke, and the
# timing of the persisted layout
O=gpurun_out/r05final; mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 600 $O/bench_n1.json; echo
python -m pytest tests -q -m gpu > $O/gpu_tests_full.log 2>&1; tail -4 $O/gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python - > $O/db_timing.txt 2>&1 <<'PY'
import os, sys, time, json
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.getcwd())
import mmseqs2_amd
from mmseqs2_amd import capi, workloads as wl
m = dict(np.load("tests/golden/matrices.npz"))
tv = np.load("tests/golden/tantan_vectors.npz")
km16 = m["vtml80_kmer"].astype(np.int16)
(qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(20000, 50, 100, seed=10)
gpu = mmseqs2_amd.MMGpu(0)
s3, i3 = capi.host_score_matrix(km16, 3)
thr = int(163.2 - 8.917 * 5.7)
out = {}
for rep in range(2):
    t0 = time.perf_counter(); gpu.load_targets(tres, toff, 21); gpu.synchronize(); t1 = time.perf_counter()
    gpu.pf_mask_targets(tv["vtml80_likelihood_ratios"], float(tv["mask_prob"]), 20); gpu.synchronize(); t2 = time.perf_counter()
    gpu.pf_build_index(6, 21, True, s3, i3, km16, thr, m["blosum62_ungapped"]); gpu.synchronize(); t3 = time.perf_counter()
    out["build_%d" % rep] = dict(upload_s=round(t1 - t0, 3), mask_s=round(t2 - t1, 3), index_s=round(t3 - t2, 3), total_s=round(t3 - t0, 3))
path = "/tmp/config3.mmgpu"
t0 = time.perf_counter(); gpu.db_save(path, 1, 3); out["save_s"] = round(time.perf_counter() - t0, 3)
out["file_GB"] = round(os.path.getsize(path) / 1e9, 2)
for rep in range(3):
    t0 = time.perf_counter(); ok = gpu.db_load(path, 1, 3, 6, 21, True, s3, i3, m["blosum62_ungapped"]); gpu.synchronize()
    out["load_with_index_%d_s" % rep] = round(time.perf_counter() - t0, 3)
t0 = time.perf_counter(); gpu.db_load(path, 1, 0); gpu.synchronize(); out["load_targets_only_s"] = round(time.perf_counter() - t0, 3)
out["what"] = "configs[2] database (1 M targets, 278 M residues): build on the device (upload of the host's sequences + tantan + index) against mmgpu_db_load of the persisted layout (page cache warm; eight reader threads, pread into pinned 16 MB chunks)"
print(json.dumps(out))
PY
tail -1 $O/db_timing.txt
MMGPU_TIMING_VARIANTS="MMGPU_DB_FILE=/tmp/t_search.mmgpu" python scripts/dropin_search_timing.py 10 > $O/search_timing.json 2> $O/search_timing_trace.txt; tail -1 $O/search_timing.json
