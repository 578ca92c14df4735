#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X-native prefilter -> align path.

Metric (BASELINE.json): "giga-cells/s (SW) + query seqs/s, 10k queries vs 1M targets, 1/2/4/8 GPU".

Workload = BASELINE.json configs[2] (and configs[3] for N > 1): 10 000 protein queries against 1 000 000 family-structured
targets (UniRef50-like lengths), `-s 5.7` (k = 6, spaced), `--max-seqs 300`, then the gapped alignment
of every hit list with start positions for the pairs passing `-e 1e-3` (what `mmseqs search` runs: prefilter + align
--alignment-mode 2).  A STEP = one pass of the whole hot path over the query set with the target DB, the k-mer index and
the prepared query batch resident in HBM:

  N = 1: prefilter kernels -> device-side hand-over of the hit lists (mmgpu_sw_prepare_from_pf) -> alignment kernels.
  N > 1 (one process per GPU, torch.distributed / RCCL): the SAME 1M targets dealt to the ranks by length bucket (strong
         scaling): prefilter of the shard -> ONE all-gather of 16-byte exchange records -> merge (result identical to the
         1-GPU run) -> every rank aligns the pairs whose target it holds -> all-gather of the owned alignment records.
         `--scaling weak` gives every rank its own 1M-target shard of an N-million database instead.

`value` = SW giga-cells/s: forward DP cells of the step (sum of qlen*tlen over the aligned pairs, Alignment.cpp:380,530
convention; reverse-scan cells are NOT counted) divided by the time of the alignment stage inside the step (HIP events on
the launch stream).  `queries_per_s` = queries / whole step time and `ms_per_step` (the whole step) stand beside it.

Secondary sections of the same line: "align_only" (configs[1]: 1000 x 100 000 all-vs-all alignment), "nucleotide_align"
(configs[4], alignment step).  CPU baselines = the real reference (oracle/_ref/libmmref.so) on the host cores, bounded
samples; they double as full-size parity checks of the timed run's results.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
# packed-int16 (VOP3P) VALU ops issue at 16 lanes/clk/SIMD on gfx950: measured 38.1 Tlane-op/s for v_pk_max_i16 /
# v_pk_sub_u16 / v_perm_b32 (profiles/r01_valu_issue_rate_probe.txt) = 256 CU x 4 SIMD x 16 lanes x 2.4 GHz = 39.3e12
VALU_LANE_OPS_PER_S = 256 * 4 * 16 * 2.4e9
PROFILE_ROUND = "r06"


def allreduce(torch, dist, values, op="sum"):
    """All-reduce of a few python floats (device tensor with nccl, host tensor with gloo)."""
    if dist is None:
        return [float(v) for v in values]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(v) for v in values], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return [float(v) for v in t.cpu()]


def csrc_digest():
    """digest of the device sources this process runs with (scripts/csrc_digest.py)"""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        from csrc_digest import csrc_digest as f
        return f(ROOT)
    except Exception:
        return None


def pmc_file_digest(path):
    """the csrc digest a committed rocprof summary was taken with (header line written by scripts/rocprof_summary.py), or None"""
    try:
        for line in open(path):
            if line.startswith("# csrc_digest:"):
                return line.split(":", 1)[1].strip()
            if not line.startswith("#"):
                break
    except OSError:
        pass
    return None


def pmc_traffic(kernels, stem, check_digest=True, per_run_of=None):
    """HBM bytes per launch of the named kernels (substrings; summed over their template instantiations) from the committed rocprofv3 PMC
    passes of this same workload (profiles/<stem>_pmc_{fetch,write}_size.txt: separate --pmc FETCH_SIZE / WRITE_SIZE
    runs, unit KB, mean per dispatch).  per_run_of: a kernel that is dispatched exactly once per run of the stage - the counters of
    every named kernel are then summed over ALL its dispatches and divided by that kernel's dispatch count, i.e. bytes per run of
    the stage also where a kernel is launched several times per run (the prefilter's stage chunks).  FETCH_SIZE is reported as measured; MI355X_MICROARCH.md notes it under-counts
    wide coalesced reads by 2x on gfx950, so this is a lower bound.  check_digest: the passes must have been taken with the device
    sources this process runs with (their header names the digest, scripts/csrc_digest.py) - counters of other kernels are not
    reported as this run's traffic (None)."""
    tot = 0.0
    for kind in ("fetch", "write"):
        path = os.path.join(ROOT, "profiles", "%s_pmc_%s_size.txt" % (stem, kind))
        if not os.path.exists(path):
            return None
        if check_digest and pmc_file_digest(path) != csrc_digest():
            return None
        found = False
        runs = None
        lines = [l for l in open(path) if "_SIZE" in l and "mean=" in l and " n=" in l and "sum=" in l]
        if per_run_of is not None:
            for line in lines:
                if per_run_of in line:
                    runs = float(line.split(" n=")[1].split()[0])
            if not runs:
                return None
        for line in lines:
            if any(k in line for k in kernels):
                if runs:
                    tot += float(line.split("sum=")[1].split()[0]) / runs * 1024.0
                else:
                    tot += float(line.split("mean=")[1]) * 1024.0
                found = True
        if not found:
            return None
    return round(tot)


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines (the real reference on the host cores; outside every timed region)
def host_cpu_topology():
    """(hardware threads, physical cores) of the box, from /proc/cpuinfo."""
    threads = os.cpu_count() or 1
    cores = set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return threads, (len(cores) or threads)


def cpu_quota_cores():
    """cores' worth of CPU time the cgroup grants this process (cpu.max = "<quota> <period>"), None if unlimited / unknown"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        return None


def host_limits():
    """What bounds the host side of this box besides its core count: cgroup CPU quota, load, CPU model."""
    out = {}
    for key, path in (("cgroup_cpu_max", "/sys/fs/cgroup/cpu.max"), ("loadavg", "/proc/loadavg")):
        try:
            out[key] = open(path).read().strip()
        except OSError:
            out[key] = None
    try:
        out["model"] = next(l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
        out["mhz_first_cpus"] = [round(float(l.split(":")[1])) for l in open("/proc/cpuinfo") if l.startswith("cpu MHz")][:4]
    except (OSError, StopIteration):
        pass
    return out


def sw_cpu_baseline_lists(matrices, qres, qoff, lists, tres, toff, budget_s, gpu_res, evalue_thr=1e-3):
    """Alignment::run's inner loop on the box's host cores as ONE native OpenMP call (oracle/ref_shim.cpp
    mmref_sw_lists_omp: one SmithWaterman per thread, schedule(dynamic, 5), Alignment.cpp:279-313): the reference's striped
    AVX2 Smith-Waterman (uint8 pass + int16 re-run) with start positions for the pairs passing -e (alignment mode
    SCORE_COV, what the timed device step computes) over the SAME hit lists; a bounded sample of the queries.  Nothing is
    compared inside the timed region: the result arrays are checked against the device's records afterwards."""
    from oracle import pyoracle
    if not pyoracle.ref_available():
        return None
    hw_threads, phys_cores = host_cpu_topology()
    nq = len(lists)
    tlen = (toff[1:] - toff[:-1]).astype(np.int64)
    qlen = (qoff[1:] - qoff[:-1]).astype(np.int64)
    ser = matrices["blosum62_serialized"]
    l_off = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.uint64)
    l_ids = np.concatenate(lists).astype(np.uint32) if nq else np.zeros(0, np.uint32)
    cells_q = np.array([int(qlen[i]) * int(tlen[lists[i]].sum()) for i in range(nq)], np.float64)

    def run(q_from, q_to, threads, mode):
        sec, used, res = pyoracle.ref_sw_lists_omp(ser, qres, qoff, l_ids, l_off, tres, toff, q_from, q_to, threads, mode=mode,
                                                   evalue_thr=evalue_thr, db_residues=int(toff[-1]))
        return sec, used, res, float(cells_q[q_from:q_to].sum())

    # calibration pass (also warms the page cache of the targets), then a sample sized for the budget
    n_cal = min(nq, max(64, hw_threads))
    sec, used, _, cells = run(0, n_cal, hw_threads, 1)
    rate = cells / max(sec, 1e-3)
    per_q = float(cells_q.mean()) if nq else 1.0
    n_s = int(min(nq - n_cal, max(4 * hw_threads, rate * (budget_s / 3.0) / max(per_q, 1.0))))
    lo, hi = (n_cal, n_cal + n_s) if n_s > 0 else (0, n_cal)
    runs = {}
    sec_a, used_a, res, cells = run(lo, hi, hw_threads, 1)
    runs["all_hardware_threads"] = {"threads": used_a, "seconds": round(sec_a, 3), "gcups": round(cells / sec_a / 1e9, 2)}
    if phys_cores != hw_threads:
        sec_p, used_p, _, _ = run(lo, hi, phys_cores, 1)
        runs["one_thread_per_physical_core"] = {"threads": used_p, "seconds": round(sec_p, 3), "gcups": round(cells / sec_p / 1e9, 2)}
    # a container's CPU quota (cgroup cpu.max) decides how many cores this process really gets: more threads than that only
    # add contention, so the reference is also run with one and two threads per core of the quota
    quota = cpu_quota_cores()
    if quota is not None:
        for mult in (1, 2):
            th = int(max(1, min(hw_threads, round(quota * mult))))
            if all(th != r["threads"] for r in runs.values()):
                sec_q, used_q, _, _ = run(lo, hi, th, 1)
                runs["%d_thread%s_per_core_of_the_cpu_quota" % (mult, "" if mult == 1 else "s")] = {
                    "threads": used_q, "seconds": round(sec_q, 3), "gcups": round(cells / sec_q / 1e9, 2)}
    timed = [k for k in runs]
    sec_s, used_s, _, _ = run(lo, hi, hw_threads, 0)
    runs["score_and_end_only_all_hardware_threads"] = {"threads": used_s, "seconds": round(sec_s, 3), "gcups": round(cells / sec_s / 1e9, 2)}
    # thread scaling on a slice of the sample: what one thread delivers on this box, and where the box stops scaling
    # (a container's CPU quota or shared host shows up here, not in the thread count the OS reports)
    scaling = {}
    n_sc = min(hi - lo, 256)
    for th in (1, 8, 32):
        if th <= hw_threads:
            q_to = lo + (max(8, n_sc // 8) if th == 1 else n_sc)
            sec_t, used_t, _, cells_t = run(lo, q_to, th, 1)
            scaling[str(used_t)] = round(cells_t / sec_t / 1e9, 2)
    best = max(timed, key=lambda k: runs[k]["gcups"])
    # parity of the timed device run against these results (after the timers)
    a, b = int(l_off[lo]), int(l_off[hi])
    g = np.concatenate([gpu_res[qi] for qi in range(lo, hi)]) if hi > lo else np.zeros(0, gpu_res[0].dtype)
    sc = res["score"][a:b].astype(np.int64)
    pos = sc > 0
    bad = int(np.count_nonzero(g["score"].astype(np.int64) != sc)) + int(np.count_nonzero((g["t_end"] != res["t_end"][a:b]) & pos)) \
        + int(np.count_nonzero((g["q_end"] != res["q_end"][a:b]) & pos))
    g_st, r_st = g["q_start"] >= 0, res["q_start"][a:b] >= 0
    both = g_st & r_st
    bad_start = int(np.count_nonzero((g["q_start"] != res["q_start"][a:b]) & both)) + int(np.count_nonzero((g["t_start"] != res["t_start"][a:b]) & both))
    usable = phys_cores if quota is None else max(1, min(phys_cores, int(round(quota))))
    return {"value": runs[best]["gcups"], "unit": "GCUPS", "cores": usable, "physical_cores": phys_cores, "cpu_quota_cores": quota,
            "threads": runs[best]["threads"],
            "hardware_threads": hw_threads, "kind": "reference",
            "gcups_per_thread": round(runs[best]["gcups"] / max(runs[best]["threads"], 1), 3),
            "what": "the reference's own ssw_init + ssw_align (AVX2 striped uint8 pass, int16 re-run, reverse scan for the pairs "
                    "passing -e 1e-3) in one native OpenMP call, one SmithWaterman per thread, schedule(dynamic,5) as Alignment.cpp:279-313; "
                    "forward cells / wall seconds inside the call, the same accounting as the device value",
            "sample": "the hit lists of queries [%d, %d) of the %d queries of the same workload (%d pairs, %.4g forward cells); runs: %s"
                      % (lo, hi, nq, b - a, cells, best),
            "runs": runs, "gcups_by_thread_count_on_a_slice": scaling, "host": host_limits(),
            "parity_vs_baseline": {"pairs_compared": b - a, "field_mismatches": bad, "fields": "score, q_end, t_end",
                                   "pairs_with_start_on_both": int(both.sum()), "start_field_mismatches": bad_start,
                                   "pairs_with_start_on_one_side_only": int(np.count_nonzero(g_st != r_st))}}


def prefilter_cpu_baseline(matrices, qres, qoff, tres, toff, kmer_thr, budget_s, gpu_lists, mask=False):
    """The reference's own prefilter query loop (QueryMatcher::matchQuery per OpenMP thread, Prefiltering.cpp:820-917)
    from oracle/_ref/libmmref.so on the host cores, bounded sample of the same queries against the same targets.
    The lists it produces are also the checker of the device lists at full size."""
    from oracle import pyoracle
    if not pyoracle.ref_available():
        return None
    cores = os.cpu_count() or 1
    ref = pyoracle.RefPrefilter(6, serialized=(matrices["vtml80_serialized"].tobytes(), matrices["blosum62_serialized"].tobytes()))
    t0 = time.time()
    mask_note = ""
    if mask:      # the reference's own tantan (Masker::maskSequence, maskTantan) over the targets, then its index over the masked ones
        tres, n_masked, _, _ = ref.tantan_mask(tres, toff, float(np.float32(0.9)))
        mask_note = "; targets masked by the reference's tantan (%d residues, %.1f s)" % (n_masked, time.time() - t0)
    ref.build_index(tres, toff, kmer_thr)
    t_index = time.time() - t0
    nq = len(qoff) - 1
    n_cal = min(nq, max(cores, 64))
    sec, hits, dbm, _ = ref.match_batch(qres[:int(qoff[n_cal])], qoff[:n_cal + 1], cores)
    rate = n_cal / max(sec, 1e-3)
    n = int(min(nq, max(n_cal, rate * budget_s)))
    sec, hits, dbm, counts, lists = ref.match_batch(qres[:int(qoff[n])], qoff[:n + 1], cores, want_lists=True)
    out = {"value": round(n / sec, 1), "unit": "queries/s (prefilter only)", "cores": cores, "kind": "reference",
           "sample": "first %d of the %d queries against the same %d targets, %.1f s wall, %d threads; reference index build "
                     "%.1f s (not counted)%s" % (n, nq, len(toff) - 1, sec, cores, t_index, mask_note),
           "db_matches_per_query": round(dbm / n), "hits_per_query": round(hits / n, 1)}
    bad = 0
    for qi in range(n):
        gi, gs, gd = gpu_lists[qi]
        c = int(counts[qi])
        same = (len(gi) == c and np.array_equal(gi, lists["ids"][qi, :c]) and np.array_equal(gs, lists["scores"][qi, :c])
                and np.array_equal(gd, lists["diags"][qi, :c]))
        bad += not same
    out["parity_vs_reference"] = {"queries_compared": n, "queries_with_different_hit_lists": int(bad),
                                  "reference_cache_bins": int(lists["bins"]),
                                  "fields": "hit ids, prefilter scores, diagonals, order"}
    return out


# ---------------------------------------------------------------------------------------------------------------------
def end_to_end_pass(gpu, matrices, qres, qoff, kmer_thr, max_res, db_residues, fused_ref, counts_ref):
    """SURVEY.md section 8d's queries/s: host numeric sequences in -> host hit_t lists + mmgpu_sw_hit records out.  Inside the
    timed region: both composition-bias passes on the host (prefilter: float over the k-mer matrix; alignment: ssw_init's int8
    over BLOSUM62), the per-query start-score thresholds from -e, the query descriptors, mmgpu_pf_prepare (host k-mer
    thresholds + upload), the prefilter kernels, the device-side hand-over, the alignment kernels, and both downloads.
    Resident from before: targets + k-mer index (their set-up time is reported beside this).  Pass A runs without any
    intermediate synchronisation (the number); pass B synchronises after every stage (the breakdown)."""
    from mmseqs2_amd import capi, evalue
    km16 = matrices["vtml80_kmer"].astype(np.int16)
    mat = matrices["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    qlens = (qoff[1:] - qoff[:-1]).astype(np.int64)
    nq = len(qlens)
    evalue.min_scores_for_evalue(1e-3, qlens[:4], db_residues)       # import scipy outside the timed region
    out = {}
    for label, sync in (("no_intermediate_sync", False), ("stage_breakdown", True)):
        stages = {}
        t_all = time.perf_counter()

        def mark(name, t0):
            if sync:
                gpu.synchronize()
                stages[name] = round((time.perf_counter() - t0) * 1e3, 2)
            return time.perf_counter()

        t0 = time.perf_counter()
        cbf, _ = capi.host_comp_bias_batch(km16, matrices["vtml80_pback"], qres, qoff, want_round=False)
        _, cbr = capi.host_comp_bias_batch(sub16, matrices["blosum62_pback"], qres, qoff, want_float=False)
        thr = evalue.min_scores_for_evalue(1e-3, qlens, db_residues)
        t0 = mark("host_composition_bias_x2_and_evalue_thresholds", t0)
        pfb = gpu.pf_prepare_flat(qres, qoff, cbf, kmer_thr, max_hits=max_res, min_diag_score=15, ref_bins=2)
        msh = gpu.sw_marshal_flat(mat, 11, 1, qres, qoff, cbr, thr)
        t0 = mark("descriptors_pf_prepare_upload", t0)
        pfb.run()
        t0 = mark("prefilter_kernels", t0)
        fb = gpu.sw_prepare_from_pf(mat, 11, 1, None, pfb, mode=1, marshalled=msh)
        fb.run()
        t0 = mark("handover_and_alignment_kernels", t0)
        hits, counts, status, _ = pfb.fetch()
        res = fb.fetch()
        t0 = mark("download_hit_lists_and_alignment_records", t0)
        wall = time.perf_counter() - t_all
        if not sync:
            res = res.reshape(nq, pfb.max_hits)
            same = bool(np.array_equal(counts, counts_ref)) and all(
                np.array_equal(res[qi, :counts[qi]], fused_ref[qi, :counts[qi]]) for qi in range(0, nq, 7))
            out.update({"seconds": round(wall, 4), "queries_per_s_end_to_end": round(nq / wall, 1),
                        "results_equal_to_the_timed_steps": same,
                        "bytes_downloaded": int(hits.nbytes + counts.nbytes + status.nbytes + res.nbytes)})
        else:
            out["stage_ms_with_synchronisation"] = stages
            out["seconds_with_synchronisation"] = round(wall, 4)
        fb.free()
        pfb.free()
    out["what"] = ("host numeric query sequences in -> host hit_t lists + mmgpu_sw_hit records out, one call sequence over the C-ABI for the "
                   "whole query set: composition bias x2 + E-value thresholds on the host, descriptors, mmgpu_pf_prepare, prefilter, "
                   "mmgpu_sw_prepare_from_pf, alignment (score, ends, starts for pairs passing -e 1e-3), both downloads; targets and k-mer "
                   "index resident (set-up reported in setup_s)")
    return out


def module_seconds(args, qres, qoff, tres, toff):
    """Wall seconds of `mmseqs prefilter` and `mmseqs align` (default flags: --mask 1; threads: see below) through the stock
    binary and through the binary with integration/mmseqs_mmgpu.patch, on the headline workload written as FASTA; the two
    alignment databases are compared entry by entry.  Needs oracle/_ref/mmseqs_{stock,mmgpu} (integration/build_mmseqs.sh)."""
    import shutil
    import subprocess
    import tempfile
    from mmseqs2_amd import workloads as wl, dbio
    stock = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock")
    patched = os.path.join(ROOT, "oracle", "_ref", "mmseqs_mmgpu")
    if not (os.path.exists(stock) and os.path.exists(patched)):
        return None
    # threads: what serves the reference best on this box - two per core of the cgroup's CPU quota when there is one (the
    # thread sweep of cpu_baseline shows more only adds contention), all hardware threads otherwise; the same for both binaries
    quota = cpu_quota_cores()
    hw = os.cpu_count() or 1
    threads = str(hw if quota is None else int(max(1, min(hw, round(2 * quota)))))
    w = tempfile.mkdtemp(prefix="mmgpu_modules_")
    try:
        wl.write_fasta(os.path.join(w, "q.fasta"), qres, qoff, "q")
        wl.write_fasta(os.path.join(w, "t.fasta"), tres, toff, "t")
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "mmseqs2_amd", "lib") + ":" + env.get("LD_LIBRARY_PATH", "")

        def run(b, a):
            t0 = time.perf_counter()
            r = subprocess.run([b] + a, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=args.module_timeout)
            if r.returncode != 0:
                raise RuntimeError("%s %s failed: %s" % (os.path.basename(b), a[0], r.stdout[-400:]))
            return time.perf_counter() - t0, r.stdout

        run(stock, ["createdb", "q.fasta", "q", "-v", "1"])
        run(stock, ["createdb", "t.fasta", "t", "-v", "1"])
        out = {"workload": "the headline workload through `mmseqs prefilter -s 5.7` + `mmseqs align --alignment-mode 2 -e 0.001` (what "
                           "`mmseqs search` runs), default flags otherwise (--mask 1, --max-seqs 300), --threads %s" % threads}
        stub = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock_stub")
        binaries = [("stock", stock), ("patched", patched)] + ([("stock_block_aligner_stubbed", stub)] if os.path.exists(stub) else [])
        for name, b in binaries:
            tp, log = run(b, ["prefilter", "q", "t", "pref_" + name, "-s", "5.7", "--threads", threads, "-v", "3"])
            ta, _ = run(b, ["align", "q", "t", "pref_" + name, "aln_" + name, "--alignment-mode", "2", "-e", "0.001", "--threads", threads, "-v", "3"])
            out[name] = {"prefilter_wall_s": round(tp, 2), "align_wall_s": round(ta, 2),
                         "queries_per_s_prefilter_plus_align": round((len(qoff) - 1) / (tp + ta), 1)}
        out["note"] = ("no Rust toolchain in this image: `stock` links the block-aligner crate's C API over a scalar C restatement of the crate "
                       "(oracle/ref_block_capi.cpp), slower than the crate's AVX2 code; `stock_block_aligner_stubbed` is the same tree with do-nothing "
                       "stubs, i.e. every int16-range hit takes the reference's Smith-Waterman fallback (rounds 1-2's stock) - the real stock "
                       "`align` time lies near it")
        n, bad, _ = dbio.diff_dbs(os.path.join(w, "aln_stock"), os.path.join(w, "aln_patched"))
        out["alignment_dbs_identical"] = bad == 0
        out["entries_compared"] = n
        # round 4 (row f2): `mmseqs search` through the patched binary runs both modules inside the search process (one device
        # context, tantan masking + index on the device, hit lists in memory, targets resident once): wall time of the whole
        # command, best of two, and its result database against the stock binary's align database above
        best = None
        for rep in range(2):
            ts_, _ = run(patched, ["search", "q", "t", "res_fused%d" % rep, "tmp_fused%d" % rep, "-s", "5.7", "--threads", threads, "-v", "3"])
            best = ts_ if best is None else min(best, ts_)
        n2, bad2, _ = dbio.diff_dbs(os.path.join(w, "aln_stock"), os.path.join(w, "res_fused0"))
        stock_total = out["stock"]["prefilter_wall_s"] + out["stock"]["align_wall_s"]
        out["patched_search_fused"] = {"wall_s": round(best, 2), "queries_per_s": round((len(qoff) - 1) / best, 1),
                                       "result_db_identical_to_stock_align_db": bad2 == 0, "entries_compared": n2,
                                       "what": "`mmseqs search q t res tmp -s 5.7` (default flags), whole command incl. process start, database "
                                               "opening, masking, index build, prefilter, alignment, result database"}
        out["speedup_search_fused_vs_stock_prefilter_plus_align"] = round(stock_total / best, 2)
        # round 5 (row f1): the same command with MMGPU_DB_FILE - the first search builds targets / masked view / index on the device
        # and saves the device layout (mmgpu_db_save), the next two load it (mmgpu_db_load: no SequenceLookup fill on the host, no
        # upload from it, no masking, no index build) and are timed
        env_plain = dict(env)
        env["MMGPU_DB_FILE"] = os.path.join(w, "t.mmgpu")
        best_db, first_db, loaded = None, None, True
        for rep in range(3):
            ts_, log_ = run(patched, ["search", "q", "t", "res_db%d" % rep, "tmp_db%d" % rep, "-s", "5.7", "--threads", threads, "-v", "3"])
            if rep == 0:
                first_db = ts_
            else:
                best_db = ts_ if best_db is None else min(best_db, ts_)
                loaded = loaded and "no sequence lookup on the host" in log_
        env.clear()
        env.update(env_plain)
        n4, bad4, _ = dbio.diff_dbs(os.path.join(w, "aln_stock"), os.path.join(w, "res_db2"))
        out["patched_search_fused_persisted_layout"] = {
            "wall_s": round(best_db, 2), "first_search_builds_and_saves_wall_s": round(first_db, 2), "queries_per_s": round((len(qoff) - 1) / best_db, 1),
            "layout_file_GB": round(os.path.getsize(os.path.join(w, "t.mmgpu")) / 1e9, 2), "loaded_without_host_lookup": loaded,
            "result_db_identical_to_stock_align_db": bad4 == 0, "entries_compared": n4,
            "what": "the same `mmseqs search` command with MMGPU_DB_FILE naming a persisted device layout of the target database (second and "
                    "third search; page cache warm): whole command incl. process start"}
        out["speedup_search_persisted_layout_vs_stock_prefilter_plus_align"] = round(stock_total / best_db, 2)
        # the same command attached to a resident mmgpu_server (mmseqs2_amd/server/, the counterpart of the reference's gpuserver:
        # targets, their masked copy and the k-mer index stay on the device between searches) through LD_PRELOAD=libmmgpu_client.so;
        # the first search fills the server, the next two are timed
        server = os.path.join(ROOT, "mmseqs2_amd", "lib", "mmgpu_server")
        client = os.path.join(ROOT, "mmseqs2_amd", "lib", "libmmgpu_client.so")
        if os.path.exists(server) and os.path.exists(client):
            import signal
            sock = os.path.join(w, "mmgpu.sock")
            srv = subprocess.Popen([server, "--socket", sock], stderr=subprocess.DEVNULL)
            try:
                t0 = time.perf_counter()
                while not os.path.exists(sock) and srv.poll() is None and time.perf_counter() - t0 < 60:
                    time.sleep(0.05)
                env_direct = dict(env)
                env.update({"LD_PRELOAD": client, "MMGPU_SERVER_SOCKET": sock})
                best_srv = None
                for rep in range(3):
                    ts_, _ = run(patched, ["search", "q", "t", "res_srv%d" % rep, "tmp_srv%d" % rep, "-s", "5.7", "--threads", threads, "-v", "3"])
                    if rep > 0:
                        best_srv = ts_ if best_srv is None else min(best_srv, ts_)
                env.clear()
                env.update(env_direct)
                n3, bad3, _ = dbio.diff_dbs(os.path.join(w, "aln_stock"), os.path.join(w, "res_srv2"))
                out["patched_search_fused_resident_server"] = {"wall_s": round(best_srv, 2), "queries_per_s": round((len(qoff) - 1) / best_srv, 1),
                                                              "result_db_identical_to_stock_align_db": bad3 == 0, "entries_compared": n3,
                                                              "what": "the same `mmseqs search` command with LD_PRELOAD=libmmgpu_client.so against a running "
                                                                      "mmgpu_server that already holds the target database (second and third search; the "
                                                                      "first one uploads): whole command incl. process start"}
                out["speedup_search_resident_server_vs_stock_prefilter_plus_align"] = round(stock_total / best_srv, 2)
            finally:
                srv.send_signal(signal.SIGTERM)
                try:
                    srv.wait(timeout=30)
                except subprocess.TimeoutExpired:
                    srv.kill()
        if "stock_block_aligner_stubbed" in out:
            sb = out["stock_block_aligner_stubbed"]
            out["speedup_search_fused_vs_stubbed_stock_prefilter_plus_align"] = round((sb["prefilter_wall_s"] + sb["align_wall_s"]) / best, 2)
        out["speedup_prefilter_plus_align"] = round((out["stock"]["prefilter_wall_s"] + out["stock"]["align_wall_s"]) /
                                                    (out["patched"]["prefilter_wall_s"] + out["patched"]["align_wall_s"]), 2)
        return out
    finally:
        shutil.rmtree(w, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------------
def choose_query_groups(n_ranks, requested=0):
    """G of the G x S layout (search_headline).  Model from the N = 1 stage times of the headline workload (ms per 10 000 queries x
    1 M targets, profiles/r05_bench_n1.json: 152.9 ms = k + e): the similar-k-mer stage k is per query, everything else e scales with the index
    entries / pairs a rank touches; a group's exchange costs x per step.  t(G, S) = k / G + e / (G * S) + x * (S > 1).  At least
    two target shards per group whenever N >= 2: the hit-list exchange over RCCL is the path BASELINE.json configs[3] names."""
    if requested > 0:
        if n_ranks % requested:
            raise SystemExit("--query-groups must divide --gpus")
        return requested
    k, e, x = 10.5, 142.4, 3.0
    best, best_t = 1, None
    for g in range(1, n_ranks + 1):
        if n_ranks % g:
            continue
        s_ = n_ranks // g
        if n_ranks >= 2 and s_ < 2:
            continue
        t = k / g + e / (g * s_) + (x if s_ > 1 else 0.0)
        if best_t is None or t < best_t - 1e-9:
            best, best_t = g, t
    return best


def search_headline(args, gpu, torch, dist, rank, world, matrices, barrier, device_index=0):
    """BASELINE.json configs[2] / configs[3]: the timed steps and everything derived from them."""
    from mmseqs2_amd import capi, evalue, workloads as wl
    from mmseqs2_amd import distributed as D
    km16 = matrices["vtml80_kmer"].astype(np.int16)
    mat = matrices["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    sens, k, max_res = 5.7, 6, 300
    kmer_thr = int(163.2 - 8.917 * sens)          # Prefiltering::getKmerThreshold, Prefiltering.cpp:1080-1095
    sharded = dist is not None                    # world > 1, or the single-rank self-test of the exchange path
    weak = args.scaling == "weak" and world > 1
    t0 = time.time()
    (qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(args.pf_families, args.pf_members, args.pf_queries, seed=10,
                                                             target_seed=(11 + 1000 * rank) if weak else None)
    t_gen = time.time() - t0
    qs = wl.split(qres, qoff)
    nq_all = len(qs)
    # ---- layout of an N-rank run: G query groups x S target shards (G * S = N) --------------------------------------------
    # Sharding the targets divides everything that scales with index entries (gather / split, replay, scoring, alignment) by S,
    # but not the similar-k-mer stage, which is per query (19.5 of the 172 ms of the N = 1 step): with S = N = 8 it is half of
    # the step and the speed-up ends near 4x.  Dealing the QUERIES to G groups divides that stage as well; inside a group the
    # database is dealt to S = N / G ranks by length bucket and the hit lists are exchanged over the group's own communicator -
    # the exchange step of north_star, between fewer ranks.  Groups never talk to each other (different queries).
    G = choose_query_groups(world, args.query_groups) if (sharded and not weak) else 1
    S = world // G
    group_index, group_rank = rank // S, rank % S
    world_all, rank_all = world, rank
    pg = None
    if G > 1:
        groups = [dist.new_group(list(range(g * S, (g + 1) * S))) for g in range(G)]      # (every rank creates every group)
        pg = groups[group_index]
        lo, hi = group_index * nq_all // G, (group_index + 1) * nq_all // G
        qs = qs[lo:hi]
        qoff_g = (qoff[lo:hi + 1] - qoff[lo]).astype(qoff.dtype)
        qres, qoff = qres[int(qoff[lo]):int(qoff[hi])], qoff_g
        world, rank = S, group_rank      # from here on: this rank's group
    nq = len(qs)
    n_local_gen = len(toff) - 1
    t0 = time.time()
    s3, i3 = capi.host_score_matrix(km16, 3)
    shard = None
    if sharded and not weak:
        # strong scaling: the same database, dealt to the ranks by length bucket (mmgpu_host_partition_targets)
        shard = D.setup_shard(gpu, rank, world, tres, toff)
        n_global, db_residues = len(toff) - 1, float(toff[-1])
    elif sharded:
        # weak scaling: rank r holds targets [r * n, (r + 1) * n) of an N-fold database (its own family members)
        n_global = n_local_gen * world
        gids = (rank * n_local_gen + np.arange(n_local_gen)).astype(np.uint32)
        shard_of = (np.arange(n_global) // n_local_gen).astype(np.uint32)
        local_id = (np.arange(n_global) % n_local_gen).astype(np.uint32)
        gpu.load_targets(tres, toff, 21)
        gpu.pf_set_shard(world, rank, n_global, gids, shard_of, local_id)
        shard = dict(shard_of=shard_of, local_id=local_id, global_ids=gids, n_global=n_global)
        db_residues = float(toff[-1]) * world
    else:
        gpu.load_targets(tres, toff, 21)
        n_global, db_residues = len(toff) - 1, float(toff[-1])
    # the exchange steps run inside the library (RCCL communicator owned by it, include/mmgpu.h); torch.distributed carries the
    # 128-byte communicator id and the barriers only.  MMGPU_BENCH_COLLECTIVES=torch keeps round 2's torch.distributed collectives.
    lib_comm = False
    comm_note = "torch.distributed"
    if sharded and os.environ.get("MMGPU_BENCH_COLLECTIVES", "library") == "library":
        try:
            if dist.get_backend() == "nccl":
                idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    idt.copy_(torch.from_numpy(gpu.comm_unique_id()))
                dist.broadcast(idt, group_index * S, group=pg)      # the group's first rank made the id
                gpu.comm_init_rank(idt.cpu().numpy(), rank, world)
                lib_comm = True
                comm_note = "library: RCCL communicator owned by libmmgpu (mmgpu_comm_init_rank), %d ranks per communicator, %d query group(s)" % (world, G)
            elif world == 1:
                lib_comm = True         # a context without a communicator is its own single rank (device copies)
                comm_note = "library: single rank, device copies"
        except Exception as e:          # report, and keep the run alive on the torch.distributed path
            comm_note = "torch.distributed (library communicator failed: %s)" % str(e)[:200]
    # --mask 1 (the default of `mmseqs prefilter`): tantan masking of the resident targets for the prefilter (the masking step of
    # IndexBuilder::fillDatabase, on the device); the k-mer index is then built in HBM from the masked view, the alignment kernels
    # keep reading the unmasked residues
    t_mask, n_masked = 0.0, 0
    if args.mask:
        tv = np.load(os.path.join(ROOT, "tests", "golden", "tantan_vectors.npz"))      # constants: VTML80 likelihood ratios, 0.9f
        gpu.synchronize()
        tm0 = time.time()
        n_masked = gpu.pf_mask_targets(tv["vtml80_likelihood_ratios"], float(tv["mask_prob"]), 20)
        t_mask = time.time() - tm0
    gpu.pf_build_index(k, 21, True, s3, i3, km16, kmer_thr, matrices["blosum62_ungapped"])
    gpu.synchronize()
    t_index = time.time() - t0
    # Strong scaling: every rank also holds the WHOLE database (3 GB at this size) in a second context of its device - the queries a
    # step flags "inexact" (a shard reached its share of the reference's databaseHits buffer: ~0.1 % of the queries) run once more
    # against it inside the step (mmgpu_pf_exchange_redo_unsplit), so that the N-rank step answers every query, like
    # Prefiltering::mergeTargetSplits does (Prefiltering.cpp:412-526)
    gpu_full, t_full = None, 0.0
    if sharded and not weak and world > 1 and os.environ.get("MMGPU_BENCH_NO_UNSPLIT_CONTEXT") is None:
        import mmseqs2_amd
        tf0 = time.time()
        gpu_full = mmseqs2_amd.MMGpu(device_index)
        gpu_full.load_targets(tres, toff, 21)
        if args.mask:
            gpu_full.pf_mask_targets(tv["vtml80_likelihood_ratios"], float(tv["mask_prob"]), 20)
        gpu_full.pf_build_index(k, 21, True, s3, i3, km16, kmer_thr, matrices["blosum62_ungapped"])
        gpu_full.synchronize()
        t_full = time.time() - tf0
    cbs = [capi.host_comp_bias(km16, matrices["vtml80_pback"], q)[0] for q in qs]
    queries = [dict(q=q, comp_bias=cb, identity_id=None) for q, cb in zip(qs, cbs)]
    pfb = gpu.pf_prepare(queries, kmer_thr, max_hits=max_res, min_diag_score=15, ref_bins=2)
    stride = pfb.max_hits
    # start positions (reverse scan) only for the pairs that pass -e 1e-3: the smallest raw score whose E-value (ALP
    # parameters of BLOSUM62 11/1, mmseqs2_amd/evalue.py == EvalueComputation.h:37-41) passes, per query length
    thr_of_len = {}
    swq = []
    for q in qs:
        L = len(q)
        if L not in thr_of_len:
            thr_of_len[L] = evalue.min_score_for_evalue(1e-3, L, db_residues)
        swq.append(dict(q=q, comp_bias=capi.host_comp_bias(sub16, matrices["blosum62_pback"], q)[1], min_start_score=thr_of_len[L]))
    msh = gpu.sw_marshal_queries(mat, 11, 1, swq)
    dev = torch.device("cuda", torch.cuda.current_device())
    res_t = torch.zeros((nq, stride, 6), dtype=torch.int32, device=dev) if sharded else None
    stat = {"pf_ms": [], "align_ms": [], "cells": 0, "pairs": 0}
    keep = {}

    def step(record):
        pfb.run()
        if not sharded:
            fb = gpu.sw_prepare_from_pf(mat, 11, 1, None, pfb, mode=1, marshalled=msh)
            fb.run()
            a_ms = fb.kernel_ms()            # synchronises: the step is complete here
            if record:
                stat["pf_ms"].append(pfb.stage_ms()[6])
                stat["align_ms"].append(a_ms)
                stat["cells"], stat["pairs"] = fb.cells, fb.pairs
            if record == "keep":
                keep["fused"] = fb.fetch().reshape(nq, stride)
            fb.free()
            return
        if lib_comm:
            # prefilter of the shard -> all-gather + merge -> owned pairs -> gather: all enqueued by the library on its stream
            dh, dc, df, _ = pfb.exchange_merge()
            if gpu_full is not None:
                rd = pfb.redo_unsplit(gpu_full)      # reads the merged flags back; re-runs what they name, replaces those lists
                if record:
                    stat["redone"], stat["redone_left"] = rd
            b = gpu.sw_prepare_owned(mat, 11, 1, None, pfb, mode=1, marshalled=msh)
            b.run()
            b.gather_owned()
            a_ms = b.kernel_ms()          # synchronises on the alignment kernels' event
            gpu.synchronize()             # ... and on the gather behind them: the step is complete here
            if record:
                stat["pf_ms"].append(pfb.stage_ms()[6])
                stat["align_ms"].append(a_ms)
                stat["cells"], stat["pairs"] = b.cells, b.pairs
            if record == "keep":
                import ctypes
                hip = ctypes.CDLL("libamdhip64.so")
                mh, mc, mf = np.zeros((nq, stride, 3), np.int32), np.zeros(nq, np.int32), np.zeros(nq, np.int32)
                for dst, src in ((mh, dh), (mc, dc), (mf, df)):
                    assert hip.hipMemcpy(dst.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(src), dst.nbytes, 2) == 0      # device -> host
                keep["merged"] = (mh, mc, mf)
                full, nrec = b.fetch_owned()
                keep["full"] = full.view(np.int32).reshape(nq, stride, 6)
                keep["records_gathered"] = nrec
            b.free()
            return
        mh_t, mc_t, mf_t = D.exchange_and_merge_device(gpu, pfb, nq, stride, group=pg)
        if gpu_full is not None:
            rd = D.rerun_flagged_unsplit(gpu_full, queries, kmer_thr, max_res, 15, 2, mh_t, mc_t, mf_t)
            if record:
                stat["redone"], stat["redone_left"] = rd
        b, lc, ls = D.align_owned_pairs(gpu, mat, 11, 1, msh, mh_t, mc_t, nq, stride, mode=1)
        b.run()
        b.fetch_device(res_t.data_ptr())
        a_ms = b.kernel_ms()
        full = D.gather_owned_results(res_t, lc, ls, nq, stride, group=pg)
        if record:
            stat["pf_ms"].append(pfb.stage_ms()[6])
            stat["align_ms"].append(a_ms)
            stat["cells"], stat["pairs"] = b.cells, b.pairs
        if record == "keep":
            keep["merged"] = (mh_t.cpu().numpy(), mc_t.cpu().numpy(), mf_t.cpu().numpy())
            keep["full"] = full.cpu().numpy()
        b.free()

    for _ in range(max(args.warmup, 1)):          # at least one untimed pass sizes the working buffers
        step(None)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = allreduce(torch, dist, [elapsed], "max")[0]
    align_ms = allreduce(torch, dist, [float(np.mean(stat["align_ms"]))], "max")[0]
    pf_ms = allreduce(torch, dist, [float(np.mean(stat["pf_ms"]))], "max")[0]
    cells, pairs = allreduce(torch, dist, [stat["cells"], stat["pairs"]])
    mem_free, mem_total = gpu.device_memory()     # after the timed steps: targets + index + every working buffer at its high-water mark
    step("keep")                                  # one more pass whose results are downloaded for the checks below

    # ---- the same batch with the semantics of `mmseqs search` (alignment mode 2, StripedSmithWaterman.cpp:857-882): start positions
    # of the hits of the uint8 pass from the reverse scan, those of the int16-range hits from the block aligner, the reverse scan only
    # for what it declines.  Timed beside the step above (whose definition `value` keeps): whole steps, wall clock, all on the device.
    search = {"align_ms": [], "block_s": [], "pf_ms": []}
    if not sharded:
        def step_search(record):
            pfb.run()
            fb = gpu.sw_prepare_from_pf(mat, 11, 1, None, pfb, mode=2, marshalled=msh)
            fb.run()
            a_ms = fb.kernel_ms()            # synchronises on the alignment kernels
            tb0 = time.perf_counter()
            sel = fb.block_starts()          # selection, block aligner, scatter, reverse scan of declined pairs: synchronises
            tb = time.perf_counter() - tb0
            if record:
                search["align_ms"].append(a_ms)
                search["block_s"].append(tb)
                search["pf_ms"].append(pfb.stage_ms()[6])
                search["selected"], search["declined"], search["too_large"] = sel
            if record == "keep":
                keep["search"] = fb.fetch().reshape(nq, stride)
            fb.free()
        step_search(None)
        gpu.synchronize()
        ts0 = time.perf_counter()
        for _ in range(args.steps):
            step_search(True)
        gpu.synchronize()
        search["elapsed_s"] = time.perf_counter() - ts0
        step_search("keep")

    # ---- per-stage counters of the prefilter (last pass) ----
    stage = np.array(pfb.stage_ms())
    pf_cells, pf_cands = pfb.last_cells()
    out = {"elapsed_s": elapsed, "align_ms": align_ms, "pf_ms": pf_ms, "cells": cells, "pairs": pairs, "nq": nq_all,
           "query_groups": G, "target_shards_per_group": S if sharded else 1,
           "n_global": n_global, "n_local": gpu.n_targets, "kmer_thr": kmer_thr, "max_res": max_res, "stage": stage,
           "pf_cells": pf_cells, "pf_cands": pf_cands, "t_gen": t_gen, "t_index": t_index, "weak": weak, "sharded": sharded,
           "thr_example": thr_of_len.get(len(qs[0])), "hbm_in_use_gb": round((mem_total - mem_free) / 2 ** 30, 1),
           "mask": int(args.mask), "t_mask": t_mask, "n_masked": int(n_masked),
           "collectives": comm_note if sharded else None, "search": search if not sharded else None,
           "rerun_unsplit": ({"queries_per_step": int(stat.get("redone", 0)), "of_those_left_to_the_host": int(stat.get("redone_left", 0)),
                              "unsplit_context_setup_s": round(t_full, 2),
                              "what": "queries whose merged list a step flags inexact, re-run inside the step against the whole database in a "
                                      "second context of the rank's device (mmgpu_pf_exchange_redo_unsplit)"} if gpu_full is not None else None)}
    if gpu_full is not None:
        gpu_full.close()
    if rank_all != 0:
        pfb.free()
        return out

    # ---- everything below: rank 0, outside the timed region ----
    tlen = (toff[1:] - toff[:-1]).astype(np.int64)
    # without the lists on the host (N > 1): sum of target lengths ~ cells / mean query length
    out["alg_bytes"] = 28.0 * pairs + cells / max(float(np.mean([len(q) for q in qs])), 1.0) + 2.0 * float(qoff[-1])
    if not sharded:
        hits, counts, status, stats = pfb.fetch()
        out["ent"] = int(stats["db_matches"].sum())
        out["sim"] = int(stats["kmer_list_len"].sum())
        out["ovf"] = int((status != 0).sum())
        out["nhits"] = int(counts.sum())
        lists = [hits[qi]["id"][:counts[qi]].copy() for qi in range(nq)]
        full_lists = [(hits[qi]["id"][:counts[qi]].copy(), hits[qi]["score"][:counts[qi]].copy(), hits[qi]["diagonal"][:counts[qi]].copy())
                      for qi in range(nq)]
        fused = keep["fused"]
        gpu_res = [fused[qi, :counts[qi]] for qi in range(nq)]
        out["alg_bytes"] = float(sum(int(tlen[l].sum()) + 28 * len(l) for l in lists)) + 2.0 * float(qoff[-1])
        # the two-call path a patched Alignment::run takes (lists through the host): timed end to end, and its results
        # must equal the fused path's slot by slot
        t0 = time.perf_counter()
        host_q = [dict(q=x["q"], comp_bias=x["comp_bias"], targets=lists[i], min_start_score=x["min_start_score"]) for i, x in enumerate(swq)]
        swb = gpu.sw_prepare(mat, 11, 1, host_q, mode=1)
        t_prep = time.perf_counter() - t0
        t0 = time.perf_counter()
        swb.run()
        sep = swb.fetch()
        t_run_fetch = time.perf_counter() - t0
        bad, off = 0, 0
        for qi in range(nq):
            n = len(lists[qi])
            a, bb = fused[qi, :n], sep[off:off + n]
            bad += int(sum(not np.array_equal(a[f], bb[f]) for f in ("score", "q_end", "t_end", "q_start", "t_start")))
            off += n
        # backtraces (Matcher::SCORE_COV_SEQID / -a) for the hit lists of the first 1000 queries
        bt_n = int(sum(len(x) for x in lists[:1000]))
        t0 = time.perf_counter()
        bt_info, _ = swb.traceback(np.arange(bt_n, dtype=np.uint32)) if bt_n else (np.zeros(0, capi.SW_BT_DTYPE), [])
        t_bt = time.perf_counter() - t0
        out["two_call"] = {"prepare_s_host_scheduling_and_upload": round(t_prep, 3), "run_and_fetch_s": round(t_run_fetch, 4),
                           "fields_differing_from_fused_path": bad,
                           "what": "mmgpu_sw_prepare (caller-supplied lists) + mmgpu_sw_run + mmgpu_sw_fetch, mode START"}
        t_bt_c = getattr(swb, "last_traceback_call_s", t_bt)
        n_cig = int((bt_info["status"] == 0).sum()) if bt_n else 0
        out["backtrace"] = {"pairs": bt_n, "with_cigar": n_cig, "s_c_abi_calls_incl_download": round(t_bt_c, 4),
                            "s_incl_python_binding": round(t_bt, 4), "pairs_per_s": round(bt_n / t_bt_c, 1) if t_bt_c > 0 else None,
                            "cigars_per_s": round(n_cig / t_bt_c, 1) if t_bt_c > 0 else None,
                            "what": "mmgpu_sw_traceback for every pair of the first 1000 queries' lists (pairs below the start-score "
                                    "threshold have no start position and are answered MMGPU_BT_NO_START)"}
        # a15: every int16-range pair (word == 1) of the workload that gets start positions, through the device's block aligner
        # (blocks up to the crate's 4096 rows: nothing is handed to the host); a sample of the answers is re-scored
        try:
            word_idx = np.nonzero((sep["word"] == 1) & (sep["q_start"] >= 0))[0].astype(np.uint32)
            t0 = time.perf_counter()
            blk, bstr = swb.block_backtrace(word_idx)
            t_blk = time.perf_counter() - t0
            tier1, tier2 = swb.block_tiers()
            pair_q = np.repeat(np.arange(nq), [len(x) for x in lists])
            pair_t = np.concatenate(lists) if lists else np.zeros(0, np.uint32)
            ok_idx = np.nonzero(blk["status"] == 0)[0]
            sample = ok_idx[::max(1, len(ok_idx) // 3000)]
            resc_ok = 0
            for k in sample:
                p = int(word_idx[k])
                qi, ti = int(pair_q[p]), int(pair_t[p])
                q, cbq = swq[qi]["q"], swq[qi]["comp_bias"]
                t = tres[int(toff[ti]):int(toff[ti + 1])]
                qp, tp, sc, prev = int(blk[k]["q_start"]), int(blk[k]["t_start"]), 0, "M"
                for ch in bstr[k]:
                    if ch == "M":
                        sc += int(mat[int(q[qp]), int(t[tp])]) + int(cbq[qp])
                        qp += 1
                        tp += 1
                    else:
                        sc -= 1 if prev == ch else 11
                        qp += ch == "I"
                        tp += ch == "D"
                    prev = ch
                resc_ok += int((sc, qp - 1, tp - 1) == (int(sep[p]["score"]), int(sep[p]["q_end"]), int(sep[p]["t_end"])))
            out["block_aligner"] = {"pairs": int(len(word_idx)), "device": int((blk["status"] == 0).sum()),
                                    "declined": int((blk["status"] == 1).sum()), "too_large": int((blk["status"] == 2).sum()),
                                    "first_tier_512_rows_lds": int(tier1), "second_tier_4096_rows": int(tier2),
                                    "rescored_sample": int(len(sample)), "rescored_equal": int(resc_ok), "s_c_abi_calls_incl_download": round(getattr(swb, "last_block_call_s", t_blk), 4),
                                    "s_incl_python_binding": round(t_blk, 4),
                                    "pinned_against": "the C restatement oracle/block_oracle.c, which passes the 23 sequence and 6 profile unit-test vectors parsed from "
                                                      "the Rust crate's own source (tests/golden/block_crate_vectors.json; the profile vectors in the upstream gap-open "
                                                      "form, DESIGN.md 4.7) and whose block lists the device equals step by step (mmgpu_sw_block_growth); NOT pinned against a Rust-linked build "
                                                      "(no cargo in this image: scripts/make_block_goldens.sh is the recipe)",
                                    "what": "mmgpu_sw_block_backtrace over every word == 1 pair of the hit lists that has a start position "
                                            "(score passes -e 1e-3); device = answered by block_kernel.hip, declined = 'Block alignment "
                                            "failed' (the reference falls back too), too_large = left to the host (must be 0); "
                                            "rescored_equal = sampled CIGARs whose path re-scores to the SW score and ends at (q_end, t_end)"}
            # the search-semantics step's records: the step's own (mode START) with the block aligner's start positions for the pairs
            # it answered - the choice MMGpuMatcher.cpp makes between the two sources on the host
            if "search" in keep:
                expect = sep.copy()
                okm = blk["status"] == 0
                expect["q_start"][word_idx[okm]] = blk["q_start"][okm]
                expect["t_start"][word_idx[okm]] = blk["t_start"][okm]
                sbad, soff = 0, 0
                for qi in range(nq):
                    n = len(lists[qi])
                    sbad += int(not np.array_equal(keep["search"][qi, :n], expect[soff:soff + n]))
                    soff += n
                out["search_parity"] = {"queries_compared": nq, "queries_with_a_differing_record": sbad, "pairs_compared": int(soff),
                                        "what": "records of the search-semantics step (MMGPU_SW_START_NOT_WORD + mmgpu_sw_block_starts) against the "
                                                "timed step's records (MMGPU_SW_START) with the start positions mmgpu_sw_block_backtrace gives for the "
                                                "int16-range pairs it answers: all six fields of every pair"}
        except Exception as e:
            out["block_aligner"] = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
        swb.free()
        started = int(sum(int((g["q_start"] >= 0).sum()) for g in gpu_res))
        out["pairs_with_start"] = started
        # cells of the reverse scan (q[0..q_end] x t[0..t_end] of every pair that got start positions, StripedSmithWaterman.cpp:1143-1175)
        out["reverse_cells"] = int(sum(int(((g["q_end"].astype(np.int64) + 1) * (g["t_end"].astype(np.int64) + 1))[g["q_start"] >= 0].sum())
                                       for g in gpu_res))
        try:
            out["end_to_end"] = end_to_end_pass(gpu, matrices, qres, qoff, kmer_thr, max_res, db_residues, fused, counts)
        except Exception as e:          # a secondary measurement must not take the headline line down with it
            out["end_to_end"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not (args.no_modules or args.headline_only):
            try:
                out["modules"] = module_seconds(args, qres, qoff, tres, toff)
            except Exception as e:
                out["modules"] = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
        if not args.no_cpu_baseline:
            out["cpu_sw"] = sw_cpu_baseline_lists(matrices, qres, qoff, lists, tres, toff, args.cpu_seconds, gpu_res)
            out["cpu_pf"] = prefilter_cpu_baseline(matrices, qres, qoff, tres, toff, kmer_thr, args.cpu_seconds, full_lists, mask=bool(args.mask))
    else:
        mh, mc, mf = keep["merged"]
        mh = mh.reshape(nq, stride * 3).view(capi.PF_HIT_DTYPE).reshape(nq, stride)
        out["nhits"] = int(mc.sum())
        out["inexact_queries"] = int((mf != 0).sum())
        full = np.ascontiguousarray(keep["full"]).reshape(-1).view(capi.SW_HIT_DTYPE).reshape(nq, stride)
        if "records_gathered" in keep:
            out["records_gathered"] = int(keep["records_gathered"])
        out["merged_lists_sorted"] = bool(all(np.all(np.diff(mh[qi]["score"][:mc[qi]].astype(np.int64)) <= 0) for qi in range(0, nq, 97)))
        out["aligned_slots_filled"] = int(sum(int((full[qi, :mc[qi]]["score"] > 0).sum()) for qi in range(nq)))
        if not weak and os.environ.get("MMGPU_BENCH_NO_UNSPLIT_CHECK") is None:
            # N ranks against ONE: rank 0 runs its group's queries once more against the whole database in a second context of its
            # device (outside the timed region) and compares hit lists and alignment records field by field; the digests go into the line
            try:
                out["parity_vs_unsplit"] = unsplit_parity(args, device_index, matrices, qs, queries, swq, tres, toff, kmer_thr, max_res, k, s3, i3,
                                                          km16, mat, mh, mc, mf, full, stride)
            except Exception as e:
                out["parity_vs_unsplit"] = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
    pfb.free()
    return out


def unsplit_parity(args, device_index, matrices, qs, queries, swq, tres, toff, kmer_thr, max_res, k, s3, i3, km16, mat, mh, mc, mf, full, stride):
    """The N-rank step's merged hit lists (global ids, scores, diagonals, order) and gathered alignment records against the same
    queries run UNSPLIT on one device: Prefiltering's own target-split mode shortens the lists, this design claims equality
     - here it is checked inside the bench run.  Queries whose merged list carries the inexact flag (an
    overflow-path element took part in a tie at the cut) are counted, not compared."""
    import zlib
    import mmseqs2_amd
    from mmseqs2_amd import capi
    g2 = mmseqs2_amd.MMGpu(device_index)
    try:
        g2.load_targets(tres, toff, 21)
        if args.mask:
            tv = np.load(os.path.join(ROOT, "tests", "golden", "tantan_vectors.npz"))
            g2.pf_mask_targets(tv["vtml80_likelihood_ratios"], float(tv["mask_prob"]), 20)
        g2.pf_build_index(k, 21, True, s3, i3, km16, kmer_thr, matrices["blosum62_ungapped"])
        pfb = g2.pf_prepare(queries, kmer_thr, max_hits=max_res, min_diag_score=15, ref_bins=2)
        pfb.run()
        msh = g2.sw_marshal_queries(mat, 11, 1, swq)
        fb = g2.sw_prepare_from_pf(mat, 11, 1, None, pfb, mode=1, marshalled=msh)
        fb.run()
        nq = len(qs)
        one = fb.fetch().reshape(nq, pfb.max_hits)
        hits, counts, status, _ = pfb.fetch()
        fb.free()
        pfb.free()
    finally:
        g2.close()
    crc_n = crc_1 = 0
    lists_diff = rec_diff = skipped = 0
    for qi in range(nq):
        if mf[qi] != 0 or status[qi] != 0:
            skipped += 1
            continue
        n1, nn = int(counts[qi]), int(mc[qi])
        a = np.ascontiguousarray(mh[qi][:nn])
        b = np.ascontiguousarray(hits[qi][:n1])
        same = n1 == nn and all(np.array_equal(a[f], b[f]) for f in ("id", "score", "diagonal"))
        lists_diff += not same
        ra = np.ascontiguousarray(full[qi, :nn])
        rb = np.ascontiguousarray(one[qi, :n1])
        fields = ("score", "q_end", "t_end", "q_start", "t_start")
        rsame = n1 == nn and all(np.array_equal(ra[f], rb[f]) for f in fields)
        rec_diff += not rsame
        for f in ("id", "score", "diagonal"):
            crc_n = zlib.crc32(np.ascontiguousarray(a[f]).tobytes(), crc_n)
            crc_1 = zlib.crc32(np.ascontiguousarray(b[f]).tobytes(), crc_1)
        for f in fields:
            crc_n = zlib.crc32(np.ascontiguousarray(ra[f]).tobytes(), crc_n)
            crc_1 = zlib.crc32(np.ascontiguousarray(rb[f]).tobytes(), crc_1)
    return {"queries_compared": int(nq - skipped), "queries_flagged_inexact_or_handed_back": int(skipped),
            "hit_lists_differing": int(lists_diff), "alignment_record_lists_differing": int(rec_diff),
            "digest_n_ranks": int(crc_n), "digest_unsplit": int(crc_1), "equal": bool(lists_diff == 0 and rec_diff == 0 and crc_n == crc_1),
            "what": "rank 0: merged hit lists (id, score, diagonal, order) and gathered alignment records (score, ends, starts) of its query "
                    "group from the N-rank step vs the same queries against the unsplit database in a second context of the same device"}


def align_only_section(args, gpu, torch, matrices, rank):
    """BASELINE.json configs[1]: 1000 x 100 000 all-vs-all alignment (score + end positions), kernel-only rate."""
    from mmseqs2_amd import workloads as wl
    from mmseqs2_amd.capi import host_comp_bias
    mat = matrices["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    (qres, qoff), (tres, toff) = wl.config2_align_only(args.queries, args.targets, seed=1 + 1000 * rank)
    qs = wl.split(qres, qoff)
    cbs = [host_comp_bias(sub16, matrices["blosum62_pback"], q)[1] for q in qs]
    gpu.load_targets(tres, toff, 21)
    ids = np.arange(args.targets, dtype=np.uint32)
    queries = [dict(q=q, comp_bias=cb, targets=ids, min_start_score=0) for q, cb in zip(qs, cbs)]
    t0 = time.time()
    batch = gpu.sw_prepare(mat, 11, 1, queries, mode=0)
    prep_s = time.time() - t0
    batch.run()
    gpu.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.align_steps):
        batch.run()
    gpu.synchronize()
    dt = (time.perf_counter() - t0) / args.align_steps
    k_ms = batch.kernel_ms_mean(args.align_steps)[0]
    if rank != 0:
        batch.free()
        return None
    t0 = time.perf_counter()
    res = batch.fetch().reshape(args.queries, args.targets)
    t_fetch = time.perf_counter() - t0
    lane_ops = batch.cells / 2.0 * 10.0
    out = {"workload": "BASELINE.json configs[1]: align-only, %d random L~N(350,35) queries x %d targets (10%% planted homologs), "
                       "all-vs-all lists, BLOSUM62 gap 11/1, comp-bias on, score + end positions" % (args.queries, args.targets),
           "gcups_kernels_only": round(batch.cells / dt / 1e9, 1), "ms_per_pass": round(dt * 1e3, 2), "kernel_ms": round(k_ms, 2),
           "pairs": int(batch.pairs), "cells": int(batch.cells),
           "end_to_end_s": {"prepare_host_scheduling_and_upload": round(prep_s, 2), "run": round(dt, 3), "fetch_2.4GB_of_results": round(t_fetch, 3),
                            "gcups_incl_prepare_and_fetch": round(batch.cells / (prep_s + dt + t_fetch) / 1e9, 1)},
           "valu_roofline_frac": round(lane_ops / (k_ms * 1e-3) / VALU_LANE_OPS_PER_S, 4),
           "score_checksum": int(res["score"].astype(np.int64).sum())}
    batch.free()
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = align_only_cpu_baseline(args, matrices, qres, qoff, tres, toff, res)
    return out


def align_only_cpu_baseline(args, matrices, qres, qoff, tres, toff, res):
    """configs[1] says "bit-exact scores vs CPU": the reference's own ssw_init + ssw_align (score + end positions, alignment mode
    SCORE_ONLY: StripedSmithWaterman.cpp:892-941) through oracle/_ref/libmmref.so on a bounded sample of the same all-vs-all
    lists (every sample query against an evenly spaced subset of the targets), timed on the host cores and compared with the
    device's records of the timed run afterwards."""
    from oracle import pyoracle
    if not pyoracle.ref_available():
        return None
    hw_threads, phys_cores = host_cpu_topology()
    quota = cpu_quota_cores()
    threads = hw_threads if quota is None else int(max(1, min(hw_threads, round(quota))))
    n_q = min(args.queries, args.align_sample_queries)
    n_t = min(args.targets, args.align_sample_targets)
    ids = np.linspace(0, args.targets - 1, n_t).astype(np.uint32)
    l_ids = np.tile(ids, n_q)
    l_off = (np.arange(n_q + 1, dtype=np.uint64) * np.uint64(n_t))
    ser = matrices["blosum62_serialized"]
    qlen = (qoff[1:] - qoff[:-1]).astype(np.int64)
    tlen = (toff[1:] - toff[:-1]).astype(np.int64)
    cells = float(qlen[:n_q].sum()) * float(tlen[ids].sum())
    sec, used, ref = pyoracle.ref_sw_lists_omp(ser, qres, qoff, l_ids, l_off, tres, toff, 0, n_q, threads, mode=0,
                                               db_residues=int(toff[-1]), want_starts=False)
    g = res[:n_q][:, ids].reshape(-1)
    sc = ref["score"].astype(np.int64)
    pos = sc > 0
    bad_score = int(np.count_nonzero(g["score"].astype(np.int64) != sc))
    bad_end = int(np.count_nonzero(((g["t_end"] != ref["t_end"]) | (g["q_end"] != ref["q_end"])) & pos))
    return {"value": round(cells / sec / 1e9, 2), "unit": "GCUPS", "cores": threads if quota is not None else phys_cores, "threads": used,
            "kind": "reference",
            "what": "the reference's ssw_init + ssw_align (AVX2 striped uint8 pass + int16 re-run, score and end positions) in one "
                    "native OpenMP call (oracle/ref_shim.cpp mmref_sw_lists_omp)",
            "sample": "queries [0, %d) x %d evenly spaced targets of the same workload (%d pairs, %.4g cells), %.2f s wall"
                      % (n_q, n_t, n_q * n_t, cells, sec),
            "parity_vs_reference": {"pairs_compared": int(n_q * n_t), "score_mismatches": bad_score,
                                    "end_position_mismatches_among_positive_scores": bad_end, "fields": "score, q_end, t_end"}}


def nucl_section(args, gpu, matrices, rank):
    """BASELINE.json configs[4], the nucleotide alignment step (BandedNucleotideAligner::align behind Alignment::run):
    reads with 10 % substitutions / 2 % indels against their source contigs (true prefilter diagonal, both strands)
    plus unrelated contigs.  The lists of this section are synthetic (the nucleotide k-mer prefilter runs in the drop-in
    tests, tests/test_mmseqs_dropin.py, not here)."""
    from mmseqs2_amd import workloads as wl
    t0 = time.time()
    queries, (tres, toff), pairs = wl.config5_nucleotide(args.nucl_contigs, args.nucl_reads, args.nucl_read_len, seed=20 + 1000 * rank)
    t_gen = time.time() - t0
    gpu.load_targets(tres, toff, 5)
    mat, rl = matrices["nucleotide"], matrices["nucleotide_reverse"]
    gpu.nucl_align(mat, rl, queries[:8], pairs[:8])            # warm-up (code load, block cache)
    gpu.synchronize()
    t0 = time.perf_counter()
    hits, strs = gpu.nucl_align(mat, rl, queries, pairs, 5, 2, 40, 4, 4)
    dt_py = time.perf_counter() - t0
    dt = gpu.last_nucl_call_s            # the C-ABI call (upload, kernel, download), as the CPU baseline is timed inside its C call
    if rank != 0:
        return None
    aligned = int(hits["bt_len"].sum())
    res = {"workload": "BASELINE.json configs[4] (alignment step only): %d reads of %d nt (10%% substitutions, 2%% indels) x "
                       "(source contig on the true diagonal + %d unrelated contigs), %d contigs ~LogNormal(20 kb), gap 5/2, "
                       "band 64, z-drop 40" % (len(queries), args.nucl_read_len, 4, args.nucl_contigs),
           "pairs": len(pairs), "pairs_per_s": round(len(pairs) / dt, 1), "s_incl_upload_and_download": round(dt, 4),
           "s_incl_python_binding": round(dt_py, 4),
           "aligned_columns": aligned, "aligned_columns_per_s": round(aligned / dt, 1),
           "true_pairs_recovered": int(sum(1 for i in range(0, len(pairs), 5) if hits[i]["bt_len"] > 0.8 * args.nucl_read_len)),
           "setup_s": {"generate": round(t_gen, 1)}}
    if not args.no_cpu_baseline:
        from oracle import pyoracle
        if pyoracle.ref_available():
            cores = os.cpu_count() or 1
            ref = pyoracle.RefNucl(serialized=matrices["nucleotide_serialized"].tobytes())
            letters = np.frombuffer(wl.NUCL_LETTERS.encode(), np.uint8)
            qoff = np.concatenate([[0], np.cumsum([len(q) for q in queries])]).astype(np.uint64)
            pa = np.array(pairs, np.int64)
            sec, out, bl = ref.batch(letters[np.concatenate(queries)], qoff, letters[tres], toff, pa[:, 0], pa[:, 1], pa[:, 2],
                                     pa[:, 3], cores)
            ref.close()
            got = np.stack([hits[f].astype(np.int64) for f in ("score", "q_start", "q_end", "t_start", "t_end", "ident")], 1)
            bad = int((np.any(got != out, axis=1) | (hits["bt_len"] != bl)).sum())
            res["cpu_baseline"] = {"value": round(len(pairs) / sec, 1), "unit": "pairs/s", "cores": cores, "kind": "reference",
                                   "sample": "all %d pairs, BandedNucleotideAligner::align on %d threads, %.2f s wall" % (len(pairs), cores, sec),
                                   "parity_vs_reference": {"pairs_compared": len(pairs), "pairs_differing": bad,
                                                           "fields": "score, start / end positions, identities, backtrace length"}}
    return res


def nucl_search_section(args, gpu, matrices, n_contigs, with_reference):
    """BASELINE.json configs[4] as a SEARCH: reads and their reverse complements (the workflow's extractframes
    step) against contigs cut at 10 000 nt (its splitsequence step, Search.cpp:194-198): exact 15-mer prefilter with the
    isNucleotide branch on the device -> the hit lists' diagonals -> the nucleotide alignment kernel.  CPU side: the real
    QueryMatcher on a sample of the reads, one thread, which is also the parity check of the hit lists."""
    from mmseqs2_amd import capi, workloads as wl
    t0 = time.time()
    queries, (tres, toff), _ = wl.config5_nucleotide(n_contigs, args.nucl_reads, args.nucl_read_len, seed=20)
    # splitsequence: pieces of at most 10 000 nt
    cut = 10000
    pieces, src = [], []
    for c in range(len(toff) - 1):
        a, e = int(toff[c]), int(toff[c + 1])
        for p in range(a, e, cut):
            pieces.append(tres[p:min(p + cut, e)])
            src.append(c)
    pres, poff = wl.seqs_from_list(pieces)
    both = []
    for q in queries:
        both.append(q)
        both.append(wl.NUCL_REVERSE[q[::-1]])
    t_gen = time.time() - t0
    mat = matrices["nucleotide"].astype(np.int8).reshape(5, 5)
    rl = matrices["nucleotide_reverse"]
    t0 = time.perf_counter()
    gpu.load_targets(pres, poff, 5)
    gpu.pf_build_index(15, 5, True, None, None, mat.astype(np.int16), 0, mat)
    gpu.synchronize()
    t_index = time.perf_counter() - t0
    pq = [dict(q=q, comp_bias=None, identity_id=None) for q in both]
    b = gpu.pf_prepare(pq, 0, max_hits=300, min_diag_score=15, ref_bins=2, exact=True, nucleotide=True)
    b.run()                      # warm-up
    gpu.synchronize()
    t0 = time.perf_counter()
    b.run()
    gpu.synchronize()
    t_pf = time.perf_counter() - t0
    hits, counts, status, _ = b.fetch()
    b.free()
    pairs = np.zeros(int(counts.sum()), capi.NUCL_PAIR_DTYPE)
    k = 0
    for qi in range(len(both)):
        n = int(counts[qi])
        pairs["query"][k:k + n] = qi
        pairs["target"][k:k + n] = hits[qi]["id"][:n]
        pairs["diagonal"][k:k + n] = hits[qi]["diagonal"][:n]
        k += n
    gpu.nucl_align(mat, rl, both[:4], pairs[:4] if len(pairs) >= 4 else pairs)
    al, _ = gpu.nucl_align(mat, rl, both, pairs, 5, 2, 40, 4, 4)
    t_al = gpu.last_nucl_call_s
    found = 0
    for qi in range(len(both)):
        # (a 10 kb read usually straddles two 10 kb pieces of its contig: count reads with a long alignment on either strand)
        if counts[qi] and al["bt_len"][pairs["query"] == qi].max(initial=0) > 0.25 * args.nucl_read_len:
            found += 1
    res = {"workload": "BASELINE.json configs[4] as a search: %d reads of %d nt + their reverse complements x %d contig pieces (<= 10 000 nt, "
                       "%d contigs ~LogNormal(20 kb)); exact 15-mer prefilter (spaced, --max-seqs 300, isNucleotide branch) -> banded "
                       "nucleotide alignment of every hit" % (len(queries), args.nucl_read_len, len(pieces), n_contigs),
           "reads": len(queries), "query_entries": len(both), "prefilter_s": round(t_pf, 4),
           "prefilter_entries_per_s": round(len(both) / t_pf, 1), "prefilter_hits": int(counts.sum()),
           "queries_handed_to_host": int((status != 0).sum()),
           "align_pairs": len(pairs), "align_s_incl_upload_and_download": round(t_al, 4),
           "reads_per_s_prefilter_plus_alignment": round(len(queries) / (t_pf + t_al), 1),
           "query_entries_with_an_alignment_over_a_quarter_of_the_read": found,
           "setup_s": {"generate_and_split": round(t_gen, 1), "upload_and_device_index_build_4^15_offsets": round(t_index, 2)}}
    if with_reference and not args.no_cpu_baseline:
        from oracle import pyoracle
        if pyoracle.ref_available():
            t0 = time.perf_counter()
            ref = pyoracle.RefNuclPrefilter(15, True, serialized=matrices["nucleotide_serialized"].tobytes())
            ref.build_index(pres, poff)
            t_ref_index = time.perf_counter() - t0
            n_s = min(len(both), 1000)
            bad = 0
            t0 = time.perf_counter()
            refl = [ref.match(both[qi], max_hits=300, force_bins=2, max_seq_len=args.nucl_read_len + 64) for qi in range(n_s)]
            sec = time.perf_counter() - t0
            for qi in range(n_s):
                if status[qi] != 0:
                    continue
                n = int(counts[qi])
                r = refl[qi]
                if not (np.array_equal(r["id"], hits[qi]["id"][:n]) and np.array_equal(r["score"], hits[qi]["score"][:n]) and
                        np.array_equal(r["diagonal"], hits[qi]["diagonal"][:n])):
                    bad += 1
            res["cpu_baseline"] = {"value": round(n_s / sec, 1), "unit": "query entries/s (prefilter only)", "cores": 1, "kind": "reference",
                                   "sample": "first %d query entries through QueryMatcher::matchQuery(isNucleotide), one thread, %.2f s; "
                                             "reference index build %.1f s (not counted)" % (n_s, sec, t_ref_index),
                                   "parity_vs_reference": {"queries_compared": n_s, "queries_with_different_hit_lists": bad}}
    return res


def translated_search_section(args):
    """BASELINE.json configs[4] as its text says - a 6-frame TRANSLATED search of 10 kb reads against contigs - through the binaries:
    `mmseqs search reads contigs res tmp --search-type 2` (data/workflow/translated_search.sh: extractorfs of both sides ->
    translatenucs -> prefilter -> align -> offsetalignment).  The frames are cut and translated by the reference's own modules on the
    host; `prefilter` and `align` of the translated ORFs run on the device through the patched binary (amino-acid kernels).  CPU
    baseline and parity: the stock binary on a sample of the reads and contigs (its result database must equal the patched
    binary's entry by entry); the full size is timed through the patched binary only."""
    import shutil
    import subprocess
    import tempfile
    from mmseqs2_amd import workloads as wl, dbio
    stock = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock")
    patched = os.path.join(ROOT, "oracle", "_ref", "mmseqs_mmgpu")
    if not (os.path.exists(stock) and os.path.exists(patched)):
        return None
    quota = cpu_quota_cores()
    hw = os.cpu_count() or 1
    threads = str(hw if quota is None else int(max(1, min(hw, round(2 * quota)))))
    s_reads, s_contigs = (int(x) for x in args.translated_sample.split("x"))
    w = tempfile.mkdtemp(prefix="mmgpu_translated_")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "mmseqs2_amd", "lib") + ":" + env.get("LD_LIBRARY_PATH", "")

    def run(b, a):
        t0 = time.perf_counter()
        r = subprocess.run([b] + a, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=args.module_timeout)
        if r.returncode != 0:
            raise RuntimeError("%s %s failed: %s" % (os.path.basename(b), a[0], r.stdout[-400:]))
        return time.perf_counter() - t0, r.stdout

    try:
        t0 = time.perf_counter()
        queries, (tres, toff), _ = wl.config5_nucleotide(args.translated_contigs, args.nucl_reads, args.nucl_read_len, seed=20)
        contigs = wl.split(tres, toff)
        wl.write_nucl_fasta(os.path.join(w, "contigs.fasta"), contigs, "c")
        wl.write_nucl_fasta(os.path.join(w, "reads.fasta"), queries, "r")
        # the sample: the first reads whose source contig lies among the first s_contigs, and those contigs
        wl.write_nucl_fasta(os.path.join(w, "s_contigs.fasta"), contigs[:s_contigs], "c")
        wl.write_nucl_fasta(os.path.join(w, "s_reads.fasta"), queries[:s_reads], "r")
        for n in ("contigs", "reads", "s_contigs", "s_reads"):
            run(stock, ["createdb", n + ".fasta", n, "-v", "1"])
        t_setup = time.perf_counter() - t0
        flags = ["--search-type", "2", "--threads", threads, "-v", "3"]
        t_full, log = run(patched, ["search", "reads", "contigs", "res_full", "tmp_full"] + flags)
        orfs = [int(l.split()[3]) for l in log.splitlines() if l.startswith(("Query database size:", "Target database size:"))][:2]
        whole = s_reads >= len(queries) and s_contigs >= len(contigs)      # (the sample is the workload: the run above is the patched side)
        t_sp = t_full if whole else run(patched, ["search", "s_reads", "s_contigs", "res_sp", "tmp_sp"] + flags)[0]
        t_ss, _ = run(stock, ["search", "s_reads", "s_contigs", "res_ss", "tmp_ss"] + flags)
        # (the stock binary's own runs of this command differ from each other in the ORDER of lines whose bit score and E-value tie -
        # `offsetalignment` merges the ORF hits of a read in a thread-dependent order - in 1 to 4 of these 100 entries at 32 threads;
        # entries are therefore compared up to the order inside such runs of tied lines, and how many needed that is reported)
        n, bad, tied, _ = dbio.diff_dbs_up_to_tie_order(os.path.join(w, "res_ss"), os.path.join(w, "res_full" if whole else "res_sp"))
        return {"workload": "BASELINE.json configs[4] as a translated search: `mmseqs search reads contigs --search-type 2` (default flags), %d reads "
                            "of %d nt x %d contigs (~LogNormal(20 kb), %d nt); --threads %s"
                            % (len(queries), args.nucl_read_len, args.translated_contigs, int(toff[-1]), threads),
                "patched_wall_s": round(t_full, 2), "reads_per_s": round(len(queries) / t_full, 1),
                "translated_orfs_query_target": orfs, "modules_on_device": log.count("MMGPU: device"),
                "cpu_path_messages": len([l for l in log.splitlines() if "using the CPU path" in l]),
                "setup_s": {"generate_fasta_createdb": round(t_setup, 1)},
                "cpu_baseline": {"value": round(s_reads / t_ss, 1), "unit": "reads/s", "cores": int(threads), "kind": "reference",
                                 "sample": "the stock binary, same command, on the first %d reads x the first %d contigs: %.2f s wall "
                                           "(patched binary on the same sample: %.2f s)" % (s_reads, s_contigs, t_ss, t_sp),
                                 "stock_wall_s": round(t_ss, 2), "patched_wall_s_same_sample": round(t_sp, 2),
                                 "parity_vs_reference": {"result_entries_compared": n, "entries_differing": bad,
                                                         "entries_equal_up_to_the_order_of_tied_lines": tied,
                                                         "note": "lines of an entry whose bit score and E-value tie are written in a thread-dependent "
                                                                 "order by the reference itself (two runs of the stock binary differ the same way)"}},
                "timeline": "profiles/%s_translated_search_timeline.txt (MMGPU_TRACE laps of this command, scripts/search_timeline.py --translated): "
                            "the align module is 7.3 s of 11.9 s, the device busy for ~6 ms of each ~490 ms bucket of 16 384 ORFs; the three largest "
                            "host-side items are the reference's own accept / sort / write loop per bucket (~0.13 s x 14), the parse of a bucket's "
                            "prefilter lists (~0.08 s x 14) and the fixed costs around the per-bucket device calls (result download + job lists of the "
                            "block aligner ~0.06 s, prepare + fetch of the alignment batch ~0.08 s; x 14)" % PROFILE_ROUND,
                "full_size": "50 000 contigs through the patched binary: profiles/r04_translated_search_50k.json (scripts/exp_translated_search.py)"}
    finally:
        shutil.rmtree(w, ignore_errors=True)


def self_launch(n):
    """`python bench.py --gpus N` started as ONE process (the driver's form of the command; the reference starts its splits from one
    command too, Prefiltering.cpp:605-689 runMpiSplits): re-run the same command line under torch.distributed.run, one rank per
    GPU, rendezvous on 127.0.0.1 at a free port.  The ranks inherit this process's stdout: rank 0 prints the one JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def main():
    if "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        pre = argparse.ArgumentParser(add_help=False)
        pre.add_argument("--gpus", type=int, default=1)
        n = pre.parse_known_args()[0].gpus
        if n > 1:
            raise SystemExit(self_launch(n))
    # exactly ONE line on stdout: libraries (RCCL prints a version banner) write to fd 1 behind Python's back, so
    # fd 1 is pointed at stderr for the run and the JSON line goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the same 1M targets dealt to the ranks (BASELINE.json configs[3]); weak = 1M targets per rank")
    ap.add_argument("--query-groups", type=int, default=0,
                    help="N > 1: G groups of ranks, each with its own slice of the queries; inside a group the targets are dealt to "
                         "N / G ranks by length bucket and the hit lists exchanged over the group's RCCL communicator.  0 = chosen "
                         "from the N = 1 stage times (the similar-k-mer stage does not shrink with the target shard), keeping at least "
                         "two target shards per group")
    ap.add_argument("--pf-families", type=int, default=20000)
    ap.add_argument("--pf-members", type=int, default=50)
    ap.add_argument("--pf-queries", type=int, default=10000)
    ap.add_argument("--queries", type=int, default=1000, help="align_only section (configs[1])")
    ap.add_argument("--align-sample-queries", type=int, default=200, help="align_only: queries of the reference's CPU run (baseline + parity)")
    ap.add_argument("--align-sample-targets", type=int, default=5000, help="align_only: targets per sample query of the reference's CPU run")
    ap.add_argument("--targets", type=int, default=100000)
    ap.add_argument("--align-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-align-only", action="store_true", help="skip the configs[1] section")
    ap.add_argument("--no-nucl", action="store_true", help="skip the configs[4] nucleotide alignment section")
    ap.add_argument("--nucl-contigs", type=int, default=50000, help="configs[4]: 50 k contigs (~1.1e9 nt)")
    ap.add_argument("--nucl-parity-contigs", type=int, default=4000,
                    help="nucleotide search: size of the second, smaller run whose hit lists are compared with the reference's matcher "
                         "(the reference's 4^15-offset index build over 50 k contigs alone takes minutes)")
    ap.add_argument("--translated-contigs", type=int, default=5000, help="configs[4] as `search --search-type 2` through the binaries")
    ap.add_argument("--translated-sample", type=str, default="1000x5000", help="reads x contigs of the stock-binary run (CPU baseline + parity)")
    ap.add_argument("--no-translated", action="store_true")
    ap.add_argument("--nucl-reads", type=int, default=1000)
    ap.add_argument("--nucl-read-len", type=int, default=10000)
    ap.add_argument("--headline-only", action="store_true", help="counter passes (scripts/collect_profiles.sh): the timed steps only")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--mask", type=int, default=1, choices=[0, 1],
                    help="1 (mmseqs' default): the targets are tantan-masked for the prefilter on the device (mmgpu_pf_mask_targets), the alignment reads them unmasked")
    ap.add_argument("--no-modules", action="store_true", help="skip the stock-vs-patched `mmseqs prefilter` / `align` wall times")
    ap.add_argument("--module-timeout", type=float, default=240.0)
    args = ap.parse_args()

    import torch
    import mmseqs2_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start `python bench.py --gpus N` (it launches its own ranks) or "
                         "python -m torch.distributed.run --nproc-per-node N bench.py --gpus N" % (args.gpus, world))
    # MMGPU_BENCH_BACKEND=gloo lets several ranks share one GPU to exercise the N>1 code path on a 1-GPU box
    backend = os.environ.get("MMGPU_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > max(torch.cuda.device_count(), 0):
        raise SystemExit("--gpus %d with the RCCL backend needs %d visible devices (%d here); MMGPU_BENCH_BACKEND=gloo lets the ranks share "
                         "a device" % (world, world, torch.cuda.device_count()))
    device_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device_index)
    dist = None
    # MMGPU_BENCH_FORCE_EXCHANGE=1: run the N > 1 path (shard description, RCCL all-gathers, merge) with a single rank
    if world > 1 or os.environ.get("MMGPU_BENCH_FORCE_EXCHANGE") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    matrices = dict(np.load(os.path.join(ROOT, "tests", "golden", "matrices.npz")))
    gpu = mmseqs2_amd.MMGpu(device_index)
    stream = torch.cuda.current_stream()
    gpu.set_stream(stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    H = search_headline(args, gpu, torch, dist, rank, world, matrices, barrier, device_index)
    side = {}
    if not args.headline_only:
        if not args.no_align_only:
            try:
                side["align_only"] = align_only_section(args, gpu, torch, matrices, rank)
            except Exception as e:      # a secondary section must not take the headline line down with it: report, do not hide
                side["align_only"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_nucl:
            try:
                side["nucleotide_align"] = nucl_section(args, gpu, matrices, rank)
                if rank == 0:
                    try:
                        side["nucleotide_search"] = nucl_search_section(args, gpu, matrices, args.nucl_contigs, args.nucl_contigs <= args.nucl_parity_contigs)
                        if args.nucl_contigs > args.nucl_parity_contigs and not args.no_cpu_baseline:
                            side["nucleotide_search"]["parity_run"] = nucl_search_section(args, gpu, matrices, args.nucl_parity_contigs, True)
                    except Exception as e:      # a secondary section must not lose the line
                        side["nucleotide_search"] = {"error": repr(e)[:300]}
                    if not (args.no_translated or args.no_modules):
                        try:
                            side["translated_search"] = translated_search_section(args)
                        except Exception as e:
                            side["translated_search"] = {"error": repr(e)[-300:]}
            except Exception as e:
                side["nucleotide_align"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        nq = H["nq"]
        ms_per_step = H["elapsed_s"] / args.steps * 1e3
        k_ms = H["align_ms"]
        value = H["cells"] / (k_ms * 1e-3) / 1e9
        stage = H["stage"]
        default_wl = (args.pf_families, args.pf_members, args.pf_queries) == (20000, 50, 10000) and world == 1
        # algorithmic HBM bytes of the alignment kernels: tlen + 28 bytes per pair + the query once
        alg_bytes = H["alg_bytes"]
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        lane_ops = H["cells"] / 2.0 * 10.0       # 10 VOP3P lane-ops per pair of cells, forward scan only
        out = {
            "metric": "sw_gcells_per_s", "value": round(value, 1), "unit": "GCUPS",
            "queries_per_s": round(nq * args.steps / H["elapsed_s"], 1),
            "queries_per_s_is": "queries / whole step with queries, targets and index resident (kernels + hand-over); the host-in/host-out rate is queries_per_s_end_to_end",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak" if H["weak"] else "strong", "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]: %d queries x %d targets (%d families x %d members, L~LogNormal(5.45,0.6)), "
                                   "-s 5.7 (k=6 spaced, k-mer threshold %d), --max-seqs %d, --mask %d%s, then Gotoh SW of every hit list "
                                   "(BLOSUM62 11/1, comp-bias on): score + end positions, start positions for pairs passing -e 1e-3"
                                   % (3 if world > 1 else 2, nq, H["n_global"], args.pf_families, args.pf_members, H["kmer_thr"], H["max_res"], H["mask"],
                                      " (tantan on the device: %d residues masked in %.3f s, set-up)" % (H["n_masked"], H["t_mask"]) if H["mask"] else ""),
                       "step": "prefilter kernels -> device-side hand-over -> alignment kernels (fused path)" if not H["sharded"] else
                               "prefilter of the shard -> all-gather of exchange records -> merge (== unsplit result) -> alignment of owned pairs -> all-gather of results",
                       "value_is": "forward DP cells of the step / alignment-stage time of the step (HIP events); queries_per_s = queries / whole step",
                       "targets_per_gpu": int(H["n_local"]), "align_pairs_per_step": int(H["pairs"]), "align_cells_per_step": int(H["cells"]),
                       "parallelism": "single GPU" if world == 1 else
                                      "1 process/GPU, %d query group(s) x %d target shards: queries dealt to the groups, inside a group the targets dealt by "
                                      "length bucket and two RCCL all-gathers per step over the group's communicator" % (H.get("query_groups", 1), H.get("target_shards_per_group", world)),
                       "collectives": H.get("collectives")},
            "ms_per_step_stages": {"prefilter_kernels": round(H["pf_ms"], 2), "align_kernels": round(k_ms, 2),
                                   "handover_exchange_and_host": round(ms_per_step - H["pf_ms"] - k_ms, 2)},
            "roofline": {"kernel": "sw_kernel<G,true> (four concurrent grids: tile shapes grouped by register need, forward + reverse scan; + sw_rev_multi_kernel)",
                         "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": pmc_traffic(("sw_kernel<", "sw_rev_multi_kernel"), PROFILE_ROUND + "_search") if default_wl else None,
                         "traffic_source": "profiles/%s_search_pmc_{fetch,write}_size.txt: FETCH_SIZE + WRITE_SIZE (KB) per launch, separate --pmc passes; "
                                           "null unless those passes were taken with the device sources of this run (csrc_digest)" % PROFILE_ROUND,
                         "csrc_digest": csrc_digest(),
                         "algorithmic_bytes_per_launch": round(alg_bytes), "kernel_ms": round(k_ms, 3),
                         "note": "Gotoh SW is VALU-issue bound (0.003 B/cell, SURVEY.md section 8d): the binding roof is valu_roofline",
                         "valu_roofline": {"achieved_lane_ops_per_s": round(lane_ops / (k_ms * 1e-3), 1),
                                           "peak_lane_ops_per_s": VALU_LANE_OPS_PER_S,
                                           "frac": round(lane_ops / (k_ms * 1e-3) / VALU_LANE_OPS_PER_S, 4),
                                           "counts": "forward cells only; the reverse scan of the pairs passing -e 1e-3 runs inside the same kernels",
                                           "reverse_scan_cells": H.get("reverse_cells"),
                                           "frac_incl_reverse_scan": (round((H["cells"] + H["reverse_cells"]) / 2.0 * 10.0 / (k_ms * 1e-3) / VALU_LANE_OPS_PER_S, 4)
                                                                      if H.get("reverse_cells") is not None else None)}},
        }
        pf = {"queries_per_s": round(nq / (H["pf_ms"] * 1e-3), 1),
              "stage_ms": {"kmers_lists": round(stage[0], 2), "gather_split": round(stage[1], 2), "replay_score_keepmax": round(stage[2], 2),
                           "large_bins_score": round(stage[3], 2), "large_bins_keepmax_and_overflow_path": round(stage[4], 2),
                           "select": round(stage[5], 2), "total": round(stage[6], 2)},
              "hbm_in_use_gb_after_steps": H["hbm_in_use_gb"],
              "ungapped_cells": int(H["pf_cells"]), "double_diagonal_candidates": int(H["pf_cands"]), "prefilter_hits": int(H.get("nhits", 0))}
        if "ent" in H:
            ent = H["ent"]
            alg = 20.0 * ent        # ~20 B per index entry touched
            ach = alg / (stage[6] * 1e-3) / 1e9 if stage[6] > 0 else 0.0          # the STAGE: every prefilter kernel of the step
            ach_split = alg / (stage[1] * 1e-3) / 1e9 if stage[1] > 0 else 0.0
            pf_kernels = ("pf_kmers", "pf_split_kernel", "pf_replay", "pf_ungapped_kernel", "pf_keepmax", "pf_select_kernel", "pf_overflow_kernel",
                          "pf_scan_kernel", "pf_tiles_kernel")
            # memory-side requests: the k-mer look-ups and the index gather are random 8 ... 40 byte reads, and the chip serves ~55 G of
            # those per second whatever their size (profiles/r05_lookup_rate_probe.txt) - the roof these two kernels actually sit under
            sim = float(H["sim"])
            pf.update({"db_matches": int(ent), "similar_kmers": int(H["sim"]), "overflow_queries": int(H["ovf"]),
                       "roofline": {"kernel": "prefilter stage (pf_kmers + pf_split + pf_replay + scoring / keepMax / select kernels)", "bound": "hbm",
                                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                                    "traffic": pmc_traffic(pf_kernels, PROFILE_ROUND + "_search", per_run_of="pf_split_kernel") if default_wl else None,
                                    "traffic_is": "counter bytes of every prefilter kernel summed over all its dispatches of a run of the stage (replay / scoring "
                                                  "/ select run once per stage chunk), per run",
                                    "kernel_ms": round(stage[6], 3), "algorithmic_bytes_per_launch": round(alg),
                                    "algorithmic_bytes_per_entry": 20,
                                    "split_kernel": {"achieved": round(ach_split, 1), "frac": round(ach_split / HBM_PEAK_GBS, 4), "kernel_ms": round(stage[1], 3),
                                                     "traffic": pmc_traffic(("pf_split_kernel",), PROFILE_ROUND + "_search") if default_wl else None},
                                    "random_request_roofline": {
                                        "what": "index lists gathered by pf_split_kernel: one memory-side request per 64-byte line a list touches - a list of "
                                                "n 8-byte entries at a random entry offset touches 1 + (n - 1) / 8 lines on average - against the measured "
                                                "rate of random requests that miss the L2 (profiles/r05_lookup_rate_probe.txt)",
                                        "lists_gathered": int(sim), "lines_touched": round(sim + (float(ent) - sim) / 8.0), "peak_requests_per_s": 55e9,
                                        "achieved_requests_per_s": round((sim + (float(ent) - sim) / 8.0) / (stage[1] * 1e-3), 1) if stage[1] > 0 else None,
                                        "frac": round((sim + (float(ent) - sim) / 8.0) / (stage[1] * 1e-3) / 55e9, 4) if stage[1] > 0 else None}}})
        out["prefilter"] = pf
        if H.get("end_to_end") is not None:
            e2e = dict(H["end_to_end"])
            e2e["setup_s_not_included"] = {"score_tables_upload_and_device_index_build": round(H["t_index"], 2)}
            if H.get("modules") is not None:
                e2e["mmseqs_modules_stock_vs_patched"] = H["modules"]
            out["end_to_end"] = e2e
            if "queries_per_s_end_to_end" in e2e:
                out["queries_per_s_end_to_end"] = e2e["queries_per_s_end_to_end"]
        S = H.get("search")
        if S and S.get("elapsed_s"):
            ss_ms = S["elapsed_s"] / args.steps * 1e3
            out["ms_per_step_search_semantics"] = round(ss_ms, 3)
            out["queries_per_s_search_semantics"] = round(nq / (ss_ms * 1e-3), 1)
            out["search_semantics"] = {
                "step": "prefilter kernels -> hand-over -> forward scan of every pair + reverse scan of the uint8-pass hits that pass -e "
                        "(MMGPU_SW_START_NOT_WORD) -> mmgpu_sw_block_starts: the device selects the int16-range hits that pass -e, the block "
                        "aligner (block4_kernel.hip) supplies their start positions, the reverse scan those of the pairs it declines; whole "
                        "steps by the wall clock, records complete on the device at the end of each",
                "stages_ms": {"prefilter_kernels": round(float(np.mean(S["pf_ms"])), 2), "align_kernels": round(float(np.mean(S["align_ms"])), 2),
                              "block_starts_call": round(float(np.mean(S["block_s"])) * 1e3, 2)},
                "int16_range_pairs_selected": int(S.get("selected", 0)), "declined_then_reverse_scanned": int(S.get("declined", 0)),
                "left_undecided": int(S.get("too_large", 0)),
                "parity": H.get("search_parity"),
                "reference": "StripedSmithWaterman.cpp:857-882 (ssw_align_private: E-value gate, block aligner for word == 1, fall-back)"}
        for kname in ("two_call", "backtrace", "block_aligner", "pairs_with_start", "inexact_queries", "rerun_unsplit", "merged_lists_sorted", "aligned_slots_filled", "records_gathered",
                      "parity_vs_unsplit"):
            if kname in H:
                out[kname] = H[kname]
        out["setup_s"] = {"generate": round(H["t_gen"], 1), "score_tables_upload_and_device_index_build": round(H["t_index"], 2)}
        if H.get("cpu_sw") is not None:
            out["cpu_baseline"] = H["cpu_sw"]
        if H.get("cpu_pf") is not None:
            out["cpu_baseline_prefilter"] = H["cpu_pf"]
        for kname, v in side.items():
            if v is not None:
                out[kname] = v
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    gpu.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
