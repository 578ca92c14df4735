#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X-native prefilter->align path.

Metric (BASELINE.json): giga-cells/s of the Gotoh Smith-Waterman stage, cells = sum over computed
alignments of qlen*tlen (forward DP only, Alignment.cpp:380,530 convention).

Workload at N=1: BASELINE.json configs[1] - "mmseqs align only: 1k random L~350 protein queries vs 100k
targets" (10 % of targets carry a planted homolog), every query against every target as `mmseqs align` sees
it behind the all-vs-all fake prefilter (data/workflow/blastp.sh:22-33): 1e8 alignments, ~1.2e13 cells per
step.  A "step" = one pass of the hot path (forward scan, score + end positions) over that batch with the
target DB, the queries and the prefilter lists already resident in HBM.

N>1 (launched by torch.distributed.run, one process per GPU): every rank owns an independent target shard of
the same size and its own prefilter lists (weak scaling, no data-path collective in this round); value =
cells of all ranks / max-over-ranks time.

The same line carries a second object, "search": BASELINE.json configs[2] (10k queries x 1M targets, -s 5.7,
UniRef50-like lengths, 50-member families) through the k-mer prefilter and the gapped alignment of its hit lists -
queries/s, per-stage milliseconds, the roofline of the HBM-bound gather kernel and the reference's own prefilter
loop timed on the host cores.  With N>1 every rank holds its own 1M-target shard (weak scaling), the per-query hit
lists are all-gathered over RCCL and merged on the device (mergeTargetSplits semantics).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
# packed-int16 (VOP3P) VALU ops issue at 16 lanes/clk/SIMD on gfx950: measured 38.1 Tlane-op/s for v_pk_max_i16 /
# v_pk_sub_u16 / v_perm_b32 (profiles/r01_valu_issue_rate_probe.txt) = 256 CU x 4 SIMD x 16 lanes x 2.4 GHz = 39.3e12
VALU_LANE_OPS_PER_S = 256 * 4 * 16 * 2.4e9


def cpu_baseline(matrices, qs, cbs, tres, toff, budget_s=15.0, gpu_res=None):
    """Reported (not optimised-against) CPU baseline on the GPU box's host cores, bounded sample.
    kind "reference": the real AVX2 striped Smith-Waterman of the reference (oracle/_ref/libmmref.so, uint8 pass
    + int16 re-run), one SmithWaterman object per thread as Alignment::run does (Alignment.cpp:279-295).
    kind "port": the scalar C oracle, when the reference library did not travel."""
    from oracle import pyoracle
    cores = os.cpu_count() or 1
    n_t = len(toff) - 1
    tlen_sum = int(toff[-1])
    use_ref = pyoracle.ref_available()
    ids = np.arange(n_t, dtype=np.uint32)
    mat = matrices["blosum62_sw"]
    if use_ref:
        ser = matrices["blosum62_serialized"]
        ctxs = [pyoracle.RefLib(serialized=ser, max_len=70000, db_residues=tlen_sum) for _ in range(cores)]
    else:
        ctxs = [pyoracle.Oracle()] * cores
    # work item = one query against a chunk of `chunk` targets; a first parallel round (one item per thread)
    # calibrates the aggregate rate under full load, then the sample is sized to ~budget_s of wall time
    chunk = min(n_t, 2000)
    chunks = [ids[k:k + chunk] for k in range(0, n_t, chunk)]
    csum = np.array([float(toff[int(c[-1]) + 1] - toff[int(c[0])]) for c in chunks])
    qlens = np.array([len(q) for q in qs], np.float64)
    items = [(qi, ci) for qi in range(len(qs)) for ci in range(len(chunks))]

    mism = [0, 0]   # pairs compared with the device results, pairs that differ (score, q_end, t_end)

    def run_items(sub):
        work = list(sub)
        lock = threading.Lock()

        def worker(ctx):
            last_q = -1
            while True:
                with lock:
                    if not work:
                        return
                    qi, ci = work.pop()
                if use_ref and qi != last_q:
                    ctx.sw_set_query(qs[qi])   # ssw_init recomputes the composition bias itself
                    last_q = qi
                if use_ref:
                    sc, qe, te = ctx.sw_batch_score(tres, toff, chunks[ci])
                else:
                    sc, qe, te, _ = ctx.sw_batch_score(qs[qi], cbs[qi], tres, toff, chunks[ci], mat, 11, 1)
                if gpu_res is not None:
                    g = gpu_res[qi, chunks[ci]]
                    # the reference's uint8 pass leaves q_end undefined (0) when the score is 0
                    pos = sc > 0
                    bad = int(np.count_nonzero(g["score"] != sc)) + int(np.count_nonzero((g["t_end"] != te) & pos)) \
                        + int(np.count_nonzero((g["q_end"] != qe) & pos))
                    with lock:
                        mism[0] += len(sc)
                        mism[1] += bad

        th = [threading.Thread(target=worker, args=(ctxs[i],)) for i in range(cores)]
        t0 = time.time()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.time() - t0, float(sum(qlens[qi] * csum[ci] for qi, ci in sub))

    cal = items[:cores]
    dt_cal, cells_cal = run_items(cal)
    rate = cells_cal / max(dt_cal, 1e-3)
    per_item = cells_cal / len(cal)
    n_items = int(min(len(items) - len(cal), max(cores, rate * budget_s / per_item)))
    sample = items[len(cal):len(cal) + n_items]
    dt, cells = run_items(sample)
    nq = len(set(qi for qi, _ in sample))
    parity = None
    if gpu_res is not None:
        parity = {"pairs_compared": mism[0], "field_mismatches": mism[1], "fields": "score, q_end, t_end"}
    return {"value": round(cells / dt / 1e9, 3), "unit": "GCUPS", "cores": cores, "parity_vs_baseline": parity,
            "kind": "reference" if use_ref else "port",
            "sample": "%d (query, %d-target chunk) items over %d queries of the same workload (%.3g cells), "
                      "%.1f s wall, %d threads" % (len(sample), chunk, nq, cells, dt, cores)}


def allreduce(torch, dist, values, op="sum"):
    """All-reduce of a few python floats (device tensor with nccl, host tensor with gloo)."""
    if dist is None:
        return [float(v) for v in values]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(v) for v in values], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return [float(v) for v in t.cpu()]


def pmc_traffic(kernel, stem="r01_prefilter_config3"):
    """HBM bytes per launch of `kernel` (summed over its template instantiations) from the committed rocprofv3 PMC
    passes of this same workload (profiles/<stem>_pmc_{fetch,write}_size.txt: separate --pmc FETCH_SIZE / WRITE_SIZE
    runs, unit KB, mean per dispatch).  FETCH_SIZE is reported as measured; MI355X_MICROARCH.md notes it under-counts
    wide coalesced reads by 2x on gfx950, so this is a lower bound."""
    tot = 0.0
    for kind in ("fetch", "write"):
        path = os.path.join(ROOT, "profiles", "%s_pmc_%s_size.txt" % (stem, kind))
        if not os.path.exists(path):
            return None
        found = False
        for line in open(path):
            if kernel in line and "_SIZE" in line and "mean=" in line:
                tot += float(line.split("mean=")[1]) * 1024.0
                found = True
        if not found:
            return None
    return {"bytes_per_launch": round(tot), "source": "profiles/%s_pmc_*_size.txt (FETCH_SIZE + WRITE_SIZE, KB)" % stem}


def search_cpu_baseline(matrices, qres, qoff, tres, toff, kmer_thr, budget_s, gpu_lists=None):
    """The reference's own prefilter query loop (QueryMatcher::matchQuery per OpenMP thread, Prefiltering.cpp:820-917)
    from oracle/_ref/libmmref.so on the host cores, bounded sample of the same queries against the same targets.
    The lists it produces are also the checker of the device lists at full size (gpu_lists = per-query
    (ids, scores, diagonals) of the timed run): bit-identical or counted as a mismatch."""
    from oracle import pyoracle
    if not pyoracle.ref_available():
        return None
    cores = os.cpu_count() or 1
    ref = pyoracle.RefPrefilter(6, serialized=(matrices["vtml80_serialized"].tobytes(), matrices["blosum62_serialized"].tobytes()))
    t0 = time.time()
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
                     "%.1f s (not counted)" % (n, nq, len(toff) - 1, sec, cores, t_index),
           "db_matches_per_query": round(dbm / n), "hits_per_query": round(hits / n, 1)}
    if gpu_lists is not None:
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


def search_section(args, gpu, torch, dist, rank, world, matrices, barrier):
    """BASELINE.json configs[2]/[3]: k-mer prefilter + gapped alignment of the hit lists, one target shard per rank."""
    from mmseqs2_amd import capi, workloads as wl
    from mmseqs2_amd import distributed as D
    km16 = matrices["vtml80_kmer"].astype(np.int16)
    sens, k, max_res = 5.7, 6, 300
    kmer_thr = int(163.2 - 8.917 * sens)          # Prefiltering::getKmerThreshold, Prefiltering.cpp:1080-1095
    t0 = time.time()
    (qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(args.pf_families, args.pf_members, args.pf_queries, seed=10,
                                                             target_seed=11 + 1000 * rank)
    t_gen = time.time() - t0
    qs = wl.split(qres, qoff)
    nq, nt = len(qs), len(toff) - 1
    t0 = time.time()
    s3, i3 = capi.host_score_matrix(km16, 3)
    gpu.load_targets(tres, toff, 21)
    # the k-mer index is built in HBM from the resident targets (IndexBuilder::fillDatabase, masking off)
    gpu.pf_build_index(k, 21, True, s3, i3, km16, kmer_thr, matrices["blosum62_ungapped"])
    gpu.synchronize()
    t_index = time.time() - t0
    cbs = [capi.host_comp_bias(km16, matrices["vtml80_pback"], q)[0] for q in qs]
    queries = [dict(q=q, comp_bias=cb, identity_id=None) for q, cb in zip(qs, cbs)]
    mh = capi.split_max_hits(max_res, world)      # Prefiltering.cpp:391-394
    bsz = args.pf_batch
    batches = [gpu.pf_prepare(queries[i:i + bsz], kmer_thr, max_hits=mh, min_diag_score=15, ref_bins=2)
               for i in range(0, nq, bsz)]
    shard_sizes = [nt] * world
    # MMGPU_BENCH_FORCE_EXCHANGE=1: run the N > 1 exchange path (RCCL all-gather of the hit lists + device merge, RCCL
    # exchange of the alignment results) on a single rank - a self-test of the collectives on a 1-GPU box
    exchange = world > 1 or (dist is not None)

    def one_pass(keep):
        merged = []
        for b in batches:
            b.run()
            if exchange:
                mh_t, mc_t = D.gather_and_merge_device(gpu, b, b.nq, mh, shard_sizes)
                if keep:
                    merged.append((mh_t, mc_t))
        return merged

    one_pass(False)                                # warm-up, sizes the working buffers
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.pf_steps):
        merged = one_pass(True)
    barrier()
    t_pf = (time.perf_counter() - t0) / args.pf_steps
    t_pf = allreduce(torch, dist, [t_pf], "max")[0]
    stage = np.zeros(7)
    ent = sim = cells = cands = nhits = ovf = 0
    lists, slot_index, full_lists = [], [], []
    for bi, b in enumerate(batches):
        stage += np.array(b.stage_ms())
        h, c, st, stats = b.fetch()
        cc = b.last_cells()
        cells += cc[0]
        cands += cc[1]
        ent += int(stats["db_matches"].sum())
        sim += int(stats["kmer_list_len"].sum())
        ovf += int((st != 0).sum())
        if exchange:
            mh_t, mc_t = merged[bi]
            hh = mh_t.cpu().numpy().reshape(b.nq, -1).view(capi.PF_HIT_DTYPE).reshape(b.nq, -1)
            cc2 = mc_t.cpu().numpy()
            for qi in range(b.nq):
                ids = hh[qi]["id"][:cc2[qi]]
                own = (ids >= rank * nt) & (ids < (rank + 1) * nt)     # pairs run on the GPU owning the target
                lists.append((ids[own] - rank * nt).astype(np.uint32))
                slot_index.append((len(lists) - 1) * world * mh + np.nonzero(own)[0])
            nhits += int(cc2.sum())
        else:
            for qi in range(b.nq):
                lists.append(h[qi]["id"][:c[qi]].copy())
                full_lists.append((h[qi]["id"][:c[qi]].copy(), h[qi]["score"][:c[qi]].copy(), h[qi]["diagonal"][:c[qi]].copy()))
            nhits += int(c.sum())

    if args.prefilter_only:      # counter passes (scripts/collect_profiles.sh): the prefilter kernels only
        for b in batches:
            b.free()
        return {"prefilter_s": round(t_pf, 4), "db_matches": int(ent), "stage_ms_total_rank0": round(float(stage[6]), 2)} if rank == 0 else None

    # ---- spot check of the prefilter lists against the oracle on a reduced copy of the problem is done by the
    # tests; here: the alignment of the lists (Alignment::run behind the prefilter DB) ----
    mat = matrices["blosum62_sw"]
    sub16 = mat.astype(np.int16)
    t0 = time.time()
    swq = []
    # Start positions (reverse scan) are computed only for pairs that pass -e 1e-3, like ssw_align_private
    # (StripedSmithWaterman.cpp:857-863).  The reference gets the E-value from ALP (host); here the threshold is the
    # Karlin-Altschul bound E = K m n exp(-lambda S) with the gapped BLOSUM62 11/1 constants (lambda 0.267, K 0.041),
    # n = residues of all shards.
    import math
    db_res = float(toff[-1]) * world
    for q, ids in zip(qs, lists):
        min_start = int(math.ceil(math.log(0.041 * len(q) * db_res / 1e-3) / 0.267))
        swq.append(dict(q=q, comp_bias=capi.host_comp_bias(sub16, matrices["blosum62_pback"], q)[1], targets=ids,
                        min_start_score=max(min_start, 1)))
    swb = gpu.sw_prepare(mat, 11, 1, swq, mode=1)
    t_handoff = time.time() - t0
    swb.run()
    barrier()
    dev = torch.device("cuda", torch.cuda.current_device())
    slot_t = local_t = None
    if exchange:
        slot_t = torch.from_numpy(np.concatenate(slot_index) if slot_index else np.zeros(0, np.int64)).to(dev)
        local_t = torch.zeros((max(swb.pairs, 1), 6), dtype=torch.int32, device=dev)

    def align_pass():
        swb.run()
        if exchange:
            # second exchange of the path: the alignment results of the merged lists stay on the device - D2D copy
            # out of the batch, scatter to the merged-list slots, one all-reduce over RCCL
            swb.fetch_device(local_t.data_ptr())
            return D.exchange_sw_results_tensor(local_t[:swb.pairs], slot_t, nq * world * mh)
        return None

    t0 = time.perf_counter()
    for _ in range(args.pf_steps):
        sw_all = align_pass()
    barrier()
    t_sw = (time.perf_counter() - t0) / args.pf_steps
    sw_cells, sw_pairs = swb.cells, swb.pairs
    fused = None
    if not exchange:
        # the whole path as one device pipeline: prefilter batch -> hit lists sorted / scheduled on the device
        # (mmgpu_sw_prepare_from_pf) -> alignment, nothing but the query descriptors crosses PCIe in between
        msh = [gpu.sw_marshal_queries(mat, 11, 1, swq[i:i + bsz]) for i in range(0, nq, bsz)]
        sep = swb.fetch()

        def fused_pass(keep):
            outs = []
            for b, m in zip(batches, msh):
                b.run()
                fb = gpu.sw_prepare_from_pf(mat, 11, 1, None, b, mode=1, marshalled=m)
                fb.run()
                if keep:
                    outs.append((fb.fetch().reshape(b.nq, -1), fb.cells, fb.pairs))
                fb.free()
            return outs

        fused_pass(False)
        gpu.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.pf_steps):
            fused_pass(False)
        gpu.synchronize()
        t_fused = (time.perf_counter() - t0) / args.pf_steps
        outs = fused_pass(True)
        # where the pipeline's time goes (separate pass with a device sync after every call; sums to more than t_fused)
        brk = np.zeros(4)
        for b, m in zip(batches, msh):
            ts = [time.perf_counter()]
            b.run(); gpu.synchronize(); ts.append(time.perf_counter())
            fb = gpu.sw_prepare_from_pf(mat, 11, 1, None, b, mode=1, marshalled=m); gpu.synchronize(); ts.append(time.perf_counter())
            fb.run(); gpu.synchronize(); ts.append(time.perf_counter())
            fb.free(); ts.append(time.perf_counter())
            brk += np.diff(ts)
        # slot by slot the same results as the two-call path through the host
        bad, off, qi = 0, 0, 0
        for fr, _, _ in outs:
            for r in range(fr.shape[0]):
                n = len(lists[qi])
                a, bb = fr[r, :n], sep[off:off + n]
                bad += int(sum(not np.array_equal(a[f], bb[f]) for f in ("score", "q_end", "t_end", "q_start", "t_start")))
                off += n
                qi += 1
        fused = {"s": round(t_fused, 4), "queries_per_s": round(nq / t_fused, 1),
                 "align_cells": int(sum(o[1] for o in outs)), "align_pairs": int(sum(o[2] for o in outs)),
                 "queries_differing_from_two_call_path": bad,
                 "synchronous_breakdown_ms": {"prefilter": round(brk[0] * 1e3, 2), "prepare_from_pf": round(brk[1] * 1e3, 2),
                                              "align": round(brk[2] * 1e3, 2), "free": round(brk[3] * 1e3, 2)},
                 "what": "per batch: prefilter, device-side list sort + job table, alignment (mmgpu_sw_prepare_from_pf); "
                         "includes the alignment batch set-up (profile build, allocation)"}
    # backtraces (Matcher::SCORE_COV_SEQID / -a) for the hit lists of the first 1000 queries
    bt_n = int(sum(len(x) for x in lists[:1000]))
    t0 = time.perf_counter()
    bt_info, _ = swb.traceback(np.arange(bt_n, dtype=np.uint32)) if bt_n else (np.zeros(0, capi.SW_BT_DTYPE), [])
    t_bt = time.perf_counter() - t0
    bt_ok = int((bt_info["status"] == 0).sum()) if bt_n else 0
    t_sw = allreduce(torch, dist, [t_sw], "max")[0]
    sw_cells, sw_pairs = allreduce(torch, dist, [sw_cells, sw_pairs])
    swb.free()
    res = None
    if rank == 0:
        # the committed PMC passes were taken on exactly the default workload (10k x 1M, one batch)
        default_wl = (args.pf_families, args.pf_members, args.pf_queries, args.pf_batch) == (20000, 50, 10000, 10000)
        traffic = pmc_traffic("pf_split_kernel") if default_wl else None
        # algorithmic HBM bytes of the gather/split kernel: ~20 B per index entry touched
        # (6 B entry gathered, 8 B written + 8 B re-read for the replay, amortised list descriptors)
        alg = 20.0 * ent
        achieved = alg / (stage[1] * 1e-3) / 1e9 if stage[1] > 0 else 0.0
        res = {
            "workload": "BASELINE.json configs[2]: %d queries x %d targets per GPU (%d families x %d members, L~LogNormal(5.45,0.6)), "
                        "-s 5.7 (k=6, k-mer thr %d), --max-seqs %d%s, then alignment of the hit lists (score, ends; starts for pairs passing -e 1e-3)"
                        % (nq, nt, args.pf_families, args.pf_members, kmer_thr, max_res,
                           " (per-split %d, Prefiltering.cpp:391-394)" % mh if world > 1 else ""),
            # N = 1: the whole device pipeline incl. the hand-over (fused_pipeline); N > 1: prefilter (with the hit-list
            # all-gather + merge) + alignment (with the result exchange), host-side list hand-over excluded
            "queries_per_s": round(nq / t_fused, 1) if fused is not None else round(nq / (t_pf + t_sw), 1),
            "queries_per_s_stages_only": round(nq / (t_pf + t_sw), 1), "prefilter_queries_per_s": round(nq / t_pf, 1),
            "prefilter_s": round(t_pf, 4), "align_s": round(t_sw, 4), "handoff_s": round(t_handoff, 2), "fused_pipeline": fused,
            "targets_total": nt * world, "n_gpus": world,
            "stage_ms": {"kmers_lists": round(stage[0], 2), "gather_split": round(stage[1], 2), "replay_score_keepmax": round(stage[2], 2),
                         "large_bins_score": round(stage[3], 2), "large_bins_keepmax_and_overflow_path": round(stage[4], 2),
                         "select": round(stage[5], 2),
                         "total_rank0": round(stage[6], 2)},
            "db_matches": int(ent), "similar_kmers": int(sim), "double_diagonal_candidates": int(cands),
            "ungapped_cells": int(cells), "prefilter_hits": int(nhits), "overflow_queries": int(ovf),
            "align_pairs": int(sw_pairs), "align_cells": int(sw_cells),
            "align_gcups": round(sw_cells / t_sw / 1e9, 1),
            "backtrace": {"pairs": bt_n, "with_cigar": bt_ok, "s_incl_download": round(t_bt, 4),
                          "pairs_per_s": round(bt_n / t_bt, 1) if t_bt > 0 else None},
            "roofline": {"kernel": "pf_split_kernel (index gather + stable bin split)", "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_ms": round(stage[1], 3), "launches": len(batches),
                         "algorithmic_bytes_per_launch": round(alg / len(batches)), "algorithmic_bytes_per_entry": 20,
                         "replay_score": {"kernel": "pf_replay_kernel (double-diagonal replay + ungapped scoring)",
                                          "bound": "latency / LDS+VALU issue", "kernel_ms": round(stage[2], 3),
                                          "entries_per_s": round(ent / (stage[2] * 1e-3), 1) if stage[2] > 0 else None,
                                          "ungapped_cells_per_s": round(cells / (stage[2] * 1e-3), 1) if stage[2] > 0 else None,
                                          "algorithmic_GBps": round((8.0 * ent + 1.0 * cells) / (stage[2] * 1e-3) / 1e9, 1) if stage[2] > 0 else None}},
            "setup_s": {"generate": round(t_gen, 1), "score_tables_upload_and_device_index_build": round(t_index, 2)},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = search_cpu_baseline(matrices, qres, qoff, tres, toff, kmer_thr, args.cpu_seconds, full_lists)
    for b in batches:
        b.free()
    return res


def nucl_section(args, gpu, matrices, rank):
    """BASELINE.json configs[4], the nucleotide alignment step (BandedNucleotideAligner::align behind Alignment::run):
    reads with 10 % substitutions / 2 % indels against their source contigs (true prefilter diagonal, both strands)
    plus unrelated contigs.  The prefilter side of the nucleotide search is not part of this round (lists synthetic)."""
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
    dt = time.perf_counter() - t0
    if rank != 0:
        return None
    aligned = int(hits["bt_len"].sum())
    res = {"workload": "BASELINE.json configs[4] (alignment step only): %d reads of %d nt (10%% substitutions, 2%% indels) x "
                       "(source contig on the true diagonal + %d unrelated contigs), %d contigs ~LogNormal(20 kb), gap 5/2, "
                       "band 64, z-drop 40" % (len(queries), args.nucl_read_len, 4, args.nucl_contigs),
           "pairs": len(pairs), "pairs_per_s": round(len(pairs) / dt, 1), "s_incl_upload_and_download": round(dt, 4),
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


def main():
    # exactly ONE line on stdout: libraries (RCCL prints a version banner) write to fd 1 behind Python's back, so
    # fd 1 is pointed at stderr for the run and the JSON line goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--targets", type=int, default=100000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-search", action="store_true", help="skip the configs[2] prefilter+align section")
    ap.add_argument("--pf-families", type=int, default=20000)
    ap.add_argument("--pf-members", type=int, default=50)
    ap.add_argument("--pf-queries", type=int, default=10000)
    ap.add_argument("--pf-batch", type=int, default=10000)
    ap.add_argument("--pf-steps", type=int, default=2)
    ap.add_argument("--no-nucl", action="store_true", help="skip the configs[4] nucleotide alignment section")
    ap.add_argument("--nucl-contigs", type=int, default=4000)
    ap.add_argument("--nucl-reads", type=int, default=1000)
    ap.add_argument("--nucl-read-len", type=int, default=10000)
    ap.add_argument("--prefilter-only", action="store_true", help="configs[2] section: stop after the prefilter (counter passes)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    import mmseqs2_amd
    from mmseqs2_amd import workloads as wl
    from mmseqs2_amd.capi import host_comp_bias

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (args.gpus, args.gpus))
    # MMGPU_BENCH_BACKEND=gloo lets several ranks share one GPU to exercise the N>1 code path on a 1-GPU box
    backend = os.environ.get("MMGPU_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1 or os.environ.get("MMGPU_BENCH_FORCE_EXCHANGE") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    matrices = dict(np.load(os.path.join(ROOT, "tests", "golden", "matrices.npz")))
    mat = matrices["blosum62_sw"]
    sub16 = mat.astype(np.int16)

    # ---- synthetic workload: same queries on every rank, rank-specific target shard (weak scaling) ----------
    (qres, qoff), (tres, toff) = wl.config2_align_only(args.queries, args.targets, seed=1 + 1000 * rank)
    if rank != 0:
        (qres, qoff), _ = wl.config2_align_only(args.queries, 1, planted_frac=0.0, seed=1)
    qs = wl.split(qres, qoff)
    cbs = [host_comp_bias(sub16, matrices["blosum62_pback"], q)[1] for q in qs]

    gpu = mmseqs2_amd.MMGpu(device_index)
    stream = torch.cuda.current_stream()
    gpu.set_stream(stream.cuda_stream)
    gpu.load_targets(tres, toff, 21)
    ids = np.arange(args.targets, dtype=np.uint32)
    queries = [dict(q=q, comp_bias=cb, targets=ids, min_start_score=0) for q, cb in zip(qs, cbs)]
    t0 = time.time()
    batch = gpu.sw_prepare(mat, 11, 1, queries, mode=0)      # H2D + scheduling: outside the timed region
    prep_s = time.time() - t0

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run()
    barrier()
    elapsed = time.perf_counter() - t0
    # HIP events recorded by the library on the launch stream around the kernels of each timed step
    kern_ms = [batch.kernel_ms_mean(args.steps)[0]]
    elapsed = allreduce(torch, dist, [elapsed], "max")[0]
    total_cells = allreduce(torch, dist, [batch.cells])[0]

    # ---- spot-check the timed batch's results against the oracle (checker only, outside the timed region) ----
    check = None
    res = None
    if rank == 0:
        from oracle.pyoracle import Oracle
        res = batch.fetch().reshape(args.queries, args.targets)
        orc = Oracle()
        rng = np.random.default_rng(0)
        bad = 0
        n_chk = 400
        for _ in range(n_chk):
            qi, ti = int(rng.integers(0, args.queries)), int(rng.integers(0, args.targets))
            r = orc.sw_align(qs[qi], cbs[qi], tres[int(toff[ti]):int(toff[ti + 1])], mat, 11, 1)
            h = res[qi, ti]
            bad += (int(h["score"]), int(h["q_end"]), int(h["t_end"]), int(h["word"])) != (r["score"], r["q_end"], r["t_end"], r["word"])
        check = {"pairs_checked_vs_oracle": n_chk, "mismatches": bad,
                 "score_checksum": int(res["score"].astype(np.int64).sum())}

    batch_cells, batch_pairs = batch.cells, batch.pairs
    batch.free()
    search = None
    if not args.no_search:
        search = search_section(args, gpu, torch, dist, rank, world, matrices, barrier)
    nucl = None
    if not args.no_nucl and not args.prefilter_only:
        try:
            nucl = nucl_section(args, gpu, matrices, rank)
        except Exception as e:      # the newest section must not take the headline line down with it: report, do not hide
            nucl = {"error": "%s: %s" % (type(e).__name__, e)} if rank == 0 else None

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_cells * args.steps / elapsed / 1e9
        k_ms = float(np.mean(kern_ms))
        # algorithmic HBM bytes of one launch: tlen + 28 bytes per alignment + the query
        # residues/bias once per workgroup-visible query
        tlen = (toff[1:] - toff[:-1]).astype(np.float64)
        alg_bytes = float(args.queries) * float((tlen + 28).sum()) + 2.0 * float(qoff[-1])
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        # packed-int16 VALU work the algorithm needs: 10 VOP3P lane-ops per pair of cells
        lane_ops = batch_cells / 2.0 * 10.0
        out = {
            "metric": "sw_gcells_per_s", "value": round(value, 2), "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: align-only, %d random L~N(350,35) queries x %d targets "
                                   "(10%% planted homologs), all-vs-all prefilter lists, BLOSUM62 gap 11/1, comp-bias on, "
                                   "score+end positions" % (args.queries, args.targets),
                       "pairs_per_gpu": int(batch_pairs), "cells_per_gpu": int(batch_cells),
                       "parallelism": "1 process/GPU, independent target shards" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6),
                         # per step = one launch of every sw_kernel<R> instantiation; PMC passes of this exact workload
                         "traffic": pmc_traffic("sw_kernel<", "r01_sw_config2") if (args.queries, args.targets) == (1000, 100000) else None,
                         "algorithmic_bytes_per_launch": round(alg_bytes),
                         "kernel_ms": round(k_ms, 3),
                         "note": "Gotoh SW is VALU-bound (0.003 B/cell); see valu_roofline for the binding resource",
                         "valu_roofline": {"achieved_lane_ops_per_s": round(lane_ops / (k_ms * 1e-3), 1),
                                           "peak_lane_ops_per_s": VALU_LANE_OPS_PER_S,
                                           "frac": round(lane_ops / (k_ms * 1e-3) / VALU_LANE_OPS_PER_S, 4)}},
            "prepare_s": round(prep_s, 2), "check": check,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(matrices, qs, cbs, tres, toff, args.cpu_seconds, res)
        if search is not None:
            out["search"] = search
        if nucl is not None:
            out["nucleotide_align"] = nucl
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    gpu.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
