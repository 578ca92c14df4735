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


def cpu_baseline(matrices, qs, cbs, tres, toff, budget_s=15.0):
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
                    ctx.sw_batch_score(tres, toff, chunks[ci])
                else:
                    ctx.sw_batch_score(qs[qi], cbs[qi], tres, toff, chunks[ci], mat, 11, 1)

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
    return {"value": round(cells / dt / 1e9, 3), "unit": "GCUPS", "cores": cores,
            "kind": "reference" if use_ref else "port",
            "sample": "%d (query, %d-target chunk) items over %d queries of the same workload (%.3g cells), "
                      "%.1f s wall, %d threads" % (len(sample), chunk, nq, cells, dt, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--targets", type=int, default=100000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    matrices = dict(np.load(os.path.join(ROOT, "tests", "golden", "matrices.npz")))
    mat = matrices["blosum62_sw"]
    sub16 = mat.astype(np.int16)

    # ---- synthetic workload: same queries on every rank, rank-specific target shard (weak scaling) ----------
    (qres, qoff), (tres, toff) = wl.config2_align_only(args.queries, args.targets, seed=1 + 1000 * rank)
    if rank != 0:
        (qres, qoff), _ = wl.config2_align_only(args.queries, 1, planted_frac=0.0, seed=1)
    qs = wl.split(qres, qoff)
    cbs = [host_comp_bias(sub16, matrices["blosum62_pback"], q)[1] for q in qs]

    gpu = mmseqs2_amd.MMGpu(local_rank)
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
    if dist is not None:
        tmax = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        cells_all = torch.tensor([float(batch.cells)], device="cuda", dtype=torch.float64)
        dist.all_reduce(cells_all)
        total_cells = float(cells_all.item())
    else:
        total_cells = float(batch.cells)

    # ---- spot-check the timed batch's results against the oracle (checker only, outside the timed region) ----
    check = None
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
        del res

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
        lane_ops = batch.cells / 2.0 * 10.0
        out = {
            "metric": "sw_gcells_per_s", "value": round(value, 2), "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: align-only, %d random L~N(350,35) queries x %d targets "
                                   "(10%% planted homologs), all-vs-all prefilter lists, BLOSUM62 gap 11/1, comp-bias on, "
                                   "score+end positions" % (args.queries, args.targets),
                       "pairs_per_gpu": int(batch.pairs), "cells_per_gpu": int(batch.cells),
                       "parallelism": "1 process/GPU, independent target shards" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                         "kernel_ms": round(k_ms, 3),
                         "note": "Gotoh SW is VALU-bound (0.003 B/cell); see valu_roofline for the binding resource",
                         "valu_roofline": {"achieved_lane_ops_per_s": round(lane_ops / (k_ms * 1e-3), 1),
                                           "peak_lane_ops_per_s": VALU_LANE_OPS_PER_S,
                                           "frac": round(lane_ops / (k_ms * 1e-3) / VALU_LANE_OPS_PER_S, 4)}},
            "prepare_s": round(prep_s, 2), "check": check,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(matrices, qs, cbs, tres, toff, args.cpu_seconds)
        print(json.dumps(out))
    batch.free()
    gpu.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
