"""The bench line the driver parses: the committed line of this round (profiles/r06_bench_n1.json, produced by
`python bench.py` on the GPU box) must carry the contract's keys, be quoted on BASELINE.json's metric configuration (10k
queries x 1M targets), and its CPU-baseline legs - which double as full-size parity checks against the real reference -
must have found no difference."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    return json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_n1.json")).read())


def test_committed_bench_line_has_the_contract_keys():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "queries_per_s", "end_to_end", "queries_per_s_end_to_end"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["steps"] >= 10
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "configs[2]" in d["config"]["workload"] and "10000 queries x 1000000 targets" in d["config"]["workload"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # value = forward cells / alignment-stage time; queries_per_s = queries / whole step
    assert abs(d["value"] - d["config"]["align_cells_per_step"] / (r["kernel_ms"] * 1e-3) / 1e9) < 0.01 * d["value"]
    assert abs(d["queries_per_s"] - 10000 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["queries_per_s"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    # the CPU legs double as full-size parity checks: they must have found nothing
    # (a bounded sample of the hit lists through the reference's own OpenMP loop: > 1 M pairs, starts included)
    assert c["parity_vs_baseline"]["field_mismatches"] == 0 and c["parity_vs_baseline"]["pairs_compared"] > 1000000
    assert c["kind"] == "reference" and c["threads"] >= c["cores"] >= 1 and "runs" in c
    assert d["cpu_baseline_prefilter"]["parity_vs_reference"]["queries_with_different_hit_lists"] == 0
    assert d["two_call"]["fields_differing_from_fused_path"] == 0
    assert d["nucleotide_align"]["cpu_baseline"]["parity_vs_reference"]["pairs_differing"] == 0
    assert d["nucleotide_search"]["parity_run"]["cpu_baseline"]["parity_vs_reference"]["queries_with_different_hit_lists"] == 0
    assert d["nucleotide_search"]["parity_run"]["cpu_baseline"]["parity_vs_reference"]["queries_compared"] >= 1000
    assert isinstance(d["roofline"]["traffic"], (int, float)) and d["roofline"]["traffic"] > d["roofline"]["algorithmic_bytes_per_launch"]
    # round 5: the prefilter roofline is quoted for the STAGE (every prefilter kernel of the step), the split kernel is a sub-field,
    # and the look-up / gather kernels are priced against the measured rate of random memory-side requests
    pr = d["prefilter"]["roofline"]
    assert abs(pr["kernel_ms"] - d["prefilter"]["stage_ms"]["total"]) < 0.01 and abs(pr["frac"] - pr["achieved"] / pr["peak"]) < 1e-3
    assert abs(pr["achieved"] - pr["algorithmic_bytes_per_launch"] / (pr["kernel_ms"] * 1e-3) / 1e9) < 0.01 * pr["achieved"]
    assert abs(pr["split_kernel"]["kernel_ms"] - d["prefilter"]["stage_ms"]["gather_split"]) < 0.01 and 0 < pr["random_request_roofline"]["frac"] < 1
    assert isinstance(pr["traffic"], (int, float)) and pr["traffic"] > 0
    # block aligner (a15): counts, and what the restatement is pinned against
    b = d["block_aligner"]
    assert b["device"] == b["pairs"] and b["too_large"] == 0 and b["rescored_equal"] == b["rescored_sample"]
    assert "pinned_against" in b and "Rust" in b["pinned_against"]
    # host numeric sequences in -> host lists out, and the two modules through the stock and the patched binary (default --mask 1)
    e = d["end_to_end"]
    assert e["results_equal_to_the_timed_steps"] is True
    assert abs(d["queries_per_s_end_to_end"] - 10000 / e["seconds"]) < 0.01 * d["queries_per_s_end_to_end"]
    m = e["mmseqs_modules_stock_vs_patched"]
    assert m["alignment_dbs_identical"] is True and m["entries_compared"] == 10000
    assert m["patched"]["prefilter_wall_s"] < m["stock"]["prefilter_wall_s"] and m["patched"]["align_wall_s"] < m["stock_block_aligner_stubbed"]["align_wall_s"]
    # round 5: the whole search, also on a persisted device layout (row f1) - result databases equal the stock binary's
    for k in ("patched_search_fused", "patched_search_fused_persisted_layout", "patched_search_fused_resident_server"):
        assert m[k]["result_db_identical_to_stock_align_db"] is True and m[k]["entries_compared"] == 10000, k
    assert m["patched_search_fused_persisted_layout"]["loaded_without_host_lookup"] is True
    # the translated search of configs[4] through both binaries: equal entries (up to the order of tied lines, which the stock
    # binary itself does not keep from run to run)
    t = d["translated_search"]["cpu_baseline"]["parity_vs_reference"]
    assert t["entries_differing"] == 0 and t["result_entries_compared"] == 1000
    assert "timeline" in d["translated_search"]
    # round 6: configs[1] (align-only) carries its own parity against the reference's ssw_align and a CPU baseline
    ao = d["align_only"]["cpu_baseline"]
    assert ao["kind"] == "reference" and ao["parity_vs_reference"]["pairs_compared"] >= 100000
    assert ao["parity_vs_reference"]["score_mismatches"] == 0 and ao["parity_vs_reference"]["end_position_mismatches_among_positive_scores"] == 0
    # round 6: the search-semantics step (block aligner inside the timed step, reverse scan only where the block aligner declines)
    ss = d["search_semantics"]
    assert ss["parity"]["queries_with_a_differing_record"] == 0 and ss["parity"]["queries_compared"] == 10000 and ss["left_undecided"] == 0
    assert abs(d["queries_per_s_search_semantics"] - 10000 / (d["ms_per_step_search_semantics"] * 1e-3)) < 0.01 * d["queries_per_s_search_semantics"]
    assert d["ms_per_step_search_semantics"] <= 200.0


def test_pmc_reader_finds_the_quoted_kernels():
    sys.path.insert(0, ROOT)
    import bench
    for kernels, stem in ((("pf_split_kernel",), "r01_prefilter_config3"), (("sw_kernel<",), "r01_sw_config2"),
                          (("pf_split_kernel",), "r02_search"), (("sw_kernel<", "sw_rev_multi_kernel"), "r02_search"),
                          (("pf_split_kernel",), "r03_search"), (("sw_kernel<", "sw_rev_multi_kernel"), "r03_search")):
        t = bench.pmc_traffic(kernels, stem, check_digest=False)      # (earlier rounds' passes carry no digest)
        assert t is not None and t > 1e9, (kernels, t)


def test_counter_traffic_is_reported_only_from_passes_taken_with_these_device_sources():
    """`roofline.traffic` is read from profiles/<round>_search_pmc_{fetch,write}_size.txt: their header names the digest of
    mmseqs2_amd/csrc they were taken with (scripts/csrc_digest.py; for a clean tree it changes exactly when
    `git rev-parse HEAD:mmseqs2_amd/csrc` does).  Either the committed passes are this tree's - then the line built from them names
    the same digest - or the kernels changed after they were collected, and bench.py must report no counter traffic at all
    (null, never another build's counters) until scripts/collect_profiles.sh has been run again."""
    sys.path.insert(0, ROOT)
    import bench
    now = bench.csrc_digest()
    assert now is not None
    paths = [os.path.join(ROOT, "profiles", "%s_search_pmc_%s_size.txt" % (bench.PROFILE_ROUND, kind)) for kind in ("fetch", "write")]
    current = all(bench.pmc_file_digest(p) == now for p in paths)
    t = bench.pmc_traffic(("sw_kernel<", "sw_rev_multi_kernel"), bench.PROFILE_ROUND + "_search")
    if current:
        assert t > 1e9
        assert _line()["roofline"]["csrc_digest"] == now
    else:
        assert t is None
        assert bench.pmc_traffic(("pf_split_kernel",), bench.PROFILE_ROUND + "_search", per_run_of="pf_split_kernel") is None


def test_stage_traffic_counts_every_dispatch_of_a_run():
    """`prefilter.roofline.traffic` is bytes per RUN of the stage: a kernel that is launched once per stage chunk (replay, scoring,
    select: three times per run in the committed passes) contributes all of its dispatches, not its mean dispatch."""
    sys.path.insert(0, ROOT)
    import bench
    stem = bench.PROFILE_ROUND + "_search"
    once = bench.pmc_traffic(("pf_split_kernel",), stem, check_digest=False)      # (how the files are read, whatever build they are of)
    assert abs(bench.pmc_traffic(("pf_split_kernel",), stem, check_digest=False, per_run_of="pf_split_kernel") - once) <= 1e-6 * once
    mean_replay = bench.pmc_traffic(("pf_replay_kernel",), stem, check_digest=False)
    run_replay = bench.pmc_traffic(("pf_replay_kernel",), stem, check_digest=False, per_run_of="pf_split_kernel")
    assert 2.5 * mean_replay < run_replay < 3.5 * mean_replay       # three stage chunks per run
    line = _line()["prefilter"]["roofline"]
    assert line["traffic"] > line["split_kernel"]["traffic"] + run_replay
    assert bench.pmc_traffic(("pf_split_kernel",), stem, check_digest=False, per_run_of="no_such_kernel") is None


def test_result_databases_compare_up_to_the_order_of_tied_lines(tmp_path):
    """mmseqs2_amd/dbio.py::diff_dbs_up_to_tie_order: lines of an entry with equal (bit score, E-value) may swap places - the stock
    binary's own translated search does that from run to run - anything else is a difference."""
    import struct
    from mmseqs2_amd import dbio

    def write(name, entries):
        data, index, off = b"", "", 0
        for k, e in entries:
            blob = e + b"\0"
            index += "%d\t%d\t%d\n" % (k, off, len(blob))
            data += blob
            off += len(blob)
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        open(p + ".index", "w").write(index)
        open(p + ".dbtype", "wb").write(struct.pack("<i", 5))
        return p

    a = write("a", [(0, b"7\t64\t0.9\t1E-10\t1\t9\n7\t64\t0.8\t1E-10\t20\t29\n3\t50\t0.7\t1E-5\t2\t8\n"), (1, b"4\t30\t0.5\t1E-3\t1\t5\n")])
    b = write("b", [(0, b"7\t64\t0.8\t1E-10\t20\t29\n7\t64\t0.9\t1E-10\t1\t9\n3\t50\t0.7\t1E-5\t2\t8\n"), (1, b"4\t30\t0.5\t1E-3\t1\t5\n")])
    c = write("c", [(0, b"7\t64\t0.9\t1E-10\t1\t9\n3\t50\t0.7\t1E-5\t2\t8\n7\t64\t0.8\t1E-10\t20\t29\n"), (1, b"4\t30\t0.5\t1E-3\t1\t6\n")])
    assert dbio.diff_dbs(a, b)[1] == 1
    assert dbio.diff_dbs_up_to_tie_order(a, b)[:3] == (2, 0, 1)
    # a tied line moved across a line with another key, and a changed field: both are differences
    assert dbio.diff_dbs_up_to_tie_order(a, c)[:3] == (2, 2, 0)
    assert dbio.diff_dbs_up_to_tie_order(a, a)[:3] == (2, 0, 0)


def test_bench_started_as_one_process_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` (the driver's form of the command) re-runs itself under torch.distributed.run, one rank per GPU,
    rendezvous on 127.0.0.1; inside a launched rank (WORLD_SIZE set) nothing is re-launched."""
    sys.path.insert(0, ROOT)
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--headline-only"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    try:
        bench.main()
        raise AssertionError("main() returned instead of leaving with the launcher's exit code")
    except SystemExit as e:
        assert e.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-5:] == ["--gpus", "4", "--steps", "3", "--headline-only"] and cmd[-6].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
