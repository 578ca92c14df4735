"""The bench line the driver parses: the committed line of this round (profiles/r05_bench_n1.json, produced by
`python bench.py` on the GPU box) must carry the contract's keys, be quoted on BASELINE.json's metric configuration (10k
queries x 1M targets), and its CPU-baseline legs - which double as full-size parity checks against the real reference -
must have found no difference."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    return json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")).read())


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
    assert t["entries_differing"] == 0 and t["result_entries_compared"] == 100


def test_pmc_reader_finds_the_quoted_kernels():
    sys.path.insert(0, ROOT)
    import bench
    for kernels, stem in ((("pf_split_kernel",), "r01_prefilter_config3"), (("sw_kernel<",), "r01_sw_config2"),
                          (("pf_split_kernel",), "r02_search"), (("sw_kernel<", "sw_rev_multi_kernel"), "r02_search"),
                          (("pf_split_kernel",), "r03_search"), (("sw_kernel<", "sw_rev_multi_kernel"), "r03_search")):
        t = bench.pmc_traffic(kernels, stem, check_digest=False)      # (earlier rounds' passes carry no digest)
        assert t is not None and t > 1e9, (kernels, t)


def test_this_rounds_pmc_passes_were_taken_with_these_device_sources():
    """`roofline.traffic` is read from profiles/r05_search_pmc_{fetch,write}_size.txt: their header names the digest of
    mmseqs2_amd/csrc they were taken with (scripts/csrc_digest.py; for a clean tree it changes exactly when
    `git rev-parse HEAD:mmseqs2_amd/csrc` does).  Kernels changed after the passes were collected = this test fails until
    scripts/collect_profiles.sh has been run again and its summaries committed."""
    sys.path.insert(0, ROOT)
    import bench
    now = bench.csrc_digest()
    assert now is not None
    for kind in ("fetch", "write"):
        path = os.path.join(ROOT, "profiles", "%s_search_pmc_%s_size.txt" % (bench.PROFILE_ROUND, kind))
        assert bench.pmc_file_digest(path) == now, (path, bench.pmc_file_digest(path), now)
    assert bench.pmc_traffic(("sw_kernel<", "sw_rev_multi_kernel"), bench.PROFILE_ROUND + "_search") > 1e9
    assert _line()["roofline"]["csrc_digest"] == now
