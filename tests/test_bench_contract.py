"""The bench line the driver parses: the committed round-1 line (profiles/r01_bench_n1.json, produced by `python bench.py`
on the GPU box) must carry the contract's keys, and bench.py's reader of the committed PMC passes must find the kernels it
quotes as `roofline.traffic`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    return json.loads(open(os.path.join(ROOT, "profiles", "r01_bench_n1.json")).read())


def test_committed_bench_line_has_the_contract_keys():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    # the two CPU legs double as full-size parity checks: they must have found nothing
    assert c["parity_vs_baseline"]["field_mismatches"] == 0
    s = d["search"]
    assert s["cpu_baseline"]["parity_vs_reference"]["queries_with_different_hit_lists"] == 0
    assert s["fused_pipeline"]["queries_differing_from_two_call_path"] == 0
    assert d["nucleotide_align"]["cpu_baseline"]["parity_vs_reference"]["pairs_differing"] == 0


def test_pmc_reader_finds_the_quoted_kernels():
    sys.path.insert(0, ROOT)
    import bench
    for kernel, stem in (("pf_split_kernel", "r01_prefilter_config3"), ("sw_kernel<", "r01_sw_config2")):
        t = bench.pmc_traffic(kernel, stem)
        assert t is not None and t["bytes_per_launch"] > 1e9, (kernel, t)
