"""The drop-in, exercised as a drop-in (b): `mmseqs prefilter` / `align` / `search` of a
reference build patched with integration/mmseqs_mmgpu.patch and linked to libmmgpu.so, against the STOCK reference binary
on the same inputs; the result DBs must agree entry by entry, byte for byte.

The binaries are built by integration/build_mmseqs.sh from the reference tree and travel to the GPU box with the snapshot,
together with the createdb output of the reference's example proteins (oracle/_ref/dropin_data/examples = BASELINE.json
configs[0]).  The Rust block-aligner crate (start / CIGAR of int16-range hits, row a15) cannot be built in this image:
  oracle/_ref/mmseqs_stock       the crate's C API served by the plain-C restatement (oracle/block_oracle.c) - what a Rust-linked
                                 build computes, as far as tests/test_block_oracle.py pins the restatement
  oracle/_ref/mmseqs_stock_stub  do-nothing stubs in the crate's place: every int16-range pair takes the reference's documented
                                 Smith-Waterman fallback (StripedSmithWaterman.cpp:873-882)
  oracle/_ref/mmseqs_mmgpu       patched + libmmgpu.so, stubs in the crate's place (round 4: nothing of oracle/*.c inside): the
                                 int16-range pairs of the device path come from the device's block aligner
Runs in which every first alignment is the device's are compared with mmseqs_stock.  Runs in which the HOST's own Matcher also
aligns int16-range pairs (--realign / --alt-ali second alignments, `cluster`) can only equal a build with the same host-side
block aligner: they run with MMGPU_BLOCK_ALIGNER=sw (the fallback on the device side too) against mmseqs_stock_stub
(`reference_for`).

 * `-m gpu` tests: the patched binary runs on the device (the real mmseqs2_amd/lib/libmmgpu.so).
 * `-m "not gpu"` tests: the same binary with the C-ABI served by the CPU stand-in oracle/_build/emu/libmmgpu.so
   (LD_PRELOAD, test infrastructure) - this pins the HOST side of the drop-in (block logic, accept / reject replay,
   key mapping, masked index hand-over, serialisation) where no GPU exists.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from mmseqs2_amd import dbio
from mmseqs2_amd import workloads as wl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STOCK = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock")
MMGPU = os.path.join(ROOT, "oracle", "_ref", "mmseqs_mmgpu")
STOCK_STUB = os.path.join(ROOT, "oracle", "_ref", "mmseqs_stock_stub")
EXAMPLES = os.path.join(ROOT, "oracle", "_ref", "dropin_data", "examples")
EMU = os.path.join(ROOT, "oracle", "_build", "emu", "libmmgpu.so")

pytestmark = pytest.mark.skipif(not (os.path.exists(STOCK) and os.path.exists(MMGPU) and os.path.exists(EXAMPLES)),
                                reason="needs oracle/_ref/mmseqs_stock, mmseqs_mmgpu, dropin_data (integration/build_mmseqs.sh)")

THREADS = str(min(os.cpu_count() or 1, 32))

ALIGN_CASES = [
    ["--alignment-mode", "1"],
    ["--alignment-mode", "2"],
    ["--alignment-mode", "3"],
    ["-a"],
    ["--alignment-mode", "3", "--max-rejected", "5", "--max-accept", "3"],
    ["-a", "-e", "10", "-c", "0.5", "--cov-mode", "2"],
    ["--alignment-mode", "2", "--min-seq-id", "0.5", "--comp-bias-corr", "0"],
    ["-a", "--add-self-matches", "1"],
    # small gap penalties: queries whose composition bias leaves the regime in which the reference's striped loop equals the
    # textbook recurrence are aligned by the host's Matcher, the others on the device (found by scripts/dropin_option_sweep.py)
    ["-a", "--gap-open", "aa:9,nucl:5", "--gap-extend", "aa:2,nucl:2"],
    ["--alignment-mode", "3", "--gap-open", "aa:8,nucl:5", "--gap-extend", "aa:2,nucl:2"],
    # --realign (the first iteration of an iterative search): second alignment of the accepted hits with the biased matrix
    ["-a", "--realign", "1"],
    ["--realign", "1", "--realign-score-bias", "-0.4", "--realign-max-seqs", "5", "-c", "0.5"],
    # --alt-ali: the list on the device, the re-alignments of the masked accepted targets by the reference's own function
    ["-a", "--alt-ali", "2"],
    ["-a", "--alt-ali", "1", "--realign", "1"],
    # --corr-score-weight: the score gains weight x the autocorrelation of the per-column scores along the backtrace, the
    # E-value is recomputed over the aligned query span (StripedSmithWaterman.cpp:1221,1249-1253) - host code after the traceback
    ["-a", "--corr-score-weight", "0.5"],
]


def run(binary, args, cwd, emulate=False, extra_env=None):
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    if emulate:
        env["LD_PRELOAD"] = EMU
    if extra_env:
        env.update(extra_env)
    r = subprocess.run([binary] + args, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, "%s %s failed:\n%s" % (os.path.basename(binary), " ".join(args), r.stdout[-3000:])
    return r.stdout


def reference_for(args):
    """-> (reference binary, extra environment of the patched binary) for a command line: see the module docstring"""
    host_side = any(a in ("--realign", "--alt-ali", "cluster", "--num-iterations") for a in args)
    if host_side:
        return STOCK_STUB, {"MMGPU_BLOCK_ALIGNER": "sw"}
    return STOCK, {}


def both_modules_on_device(log):
    """a search whose prefilter and align modules both ran on the device: two device openings (the workflow script's child
    processes), or one for the fused run (plain sequence / profile-query searches, MMGpuFusedSearch)"""
    return log.count("MMGPU: device") >= 2 or ("prefilter and align run inside this process" in log and log.count("MMGPU: device") == 1)


def same(a, b):
    n, bad, msgs = dbio.diff_dbs(a, b)
    assert bad == 0, "%d of %d entries differ: %s" % (bad, n, "; ".join(msgs))
    return n


def copy_db(src, dst):
    d = os.path.dirname(src)
    base = os.path.basename(src)
    for f in os.listdir(d):
        if f == base or f.startswith(base + ".") or f.startswith(base + "_h"):
            shutil.copy(os.path.join(d, f), os.path.join(os.path.dirname(dst), os.path.basename(dst) + f[len(base):]))


def examples_pipeline(tmp, emulate):
    """QUERY.fasta vs itself with the defaults of `mmseqs search` (-s 5.7, --mask 1, --max-seqs 300, comp. bias on)"""
    w = str(tmp)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    run(STOCK, ["prefilter", "q", "q", "pref_s", "-s", "5.7", "--threads", THREADS, "-v", "2"], w)
    log = run(MMGPU, ["prefilter", "q", "q", "pref_g", "-s", "5.7", "--threads", THREADS, "-v", "3"], w, emulate)
    assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
    assert same(os.path.join(w, "pref_s"), os.path.join(w, "pref_g")) == 500
    for i, case in enumerate(ALIGN_CASES):
        ref, env = reference_for(case)
        run(ref, ["align", "q", "q", "pref_s", "aln_s%d" % i] + case + ["--threads", THREADS, "-v", "2"], w)
        log = run(MMGPU, ["align", "q", "q", "pref_s", "aln_g%d" % i] + case + ["--threads", THREADS, "-v", "3"], w, emulate, extra_env=env)
        assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
        assert same(os.path.join(w, "aln_s%d" % i), os.path.join(w, "aln_g%d" % i)) == 500, case


def test_examples_prefilter_align_host_side_emulated(tmp_path):
    """host side of both seams on real proteins, device replaced by the CPU stand-in"""
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    examples_pipeline(tmp_path, emulate=True)


def persisted_layout_pipeline(tmp, emulate):
    """MMGPU_DB_FILE (the reference's makepaddedseqdb / createindex as one file in the device layout): the first prefilter run
    builds and saves, the second loads - no upload of the lookup, no masking, no index build - and writes the same database; a
    run with other index parameters refuses the file, builds, and replaces it."""
    w = str(tmp)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    run(STOCK, ["prefilter", "q", "q", "pref_s", "-s", "5.7", "--threads", THREADS, "-v", "2"], w)
    env = {"MMGPU_DB_FILE": os.path.join(w, "q.mmgpu")}
    log = run(MMGPU, ["prefilter", "q", "q", "pref_g1", "-s", "5.7", "--threads", THREADS, "-v", "3"], w, emulate, extra_env=env)
    assert "not usable" in log and "device layout saved to" in log, log[-2000:]
    assert os.path.getsize(env["MMGPU_DB_FILE"]) > 100000
    log = run(MMGPU, ["prefilter", "q", "q", "pref_g2", "-s", "5.7", "--threads", THREADS, "-v", "3"], w, emulate, extra_env=env)
    assert "k-mer index loaded from" in log and "device layout saved to" not in log, log[-2000:]
    # ... before a sequence is mapped: the reference's fillDatabase ("Index table: counting k-mers" / "Masked residues") never ran
    assert "no sequence lookup on the host" in log and "Masked residues" not in log, log[-2000:]
    assert same(os.path.join(w, "pref_s"), os.path.join(w, "pref_g1")) == 500
    assert same(os.path.join(w, "pref_s"), os.path.join(w, "pref_g2")) == 500
    # another sensitivity = another k-mer threshold of the index: the file is refused, the run builds its own and saves it
    run(STOCK, ["prefilter", "q", "q", "pref_s4", "-s", "4", "--threads", THREADS, "-v", "2"], w)
    log = run(MMGPU, ["prefilter", "q", "q", "pref_g4", "-s", "4", "--threads", THREADS, "-v", "3"], w, emulate, extra_env=env)
    assert "made from another database or with other index parameters" in log and "device layout saved to" in log, log[-2000:]
    assert same(os.path.join(w, "pref_s4"), os.path.join(w, "pref_g4")) == 500
    # the whole search on the layout: the alignment module of the fused run finds the targets on the device and holds no copy of its
    # own - the self hits (identity pairs score their diagonal on the host) map their sequence on demand; -a: backtraces; the
    # correlation score reads every aligned target on the host
    for extra in (["-a"], ["--corr-score-weight", "0.5"]):
        tag = str(len(extra))
        run(STOCK, ["search", "q", "q", "res_s" + tag, "tmp_s" + tag, "-s", "4"] + extra + ["--threads", THREADS, "-v", "2"], w)
        log = run(MMGPU, ["search", "q", "q", "res_g" + tag, "tmp_g" + tag, "-s", "4"] + extra + ["--threads", THREADS, "-v", "3"], w, emulate,
                  extra_env=dict(env, MMGPU_TRACE="1"))
        assert "no sequence lookup on the host" in log and "targets already resident (fused search)" in log, log[-3000:]
        assert "using the CPU path" not in log, log[-3000:]
        assert same(os.path.join(w, "res_s" + tag), os.path.join(w, "res_g" + tag)) == 500
    # another target database (the same proteins, the last hundred dropped, two residues of the first one exchanged): keys and
    # lengths of the first 400 agree with the file's, the count and the sampled bytes do not - refused, rebuilt, replaced
    run(STOCK, ["convert2fasta", "q", "q.fasta", "-v", "1"], w)
    recs = open(os.path.join(w, "q.fasta")).read().split(">")[1:401]
    head, seq = recs[0].split("\n", 1)
    recs[0] = head + "\n" + seq[1] + seq[0] + seq[2:]
    open(os.path.join(w, "t2.fasta"), "w").write("".join(">" + r for r in recs))
    run(STOCK, ["createdb", "t2.fasta", "t2", "-v", "1"], w)
    run(STOCK, ["prefilter", "q", "t2", "pref_t2_s", "-s", "4", "--threads", THREADS, "-v", "2"], w)
    log = run(MMGPU, ["prefilter", "q", "t2", "pref_t2_g", "-s", "4", "--threads", THREADS, "-v", "3"], w, emulate, extra_env=env)
    assert "made from another database or with other index parameters" in log and "device layout saved to" in log, log[-2000:]
    assert same(os.path.join(w, "pref_t2_s"), os.path.join(w, "pref_t2_g")) == 500
    log = run(MMGPU, ["prefilter", "q", "t2", "pref_t2_g2", "-s", "4", "--threads", THREADS, "-v", "3"], w, emulate, extra_env=env)
    assert "no sequence lookup on the host" in log
    assert same(os.path.join(w, "pref_t2_s"), os.path.join(w, "pref_t2_g2")) == 500
    # `mmseqs makemmgpudb` (round 6; the reference's makepaddedseqdb / createindex as one command): the file is written without a
    # search, the FIRST search with it loads (no build, no save); other prefilter options = another index = refused as above
    made = os.path.join(w, "made.mmgpu")
    log = run(MMGPU, ["makemmgpudb", "q", made, "-s", "5.7", "--threads", THREADS, "-v", "3"], w, emulate)
    assert "index entries (k = 6)" in log and os.path.getsize(made) > 100000, log[-2000:]
    log = run(MMGPU, ["prefilter", "q", "q", "pref_made", "-s", "5.7", "--threads", THREADS, "-v", "3"], w, emulate, extra_env={"MMGPU_DB_FILE": made})
    assert "no sequence lookup on the host" in log and "device layout saved to" not in log, log[-2000:]
    assert same(os.path.join(w, "pref_s"), os.path.join(w, "pref_made")) == 500
    log = run(MMGPU, ["makemmgpudb", "q", made, "-s", "5.7", "--threads", THREADS, "-v", "3"], w, emulate)
    assert "already holds this database" in log, log[-2000:]
    log = run(MMGPU, ["search", "q", "q", "res_made", "tmp_made", "-s", "4", "--threads", THREADS, "-v", "3"], w, emulate, extra_env={"MMGPU_DB_FILE": made})
    assert "made from another database or with other index parameters" in log, log[-2000:]


def test_persisted_device_layout_host_side_emulated(tmp_path):
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    persisted_layout_pipeline(tmp_path, emulate=True)


@pytest.mark.gpu
def test_persisted_device_layout_on_device(tmp_path):
    persisted_layout_pipeline(tmp_path, emulate=False)


def test_disabled_binary_is_the_stock_path(tmp_path):
    """MMGPU_DISABLE=1: the patched binary must take the reference's own loops (and needs no device)"""
    w = str(tmp_path)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    run(STOCK, ["prefilter", "q", "q", "pref_s", "-s", "4", "--threads", THREADS, "-v", "2"], w)
    log = run(MMGPU, ["prefilter", "q", "q", "pref_g", "-s", "4", "--threads", THREADS, "-v", "3"], w, extra_env={"MMGPU_DISABLE": "1"})
    assert "MMGPU: device" not in log
    same(os.path.join(w, "pref_s"), os.path.join(w, "pref_g"))


@pytest.mark.gpu
def test_examples_prefilter_align_on_device(tmp_path):
    """BASELINE.json configs[0] inputs through the device: prefilter DB and 8 align configurations identical to stock"""
    examples_pipeline(tmp_path, emulate=False)


def search_workflow(tmp, emulate):
    """`mmseqs search` unchanged on the command line.  Round 4 (row f2): the plain sequence search runs its prefilter and align
    modules INSIDE the search process (integration/MMGpuFusedSearch.cpp) - one device context, the hit lists kept in memory, no
    pref_0 database on disk; MMGPU_FUSED=0 keeps blastp.sh with its two child processes; MMGPU_FUSED_PREF_ON_DISK=1 keeps the
    prefilter result as a database.  All three must give the stock binary's result database."""
    w = str(tmp)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    for s, extra in (("1", ["-a"]), ("5.7", ["-a"]), ("4", []), ("5.7", ["-e", "10", "-c", "0.5", "--cov-mode", "2", "--alignment-mode", "3"])):
        tag = s + "_" + str(len(extra))
        run(STOCK, ["search", "q", "q", "res_s" + tag, "tmp_s" + tag, "-s", s] + extra + ["--threads", THREADS, "-v", "2"], w)
        log = run(MMGPU, ["search", "q", "q", "res_g" + tag, "tmp_g" + tag, "-s", s] + extra + ["--threads", THREADS, "-v", "3"], w, emulate)
        assert "prefilter and align run inside this process" in log and log.count("MMGPU: device") == 1, log[-3000:]
        assert "using the CPU path" not in log, log[-3000:]
        assert same(os.path.join(w, "res_s" + tag), os.path.join(w, "res_g" + tag)) == 500
        assert not os.path.exists(os.path.join(w, "tmp_g" + tag, "latest", "pref_0.dbtype"))
    # MMGPU_FUSED_OVERLAP=1: the two modules side by side, with blocks small enough to interleave - the alignment module takes
    # buckets of 100 queries as soon as the prefilter module has written them (blocks of 64)
    log = run(MMGPU, ["search", "q", "q", "res_small", "tmp_small", "-s", "5.7", "-a", "--threads", THREADS, "-v", "3"], w, emulate,
              extra_env={"MMGPU_FUSED_OVERLAP": "1", "MMGPU_PREF_BLOCK_QUERIES": "64", "MMGPU_FUSED_BUCKET_QUERIES": "100"})
    assert "the alignment module starts while the prefilter module runs" in log, log[-3000:]
    assert same(os.path.join(w, "res_s5.7_1"), os.path.join(w, "res_small")) == 500
    # the workflow script with its child processes, and the fused run with the prefilter result on disk
    log = run(MMGPU, ["search", "q", "q", "res_script", "tmp_script", "-s", "5.7", "-a", "--threads", THREADS, "-v", "3"], w, emulate,
              extra_env={"MMGPU_FUSED": "0"})
    assert "inside this process" not in log and log.count("MMGPU: device") >= 2, log[-3000:]
    assert same(os.path.join(w, "res_s5.7_1"), os.path.join(w, "res_script")) == 500
    log = run(MMGPU, ["search", "q", "q", "res_disk", "tmp_disk", "-s", "5.7", "-a", "--threads", THREADS, "-v", "3"], w, emulate,
              extra_env={"MMGPU_FUSED_PREF_ON_DISK": "1"})
    assert "inside this process" in log and os.path.exists(os.path.join(w, "tmp_disk", "latest", "pref_0.dbtype")), log[-3000:]
    assert same(os.path.join(w, "res_s5.7_1"), os.path.join(w, "res_disk")) == 500
    # --remove-tmp-files and a second search into the same tmp directory
    run(MMGPU, ["search", "q", "q", "res_rm", "tmp_disk", "-s", "4", "--remove-tmp-files", "1", "--threads", THREADS, "-v", "3"], w, emulate)
    assert same(os.path.join(w, "res_s4_0"), os.path.join(w, "res_rm")) == 500


def test_search_workflow_host_side_emulated(tmp_path):
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    search_workflow(tmp_path, emulate=True)


@pytest.mark.gpu
def test_examples_search_workflow_on_device(tmp_path):
    search_workflow(tmp_path, emulate=False)
    w = str(tmp_path)
    # iterative profile search (blastpgp.sh): realigned first iteration, profile queries in the second - every prefilter /
    # align call of the workflow on the device
    it = ["--num-iterations", "2", "-s", "4", "--threads", THREADS]
    ref, env = reference_for(it)
    run(ref, ["search", "q", "q", "res_it_s", "tmp_it_s"] + it + ["-v", "2"], w)
    log = run(MMGPU, ["search", "q", "q", "res_it_g", "tmp_it_g"] + it + ["-v", "3"], w, extra_env=env)
    assert log.count("MMGPU: device") >= 4 and "using the CPU path" not in log, log[-3000:]
    assert same(os.path.join(w, "res_it_s"), os.path.join(w, "res_it_g")) == 500


def _config3_tenth(w):
    (qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=2000, members=50, n_queries=1000, seed=10)
    wl.write_fasta(os.path.join(w, "q.fasta"), qres, qoff, "q")
    wl.write_fasta(os.path.join(w, "t.fasta"), tres, toff, "t")
    run(STOCK, ["createdb", "q.fasta", "q", "-v", "1"], w)
    run(STOCK, ["createdb", "t.fasta", "t", "-v", "1"], w)


@pytest.mark.gpu
def test_config3_tenth_scale_on_device(tmp_path):
    """BASELINE.json configs[2] at 1/10 scale (1000 queries vs 100 000 family-structured targets, -s 5.7, default masking):
    prefilter DB, alignment DBs (modes 2 and 3 with backtraces) and the search result identical to the stock binary"""
    w = str(tmp_path)
    _config3_tenth(w)
    run(STOCK, ["prefilter", "q", "t", "pref_s", "-s", "5.7", "--threads", THREADS, "-v", "2"], w)
    log = run(MMGPU, ["prefilter", "q", "t", "pref_g", "-s", "5.7", "--threads", THREADS, "-v", "3"], w)
    assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
    assert same(os.path.join(w, "pref_s"), os.path.join(w, "pref_g")) == 1000
    for i, case in enumerate([["--alignment-mode", "2"], ["-a"]]):
        run(STOCK, ["align", "q", "t", "pref_s", "aln_s%d" % i] + case + ["--threads", THREADS, "-v", "2"], w)
        log = run(MMGPU, ["align", "q", "t", "pref_s", "aln_g%d" % i] + case + ["--threads", THREADS, "-v", "3"], w)
        assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
        assert same(os.path.join(w, "aln_s%d" % i), os.path.join(w, "aln_g%d" % i)) == 1000


def test_small_synthetic_host_side_emulated(tmp_path):
    """family-structured synthetic set small enough for the CPU stand-in: masked index hand-over, self hits across two
    different DBs, lists cut by --max-seqs"""
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    w = str(tmp_path)
    (qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=60, members=20, n_queries=40, seed=5)
    # low-complexity stretches so that tantan masks something
    rng = np.random.default_rng(3)
    tl = wl.split(tres, toff)
    for i in rng.choice(len(tl), 200, replace=False):
        p = int(rng.integers(0, max(1, len(tl[i]) - 40)))
        tl[i][p:p + 40] = np.tile(rng.choice(20, 2), 20)[:len(tl[i][p:p + 40])]
    tres, toff = wl.seqs_from_list(tl)
    wl.write_fasta(os.path.join(w, "q.fasta"), qres, qoff, "q")
    wl.write_fasta(os.path.join(w, "t.fasta"), tres, toff, "t")
    run(STOCK, ["createdb", "q.fasta", "q", "-v", "1"], w)
    run(STOCK, ["createdb", "t.fasta", "t", "-v", "1"], w)
    for k, extra in enumerate([[], ["--max-seqs", "7"], ["--mask", "0", "--comp-bias-corr", "0", "--min-ungapped-score", "30"]]):
        run(STOCK, ["prefilter", "q", "t", "pref_s%d" % k, "-s", "5.7", "--threads", THREADS, "-v", "2"] + extra, w)
        log = run(MMGPU, ["prefilter", "q", "t", "pref_g%d" % k, "-s", "5.7", "--threads", THREADS, "-v", "3"] + extra, w, True)
        assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
        same(os.path.join(w, "pref_s%d" % k), os.path.join(w, "pref_g%d" % k))
    run(STOCK, ["align", "q", "t", "pref_s0", "aln_s", "-a", "--threads", THREADS, "-v", "2"], w)
    run(MMGPU, ["align", "q", "t", "pref_s0", "aln_g", "-a", "--threads", THREADS, "-v", "3"], w, True)
    same(os.path.join(w, "aln_s"), os.path.join(w, "aln_g"))


def _long_case(w):
    """a database with a 40 000-residue target that carries homologs of two queries, and a 33 000-residue query"""
    rng = np.random.default_rng(11)
    (qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=40, members=20, n_queries=12, seed=8)
    qs, tl = wl.split(qres, qoff), wl.split(tres, toff)
    big = rng.choice(20, size=40000, p=wl.BACKGROUND).astype(np.uint8)
    for k, at in ((0, 1000), (1, 34000)):
        h = wl.mutate(rng, qs[k], 0.8)
        big[at:at + len(h)] = h
    tl.append(big)
    longq = rng.choice(20, size=33000, p=wl.BACKGROUND).astype(np.uint8)
    longq[5000:5000 + len(tl[3])] = tl[3]
    qs.append(longq)
    wl.write_fasta(os.path.join(w, "q.fasta"), *wl.seqs_from_list(qs), "q")
    wl.write_fasta(os.path.join(w, "t.fasta"), *wl.seqs_from_list(tl), "t")
    run(STOCK, ["createdb", "q.fasta", "q", "-v", "1"], w)
    run(STOCK, ["createdb", "t.fasta", "t", "-v", "1"], w)


def _long_pipeline(w, emulate):
    # 13 queries on 48 threads: Alignment lowers its own thread count to the number of queries (Alignment.cpp:128) while
    # the OpenMP default stays at --threads - per-thread state of the device path must follow the former
    THREADS = "48"
    _long_case(w)
    run(STOCK, ["prefilter", "q", "t", "pref_s", "-s", "5.7", "--threads", THREADS, "-v", "2"], w)
    log = run(MMGPU, ["prefilter", "q", "t", "pref_g", "-s", "5.7", "--threads", THREADS, "-v", "3"], w, emulate)
    assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
    assert same(os.path.join(w, "pref_s"), os.path.join(w, "pref_g")) == 13
    # the long target is found by the queries planted in it, the long query finds its planted target
    pref = dbio.read_db(os.path.join(w, "pref_s"))
    long_key = [line.split("\t")[0] for line in open(os.path.join(w, "t.index")) if int(line.split("\t")[2]) > 40000][0].encode()
    assert sum(any(l.split(b"\t")[0] == long_key for l in e.split(b"\n") if l) for e in pref.values()) >= 2
    run(STOCK, ["align", "q", "t", "pref_s", "aln_s", "-a", "--threads", THREADS, "-v", "2"], w)
    log = run(MMGPU, ["align", "q", "t", "pref_s", "aln_g", "-a", "--threads", THREADS, "-v", "3"], w, emulate)
    assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
    assert same(os.path.join(w, "aln_s"), os.path.join(w, "aln_g")) == 13


def test_long_sequences_host_side_emulated(tmp_path):
    """sequences of 32768 residues or more: the prefilter hands those queries back to the host matcher (computeLongScore),
    the alignment runs them on the device path"""
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    _long_pipeline(str(tmp_path), True)


@pytest.mark.gpu
def test_long_sequences_on_device(tmp_path):
    _long_pipeline(str(tmp_path), False)


def _profile_pipeline(w, emulate):
    """Profile queries through `mmseqs align` (alignment half): a profile database made by the stock
    binary (search -> result2profile) searched against the sequences: prefilter (similar k-mers from the profile's own
    sorted score rows, ungapped scores from its alignment profile) and alignment (the profile's score rows) on the device."""
    (qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=50, members=20, n_queries=30, seed=21)
    wl.write_fasta(os.path.join(w, "q.fasta"), qres, qoff, "q")
    wl.write_fasta(os.path.join(w, "t.fasta"), tres, toff, "t")
    run(STOCK, ["createdb", "q.fasta", "q", "-v", "1"], w)
    run(STOCK, ["createdb", "t.fasta", "t", "-v", "1"], w)
    run(STOCK, ["search", "q", "t", "res0", "tmp0", "-s", "5.7", "-a", "--threads", THREADS, "-v", "1"], w)
    run(STOCK, ["result2profile", "q", "t", "res0", "prof", "--threads", THREADS, "-v", "1"], w)
    run(STOCK, ["prefilter", "prof", "t", "pref_p", "-s", "5.7", "--threads", THREADS, "-v", "1"], w)
    # the prefilter of profile queries: k-mers from the profile's own score rows, index built with threshold 0
    for k, extra in enumerate([[], ["--max-seqs", "9"]]):
        run(STOCK, ["prefilter", "prof", "t", "ppref_s%d" % k, "-s", "5.7", "--threads", THREADS, "-v", "2"] + extra, w)
        log = run(MMGPU, ["prefilter", "prof", "t", "ppref_g%d" % k, "-s", "5.7", "--threads", THREADS, "-v", "3"] + extra, w, emulate)
        assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
        assert same(os.path.join(w, "ppref_s%d" % k), os.path.join(w, "ppref_g%d" % k)) == 30
    for i, case in enumerate([["--alignment-mode", "1"], ["-a"], ["--alignment-mode", "3", "-e", "10", "-c", "0.3"]]):
        run(STOCK, ["align", "prof", "t", "pref_p", "paln_s%d" % i] + case + ["--threads", THREADS, "-v", "2"], w)
        log = run(MMGPU, ["align", "prof", "t", "pref_p", "paln_g%d" % i] + case + ["--threads", THREADS, "-v", "3"], w, emulate)
        assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
        assert same(os.path.join(w, "paln_s%d" % i), os.path.join(w, "paln_g%d" % i)) == 30, case
    # profile TARGETS: sequences against the profile database - the index holds the profiles' similar k-mers (host-built, handed
    # over), the queries match exactly, the ungapped scores are taken on the consensus sequences
    for k, extra in enumerate([[], ["--max-seqs", "7"]]):
        run(STOCK, ["prefilter", "t", "prof", "tpref_s%d" % k, "-s", "5.7", "--threads", THREADS, "-v", "2"] + extra, w)
        log = run(MMGPU, ["prefilter", "t", "prof", "tpref_g%d" % k, "-s", "5.7", "--threads", THREADS, "-v", "3"] + extra, w, emulate)
        assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
        assert same(os.path.join(w, "tpref_s%d" % k), os.path.join(w, "tpref_g%d" % k)) == 1000
    run(STOCK, ["search", "t", "prof", "tres_s", "ttmp_s", "-s", "5.7", "-a", "--threads", THREADS, "-v", "1"], w)
    log = run(MMGPU, ["search", "t", "prof", "tres_g", "ttmp_g", "-s", "5.7", "-a", "--threads", THREADS, "-v", "3"], w, emulate)
    assert "MMGPU: device" in log, log[-3000:]
    assert same(os.path.join(w, "tres_s"), os.path.join(w, "tres_g")) == 1000
    # the whole profile search through the patched binary: both stages on the device
    run(STOCK, ["search", "prof", "t", "pres_s", "ptmp_s", "-s", "5.7", "-a", "--threads", THREADS, "-v", "1"], w)
    log = run(MMGPU, ["search", "prof", "t", "pres_g", "ptmp_g", "-s", "5.7", "-a", "--threads", THREADS, "-v", "3"], w, emulate)
    assert both_modules_on_device(log) and "using the CPU path" not in log, log[-3000:]
    assert same(os.path.join(w, "pres_s"), os.path.join(w, "pres_g")) == 30


def test_profile_queries_host_side_emulated(tmp_path):
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    _profile_pipeline(str(tmp_path), emulate=True)


@pytest.mark.gpu
def test_profile_queries_on_device(tmp_path):
    _profile_pipeline(str(tmp_path), emulate=False)


def _translated_pipeline(w, emulate):
    """Translated search (the 6-frame half): nucleotide queries against protein targets run
    extractorfs -> translatenucs -> prefilter -> align -> offsetalignment (data/workflow/translated_search.sh); the two
    middle stages are amino-acid databases and go through the device hooks unchanged."""
    rng = np.random.default_rng(17)
    (qres, qoff), (tres, toff), _, _ = wl.config3_prefilter(n_families=40, members=15, n_queries=24, seed=33)
    # back-translate the protein queries (one codon per residue, random synonymous choice), half of them reverse-complemented,
    # embedded in random flanks
    aa = "ACDEFGHIKLMNPQRSTVWYX"
    codons = {"A": ["GCT", "GCC", "GCA", "GCG"], "C": ["TGT", "TGC"], "D": ["GAT", "GAC"], "E": ["GAA", "GAG"], "F": ["TTT", "TTC"],
              "G": ["GGT", "GGC", "GGA", "GGG"], "H": ["CAT", "CAC"], "I": ["ATT", "ATC", "ATA"], "K": ["AAA", "AAG"],
              "L": ["TTA", "TTG", "CTT", "CTC", "CTA", "CTG"], "M": ["ATG"], "N": ["AAT", "AAC"], "P": ["CCT", "CCC", "CCA", "CCG"],
              "Q": ["CAA", "CAG"], "R": ["CGT", "CGC", "CGA", "CGG", "AGA", "AGG"], "S": ["TCT", "TCC", "TCA", "TCG", "AGT", "AGC"],
              "T": ["ACT", "ACC", "ACA", "ACG"], "V": ["GTT", "GTC", "GTA", "GTG"], "W": ["TGG"], "Y": ["TAT", "TAC"], "X": ["NNN"]}
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    order = wl.NUM2AA
    qs = wl.split(qres, qoff)
    with open(os.path.join(w, "qn.fasta"), "w") as f:
        for i, q in enumerate(qs):
            prot = "".join(order[int(c)] if order[int(c)] in codons else "X" for c in q)
            nt = "".join(codons[c][int(rng.integers(len(codons[c])))] for c in prot)
            nt = "".join(rng.choice(list("ACGT"), 30)) + "ATG" + nt + "TAA" + "".join(rng.choice(list("ACGT"), 30))
            if i % 2:
                nt = "".join(comp[c] for c in reversed(nt))
            f.write(">qn%d\n%s\n" % (i, nt))
    wl.write_fasta(os.path.join(w, "t.fasta"), tres, toff, "t")
    run(STOCK, ["createdb", "qn.fasta", "qn", "-v", "1"], w)
    run(STOCK, ["createdb", "t.fasta", "t", "-v", "1"], w)
    run(STOCK, ["search", "qn", "t", "tres_s", "ttmp_s", "-s", "5.7", "-a", "--threads", THREADS, "-v", "1"], w)
    log = run(MMGPU, ["search", "qn", "t", "tres_g", "ttmp_g", "-s", "5.7", "-a", "--threads", THREADS, "-v", "3"], w, emulate)
    assert both_modules_on_device(log) and "using the CPU path" not in log, log[-3000:]
    # (entry by entry; lines of an entry whose bit score and E-value tie are written by `offsetalignment` in a thread-dependent order -
    # two runs of the STOCK binary differ that way at configs[4]'s size, bench.py `translated_search` - so such runs of lines are
    # compared as sets)
    n, bad, _, msgs = dbio.diff_dbs_up_to_tie_order(os.path.join(w, "tres_s"), os.path.join(w, "tres_g"))
    assert bad == 0, msgs
    assert n == 24
    # the search found the planted proteins: every query has hits
    d = dbio.read_db(os.path.join(w, "tres_g"))
    assert sum(1 for v in d.values() if len(v) > 1) >= 20


def test_translated_search_host_side_emulated(tmp_path):
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    _translated_pipeline(str(tmp_path), emulate=True)


@pytest.mark.gpu
def test_translated_search_on_device(tmp_path):
    _translated_pipeline(str(tmp_path), emulate=False)


def _nucleotide_pipeline(w, emulate, k="15"):
    """Nucleotide search: `mmseqs prefilter` of nucleotide databases with the search's parameters
    (exact k-mers, k = 15, Search.cpp:180-198) on the device, and the whole blastn workflow (`search --search-type 3`:
    extractframes / splitsequence -> prefilter -> align -> offsetalignment) through the patched binary: prefilter on the
    device, the banded nucleotide alignment on the device as well (the reference's loop where its conditions are not met)."""
    rng = np.random.default_rng(23)
    letters = np.array(list("ACGT"))
    targets = ["".join(letters[rng.integers(0, 4, int(rng.integers(3000, 12000)))]) for _ in range(60)]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    queries = []
    for i in range(30):
        t = targets[int(rng.integers(0, len(targets)))]
        L = int(rng.integers(300, 2500))
        a = int(rng.integers(0, len(t) - L))
        piece = list(t[a:a + L])
        for p in np.nonzero(rng.random(L) < 0.06)[0]:
            piece[p] = "ACGT"[int(rng.integers(0, 4))]
        s = "".join(piece)
        if i % 2:
            s = "".join(comp[c] for c in reversed(s))
        queries.append(s)
    with open(os.path.join(w, "nq.fasta"), "w") as f:
        for i, s in enumerate(queries):
            f.write(">q%d\n%s\n" % (i, s))
    with open(os.path.join(w, "nt.fasta"), "w") as f:
        for i, s in enumerate(targets):
            f.write(">t%d\n%s\n" % (i, s))
    run(STOCK, ["createdb", "nq.fasta", "nq", "-v", "1"], w)
    run(STOCK, ["createdb", "nt.fasta", "nt", "-v", "1"], w)
    # the reverse strands as separate query entries, like the workflow's extractframes step
    run(STOCK, ["extractframes", "nq", "nqf", "--forward-frames", "1", "--reverse-frames", "1", "--threads", THREADS, "-v", "1"], w)
    # (k: the search's own 15 on the device; the CPU stand-in runs the same code at k = 12 - 4^15 offsets per process are slow there)
    for n, extra in enumerate([[], ["--max-seqs", "4"], ["-k", "11", "--spaced-kmer-mode", "0"]]):
        args = ["--exact-kmer-matching", "1", "--max-seq-len", "10000", "--threads", THREADS]
        if "-k" not in extra:
            args += ["-k", k]
        run(STOCK, ["prefilter", "nqf", "nt", "npref_s%d" % n] + args + extra + ["-v", "2"], w)
        log = run(MMGPU, ["prefilter", "nqf", "nt", "npref_g%d" % n] + args + extra + ["-v", "3"], w, emulate)
        assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
        assert same(os.path.join(w, "npref_s%d" % n), os.path.join(w, "npref_g%d" % n)) == 60
    d = dbio.read_db(os.path.join(w, "npref_g0"))
    assert sum(1 for v in d.values() if len(v) > 1) >= 25          # every read finds its contig on one of the two strands
    # `mmseqs align` of the nucleotide lists: BandedNucleotideAligner::align on the device (integration/MMGpuNuclAlignRun.cpp).  The
    # reference reads one residue past the end of its per-thread buffers, so its own output depends on what a thread mapped
    # before: the comparison is with its one-thread run (which is what the hook replays), whatever --threads the patched binary has
    # (--wrapped-scoring: the queries go to the aligner written twice - circular sequences; round 4: on the device)
    for n, extra in enumerate([["-a"], ["--alignment-mode", "3", "-c", "0.5"], ["-e", "1e-5", "--min-seq-id", "0.9", "-a"],
                               ["--cov-mode", "2", "-c", "0.3", "--alignment-output-mode", "0"], ["--wrapped-scoring", "1", "-a"],
                               ["--wrapped-scoring", "1", "-c", "0.8", "--cov-mode", "2"]]):
        run(STOCK, ["align", "nqf", "nt", "npref_s0", "naln_s%d" % n, "--threads", "1", "-v", "1"] + extra, w)
        log = run(MMGPU, ["align", "nqf", "nt", "npref_s0", "naln_g%d" % n, "--threads", THREADS, "-v", "3"] + extra, w, emulate)
        assert "MMGPU: nucleotide alignment on the device" in log, log[-2000:]
        assert same(os.path.join(w, "naln_s%d" % n), os.path.join(w, "naln_g%d" % n)) == 60
    d = dbio.read_db(os.path.join(w, "naln_g0"))
    assert sum(1 for v in d.values() if len(v) > 1) >= 25
    # buckets of seven queries: what shows through a buffer comes from queries of buckets that are gone by then (the hook's
    # histories keep their own copies)
    log = run(MMGPU, ["align", "nqf", "nt", "npref_s0", "naln_g0_buckets", "--threads", THREADS, "-v", "3", "-a"], w, emulate,
              extra_env={"MMGPU_ALIGN_BLOCK_QUERIES": "7"})
    assert "MMGPU: nucleotide alignment on the device" in log, log[-2000:]
    assert same(os.path.join(w, "naln_s0"), os.path.join(w, "naln_g0_buckets")) == 60
    # where the loop stops decides what the reference's buffers hold afterwards: finite --max-rejected stays on the CPU loop
    run(STOCK, ["align", "nqf", "nt", "npref_s0", "naln_s9", "--threads", "1", "--max-rejected", "3", "-v", "1"], w)
    log = run(MMGPU, ["align", "nqf", "nt", "npref_s0", "naln_g9", "--threads", "1", "--max-rejected", "3", "-v", "3"], w, emulate)
    assert "MMGPU: nucleotide alignment on the device" not in log and "using the CPU path" in log, log[-2000:]
    assert same(os.path.join(w, "naln_s9"), os.path.join(w, "naln_g9")) == 60
    run(STOCK, ["search", "nq", "nt", "nres_s", "ntmp_s", "--search-type", "3", "-k", k, "-a", "--threads", "1", "-v", "1"], w)
    log = run(MMGPU, ["search", "nq", "nt", "nres_g", "ntmp_g", "--search-type", "3", "-k", k, "-a", "--threads", THREADS, "-v", "3"], w, emulate)
    assert "MMGPU: device" in log and "MMGPU: nucleotide alignment on the device" in log, log[-3000:]
    assert same(os.path.join(w, "nres_s"), os.path.join(w, "nres_g")) == 30


def test_nucleotide_search_host_side_emulated(tmp_path):
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    _nucleotide_pipeline(str(tmp_path), emulate=True, k="12")


@pytest.mark.gpu
def test_nucleotide_search_on_device(tmp_path):
    _nucleotide_pipeline(str(tmp_path), emulate=False)


# ---------------------------------------------------------------------------------------------------------------------
# --diag-score 0: the prefilter scores are double-k-mer-match counts (no ungapped scoring; mmgpu_pf_params::kmer_score)
def kmer_score_pipeline(tmp, emulate):
    w = str(tmp)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    n_entries = 0
    for i, extra in enumerate((["--min-ungapped-score", "15"], ["--min-ungapped-score", "2", "--max-seqs", "20"],
                               ["--min-ungapped-score", "1", "--mask", "0", "--comp-bias-corr", "0"],
                               # what `mmseqs cluster` runs first (Cluster.cpp:225-228)
                               ["--min-ungapped-score", "0", "--comp-bias-corr", "0", "-s", "1"])):
        args = (["-s", "5.7"] if "-s" not in extra else []) + ["--diag-score", "0"] + extra + ["--threads", THREADS]
        run(STOCK, ["prefilter", "q", "q", "pref_s%d" % i] + args + ["-v", "2"], w)
        log = run(MMGPU, ["prefilter", "q", "q", "pref_g%d" % i] + args + ["-v", "3"], w, emulate)
        assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
        assert same(os.path.join(w, "pref_s%d" % i), os.path.join(w, "pref_g%d" % i)) == 500
        n_entries += sum(int(line.split()[2]) > 1 for line in open(os.path.join(w, "pref_g%d.index" % i)))
    assert n_entries > 1000      # most queries have a non-empty list (at least their self hit)


def test_kmer_score_prefilter_host_side_emulated(tmp_path):
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    kmer_score_pipeline(tmp_path, emulate=True)


@pytest.mark.gpu
def test_kmer_score_prefilter_on_device(tmp_path):
    kmer_score_pipeline(tmp_path, emulate=False)


# ---------------------------------------------------------------------------------------------------------------------
# --split: the reference's way to a database larger than memory - runSplit once per target split (reduced list length per
# split, merged by the reference's mergeTargetSplits afterwards) or per query split; the hook sees one split at a time
SPLIT_CASES = [["--split", "3", "--split-mode", "0"], ["--split", "2", "--split-mode", "1"],
               ["--split", "4", "--split-mode", "0", "--max-seqs", "50"],
               # the index holds the similar k-mers of the targets, the queries match exactly (IndexBuilder.cpp:63): the host's
               # index is handed over as it is (found by scripts/dropin_option_sweep.py: the device-built index was the plain one)
               # other matrices / alphabets / k through the same seam
               ["--seed-sub-mat", "aa:VTML40.out,nucl:nucleotide.out", "--alph-size", "aa:13,nucl:5"]]
SLOW_CASES = [["-k", "7", "--spaced-kmer-mode", "0", "-s", "4"],      # k = 7 tables: minutes through the CPU stand-in, device only
              ["--target-search-mode", "1"]]                           # (30 s through the stand-in: the similar-k-mer index is large)


def split_pipeline(tmp, emulate):
    w = str(tmp)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    for i, extra in enumerate(SPLIT_CASES + ([] if emulate else SLOW_CASES)):
        args = (["-s", "5.7"] if "-s" not in extra else []) + extra + ["--threads", THREADS]
        run(STOCK, ["prefilter", "q", "q", "pref_s%d" % i] + args + ["-v", "2"], w)
        log = run(MMGPU, ["prefilter", "q", "q", "pref_g%d" % i] + args + ["-v", "3"], w, emulate)
        assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
        assert same(os.path.join(w, "pref_s%d" % i), os.path.join(w, "pref_g%d" % i)) == 500, extra


def test_split_modes_host_side_emulated(tmp_path):
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    split_pipeline(tmp_path, emulate=True)


@pytest.mark.gpu
def test_split_modes_on_device(tmp_path):
    split_pipeline(tmp_path, emulate=False)


def test_block_too_large_for_the_device_is_cut_and_retried(tmp_path):
    """a batch the device cannot hold (out of HBM / 2^32 index entries) is halved and retried by the prefilter hook; the
    stand-in plays a device that holds at most 37 queries per batch (MMGPU_EMU_MAX_BATCH)"""
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    w = str(tmp_path)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    args = ["-s", "4", "--threads", THREADS]
    run(STOCK, ["prefilter", "q", "q", "pref_s"] + args + ["-v", "2"], w)
    log = run(MMGPU, ["prefilter", "q", "q", "pref_g"] + args + ["-v", "3"], w, True, extra_env={"MMGPU_EMU_MAX_BATCH": "37"})
    assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
    assert same(os.path.join(w, "pref_s"), os.path.join(w, "pref_g")) == 500


# ---------------------------------------------------------------------------------------------------------------------
# `mmseqs cluster` (cascaded clustering): its first prefilter runs with --diag-score 0 --min-ungapped-score 0 --comp-bias-corr 0
# (Cluster.cpp:225-228), the later steps with rising sensitivity; every prefilter / align call of the workflow on the device
def cluster_pipeline(tmp, emulate):
    w = str(tmp)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    for i, a in enumerate((["--min-seq-id", "0.3", "-s", "4"], ["-c", "0.9", "--cov-mode", "1", "--cluster-mode", "2"])):
        ref, env = reference_for(["cluster"])
        run(ref, ["cluster", "q", "clu_s%d" % i, "tmp_s%d" % i] + a + ["--threads", THREADS, "-v", "2"], w)
        log = run(MMGPU, ["cluster", "q", "clu_g%d" % i, "tmp_g%d" % i] + a + ["--threads", THREADS, "-v", "3"], w, emulate, extra_env=env)
        assert log.count("MMGPU: device") >= 6 and "using the CPU path" not in log, log[-3000:]
        assert same(os.path.join(w, "clu_s%d" % i), os.path.join(w, "clu_g%d" % i)) > 400


def test_cluster_workflow_host_side_emulated(tmp_path):
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    cluster_pipeline(tmp_path, emulate=True)


@pytest.mark.gpu
def test_cluster_workflow_on_device(tmp_path):
    cluster_pipeline(tmp_path, emulate=False)


# ---------------------------------------------------------------------------------------------------------------------
# a target database with a precomputed index (`mmseqs createindex`): Prefiltering reads IndexTable / SequenceLookup from the
# .idx file (PrefilteringIndexReader) and the hook hands them over as they are
def indexed_target_pipeline(tmp, emulate):
    w = str(tmp)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    copy_db(EXAMPLES, os.path.join(w, "t"))
    run(STOCK, ["createindex", "t", "tmpidx", "--threads", THREADS, "-v", "2"], w)
    assert os.path.exists(os.path.join(w, "t.idx"))
    args = ["-s", "5.7", "-a", "--threads", THREADS]
    run(STOCK, ["search", "q", "t", "res_s", "tmp_s"] + args + ["-v", "2"], w)
    log = run(MMGPU, ["search", "q", "t", "res_g", "tmp_g"] + args + ["-v", "3"], w, emulate)
    assert both_modules_on_device(log) and "using the CPU path" not in log, log[-3000:]
    assert same(os.path.join(w, "res_s"), os.path.join(w, "res_g")) == 500


def test_indexed_target_db_host_side_emulated(tmp_path):
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    indexed_target_pipeline(tmp_path, emulate=True)


@pytest.mark.gpu
def test_indexed_target_db_on_device(tmp_path):
    indexed_target_pipeline(tmp_path, emulate=False)


# ---------------------------------------------------------------------------------------------------------------------
# `mmseqs lcaalign` (the alignment module of `mmseqs taxonomy`'s approximate 2bLCA mode, Alignment.cpp:37-43,444-498): score-only
# pass over the list, realignment of the top hit, then the aligned stretch of the top hit's target as the query against
# every entry of the list under the top hit's E-value - three device calls per block
def lca_pipeline(tmp, emulate):
    w = str(tmp)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    run(STOCK, ["prefilter", "q", "q", "pref", "-s", "5.7", "--threads", THREADS, "-v", "2"], w)
    filled = 0
    for i, extra in enumerate(([], ["-e", "1e-5", "--max-rejected", "5"], ["-c", "0.5", "--cov-mode", "0"])):
        args = extra + ["--threads", THREADS]
        run(STOCK, ["lcaalign", "q", "q", "pref", "lca_s%d" % i] + args + ["-v", "2"], w)
        log = run(MMGPU, ["lcaalign", "q", "q", "pref", "lca_g%d" % i] + args + ["-v", "3"], w, emulate)
        assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
        assert same(os.path.join(w, "lca_s%d" % i), os.path.join(w, "lca_g%d" % i)) == 500, extra
        filled += sum(int(line.split()[2]) > 1 for line in open(os.path.join(w, "lca_g%d.index" % i)))
    assert filled >= 900       # the results are not empty


def test_lcaalign_host_side_emulated(tmp_path):
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    lca_pipeline(tmp_path, emulate=True)


@pytest.mark.gpu
def test_lcaalign_on_device(tmp_path):
    lca_pipeline(tmp_path, emulate=False)


# ---------------------------------------------------------------------------------------------------------------------
# compressed databases (--compressed 1, zstd per entry): inputs read through the reference's DBReader, outputs written through
# its DBWriter; compared after `mmseqs decompress` (the index of a compressed DB holds uncompressed lengths)
def compressed_pipeline(tmp, emulate):
    w = str(tmp)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    run(STOCK, ["compress", "q", "qc", "--threads", THREADS, "-v", "1"], w)
    for f in os.listdir(w):
        if f.startswith("q_h"):
            shutil.copy(os.path.join(w, f), os.path.join(w, "qc" + f[1:]))
    run(STOCK, ["prefilter", "qc", "qc", "pref_s", "-s", "5.7", "--compressed", "1", "--threads", THREADS, "-v", "2"], w)
    log = run(MMGPU, ["prefilter", "qc", "qc", "pref_g", "-s", "5.7", "--compressed", "1", "--threads", THREADS, "-v", "3"], w, emulate)
    assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
    run(STOCK, ["align", "qc", "qc", "pref_s", "aln_s", "-a", "--compressed", "1", "--threads", THREADS, "-v", "2"], w)
    log = run(MMGPU, ["align", "qc", "qc", "pref_s", "aln_g", "-a", "--compressed", "1", "--threads", THREADS, "-v", "3"], w, emulate)
    assert "MMGPU: device" in log and "using the CPU path" not in log, log[-2000:]
    for a, b in (("pref_s", "pref_g"), ("aln_s", "aln_g")):
        run(STOCK, ["decompress", a, a + "_d", "--threads", THREADS, "-v", "1"], w)
        run(STOCK, ["decompress", b, b + "_d", "--threads", THREADS, "-v", "1"], w)
        assert same(os.path.join(w, a + "_d"), os.path.join(w, b + "_d")) == 500


def test_compressed_databases_host_side_emulated(tmp_path):
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    compressed_pipeline(tmp_path, emulate=True)


@pytest.mark.gpu
def test_compressed_databases_on_device(tmp_path):
    compressed_pipeline(tmp_path, emulate=False)


def _several_device_contexts(tmp_path, emulate):
    """MMGPU_DEVICES=0,0,0: three device contexts in one `mmseqs` process (the box has one GPU; the library then uses its copy
    transport instead of RCCL).  `prefilter`: the targets dealt to the contexts by length bucket, one k-mer index each, the hit
    lists exchanged and merged inside the library - the result database must equal the stock binary's (= the unsplit run).
    `align -a`: the targets on every context, the queries of each block dealt to them."""
    w = str(tmp_path)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    env = {"MMGPU_DEVICES": "0,0,0"}
    run(STOCK, ["prefilter", "q", "q", "pref_s", "-s", "5.7", "--threads", THREADS, "-v", "2"], w)
    log = run(MMGPU, ["prefilter", "q", "q", "pref_g", "-s", "5.7", "--threads", THREADS, "-v", "3"], w, emulate, extra_env=env)
    assert "3 device contexts" in log and "runs on one device" not in log and "using the CPU path" not in log, log[-2000:]
    assert same(os.path.join(w, "pref_s"), os.path.join(w, "pref_g")) == 500
    for i, case in enumerate([["-a"], ["--alignment-mode", "2"]]):
        run(STOCK, ["align", "q", "q", "pref_s", "aln_s%d" % i] + case + ["--threads", THREADS, "-v", "2"], w)
        log = run(MMGPU, ["align", "q", "q", "pref_s", "aln_g%d" % i] + case + ["--threads", THREADS, "-v", "3"], w, emulate, extra_env=env)
        assert "3 device contexts" in log and "using the CPU path" not in log, log[-2000:]
        assert same(os.path.join(w, "aln_s%d" % i), os.path.join(w, "aln_g%d" % i)) == 500, case


def test_several_device_contexts_host_side_emulated(tmp_path):
    """(the stand-in library answers a multi-context prefilter from one context - what runs here is the hooks' side: context
    layout, shard-capable / not, the alignment hook dealing its queries to three contexts)"""
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    _several_device_contexts(tmp_path, emulate=True)


@pytest.mark.gpu
def test_several_device_contexts_in_one_process(tmp_path):
    _several_device_contexts(tmp_path, emulate=False)


@pytest.mark.gpu
def test_sharded_prefilter_reruns_flagged_queries_on_the_device(tmp_path):
    """MMGPU_DEVICES=0,0 with the database-hit buffer cut down (MMGPU_PF_MAX_DB_MATCHES, the library's test aid) so that queries of the
    500-sequence example database reach a shard's share of it: such queries come back flagged "shard-dependent order" - and are run once
    more against the unsplit database ON THE DEVICE (mmgpu_multi_pf_run / mmgpu_multi_pf_fetch), never through a host index
    (one flagged query made the run build the host's IndexTable).  The result equals the one-context run under the same
    buffer size (the stock binary has no such knob)."""
    w = str(tmp_path)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    small = {"MMGPU_PF_MAX_DB_MATCHES": "1500"}
    log1 = run(MMGPU, ["prefilter", "q", "q", "pref_1", "-s", "5.7", "--threads", THREADS, "-v", "3"], w, extra_env=small)
    assert "using the CPU path" not in log1, log1[-2000:]
    log2 = run(MMGPU, ["prefilter", "q", "q", "pref_2", "-s", "5.7", "--threads", THREADS, "-v", "3"], w,
               extra_env=dict(small, MMGPU_DEVICES="0,0", MMGPU_QUERY_GROUPS="1"))
    assert "1 query group x 2 target shards" in log2, log2[-2500:]
    assert "ran once more against the unsplit database on the device" in log2, log2[-2500:]
    assert "building the host index" not in log2 and "shard-dependent order" not in log2, log2[-2500:]
    assert same(os.path.join(w, "pref_1"), os.path.join(w, "pref_2")) == 500


def _query_groups_by_target_shards(tmp_path, emulate):
    """MMGPU_DEVICES=0,0,0,0: the prefilter hook lays the four contexts out as G query groups x S target shards (MMGpuRun::queryGroups:
    2 x 2 by the stage model, or MMGPU_QUERY_GROUPS).  Every group holds the whole database dealt to its S contexts - tantan-masked on
    the device shard by shard, one k-mer index per context, lists exchanged and merged inside the group - and the query blocks (64
    queries here, so that every group gets several) are dealt to the groups.  The result database must equal the stock binary's for
    every layout; configurations without target shards (--diag-score 0, a profile query database) run as four groups of one."""
    w = str(tmp_path)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    run(STOCK, ["prefilter", "q", "q", "pref_s", "-s", "5.7", "--threads", THREADS, "-v", "2"], w)
    for groups, layout in ((None, "2 query groups x 2 target shards"), ("1", "1 query group x 4 target shards"),
                           ("4", "4 query groups x 1 target shard,")):
        env = {"MMGPU_DEVICES": "0,0,0,0", "MMGPU_PREF_BLOCK_QUERIES": "64"}
        if groups is not None:
            env["MMGPU_QUERY_GROUPS"] = groups
        out = "pref_g" + (groups or "auto")
        log = run(MMGPU, ["prefilter", "q", "q", out, "-s", "5.7", "--threads", THREADS, "-v", "3"], w, emulate, extra_env=env)
        assert layout in log and "using the CPU path" not in log and "tantan on the device" in log, log[-2500:]
        assert same(os.path.join(w, "pref_s"), os.path.join(w, out)) == 500, layout
    env = {"MMGPU_DEVICES": "0,0,0,0", "MMGPU_PREF_BLOCK_QUERIES": "64"}
    run(STOCK, ["prefilter", "q", "q", "pref_s0", "-s", "4", "--diag-score", "0", "--threads", THREADS, "-v", "2"], w)
    log = run(MMGPU, ["prefilter", "q", "q", "pref_g0", "-s", "4", "--diag-score", "0", "--threads", THREADS, "-v", "3"], w, emulate, extra_env=env)
    assert "runs without target shards" in log and "4 query groups x 1 target shard," in log and "using the CPU path" not in log, log[-2500:]
    assert same(os.path.join(w, "pref_s0"), os.path.join(w, "pref_g0")) == 500
    # a target split with more sequences than one context indexes (8 388 608; 200 for this test) on ONE device: the hook opens as
    # many contexts on it as it takes and merges their lists
    log = run(MMGPU, ["prefilter", "q", "q", "pref_v", "-s", "5.7", "--threads", THREADS, "-v", "3"], w, emulate,
              extra_env={"MMGPU_TEST_MAX_TARGETS": "200", "MMGPU_PREF_BLOCK_QUERIES": "64"})
    assert "3 device contexts (one device" in log and "1 query group x 3 target shards" in log and "using the CPU path" not in log, log[-2500:]
    assert same(os.path.join(w, "pref_s"), os.path.join(w, "pref_v")) == 500
    log = run(MMGPU, ["search", "q", "q", "res_v", "tmp_v", "-s", "5.7", "-a", "--threads", THREADS, "-v", "3"], w, emulate,
              extra_env={"MMGPU_TEST_MAX_TARGETS": "200"})
    assert "1 query group x 3 target shards" in log and "using the CPU path" not in log, log[-2500:]
    # the whole search: the alignment module of the same process deals its queries to the four contexts the prefilter opened
    run(STOCK, ["search", "q", "q", "res_s", "tmp_s", "-s", "5.7", "-a", "--threads", THREADS, "-v", "2"], w)
    log = run(MMGPU, ["search", "q", "q", "res_g", "tmp_g", "-s", "5.7", "-a", "--threads", THREADS, "-v", "3"], w, emulate, extra_env=env)
    assert "2 query groups x 2 target shards" in log and "using the CPU path" not in log, log[-2500:]
    assert same(os.path.join(w, "res_s"), os.path.join(w, "res_g")) == 500
    assert same(os.path.join(w, "res_s"), os.path.join(w, "res_v")) == 500


def test_query_groups_by_target_shards_host_side_emulated(tmp_path):
    """(as above: the hooks' layout of groups, helper threads and blocks, the large-split path, the fused search over four contexts)"""
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    _query_groups_by_target_shards(tmp_path, emulate=True)


@pytest.mark.gpu
def test_query_groups_by_target_shards_in_one_process(tmp_path):
    _query_groups_by_target_shards(tmp_path, emulate=False)


def _block_aligner_modes(tmp, emulate):
    """int16-range hits (most true homologs): the stock binary of this image runs the RESTATED block aligner behind the crate's C
    API (integration/build_mmseqs.sh).  The patched binary is linked with do-nothing stubs in the crate's place (round 4: nothing of
    oracle/*.c on its link line), so every such pair is answered by the device's block aligner - blocks up to the crate's 4096
    rows - and the databases must equal the stock binary's.  MMGPU_BLOCK_ALIGNER=sw selects the reference's documented fallback
    (reverse scan + banded traceback) for all of them: a different, equally scoring answer for some."""
    w = str(tmp)
    copy_db(EXAMPLES, os.path.join(w, "q"))
    run(STOCK, ["prefilter", "q", "q", "pref_s", "-s", "4", "--threads", THREADS, "-v", "2"], w)
    run(STOCK, ["align", "q", "q", "pref_s", "aln_s", "-a", "--threads", THREADS, "-v", "2"], w)
    log = run(MMGPU, ["align", "q", "q", "pref_s", "aln_device", "-a", "--threads", THREADS, "-v", "3"], w, emulate)
    assert "using the CPU path" not in log and "Block alignment failed" not in log, log[-2000:]
    assert same(os.path.join(w, "aln_s"), os.path.join(w, "aln_device")) == 500
    run(MMGPU, ["align", "q", "q", "pref_s", "aln_sw", "-a", "--threads", THREADS, "-v", "3"], w, emulate, extra_env={"MMGPU_BLOCK_ALIGNER": "sw"})
    n, bad, _ = dbio.diff_dbs(os.path.join(w, "aln_s"), os.path.join(w, "aln_sw"))
    assert n == 500 and bad > 0


def test_patched_binary_links_nothing_of_the_oracle():
    """the product-side binary: reference + patch + libmmgpu.so; the block-aligner crate's place is taken by do-nothing stubs, no
    restated code (symbols of oracle/block_oracle.c / ref_block_capi.cpp) inside"""
    if not os.path.exists(MMGPU):
        pytest.skip("oracle/_ref/mmseqs_mmgpu not built")
    import subprocess
    syms = subprocess.run(["nm", "-C", "--defined-only", MMGPU], capture_output=True, text=True).stdout
    for name in ("mmo_block_align", "mmo_sw_block_backtrace", "mmo_block_align_table", "mmo_block_prefix_scan"):
        assert name not in syms, name
    assert "block_align_aa_trace_xdrop_posbias" in syms      # the stub of the crate's C API


def test_block_aligner_modes_host_side_emulated(tmp_path):
    if not os.path.exists(EMU):
        pytest.skip("oracle/_build/emu/libmmgpu.so not built (make -C oracle emu)")
    _block_aligner_modes(tmp_path, emulate=True)


@pytest.mark.gpu
def test_block_aligner_modes_on_device(tmp_path):
    _block_aligner_modes(tmp_path, emulate=False)
