"""Stage-by-stage comparison of the device prefilter with the oracle (test infrastructure; used by
tests/test_prefilter_gpu.py and scripts/gpu_check_prefilter.py)."""
import numpy as np

from mmseqs2_amd import capi
from tests import pf_common as pc



def load_case(gpu, g_or_mats, tres, toff, kmer_thr, k=6, spaced=True):
    """Build the tables with the product's host builders and make them resident; returns the table dict."""
    km16 = g_or_mats["vtml80_kmer16"]
    um8 = g_or_mats["blosum62_ungapped"]
    s3, i3 = capi.host_score_matrix(km16, 3, lib=gpu.L)
    s2, i2 = capi.host_score_matrix(km16, 2, lib=gpu.L) if k != 6 else (None, None)
    off, ids, pos = capi.host_index_build(tres, toff, km16, k, spaced, kmer_thr, lib=gpu.L)
    gpu.load_targets(tres, toff, 21)
    gpu.pf_load_index(k, 21, spaced, s3, i3, off, ids, pos, um8, score2=s2, index2=i2)
    return dict(offsets=off, ids=ids, pos=pos)


def check(gpu, orc, queries, max_hits, ref_bins, min_diag_score=15, stages=True, label=""):
    """Runs one batch on the device and the oracle; returns (ok, report lines)."""
    rep = []
    ok = True
    b = gpu.pf_prepare(queries, orc.kmer_thr, max_hits=max_hits, min_diag_score=min_diag_score, ref_bins=ref_bins)
    b.run()
    hits, counts, status, stats = b.fetch()
    ms = b.stage_ms()
    rep.append("%s nq=%d max_hits=%d ref_bins=%d stage ms %s" % (label, len(queries), max_hits, ref_bins,
                                                               ["%.2f" % x for x in ms]))
    dbg = {}
    if stages:
        for w in ("nsim", "peb", "split", "bin_off", "cand_base", "surv", "surv_count", "bins"):
            dbg[w] = b.debug(w)
        bins = int(dbg["bins"][0])
        PF_T = int(dbg["bins"][2])
        rep.append("device bins %d, reference bins %d" % (bins, int(dbg["bins"][1])))
    qoff = np.concatenate([[0], np.cumsum([len(q["q"]) for q in queries])]).astype(np.int64)
    tile_base = 0
    for qi, qd in enumerate(queries):
        o = orc.match(qd["q"], qd.get("comp_bias"), ref_bins, max_hits=max_hits, min_diag_score=min_diag_score,
                      identity_id=qd.get("identity_id"), dump=stages)
        if status[qi] == 1 and o["stats"]["overflow"] > 62:
            continue          # more flushes than the device emulates: handed back to the host, by contract
        if status[qi] != 0:
            ok = False
            rep.append("q%d: device status %d" % (qi, status[qi]))
            continue
        n = int(counts[qi])
        fin = (n == len(o["id"]) and np.array_equal(hits[qi]["id"][:n], o["id"])
               and np.array_equal(hits[qi]["score"][:n], o["score"])
               and np.array_equal(hits[qi]["diagonal"][:n], o["diagonal"]))
        if int(stats[qi]["db_matches"]) != o["stats"]["db_matches"]:
            ok = False
            rep.append("q%d: db_matches %d vs oracle %d" % (qi, stats[qi]["db_matches"], o["stats"]["db_matches"]))
        if int(stats[qi]["kmer_list_len"]) != o["stats"]["kmer_list_len"]:
            ok = False
            rep.append("q%d: kmer_list_len %d vs oracle %d" % (qi, stats[qi]["kmer_list_len"], o["stats"]["kmer_list_len"]))
        if stages:
            L = len(qd["q"])
            ns = dbg["nsim"][qoff[qi]:qoff[qi] + L]
            if not np.array_equal(ns, o["nsim"]):
                ok = False
                bad = np.nonzero(ns != o["nsim"])[0]
                rep.append("q%d: nsim differs at %d positions, first %d: dev %d oracle %d thr %d" % (
                    qi, len(bad), bad[0], ns[bad[0]], o["nsim"][bad[0]], o["thr"][bad[0]]))
            # arrival stream from the split tiles
            ne = o["stats"]["db_matches"]
            nt = (ne + PF_T - 1) // PF_T
            arr_id = np.zeros(ne, np.uint32)
            arr_dg = np.zeros(ne, np.uint16)
            seen = np.zeros(ne, bool)
            order_ok = True
            for t in range(nt):
                tn = min(PF_T, ne - t * PF_T)
                e = dbg["split"][(tile_base + t) * PF_T:(tile_base + t) * PF_T + tn]
                bo = dbg["bin_off"][(tile_base + t) * (bins + 1):(tile_base + t + 1) * (bins + 1)].astype(np.int64)
                slot = (e >> np.uint64(48)).astype(np.int64)
                ids = (e & np.uint64(0xFFFFFFFF)).astype(np.uint32)
                dg = ((e >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.uint16)
                if bo[-1] != tn or np.any(np.diff(bo) < 0):
                    order_ok = False
                for bb in range(bins):
                    seg = slice(bo[bb], bo[bb + 1])
                    if np.any((ids[seg] & (bins - 1)) != bb) or np.any(np.diff(slot[seg]) <= 0):
                        order_ok = False
                good = slot < tn
                arr_id[t * PF_T + slot[good]] = ids[good]
                arr_dg[t * PF_T + slot[good]] = dg[good]
                seen[t * PF_T + slot[good]] = True
            tile_base += nt
            if o["stats"]["overflow"]:
                pass   # the oracle keeps only the last segment of an overflowing query's arrival stream
            elif not (seen.all() and np.array_equal(arr_id, o["arr_id"]) and np.array_equal(arr_dg, o["arr_diag"])):
                ok = False
                bad = np.nonzero(~seen | (arr_id != o["arr_id"]) | (arr_dg != o["arr_diag"]))[0]
                rep.append("q%d: arrival stream differs in %d of %d entries, first at %d (seen %s dev (%d,%d) oracle (%d,%d))" % (
                    qi, len(bad), ne, bad[0], seen[bad[0]], arr_id[bad[0]], arr_dg[bad[0]], o["arr_id"][bad[0]], o["arr_diag"][bad[0]]))
            if not order_ok:
                ok = False
                rep.append("q%d: split tiles are not stably grouped by bin" % qi)
            # survivors
            sb = int(dbg["cand_base"][qi * bins])
            sc = int(dbg["surv_count"][qi])
            sv = dbg["surv"][sb:sb + sc]
            exp = pc.keepmax_reference(o["dd_id"], o["dd_diag"], o["dd_count"], min_diag_score)
            got = {int(r["id"]): (int(r["diag"]), min(255, int(r["score"]))) for r in sv}
            if got != exp or len(sv) != len(exp):
                ok = False
                miss = [k for k in exp if k not in got]
                extra = [k for k in got if k not in exp]
                diff = [k for k in exp if k in got and got[k] != exp[k]]
                rep.append("q%d: survivors differ: dev %d oracle %d (double hits %d); missing %s extra %s differ %s" % (
                    qi, len(sv), len(exp), len(o["dd_id"]), miss[:5], extra[:5],
                    [(k, got[k], exp[k]) for k in diff[:5]]))
        if not fin:
            ok = False
            rep.append("q%d (L=%d): final hits differ: dev n=%d %s / %s / %s ; oracle n=%d %s / %s / %s ; thr dev %x oracle %d trunc %d" % (
                qi, len(qd["q"]), n, hits[qi]["id"][:min(n, 8)], hits[qi]["score"][:min(n, 8)], hits[qi]["diagonal"][:min(n, 8)],
                len(o["id"]), o["id"][:8], o["score"][:8], o["diagonal"][:8], stats[qi]["diag_thr"], o["stats"]["diag_thr"],
                o["stats"]["truncated"]))
    b.free()
    rep.append("%s -> %s" % (label, "OK" if ok else "MISMATCH"))
    return ok, rep
