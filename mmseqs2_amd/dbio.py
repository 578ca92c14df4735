"""Minimal reader for MMseqs2 databases (`name` data file(s), `name.index` = key \\t offset \\t length,
`name.dbtype`), used by the drop-in tests and scripts to compare result DBs entry by entry.  Split data files
(`name.0 ... name.N`, written by multi-threaded DBWriter runs that were not merged) are read as one concatenated file, the
way DBReader does (src/commons/DBReader.cpp)."""
import os
import re

import numpy as np


def _data_files(name):
    if os.path.exists(name):
        return [name]
    d, b = os.path.split(name)
    pat = re.compile(re.escape(b) + r"\.(\d+)$")
    parts = sorted((int(m.group(1)), os.path.join(d or ".", f)) for f in os.listdir(d or ".") for m in [pat.match(f)] if m)
    return [p for _, p in parts]


def read_db(name):
    """-> dict key -> bytes of the entry (without the terminating NUL)"""
    files = _data_files(name)
    if not files:
        raise FileNotFoundError(name)
    data = b"".join(open(f, "rb").read() for f in files)
    out = {}
    with open(name + ".index") as fh:
        for line in fh:
            k, off, ln = line.split("\t")[:3]
            off, ln = int(off), int(ln)
            out[int(k)] = data[off:off + max(ln - 1, 0)]
    return out


def dbtype(name):
    return int(np.frombuffer(open(name + ".dbtype", "rb").read()[:4], dtype="<i4")[0])


def diff_dbs(a, b, limit=5):
    """Entry-by-entry comparison of two DBs.  -> (n_entries, n_different, [descriptions of the first differences])"""
    da, db = read_db(a), read_db(b)
    msgs = []
    bad = 0
    if dbtype(a) != dbtype(b):
        bad += 1
        msgs.append("dbtype %d != %d" % (dbtype(a), dbtype(b)))
    for k in sorted(set(da) | set(db)):
        if k not in da or k not in db:
            bad += 1
            if len(msgs) < limit:
                msgs.append("key %d only in %s" % (k, a if k in da else b))
        elif da[k] != db[k]:
            bad += 1
            if len(msgs) < limit:
                la, lb = da[k].split(b"\n"), db[k].split(b"\n")
                for i in range(max(len(la), len(lb))):
                    x = la[i] if i < len(la) else b"<none>"
                    y = lb[i] if i < len(lb) else b"<none>"
                    if x != y:
                        msgs.append("key %d line %d: %r != %r" % (k, i, x[:200], y[:200]))
                        break
    return len(da), bad, msgs


def diff_dbs_up_to_tie_order(a, b, tie_fields=(1, 3), limit=5):
    """diff_dbs for result databases whose lines the reference itself writes in a thread-dependent order where their sort keys tie
    (`offsetalignment` of a translated search: two ORF hits of one read with the same bit score and E-value swap places from run to run
    of the STOCK binary): entries are equal when the sequences of tie keys (tab-separated fields `tie_fields`) are equal and every run
    of lines with one key holds the same lines.  -> (n_entries, n_different, n_equal_only_up_to_tie_order, [first differences])"""
    da, db = read_db(a), read_db(b)
    msgs, bad, ties = [], 0, 0

    def runs(entry):
        out = []
        for line in entry.split(b"\n"):
            f = line.split(b"\t")
            key = tuple(f[i] if i < len(f) else b"" for i in tie_fields)
            if out and out[-1][0] == key:
                out[-1][1].append(line)
            else:
                out.append((key, [line]))
        return [(k, sorted(v)) for k, v in out]

    if dbtype(a) != dbtype(b):
        bad += 1
        msgs.append("dbtype %d != %d" % (dbtype(a), dbtype(b)))
    for k in sorted(set(da) | set(db)):
        if k not in da or k not in db:
            bad += 1
            if len(msgs) < limit:
                msgs.append("key %d only in %s" % (k, a if k in da else b))
        elif da[k] != db[k]:
            if runs(da[k]) == runs(db[k]):
                ties += 1
            else:
                bad += 1
                if len(msgs) < limit:
                    msgs.append("key %d differs beyond the order of tied lines" % k)
    return len(da), bad, ties, msgs
