"""Records the block-aligner crate's OWN unit-test vectors (lib/block-aligner 0.4.0, the Rust code behind SURVEY.md section 8 row
a15) as a fixture: tests/golden/block_crate_vectors.json.

The crate cannot be built in this image (no rustc), so its tests cannot be RUN here; what can be done is to read the expected
values its authors wrote down.  This script parses, from the reference tree,

  /root/reference/lib/block-aligner/src/scan_block.rs   #[cfg(test)] mod tests: test_no_x_drop, test_x_drop, test_trace, test_bytes, test_profile
  /root/reference/lib/block-aligner/matrices/BLOSUM62   the AAMatrix table behind `BLOSUM62` (scores.rs:303)
  /root/reference/lib/block-aligner/src/scores.rs       NucMatrix::new_simple's index rule (:161-176), NW1 / BYTES1 (:293, :340)

and writes every `a.align(..)` call with the assertion(s) that follow it: sequences, matrix, gaps, size range, x-drop, the Block's
<TRACE, X_DROP> parameters, expected score / end position / CIGAR.  tests/test_block_oracle.py replays them on oracle/block_oracle.c.

Run from the repo root in the build container (it reads /root/reference):  python tests/golden/make_block_crate_vectors.py
"""
import json
import os
import re

CRATE = "/root/reference/lib/block-aligner"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "block_crate_vectors.json")


def blosum62_table():
    txt = open(os.path.join(CRATE, "matrices", "BLOSUM62")).read()
    vals = [int(x) for x in re.findall(r"-?\d+", txt)]
    assert len(vals) == 27 * 32, len(vals)
    return vals


def nuc_simple(match, mismatch):
    """NucMatrix::new_simple (scores.rs:161-176)"""
    t = [-128] * (8 * 16)
    alpha = b"ATCGN"
    for i, a in enumerate(alpha):
        for j, b in enumerate(alpha):
            t[(a & 7) * 16 + (b & 15)] = match if i == j else mismatch
    return t


def main():
    src = open(os.path.join(CRATE, "src", "scan_block.rs")).read().split("\n")
    scores_rs = open(os.path.join(CRATE, "src", "scores.rs")).read()
    assert "pub static NW1: NucMatrix = NucMatrix::new_simple(1, -1);" in scores_rs
    assert "pub static BYTES1: ByteMatrix = ByteMatrix::new_simple(1, -1);" in scores_rs
    matrices = {
        "BLOSUM62": {"kind": 0, "table": blosum62_table()},
        "NW1": {"kind": 1, "table": nuc_simple(1, -1)},
        "BYTES1": {"kind": 2, "table": [1, -1]},
    }
    vectors = []
    wanted = ("test_no_x_drop", "test_x_drop", "test_trace", "test_bytes")
    fn = None
    st = {}
    for no, line in enumerate(src, 1):
        m = re.match(r"\s*fn (test_\w+)\(\)", line)
        if m:
            fn = m.group(1) if m.group(1) in wanted else None
            st = {"gaps": {}, "seq": {}, "long": {}, "block": None}
            continue
        if fn is None:
            continue
        m = re.search(r"let (\w+) = Gaps \{ open: (-?\d+), extend: (-?\d+) \};", line)
        if m:
            st["gaps"][m.group(1)] = (int(m.group(2)), int(m.group(3)))
        m = re.search(r"let mut a = Block::<(true|false), (true|false)>::new\(", line)
        if m:
            st["block"] = (m.group(1) == "true", m.group(2) == "true")
        m = re.search(r"let (\w+) = std::iter::repeat\(b'(\w)'\)\.take\((\d+)\)", line)
        if m:
            st["long"][m.group(1)] = m.group(2) * int(m.group(3))
        m = re.search(r"let (\w+) = PaddedBytes::from_bytes::<(\w+)>\((?:b\"([^\"]*)\"|&(\w+)), (\d+)\);", line)
        if m:
            st["seq"][m.group(1)] = (m.group(3) if m.group(3) is not None else st["long"][m.group(4)], m.group(2))
        m = re.search(r"let (\w+) = NucMatrix::new_simple\((-?\d+), (-?\d+)\);", line)
        if m:
            name = "NUC_%s_%s" % (m.group(2), m.group(3))
            matrices[name] = {"kind": 1, "table": nuc_simple(int(m.group(2)), int(m.group(3)))}
            st["matrix_var"] = (m.group(1), name)
        m = re.search(r"a\.align\(&(\w+), &(\w+), &(\w+), (\w+), (\d+)\.\.=(\d+), (\d+)\);", line)
        if m:
            mat = m.group(3)
            if "matrix_var" in st and mat == st["matrix_var"][0]:
                mat = st["matrix_var"][1]
            q, qk = st["seq"][m.group(1)]
            r, rk = st["seq"][m.group(2)]
            assert qk == rk
            vectors.append({"test": fn, "line": no, "trace": st["block"][0], "x_drop_mode": st["block"][1], "q": q, "r": r,
                            "alphabet": qk, "matrix": mat, "gap_open": st["gaps"][m.group(4)][0], "gap_extend": st["gaps"][m.group(4)][1],
                            "min_size": int(m.group(5)), "max_size": int(m.group(6)), "x_drop": int(m.group(7))})
            continue
        if not vectors or vectors[-1]["test"] != fn:
            continue
        v = vectors[-1]
        m = re.search(r"assert_eq!\(a\.res\(\)\.score, (-?\d+)\);", line)
        if m:
            v["score"] = int(m.group(1))
        m = re.search(r"assert_eq!\((?:a\.res\(\)|res), AlignResult \{ score: (-?\d+), query_idx: (\d+), reference_idx: (\d+) \}\);", line)
        if m:
            v["score"], v["query_idx"], v["reference_idx"] = int(m.group(1)), int(m.group(2)), int(m.group(3))
        m = re.search(r"a\.trace\(\)\.(cigar_eq|cigar)\(", line)
        if m:
            v["eq"] = m.group(1) == "cigar_eq"
        m = re.search(r"assert_eq!\(cigar\.to_string\(\), \"([^\"]*)\"\);", line)
        if m:
            v["cigar"] = m.group(1)
    for v in vectors:
        assert "score" in v, v
    counts = {t: sum(v["test"] == t for v in vectors) for t in wanted}
    assert counts == {"test_no_x_drop": 13, "test_x_drop": 3, "test_trace": 5, "test_bytes": 2}, counts
    profile_vectors = parse_test_profile(src)
    assert len(profile_vectors) == 6, len(profile_vectors)
    json.dump({"source": "lib/block-aligner 0.4.0 src/scan_block.rs #[cfg(test)] (parsed, not executed)", "matrices": matrices,
               "vectors": vectors, "profile_vectors": profile_vectors}, open(OUT, "w"), indent=0)
    print("wrote", OUT, counts, "test_profile:", len(profile_vectors))


def parse_test_profile(src):
    """test_profile (scan_block.rs:2432-2477): every a.align_profile(&q, &r, ..) with the AAProfile::from_bytes arguments of r
    (bytes, block_size, match, mismatch, gap_open_C, gap_close_C, gap_open_R, gap_extend), the set_gap_close_C calls that follow it,
    and the assertions."""
    out = []
    inside = False
    block = None
    prof = {}
    seq = {}
    for no, line in enumerate(src, 1):
        m = re.match(r"\s*fn (test_\w+)\(\)", line)
        if m:
            inside = m.group(1) == "test_profile"
            continue
        if not inside:
            continue
        m = re.search(r"let mut a = Block::<(true|false), (true|false)>::new\(", line)
        if m:
            block = (m.group(1) == "true", m.group(2) == "true")
        m = re.search(r"let (?:mut )?(\w+) = AAProfile::from_bytes\(b\"(\w+)\", (\d+), (-?\d+), (-?\d+), (-?\d+), (-?\d+), (-?\d+), (-?\d+)\);", line)
        if m:
            prof[m.group(1)] = {"bytes": m.group(2), "block_size": int(m.group(3)), "match": int(m.group(4)), "mismatch": int(m.group(5)),
                                "gap_open_C": int(m.group(6)), "gap_close_C": int(m.group(7)), "gap_open_R": int(m.group(8)),
                                "gap_extend": int(m.group(9)), "gap_close_C_at": []}
        m = re.search(r"(\w+)\.set_gap_close_C\((\d+), (-?\d+)\);", line)
        if m:
            prof[m.group(1)]["gap_close_C_at"].append([int(m.group(2)), int(m.group(3))])
        m = re.search(r"let (\w+) = PaddedBytes::from_bytes::<AAMatrix>\(b\"(\w+)\", (\d+)\);", line)
        if m:
            seq[m.group(1)] = m.group(2)
        m = re.search(r"a\.align_profile\(&(\w+), &(\w+), (\d+)\.\.=(\d+), (\d+)\);", line)
        if m:
            v = {"line": no, "trace": block[0], "x_drop_mode": block[1], "q": seq[m.group(1)], "min_size": int(m.group(3)),
                 "max_size": int(m.group(4)), "x_drop": int(m.group(5))}
            v.update({k: (list(map(list, x)) if k == "gap_close_C_at" else x) for k, x in prof[m.group(2)].items()})
            out.append(v)
            continue
        if not out:
            continue
        v = out[-1]
        m = re.search(r"assert_eq!\(a\.res\(\)\.score, (-?\d+)\);", line)
        if m:
            v["score"] = int(m.group(1))
        m = re.search(r"assert_eq!\(res, AlignResult \{ score: (-?\d+), query_idx: (\d+), reference_idx: (\d+) \}\);", line)
        if m:
            v["score"], v["query_idx"], v["reference_idx"] = int(m.group(1)), int(m.group(2)), int(m.group(3))
        m = re.search(r"assert_eq!\(cigar\.to_string\(\), \"([^\"]*)\"\);", line)
        if m:
            v["cigar"] = m.group(1)
    return out


if __name__ == "__main__":
    main()
