"""Re-scoring of a backtrace (test infrastructure): the score of the alignment path a CIGAR describes, under the Smith-Waterman
scoring of the run (matrix + per-query-position composition bias, gap open = cost of the first gap residue, gap extend).
SURVEY.md section 8c's contract for int16-range hits (s_align::word == 1), whose start position and CIGAR the stock reference
takes from the Rust block-aligner: whatever produced the path, re-scoring it must reproduce score1."""
import numpy as np


def rescore(q, cb, t, mat, gap_open, gap_extend, q_start, t_start, bt):
    """-> (score of the path, q_end, t_end) ; bt letters: M (both advance), I (query advances), D (target advances)"""
    qp, tp, score, prev = int(q_start), int(t_start), 0, "M"
    for c in bt:
        if c == "M":
            score += int(mat[int(q[qp]), int(t[tp])]) + (int(cb[qp]) if cb is not None else 0)
            qp += 1
            tp += 1
        else:
            score -= gap_extend if prev == c else gap_open
            if c == "I":
                qp += 1
            else:
                tp += 1
        prev = c
    return score, qp - 1, tp - 1
