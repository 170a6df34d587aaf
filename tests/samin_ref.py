"""Literal per-base restatement of the alignment pileup (cmseq get_base_stats over a BAM, [NOT IN TREE]) that the tests
compare mlst_pileup_alignments with.  Test infrastructure only."""
import numpy as np


def pileup_python(index, sample, chosen, minqual: int = 20) -> dict:
    """The same pileup as a literal per-base loop (what the tests compare mlst_pileup_alignments with)."""
    a = sample.args
    out = {int(c): np.zeros((int(index.off[c + 1] - index.off[c]), 4), np.uint32) for c in chosen}
    for ai, pos, AS, XM, cig, seq, qual in sample._rec:
        if ai not in out or AS < a.minscore or XM > a.max_xM or seq == "*":
            continue
        r, q = pos, 0
        for o in cig:
            ln, op = o >> 4, o & 15
            if op in (0, 7, 8):                                          # M = X: aligned columns
                for k in range(ln):
                    b = "ACGT".find(seq[q + k].upper())
                    ph = (ord(qual[q + k]) - 33) if qual != "*" else 0
                    if b >= 0 and ph >= minqual and 0 <= r + k < out[ai].shape[0]:
                        out[ai][r + k, b] += 1
                r += ln; q += ln
            elif op in (1, 4):                                           # I S: read only
                q += ln
            elif op in (2, 3):                                           # D N: reference only
                r += ln
    return out
