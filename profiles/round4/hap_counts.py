"""How many distinct 32-base (and 16-base) block haplotypes do the alleles of a locus have?

VERDICT r3 item 1: measured on three generators before k_extend was rebuilt on per-block haplotype summaries
(/root/reference/metamlst.py:133-151 needs only sum(score) / hits per allele; alleles of a locus differ by a few SNPs).
Run: python profiles/round4/hap_counts.py   (CPU, numpy only; writes profiles/round4/hap_counts.md)
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metamlst_amd.synth import gen_locus_alleles


def hap_stats(alleles, bs):
    L = max(len(a) for a in alleles)
    nb = (L + bs - 1) // bs
    per_block = []
    for q in range(nb):
        seen = set()
        for a in alleles:
            seen.add((a[q * bs:(q + 1) * bs].tobytes()))
        per_block.append(len(seen))
    return per_block


def window_distinct(alleles, width=150):
    L = min(len(a) for a in alleles)
    out = []
    for s in range(0, L - width + 1, 37):
        out.append(len({a[s:s + width].tobytes() for a in alleles}))
    return float(np.mean(out))


def main():
    rows = []
    rng = np.random.default_rng(20221)
    cases = [
        ("tree 1-8 SNPs/edge, 3 % (cfg3 locus)", dict(length=500, n_alleles=300, snp_lo=1, snp_hi=8)),
        ("tree 1-8 SNPs/edge, 3 % (cfg2 locus)", dict(length=500, n_alleles=1430, snp_lo=1, snp_hi=8)),
        ("tree 1-2 SNPs/edge, 3 %", dict(length=500, n_alleles=300, snp_lo=1, snp_hi=2)),
        ("tree 1-2 SNPs/edge, 3 %", dict(length=500, n_alleles=1430, snp_lo=1, snp_hi=2)),
        ("skewed: 10 alleles", dict(length=450, n_alleles=10)),
        ("skewed: 100 alleles", dict(length=450, n_alleles=100)),
        ("skewed: 3,000 alleles", dict(length=450, n_alleles=3000)),
        ("skewed: 10,000 alleles", dict(length=600, n_alleles=10000)),
    ]
    for name, kw in cases:
        al = gen_locus_alleles(rng, **kw)
        n = len(al)
        h32 = hap_stats(al, 32); h16 = hap_stats(al, 16)
        # a 150-base read covers 6 blocks of 32 (worst alignment) / 11 of 16
        w32 = max(sum(h32[i:i + 6]) for i in range(max(1, len(h32) - 5)))
        rows.append((name, n, kw["length"], float(np.mean(h32)), max(h32), float(np.mean(h16)), max(h16), w32, window_distinct(al)))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hap_counts.md"), "w") as f:
        f.write("# Distinct block haplotypes per locus (profiles/round4/hap_counts.py)\n\n")
        f.write("| generator | alleles | length | mean H (32-base block) | max H32 | mean H (16-base) | max H16 | sum of H32 over the 6 blocks a 150-base read covers (worst window) | distinct 150-base windows |\n|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | %d | %d | %.1f | %d | %.1f | %d | %d | %.0f |\n" % r)
    print(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hap_counts.md")).read())


if __name__ == "__main__":
    main()
