"""Stand-in for SegataLab/cmseq (empty submodule in /root/reference): serves a consensus derived
from a supplied per-locus count table <bam>.counts.json and logs how it was called."""
import json


class _Handle(object):
    def close(self):
        pass


class _Contig(object):
    def __init__(self, owner, label):
        self.owner, self.label = owner, label

    def reference_free_consensus(self, **kw):
        self.owner.calls.append({"label": self.label, "kwargs": {k: (list(map(list, v)) if k == "BAM_tagFilter" else v) for k, v in kw.items()}})
        with open(self.owner.path + ".cmseq_calls.json", "w") as f:
            json.dump(self.owner.calls, f)
        sp, gene, _ = self.label.split("_")
        counts = self.owner.counts[sp + "_" + gene]
        out = []
        for row in counts:
            if sum(row) < kw.get("mincov", 1):
                out.append(kw.get("noneCharacter", "N"))
            else:
                out.append("ACGT"[max(range(4), key=lambda k: (row[k], -k))])
        return "".join(out)


class BamFile(object):
    def __init__(self, path, filterInputList=None, **kw):
        self.path = path
        self.counts = json.load(open(path + ".counts.json"))
        self.calls = []
        self.bam_handle = _Handle()

    def get_contig_by_label(self, label):
        return _Contig(self, label)
