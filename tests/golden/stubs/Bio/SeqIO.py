"""Minimal FASTA reader / writer standing in for Biopython's SeqIO (golden generation only)."""
from .Seq import Seq
from .SeqRecord import SeqRecord


def write(records, path, fmt):
    with open(path, "w") as f:
        for r in records:
            f.write(">%s\n%s\n" % (r.id, r.seq))
    return len(records)


def parse(path, fmt):
    name, chunks = None, []
    for line in open(path):
        line = line.rstrip("\r\n")
        if line.startswith(">"):
            if name is not None:
                yield SeqRecord(Seq("".join(chunks)), id=name, description="")
            name, chunks = (line[1:].split() or [""])[0], []
        elif name is not None:
            chunks.append(line.strip())
    if name is not None:
        yield SeqRecord(Seq("".join(chunks)), id=name, description="")
