"""Minimal FASTA reader / writer standing in for Biopython's SeqIO (golden generation only)."""
from .Seq import Seq
from .SeqRecord import SeqRecord


def write(records, path, fmt):
    """Biopython's FastaWriter [NOT IN TREE]: title '>id description' (id alone when the description is empty), the
    sequence wrapped at 60 columns.  `path` may be a file name or an open handle."""
    records = list(records)
    f = open(path, "w") if isinstance(path, str) else path
    for r in records:
        f.write(">%s\n" % (("%s %s" % (r.id, r.description)) if r.description else r.id))
        s = str(r.seq)
        for at in range(0, len(s), 60):
            f.write(s[at:at + 60] + "\n")
    if isinstance(path, str):
        f.close()
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
