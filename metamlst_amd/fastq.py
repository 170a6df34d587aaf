"""FASTQ -> (bases, quals, offsets) batches for mlst_submit_reads (host part).

The reference never reads FASTQ itself (bowtie2 does, outside the tree); this is the minimal host
reader so the engine can be fed the same files.  Plain or gzip, 4-line records, Phred+33."""
from __future__ import annotations

import gzip

import numpy as np


def _open(path: str):
    return gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")


def read_batches(path: str, batch_reads: int = 2_000_000, max_len: int = 320):
    """Yield (bases uint8[], quals uint8[], off uint64[n+1], names list[bytes]) per batch.
    Reads longer than max_len are truncated (the packed format holds 320 bases)."""
    seqs, quals, names = [], [], []
    with _open(path) as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().rstrip(b"\r\n")
            f.readline()
            q = f.readline().rstrip(b"\r\n")
            if len(q) != len(s):
                raise ValueError("FASTQ record with different sequence and quality lengths: %r" % h[:60])
            names.append(h[1:].split()[0] if len(h) > 1 else b"")
            seqs.append(s[:max_len])
            quals.append(q[:max_len])
            if len(seqs) == batch_reads:
                yield _pack(seqs, quals, names)
                seqs, quals, names = [], [], []
    if seqs:
        yield _pack(seqs, quals, names)


def _pack(seqs, quals, names):
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return (np.frombuffer(b"".join(seqs), np.uint8).copy(), np.frombuffer(b"".join(quals), np.uint8).copy(), off, names)


def interleave(path1: str, path2: str, batch_pairs: int = 1_000_000, max_len: int = 320):
    """Paired files -> batches with mates at rows 2k, 2k+1 (the layout mlst_submit_reads(paired=1) expects)."""
    for (b1, q1, o1, n1), (b2, q2, o2, n2) in zip(read_batches(path1, batch_pairs, max_len), read_batches(path2, batch_pairs, max_len)):
        if len(o1) != len(o2):
            raise ValueError("paired FASTQ files have different numbers of reads")
        seqs, quals = [], []
        for k in range(len(o1) - 1):
            seqs += [b1[int(o1[k]):int(o1[k + 1])].tobytes(), b2[int(o2[k]):int(o2[k + 1])].tobytes()]
            quals += [q1[int(o1[k]):int(o1[k + 1])].tobytes(), q2[int(o2[k]):int(o2[k + 1])].tobytes()]
        yield _pack(seqs, quals, [x for p in zip(n1, n2) for x in p])


def text_chunks(path: str, chunk_bytes: int = 256 << 20):
    """Yield byte chunks of an uncompressed (or .gz) FASTQ that each hold whole records, for Engine.submit_fastq.
    The only host work is cutting after a multiple of four lines; parsing happens on the GPU."""
    carry = b""
    lines_in_carry = 0
    with _open(path) as f:
        while True:
            block = f.read(chunk_bytes)
            if not block:
                break
            data = carry + block
            arr = np.frombuffer(data, np.uint8)
            nl = np.flatnonzero(arr == 10)
            whole = (len(nl) // 4) * 4
            if whole == 0:
                carry = data
                continue
            cut = int(nl[whole - 1]) + 1
            yield data[:cut]
            carry = data[cut:]
    if carry.strip():
        yield carry
