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


def tile_fasta(path: str, read_len: int = 150, stride: int = 25, min_len: int = 50, chunk_reads: int = 500_000):
    """Contigs / an assembled genome as input (the modality of the reference's mlst.py, which BLASTs contigs against
    the alleles; BLAST is not in the tree -- here the contigs go through the same alignment path as reads): every
    contig is cut into overlapping windows (read_len bases every stride bases, the last window flush with the contig
    end), written as FASTQ text with Phred 40 and handed to Engine.submit_fastq in chunks.  Contigs shorter than
    min_len are skipped, shorter than read_len become one read.  Read names: <contig index>_<0-based start>."""
    if read_len < 1 or stride < 1:
        raise ValueError("read_len and stride must be positive")

    def contigs():
        name, parts = None, []
        with _open(path) as f:
            for line in f:
                if line.startswith(b">"):
                    if name is not None:
                        yield b"".join(parts)
                    name, parts = line, []
                elif name is not None:
                    parts.append(line.strip())
        if name is not None:
            yield b"".join(parts)

    out, n_out = [], 0
    for ci, seq in enumerate(contigs()):
        seq = seq.upper()
        n = len(seq)
        if n < min_len:
            continue
        starts = list(range(0, max(n - read_len, 0) + 1, stride))
        if n > read_len and starts[-1] != n - read_len:
            starts.append(n - read_len)
        for st in starts:
            w = seq[st:st + read_len]
            out.append(b"@%d_%d\n%s\n+\n%s\n" % (ci, st, w, b"I" * len(w)))
            n_out += 1
            if n_out == chunk_reads:
                yield b"".join(out)
                out, n_out = [], 0
    if out:
        yield b"".join(out)


def _bgzf_block_size(buf, off: int) -> int:
    """Total size of the BGZF block that starts at buf[off], 0 if the header is incomplete, -1 if it is not BGZF."""
    if len(buf) - off < 18:
        return 0
    if buf[off] != 0x1F or buf[off + 1] != 0x8B or buf[off + 2] != 8 or not (buf[off + 3] & 4):
        return -1
    xlen = buf[off + 10] | (buf[off + 11] << 8)
    if len(buf) - off < 12 + xlen:
        return 0
    o = 0
    while o + 4 <= xlen:
        p = off + 12 + o
        slen = buf[p + 2] | (buf[p + 3] << 8)
        if buf[p] == 66 and buf[p + 1] == 67 and slen == 2:
            return (buf[p + 4] | (buf[p + 5] << 8)) + 1
        o += 4 + slen
    return -1


def is_bgzf(path: str) -> bool:
    with open(path, "rb") as f:
        return _bgzf_block_size(f.read(4096), 0) > 0


def bgzf_chunks(path: str, chunk_bytes: int = 256 << 20):
    """Yield (compressed bytes holding whole BGZF blocks, is_last) for Engine.submit_fastq_bgzf.  The only host work is
    walking the block headers; inflating and parsing happen on the GPU."""
    carry = b""
    pending = None
    with open(path, "rb") as f:
        while True:
            block = f.read(chunk_bytes)
            data = carry + block
            off = 0
            while True:
                n = _bgzf_block_size(data, off)
                if n < 0:
                    raise ValueError("%s: not a BGZF block at byte offset %d of a chunk" % (path, off))
                if n == 0 or off + n > len(data):
                    break
                off += n
            if pending is not None:
                yield pending, False
            pending = data[:off] if off else None
            carry = data[off:]
            if not block:
                break
    if carry:
        raise ValueError("%s: truncated BGZF block at the end of the file" % path)
    yield (pending if pending is not None else b""), True
