"""FASTQ -> (bases, quals, offsets) batches for mlst_submit_reads (host part).

The reference never reads FASTQ itself (bowtie2 does, outside the tree); this is the minimal host
reader so the engine can be fed the same files.  Plain or gzip, 4-line records, Phred+33."""
from __future__ import annotations

import gzip
import os
import threading

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


def record_start(buf, pos: int = 0) -> int:
    """Offset of the first FASTQ record that starts at or after `pos` in `buf` (bytes-like), -1 if none can be told.
    A record starts at a line that begins with '@' whose next-but-one line begins with '+': a quality line may begin
    with '@' too, but then the line two further on is a sequence line, which never begins with '+' (4-line records)."""
    n = len(buf)
    if pos <= 0:
        p = 0
    else:
        p = buf.find(b"\n", pos - 1) + 1          # first line start at or after pos (pos itself when buf[pos-1] is a newline)
        if p == 0:
            return -1
    while p < n:
        e1 = buf.find(b"\n", p)
        if e1 < 0:
            return -1
        e2 = buf.find(b"\n", e1 + 1)
        if e2 < 0 or e2 + 1 >= n:
            return -1
        if buf[p] == 0x40 and buf[e2 + 1] == 0x2B:
            return p
        p = e1 + 1
    return -1


def last_record_start(buf, window: int = 1 << 16) -> int:
    """Offset of the last record that starts in `buf` (so that buf[:offset] holds whole records), -1 if none.  Only the
    tail of the buffer is looked at: the window doubles until it holds a record start."""
    n = len(buf)
    w = min(window, n)
    while True:
        lo = n - w
        best, p = -1, record_start(buf, lo) if lo else record_start(buf, 0)
        while p >= 0:
            best = p
            nxt = buf.find(b"\n", p)
            p = record_start(buf, nxt + 1) if nxt >= 0 else -1
        if best >= 0 or w == n:
            return best
        w = min(n, w * 4)


# ---- plain files: positional reads by a few threads into buffers that are used again
# One Python thread moves ~1 GB/s from the page cache through f.read() + slicing (two or three copies of every byte): a
# sample of a million reads took 0.3 s to reach the library, which needs 7 ms for it.  os.preadv releases the GIL, so a few
# threads fill one numpy buffer side by side (no copy afterwards: the chunk handed to the library is a view of it), and
# the buffers go round (a fresh 256 MB array costs its page faults again).
_READ_PIECE = 4 << 20
_pool_lock = threading.Lock()
_pool_bufs: list = []          # idle buffers (uint8 arrays), at most _POOL_MAX of them
_POOL_MAX = int(os.environ.get("MLST_READ_POOL", "20"))      # (four feeder threads keep four buffers each in flight: a pool that holds fewer
                                                             # frees and allocates page-locked memory between samples -- ~50 ms per 256 MB buffer)
_readers = None


def _reader_pool():
    global _readers
    if _readers is None:
        with _pool_lock:      # (first called by four feeder threads at once: one pool, not four)
            if _readers is None:
                from concurrent.futures import ThreadPoolExecutor
                _readers = ThreadPoolExecutor(max(2, min(int(os.environ.get("MLST_READ_THREADS", "32")), (os.cpu_count() or 2) // 2)), thread_name_prefix="fastq-read")
    return _readers


_alloc = None                  # set_buffer_allocator: n bytes -> uint8 array (page-locked memory when a GPU runtime is there)


def set_buffer_allocator(fn) -> None:
    """Where the read buffers of text_chunks(reuse=True) come from: fn(n_bytes) -> uint8 array, None = numpy.  The CLI passes
    engine.pinned_array: a chunk then crosses the link as one DMA transfer instead of through the runtime's staging copy."""
    global _alloc
    with _pool_lock:
        if fn is _alloc:
            return
        _alloc = fn
        old = list(_pool_bufs)      # (dropped after the lock is released: freeing page-locked memory takes milliseconds per buffer)
        _pool_bufs.clear()
    del old


def _take_buffer(n: int) -> np.ndarray:
    with _pool_lock:
        for k, b in enumerate(_pool_bufs):
            if b.size >= n:
                return _pool_bufs.pop(k)
        if len(_pool_bufs) >= _POOL_MAX:
            _pool_bufs.pop(0)
        fn = _alloc
    size = max(1 << 20, 1 << (int(n) - 1).bit_length())      # powers of two: a buffer fits the next file's chunks too
    if fn is not None:
        try:
            return fn(size)
        except Exception:      # noqa: BLE001 -- no page-locked memory to be had: ordinary memory does
            pass
    return np.empty(size, np.uint8)


def _give_buffer(b: np.ndarray) -> None:
    with _pool_lock:
        if len(_pool_bufs) < _POOL_MAX:
            _pool_bufs.append(b)


def _pread_into(fd: int, buf: np.ndarray, lo: int, hi: int) -> None:
    """bytes [lo, hi) of the file into buf[:hi - lo], pieces of 4 MB read side by side"""
    mv = memoryview(buf)

    def piece(a: int) -> None:
        e = min(a + _READ_PIECE, hi)
        at = a
        while at < e:
            got = os.preadv(fd, [mv[at - lo:e - lo]], at)
            if got <= 0:
                raise OSError("short read at byte %d" % at)
            at += got

    starts = range(lo, hi, _READ_PIECE)
    if len(starts) <= 1:
        for a in starts:
            piece(a)
    else:
        list(_reader_pool().map(piece, starts))


def raw_chunks(path: str, chunk_bytes: int = 256 << 20, lo: int = 0, hi: int | None = None, margin: int = 1 << 17,
               reuse: bool = False, ring: list | None = None):
    """Bytes [lo, hi) of a file in pieces of chunk_bytes, cut anywhere: yields (buf, n) with the piece at buf[margin:margin + n]
    and `margin` free bytes in front of it -- room for what the consumer carries over from the piece before (the cut-off BGZF
    block of Engine.submit_fastq_bgzf_file).  Positional reads by a few threads; reuse / ring as in text_chunks."""
    size = os.path.getsize(path)
    hi = size if hi is None else min(hi, size)
    if ring is None:
        ring = []
    fd = os.open(path, os.O_RDONLY)
    try:
        at = lo
        while at < hi:
            n = min(chunk_bytes, hi - at)
            if reuse and len(ring) >= 4:
                buf = ring.pop(0)
                if buf.size < margin + n:
                    _give_buffer(buf)
                    buf = _take_buffer(margin + n)
            else:
                buf = _take_buffer(margin + n) if reuse else np.empty(margin + n, np.uint8)
            if reuse:
                ring.append(buf)
            _pread_into(fd, buf[margin:], at, at + n)
            at += n
            yield buf, n
    finally:
        os.close(fd)


def release_buffers(ring: list) -> None:
    """The consumer is done with every chunk of a text_chunks(reuse=True, ring=ring) walk: its buffers may serve the next file."""
    while ring:
        _give_buffer(ring.pop())


def _plain_chunks(path: str, chunk_bytes: int, start: int, end: int | None, reuse: bool, ring: list | None = None):
    """text_chunks for an uncompressed file (same chunks, same byte ranges): positional reads, chunks are uint8 arrays.
    reuse=True: a chunk's memory is used again once the consumer has taken the chunk after the next THREE (prefetch() holds two,
    the consumer one) -- for consumers that are done with a chunk when they ask for the next; reuse=False: every chunk its own array.
    The walk never hands its buffers to another walk by itself -- its last chunks may still be in the consumer's hands when it
    ends: a consumer that passes `ring` gets the buffers listed there and gives them back with release_buffers(ring)."""
    size = os.path.getsize(path)
    if size == 0:
        return
    fd = os.open(path, os.O_RDONLY)
    if ring is None:
        ring = []
    try:
        def peek(lo: int, n: int) -> bytes:
            return os.pread(fd, max(0, min(n, size - lo)), lo)

        examined = [0]

        def first_start(at: int) -> int:
            """first record start at or after byte `at` (absolute), -1 if none can be told"""
            if at <= 0:
                return 0
            w = 1 << 16
            while True:
                head = peek(at - 1, w)                       # one byte early: a record that starts exactly at `at` counts
                examined[0] = at - 1 + len(head)
                s0 = record_start(head, 1)
                if s0 >= 0:
                    return at - 1 + s0
                if examined[0] >= size or w >= (64 << 20):
                    return -1
                w *= 4

        pos = first_start(start) if start else 0
        if pos < 0:
            if end is not None and examined[0] < end:
                raise ValueError("%s: no FASTQ record boundary after byte %d" % (path, start))
            return                                           # the range holds the inside of the file's last record only
        if end is not None and pos >= end:
            return                                           # the first record at or after `start` belongs to the next range
        stop = size
        if end is not None and end < size:
            s1 = first_start(end)                            # the record that starts last before `end` is ours in full
            stop = s1 if s1 >= 0 else size
        while pos < stop:
            e = min(pos + chunk_bytes, stop)
            cut = e
            if e < stop:                                     # cut at the last record start inside [pos, e)
                w = 1 << 16
                while True:
                    lo = max(pos, e - w)
                    k = last_record_start(peek(lo, e - lo))
                    if k > 0 or (k == 0 and lo > pos):
                        cut = lo + k
                        break
                    if lo == pos:                            # no whole record in the chunk (a record longer than it): read on
                        chunk_bytes *= 2
                        cut = -1
                        break
                    w *= 4
                if cut < 0:
                    continue
            n = cut - pos
            if reuse and len(ring) >= 4:
                buf = ring.pop(0)
                if buf.size < n:
                    _give_buffer(buf)
                    buf = _take_buffer(n)
            else:
                buf = _take_buffer(n) if reuse else np.empty(n, np.uint8)
            _pread_into(fd, buf, pos, cut)
            if reuse:
                ring.append(buf)
            chunk = buf[:n]
            if cut < size or n > 4096 or chunk.tobytes().strip():      # (a file's whitespace-only tail is not a chunk)
                yield chunk
            pos = cut
    finally:
        os.close(fd)


def text_chunks(path: str, chunk_bytes: int = 256 << 20, start: int = 0, end: int | None = None, reuse: bool = False, ring: list | None = None):
    """Yield byte chunks of an uncompressed (or .gz) FASTQ that each hold whole records, for Engine.submit_fastq.
    The host never scans the text: a chunk is cut at the last record start found in its tail (record_start); parsing
    happens on the GPU.  start / end (plain files only) restrict the walk to the records that START in the byte range
    [start, end): the first record is found by resynchronising at `start`, the last one is completed beyond `end` --
    N ranks given consecutive ranges read every record exactly once and touch only their share of the file.
    Plain files are read by positional reads of a few threads (_plain_chunks; reuse: see there); .gz through gzip."""
    gz = path.endswith(".gz")
    if gz and (start or end is not None):
        raise ValueError("byte ranges need an uncompressed FASTQ (gzip has no random access; bgzip files are sharded by block)")
    if not gz:
        yield from _plain_chunks(path, chunk_bytes, start, end, reuse, ring)
        return
    carry = b""
    with _open(path) as f:
        pos = 0
        if start:
            f.seek(start - 1)                      # one byte early: a record that starts exactly at `start` is ours
            pos = start - 1
            head = f.read(1 << 16)
            while True:
                s0 = record_start(head, 1)
                if s0 >= 0 or len(head) >= (64 << 20):
                    break
                more = f.read(len(head))
                if not more:
                    break
                head += more
            if s0 < 0:
                if end is not None and pos + len(head) < end:
                    raise ValueError("%s: no FASTQ record boundary after byte %d" % (path, start))
                return                              # the range holds the inside of the file's last record only
            if end is not None and pos + s0 >= end:
                return                              # the first record at or after `start` belongs to the next range
            carry = head[s0:]
            pos += len(head)
        done = False
        while not done:
            want = chunk_bytes - len(carry) if len(carry) < chunk_bytes else 0
            block = f.read(want) if want else b""
            data = carry + block if carry else block
            base = pos - len(carry)                 # file offset of data[0]
            pos += len(block)
            eof = want > 0 and len(block) < want
            if end is not None and base + len(data) > end:
                # the record that starts last before `end` is ours in full; what follows belongs to the next range
                cut = record_start(data, max(end - base, 0))
                while cut < 0 and not eof:          # the boundary record is not complete yet
                    more = f.read(1 << 16)
                    if not more:
                        eof = True
                        break
                    data += more
                    pos += len(more)
                    cut = record_start(data, max(end - base, 0))
                if cut < 0:
                    cut = len(data)                 # nothing starts after `end`: the range runs to the end of the file
                if cut:
                    yield data[:cut]
                return
            if eof:
                if data.strip():
                    yield data
                return
            cut = last_record_start(data)
            if cut <= 0:
                carry = data                        # no whole record yet (a record longer than the chunk): read on
                chunk_bytes *= 2
                continue
            yield data[:cut]
            carry = data[cut:]


def _count_newlines(buf: np.ndarray, n: int) -> tuple[list, list]:
    """newlines of buf[:n] per piece of 4 MB, counted side by side (numpy's comparison and count release the GIL)"""
    starts = list(range(0, n, _READ_PIECE))

    def one(a: int) -> int:
        return int(np.count_nonzero(buf[a:min(a + _READ_PIECE, n)] == 10))

    counts = [one(a) for a in starts] if len(starts) <= 1 else list(_reader_pool().map(one, starts))
    return starts, counts


def _after_newline(buf: np.ndarray, n: int, starts: list, counts: list, which: int) -> int:
    """offset just behind the which-th newline (1-based) of buf[:n]"""
    seen = 0
    for a, c in zip(starts, counts):
        if seen + c >= which:
            piece = buf[a:min(a + _READ_PIECE, n)]
            return a + int(np.flatnonzero(piece == 10)[which - seen - 1]) + 1
        seen += c
    raise AssertionError("fewer newlines than counted")


def _plain_pair_chunks(path1: str, path2: str, chunk_bytes: int, reuse: bool, ring: list | None):
    """pair_chunks for two uncompressed files: the same windows and cuts, read by positional reads of a few threads into
    buffers (reuse / ring: as in _plain_chunks, eight buffers deep -- two per pair), newlines counted side by side."""
    s1, s2 = os.path.getsize(path1), os.path.getsize(path2)
    fd1, fd2 = os.open(path1, os.O_RDONLY), os.open(path2, os.O_RDONLY)
    if ring is None:
        ring = []
    o1 = o2 = 0

    def window(fd, off, size):
        n = min(chunk_bytes, size - off)
        if reuse and len(ring) >= 8:
            buf = ring.pop(0)
            if buf.size < n + 1:
                _give_buffer(buf)
                buf = _take_buffer(n + 1)
        else:
            buf = _take_buffer(n + 1) if reuse else np.empty(n + 1, np.uint8)
        if reuse:
            ring.append(buf)
        if n:
            _pread_into(fd, buf, off, off + n)
        eof = off + n == size
        nv = n
        if eof and n and buf[n - 1] != 10:
            buf[n] = 10                                     # an unterminated last line ends here
            nv = n + 1
        return buf, n, nv, eof

    try:
        while True:
            A, n1, v1, e1 = window(fd1, o1, s1)
            B, n2, v2, e2 = window(fd2, o2, s2)
            st1, c1 = _count_newlines(A, v1)
            st2, c2 = _count_newlines(B, v2)
            k = min(sum(c1), sum(c2)) // 4
            if k == 0:
                blank1, blank2 = not A[:v1].tobytes().strip(), not B[:v2].tobytes().strip()
                if e1 and e2:
                    if not (blank1 and blank2):
                        raise ValueError("paired FASTQ files hold different numbers of records (%s, %s)" % (path1, path2))
                    return
                if (e1 and blank1) or (e2 and blank2):
                    raise ValueError("paired FASTQ files hold different numbers of records (%s, %s)" % (path1, path2))
                chunk_bytes *= 2
                continue
            a, b = _after_newline(A, v1, st1, c1, 4 * k), _after_newline(B, v2, st2, c2, 4 * k)
            yield A[:a], B[:b]
            o1 += min(a, n1); o2 += min(b, n2)
    finally:
        os.close(fd1); os.close(fd2)


def pair_chunks(path1: str, path2: str, chunk_bytes: int = 128 << 20, reuse: bool = False, ring: list | None = None):
    """Two FASTQ files of mates -> (chunk1, chunk2) pairs holding the SAME number of whole records each, in file order,
    for Engine.submit_fastq_pair (the k-th record of one file is the mate of the k-th record of the other).  Here the
    host does count lines -- the two cuts have to fall after the same record number.  Uncompressed files: _plain_pair_chunks
    (threads; reuse / ring as in text_chunks); gzip: one thread through gzip."""
    if not path1.endswith(".gz") and not path2.endswith(".gz"):
        yield from _plain_pair_chunks(path1, path2, chunk_bytes, reuse, ring)
        return

    def newlines(buf):
        return np.flatnonzero(np.frombuffer(buf, np.uint8) == 10)

    with _open(path1) as f1, _open(path2) as f2:
        c1 = c2 = b""
        e1 = e2 = False
        while True:
            if not e1 and len(c1) < chunk_bytes:
                b = f1.read(chunk_bytes - len(c1)); e1 = not b; c1 += b
            if not e2 and len(c2) < chunk_bytes:
                b = f2.read(chunk_bytes - len(c2)); e2 = not b; c2 += b
            if e1 and c1 and not c1.endswith(b"\n"):
                c1 += b"\n"
            if e2 and c2 and not c2.endswith(b"\n"):
                c2 += b"\n"
            n1, n2 = newlines(c1), newlines(c2)
            k = min(len(n1), len(n2)) // 4
            if k == 0:
                if e1 and e2:
                    if c1.strip() or c2.strip():
                        raise ValueError("paired FASTQ files hold different numbers of records (%s, %s)" % (path1, path2))
                    return
                if (e1 and not c1.strip()) or (e2 and not c2.strip()):
                    raise ValueError("paired FASTQ files hold different numbers of records (%s, %s)" % (path1, path2))
                chunk_bytes *= 2
                continue
            a, b = int(n1[4 * k - 1]) + 1, int(n2[4 * k - 1]) + 1
            yield c1[:a], c2[:b]
            c1, c2 = c1[a:], c2[b:]


def pair_cuts(path1: str, path2: str, chunk_bytes: int = 128 << 20) -> list[tuple[int, int, int, int]]:
    """(offset 1, bytes 1, offset 2, bytes 2) of the chunk pairs pair_chunks yields, for uncompressed files: ONE rank walks the
    mate files (the cuts have to fall after the same record number in both, so somebody has to count lines) and the others
    read their chunks by offset (multigpu.submit_fastq_shard)."""
    cuts, o1, o2 = [], 0, 0
    s1, s2 = os.path.getsize(path1), os.path.getsize(path2)
    for c1, c2 in pair_chunks(path1, path2, chunk_bytes):
        n1, n2 = min(len(c1), s1 - o1), min(len(c2), s2 - o2)      # (a newline added behind an unterminated last line is not in the file)
        cuts.append((o1, n1, o2, n2))
        o1 += n1; o2 += n2
    return cuts


def read_pair_cut(path1: str, path2: str, cut: tuple[int, int, int, int]) -> tuple[bytes, bytes]:
    out = []
    for path, off, n in ((path1, cut[0], cut[1]), (path2, cut[2], cut[3])):
        with open(path, "rb") as f:
            f.seek(off)
            b = f.read(n)
        out.append(b if b.endswith(b"\n") or not b else b + b"\n")
    return out[0], out[1]


def mates_share_names(path1: str, path2: str) -> bool:
    """True when the first records of the two mate files carry the same read name (the part of the header line before
    the first blank): `@x 1:N:0` / `@x 2:N:0` do, `@x/1` / `@x/2` do not -- and neither would they share a QNAME in the
    SAM that bowtie2 -U writes."""
    with _open(path1) as f1, _open(path2) as f2:
        h1, h2 = f1.readline().split(), f2.readline().split()
    return bool(h1) and bool(h2) and h1[0] == h2[0]


class prefetch:
    """Run the iterator `it` in a thread, `depth` items ahead: file reads (which release the GIL) overlap the consumer's
    GPU submissions.  Exceptions of the producer are raised in the consumer.  The thread starts at once (not at the first
    next()): a caller that opens the reader of sample k + 1 before it consumes sample k has the first chunks of k + 1
    read meanwhile."""

    def __init__(self, it, depth: int = 2):
        import queue
        self._q: queue.Queue = queue.Queue(maxsize=depth)
        self._end = object()
        self._done = False

        def work():
            try:
                for x in it:
                    self._q.put(x)
                self._q.put(self._end)
            except BaseException as e:      # noqa: BLE001 -- handed to the consumer
                self._q.put(e)

        self._t = threading.Thread(target=work, daemon=True)
        self._t.start()

    def __iter__(self):
        return self

    def __next__(self):
        if self._done:
            raise StopIteration
        x = self._q.get()
        if x is self._end:
            self._done = True
            raise StopIteration
        if isinstance(x, BaseException):
            self._done = True
            raise x
        return x


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


# ---- byte ranges of a bgzip'd FASTQ (one range per GPU: metamlst_amd/multigpu.py) -------------------------------------
def bgzf_find_block(f, pos: int, size: int) -> int:
    """Offset of the first BGZF block that starts at or after `pos` (`size` when there is none).  A candidate -- the gzip
    magic with the FEXTRA flag and a 'BC' subfield -- counts when the block it announces is followed by another valid
    header or by the end of the file (deflate data may contain the magic bytes by chance)."""
    if pos <= 0:
        return 0
    at = pos
    while at < size:
        f.seek(at)
        win = f.read((1 << 17) + 64)
        if len(win) < 18:
            return size
        i = 0
        while True:
            i = win.find(b"\x1f\x8b\x08\x04", i)
            if i < 0 or i + 18 > len(win):
                break
            n = _bgzf_block_size(win, i)
            if n > 0:
                nxt = at + i + n
                if nxt == size:
                    return at + i
                if nxt < size:
                    f.seek(nxt)
                    h2 = f.read(64)
                    n2 = _bgzf_block_size(h2, 0)
                    if n2 > 0 or (n2 == 0 and len(h2) >= 18 and h2[:4] == b"\x1f\x8b\x08\x04"):
                        return at + i
            i += 1
        at += 1 << 17
    return size


def _bgzf_read_block(f, off: int):
    """(total compressed size, inflated text) of the block at `off`; (0, b"") at the end of the file."""
    import zlib
    f.seek(off)
    head = f.read(18)
    if len(head) < 18:
        return 0, b""
    n = _bgzf_block_size(head, 0)
    if n == 0:                                     # extra field longer than usual: read more of the header
        f.seek(off)
        head = f.read(4096)
        n = _bgzf_block_size(head, 0)
    if n <= 0:
        raise ValueError("not a BGZF block at byte %d" % off)
    f.seek(off)
    blk = f.read(n)
    xlen = blk[10] | (blk[11] << 8)
    return n, zlib.decompress(blk[12 + xlen:n - 8], -15)


def bgzf_split(f, block_off: int, size: int):
    """Where the records of two neighbouring byte ranges part, for the range boundary at the block that starts at
    `block_off`: -> (end, text, p).  `text` is the inflated text of the blocks block_off .. end, `p` the offset in it of
    the first record that starts behind the first line break (record_start(text, 1)): text[:p] completes the last record
    of the range before, text[p:] opens the range behind.  Both ranks evaluate this same function, so they agree.  At
    the end of the file p = len(text)."""
    text, at = b"", block_off
    while at < size:
        n, t = _bgzf_read_block(f, at)
        if n == 0:
            break
        at += n
        text += t
        p = record_start(text, 1)
        if p >= 0:
            return at, text, p
    return at, text, len(text)


def bgzf_range_plan(path: str, lo: int, hi: int):
    """What the rank owning compressed bytes [lo, hi) of a bgzip'd FASTQ submits: {head: text to submit first (host
    inflated), mid: (first, end) compressed byte range of whole blocks to submit as BGZF, tail: text that completes the
    last record}.  A range without a block start owns nothing."""
    size = os.path.getsize(path)
    hi = min(hi, size)
    with open(path, "rb") as f:
        b_lo = bgzf_find_block(f, lo, size)
        b_hi = bgzf_find_block(f, hi, size) if hi < size else size
        if b_lo >= b_hi:
            return {"head": b"", "mid": (b_lo, b_lo), "tail": b""}
        if lo > 0:
            e_lo, t_lo, p_lo = bgzf_split(f, b_lo, size)
        else:
            e_lo, t_lo, p_lo = b_lo, b"", 0
        if b_hi < size:
            e_hi, t_hi, p_hi = bgzf_split(f, b_hi, size)
        else:
            e_hi, t_hi, p_hi = size, b"", 0
        if e_lo > b_hi:
            # the blocks inflated for the head reach into the next range (tiny ranges): both cuts lie in t_lo.
            # text position of block b_hi inside t_lo = inflated size of the blocks b_lo .. b_hi
            pos, at = 0, b_lo
            while at < b_hi:
                n, t = _bgzf_read_block(f, at)
                at += n
                pos += len(t)
            cut = pos + p_hi if b_hi < size else len(t_lo)
            if e_hi > e_lo:                       # the next boundary's record start lies beyond what the head inflated
                f_text, at = t_lo, e_lo
                while at < e_hi:
                    n, t = _bgzf_read_block(f, at)
                    at += n
                    f_text += t
                t_lo = f_text
            return {"head": t_lo[p_lo:max(cut, p_lo)], "mid": (b_hi, b_hi), "tail": b""}
        return {"head": t_lo[p_lo:], "mid": (e_lo, b_hi), "tail": t_hi[:p_hi]}
