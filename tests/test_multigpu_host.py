"""Host-side pieces of the multi-GPU paths that need no GPU: the rank launcher (`bench.py --gpus N` / `cli type --gpus N` start
their ranks with it before touching a device) and the FASTQ chunk dealing of `cli type --gpus N`."""
import os
import subprocess
import sys

import numpy as np

from metamlst_amd import multigpu
from metamlst_amd.fastq import text_chunks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spawn_ranks_exports_the_rendezvous_environment(tmp_path):
    out = tmp_path / "r"
    code = ("import os; open(r'%s' + os.environ['RANK'], 'w').write(' '.join(os.environ[k] for k in "
            "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')))" % str(out))
    assert multigpu.spawn_ranks(3, [sys.executable, "-c", code]) == 0
    seen = [open(str(out) + str(r)).read().split() for r in range(3)]
    assert [s[0] for s in seen] == ["0", "1", "2"] and all(s[1] == s[0] and s[2] == "3" and s[3] == "127.0.0.1" for s in seen)
    assert len({s[4] for s in seen}) == 1 and int(seen[0][4]) > 0


def test_spawn_ranks_returns_the_first_failure_and_ends_the_others():
    code = "import os, sys, time; r = int(os.environ['RANK']); sys.exit(7) if r == 1 else time.sleep(60)"
    rc = multigpu.spawn_ranks(3, [sys.executable, "-c", code])
    assert rc == 7


class FakeEngine:
    def __init__(self):
        self.calls = []
        self.base = 0

    def set_read_index_base(self, b):
        self.base = b

    def submit_fastq(self, chunk, paired=False):
        self.calls.append((self.base, bytes(chunk)))
        self.base += bytes(chunk).count(b"\n") // 4

    def submit_fastq_pair(self, c1, c2):
        self.calls.append((self.base, bytes(c1), bytes(c2)))


def make_fastq(path, n, seed, tag=b"r", name_suffix=b""):
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        for k in range(n):
            L = int(rng.integers(30, 151))
            q = bytes(rng.choice(np.frombuffer(b"@+I5#", np.uint8), L))      # quality lines that start like headers
            f.write(b"@%s%d%s\n%s\n+\n%s\n" % (tag, k, name_suffix, b"ACGT" * (L // 4) + b"A" * (L % 4), q))


def test_every_rank_reads_only_its_byte_range_and_the_order_keys_follow_the_file(tmp_path):
    """cli type --gpus N on plain FASTQ (VERDICT r2, item 3 i): rank r opens the file at size r / N, resynchronises on a
    record boundary and stops behind the record that starts last before size (r + 1) / N; bases are order keys."""
    p = tmp_path / "s.fastq"
    make_fastq(p, 5000, 3)
    whole = open(p, "rb").read()
    for world in (2, 3, 8):
        engines = [FakeEngine() for _ in range(world)]
        for r in range(world):
            multigpu.submit_fastq_shard(engines[r], [str(p)], r, world, 64 << 10)
        calls = sorted(c for e in engines for c in e.calls)
        assert b"".join(c[1] for c in calls) == whole                      # every record exactly once, in file order by key
        for r, e in enumerate(engines):
            got = sum(len(c[1]) for c in e.calls)
            assert abs(got - len(whole) / world) < 400                        # its share, give or take a record
            assert all((c[0] >> multigpu.ORDER_SHIFT) == r for c in e.calls)
            assert all(c[1][:1] == b"@" and c[1].count(b"\n") % 4 == 0 for c in e.calls)


def test_two_files_of_one_sample_keep_file_order(tmp_path):
    a, b = tmp_path / "a.fastq", tmp_path / "b.fastq"
    make_fastq(a, 700, 1, b"a"); make_fastq(b, 900, 2, b"b")
    engines = [FakeEngine() for _ in range(4)]
    for r in range(4):
        multigpu.submit_fastq_shard(engines[r], [str(a), str(b)], r, 4, 16 << 10)
    calls = sorted(c for e in engines for c in e.calls)
    assert b"".join(c[1] for c in calls) == open(a, "rb").read() + open(b, "rb").read()


def test_mate_files_are_dealt_in_pairs_of_chunks(tmp_path):
    a, b = tmp_path / "r1.fastq", tmp_path / "r2.fastq"
    make_fastq(a, 1200, 5, b"p", b" 1:N:0"); make_fastq(b, 1200, 6, b"p", b" 2:N:0")
    from metamlst_amd.fastq import mates_share_names
    assert mates_share_names(str(a), str(b))
    engines = [FakeEngine() for _ in range(3)]
    for r in range(3):
        multigpu.submit_fastq_shard(engines[r], [str(a), str(b)], r, 3, 32 << 10, paired=True)
    calls = sorted(c for e in engines for c in e.calls)
    assert all(c[0] % 2 == 0 and c[1].count(b"\n") == c[2].count(b"\n") for c in calls)
    assert b"".join(c[1] for c in calls) == open(a, "rb").read() and b"".join(c[2] for c in calls) == open(b, "rb").read()
    assert all(len(e.calls) > 0 for e in engines)
    c, d = tmp_path / "x1.fastq", tmp_path / "x2.fastq"
    make_fastq(c, 3, 5, b"p", b"/1"); make_fastq(d, 3, 6, b"p", b"/2")
    assert not mates_share_names(str(c), str(d))                             # @p0/1 and @p0/2: two QNAMEs in bowtie2 -U's SAM


def test_record_boundaries_survive_quality_lines_that_look_like_headers(tmp_path):
    from metamlst_amd.fastq import last_record_start, record_start
    rec = [b"@r0\nACGT\n+\n@+@+\n", b"@r1\nAC\n+\n@@\n", b"@r2\nA\n+\n+\n", b"@r3\nACG\n+r3\n@II\n"]
    buf = b"".join(rec)
    starts = np.cumsum([0] + [len(r) for r in rec])[:-1].tolist()
    for pos in range(len(buf)):
        want = next((s for s in starts if s >= pos), -1)
        assert record_start(buf, pos) == want, pos
    assert last_record_start(buf + b"@r4\nA") == starts[-1]           # a record whose '+' line is not in the buffer yet cannot be told
    for crlf in (buf.replace(b"\n", b"\r\n"),):
        assert record_start(crlf, 1) == crlf.index(b"@r1")


def test_bgzf_ranges_partition_the_records(tmp_path):
    """cli type --gpus N on bgzip'd FASTQ: whole BGZF blocks per rank, the text around every range boundary split at the
    first record start behind the boundary block's first line break (fastq.bgzf_range_plan) -- checked here with zlib."""
    import struct
    import zlib
    from metamlst_amd.fastq import _bgzf_block_size, bgzf_range_plan
    p = tmp_path / "t.fastq"
    make_fastq(p, 3000, 9)
    text = open(p, "rb").read()

    def block(data):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(data) + c.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp
                + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))

    def inflate(raw, a, b):
        out = b""
        while a < b:
            n = _bgzf_block_size(raw, a)
            xlen = raw[a + 10] | (raw[a + 11] << 8)
            out += zlib.decompress(raw[a + 12 + xlen:a + n - 8], -15)
            a += n
        return out

    for bs in (700, 65280):
        raw = b"".join(block(text[i:i + bs]) for i in range(0, len(text), bs)) + block(b"")
        z = tmp_path / ("t%d.fastq.gz" % bs)
        z.write_bytes(raw)
        for world in (1, 2, 5, 8, 40):
            got = b""
            for r in range(world):
                plan = bgzf_range_plan(str(z), len(raw) * r // world, len(raw) * (r + 1) // world if r + 1 < world else len(raw))
                part = plan["head"] + inflate(raw, *plan["mid"]) + plan["tail"]
                assert part == b"" or (part[:1] == b"@" and part.count(b"\n") % 4 == 0)
                got += part
            assert got == text, (bs, world)


def test_whole_samples_are_dealt_by_size():
    assert multigpu.deal_samples([5, 9, 3, 3, 7], 2) == [1, 0, 0, 0, 1]
    assert multigpu.deal_samples([4, 4, 4], 8) == [0, 1, 2]
    owner = multigpu.deal_samples(list(range(1, 41)), 8)
    load = [sum(s for s, o in zip(range(1, 41), owner) if o == r) for r in range(8)]
    assert max(load) - min(load) <= 8


def test_spawn_ranks_leaves_no_rank_behind_when_the_parent_is_interrupted(tmp_path):
    """A KeyboardInterrupt in the launcher (or any exception) ends the ranks: terminate, wait, kill."""
    pidfile = tmp_path / "pids"
    code = "import os, time; open(r'%s', 'a').write(str(os.getpid()) + '\\n'); time.sleep(120)" % str(pidfile)
    driver = ("import sys, threading, _thread, time; sys.path.insert(0, r'%s'); from metamlst_amd import multigpu\n"
              "threading.Timer(1.5, _thread.interrupt_main).start()\n"
              "try:\n    multigpu.spawn_ranks(3, [sys.executable, '-c', %r])\nexcept KeyboardInterrupt:\n    pass\n" % (ROOT, code))
    subprocess.run([sys.executable, "-c", driver], timeout=60, check=True)
    pids = [int(x) for x in open(pidfile).read().split()]
    assert len(pids) == 3
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except OSError:
            alive = False
        assert not alive


def test_pair_cuts_are_the_chunks_pair_chunks_yields(tmp_path):
    """Mate files at N > 1: rank 0 walks the two files once and broadcasts byte offsets (fastq.pair_cuts); what a rank
    reads by offset must be exactly the chunk pair pair_chunks yields there, also when the second file ends without a
    newline and when the records of the two files have different lengths (`-U r1,r2` of /root/reference/README.md:20)."""
    from metamlst_amd import fastq
    a, b = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    with open(a, "wb") as f:
        for k in range(700):
            f.write(b"@x%d 1:N:0\n" % k + b"ACGT" * (5 + k % 7) + b"\n+\n" + b"I" * (4 * (5 + k % 7)) + b"\n")
    with open(b, "wb") as f:
        for k in range(700):
            f.write(b"@x%d 2:N:0\n" % k + b"TTGCA" * (3 + k % 5) + b"\n+\n" + b"@" * (5 * (3 + k % 5)) + b"\n")
        f.seek(-1, 2)
        f.truncate()                                     # no newline behind the last quality line
    for chunk in (2048, 5000, 1 << 20):
        want = [(bytes(x), bytes(y)) for x, y in fastq.pair_chunks(a, b, chunk)]
        cuts = fastq.pair_cuts(a, b, chunk)
        assert [fastq.read_pair_cut(a, b, c) for c in cuts] == want
        assert sum(c[1] for c in cuts) == os.path.getsize(a) and sum(c[3] for c in cuts) == os.path.getsize(b)
        assert all(x.count(b"\n") == y.count(b"\n") and x.count(b"\n") % 4 == 0 for x, y in want)
        assert b"".join(x for x, _ in want) == open(a, "rb").read() and b"".join(y for _, y in want) == open(b, "rb").read() + b"\n"
    # the threaded walk of plain files cuts where the one-thread walk (kept for gzip) cuts
    import gzip
    for src in (a, b):
        with open(src, "rb") as f, gzip.open(src + ".gz", "wb") as g:
            g.write(f.read())
    for chunk in (2048, 5000):
        p, q = [(bytes(x), bytes(y)) for x, y in fastq.pair_chunks(a, b, chunk)], list(fastq.pair_chunks(a + ".gz", b + ".gz", chunk))
        # (the one-thread walk learns of a file's end one read later and hands the last record over in a chunk of its own)
        assert p[:-1] == q[:len(p) - 1] and b"".join(x for x, _ in p) == b"".join(x for x, _ in q) and b"".join(y for _, y in p) == b"".join(y for _, y in q)
    # buffers that go round behind a reader thread; files with different numbers of records are refused
    ring = []
    got = [(bytes(x), bytes(y)) for x, y in fastq.prefetch(fastq.pair_chunks(a, b, 3000, reuse=True, ring=ring))]
    fastq.release_buffers(ring)
    assert got == [(bytes(x), bytes(y)) for x, y in fastq.pair_chunks(a, b, 3000)]
    with open(b, "ab") as f:
        f.write(b"\n@extra\nAC\n+\nII\n")
    try:
        list(fastq.pair_chunks(a, b, 4096))
        raise AssertionError("different record counts went unnoticed")
    except ValueError:
        pass


def test_plain_file_chunks_partition_the_file_at_any_cut_and_buffers_go_round_safely(tmp_path):
    """fastq.text_chunks on plain files (positional reads by a few threads, chunks are views of buffers that are used again
    with reuse=True): whatever the chunk size and the byte ranges, every record comes exactly once, whole, in file order."""
    import time
    from metamlst_amd.fastq import prefetch
    p = tmp_path / "t.fastq"
    make_fastq(p, 900, 11)
    whole = open(p, "rb").read()
    for cb in (50, 300, 4096, 1 << 20):                                      # 50: shorter than a record -> the chunk size doubles
        chunks = [bytes(c) for c in text_chunks(str(p), cb)]
        assert b"".join(chunks) == whole
        assert all(c[:1] == b"@" and c.count(b"\n") % 4 == 0 for c in chunks)
    rng = np.random.default_rng(5)
    for _ in range(40):                                                      # arbitrary cuts, also in the middle of lines and at line starts
        cuts = sorted({0, len(whole)} | {int(x) for x in rng.integers(1, len(whole), size=int(rng.integers(1, 6)))})
        got = [b"".join(bytes(c) for c in text_chunks(str(p), 2000, lo, hi if hi < len(whole) else None)) for lo, hi in zip(cuts[:-1], cuts[1:])]
        assert b"".join(got) == whole, cuts
        assert all(g[:1] in (b"@", b"") for g in got)
    # no final newline, CRLF, trailing blank lines, an empty file
    for name, data in (("a", whole[:-1]), ("b", whole.replace(b"\n", b"\r\n")), ("c", whole + b"\n\n"), ("d", b"")):
        q = tmp_path / (name + ".fastq")
        q.write_bytes(data)
        got = b"".join(bytes(c) for c in text_chunks(str(q), 3000))
        assert got == (data.rstrip(b"\n") + b"\n" if name == "c" else data) or (name == "c" and got.rstrip() == data.rstrip())
    # buffers that go round: a slow consumer behind prefetch() still sees every chunk intact
    seen = []
    for c in prefetch(text_chunks(str(p), 1500, reuse=True)):
        time.sleep(0.001)
        seen.append(bytes(c))
    assert b"".join(seen) == whole


def test_read_ahead_of_the_next_sample_never_writes_into_chunks_still_in_the_consumers_hands(tmp_path):
    """cli.open_sample_reader starts the reader of sample k + 1 before sample k is fed; a walk's buffers go to the next walk
    only when the consumer has closed it (a walk that gave them away at its own end overwrote the last chunks of a sample
    that was still being submitted: a wrong count in one .nfo line out of five, found by the two-rank CLI test)."""
    import time
    from metamlst_amd import fastq
    from metamlst_amd.cli import open_sample_reader
    paths = []
    for k in range(4):
        p = tmp_path / ("s%d.fastq" % k)
        make_fastq(p, 300 + 40 * k, 20 + k, tag=b"s%d_" % k)
        paths.append(str(p))
    fastq.set_buffer_allocator(None)
    nxt = open_sample_reader([paths[0]], False, 3000)
    for k, path in enumerate(paths):
        mine, nxt = nxt, (open_sample_reader([paths[k + 1]], False, 3000) if k + 1 < len(paths) else None)
        time.sleep(0.02)                                                     # the next file's reader fills its queue meanwhile
        got = []
        for c in mine:
            view = c                                                         # what submit_fastq would read from
            time.sleep(0.0005)
            got.append(bytes(view))
        mine.close()
        assert b"".join(got) == open(path, "rb").read(), k


def test_raw_chunks_leave_a_margin_in_front_of_every_piece(tmp_path):
    """fastq.raw_chunks (the reader of bgzip'd files): pieces cut anywhere, each behind `margin` free bytes of its buffer."""
    from metamlst_amd import fastq
    p = tmp_path / "x.bin"
    data = np.random.default_rng(3).integers(0, 256, 300_001, dtype=np.uint8).tobytes()
    p.write_bytes(data)
    for lo, hi in ((0, None), (1234, 250_000), (299_990, None), (5, 5)):
        got, ring = [], []
        for buf, n in fastq.prefetch(fastq.raw_chunks(str(p), 70_000, lo, hi, margin=4096, reuse=True, ring=ring)):
            assert buf.size >= 4096 + n and 0 < n <= 70_000
            buf[4096 - 7:4096] = 1                                           # the consumer writes its carry in front: that is what the margin is for
            got.append(bytes(buf[4096:4096 + n]))
        fastq.release_buffers(ring)
        assert b"".join(got) == data[lo:hi]
