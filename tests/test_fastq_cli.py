"""FASTQ reader (CPU) and the two CLI entry points end to end (GPU)."""
import gzip
import os
import tempfile

import numpy as np
import pytest

import fixtures as fx
from metamlst_amd import synth
from metamlst_amd.fastq import interleave, read_batches, tile_fasta


def write_fastq(path, bases, quals, gz=False):
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for k in range(bases.shape[0]):
            f.write(b"@r%d extra\n" % k + bases[k].tobytes() + b"\n+\n" + quals[k].tobytes() + b"\n")


def test_fastq_roundtrip_plain_gz_and_batches():
    rng = np.random.default_rng(0)
    b = synth._ACGT[rng.integers(0, 4, size=(25, 60))]
    q = (rng.integers(2, 41, size=(25, 60)) + 33).astype(np.uint8)
    d = tempfile.mkdtemp()
    write_fastq(d + "/a.fastq", b, q)
    write_fastq(d + "/a.fastq.gz", b, q, gz=True)
    for path in (d + "/a.fastq", d + "/a.fastq.gz"):
        got = list(read_batches(path, batch_reads=10))
        assert [len(g[2]) - 1 for g in got] == [10, 10, 5]
        assert np.array_equal(np.concatenate([g[0] for g in got]), b.reshape(-1))
        assert np.array_equal(np.concatenate([g[1] for g in got]), q.reshape(-1))
        assert got[0][3][3] == b"r3"
    pairs = list(interleave(d + "/a.fastq", d + "/a.fastq.gz", batch_pairs=100))[0]
    assert len(pairs[2]) - 1 == 50 and np.array_equal(pairs[0][:60], pairs[0][60:120])


def test_tile_fasta_covers_every_base_of_every_contig():
    rng = np.random.default_rng(5)
    lens = [1000, 149, 150, 151, 49, 50, 333]
    seqs = [synth._ACGT[rng.integers(0, 4, size=n)].tobytes() for n in lens]
    d = tempfile.mkdtemp()
    with open(d + "/c.fa", "wb") as f:
        for k, s in enumerate(seqs):
            f.write(b">c%d desc\n" % k + b"\n".join(s[i:i + 70].lower() if k == 0 else s[i:i + 70] for i in range(0, len(s), 70)) + b"\n")
    text = b"".join(tile_fasta(d + "/c.fa", 150, 25, 50, chunk_reads=7))
    recs = text.split(b"\n")[:-1]
    assert len(recs) % 4 == 0
    cover = [np.zeros(n, int) for n in lens]
    for k in range(0, len(recs), 4):
        ci, st = (int(x) for x in recs[k][1:].split(b"_"))
        w = recs[k + 1]
        assert recs[k + 2] == b"+" and recs[k + 3] == b"I" * len(w)
        assert w == seqs[ci][st:st + len(w)] and len(w) == min(150, lens[ci])
        cover[ci][st:st + len(w)] += 1
    assert cover[4].sum() == 0                                   # 49 bases: below min_len
    for ci in (0, 1, 2, 3, 5, 6):
        assert cover[ci].min() >= 1
    assert cover[0][200:800].min() == 6                          # 150 / 25 windows over every interior base


@pytest.mark.gpu
def test_cli_types_an_assembled_genome_from_its_contigs():
    """--contigs: the genome itself (three contigs, one reverse-complemented) instead of reads; the planted ST is called."""
    from metamlst_amd.cli import main
    db, idx = fx.ecoli_small(80)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][11], size=200_000)
    comp = np.zeros(256, np.uint8)
    for x, y in zip(b"ACGT", b"TGCA"):
        comp[x] = y
    parts = [g[:70_000], comp[g[70_000:140_000]][::-1], g[140_000:]]
    d = tempfile.mkdtemp()
    with open(d + "/asm12.fna", "wb") as f:
        for k, s in enumerate(parts):
            f.write(b">contig%d\n" % k + s.tobytes() + b"\n")
    assert main(["type", d + "/asm12.fna", "--contigs", "-d", db.path, "-o", d + "/out", "--quiet"]) == 0
    assert main(["merge", d + "/out", "-d", db.path]) == 0
    rep = open(d + "/out/merged/ecoli_report.txt").read().splitlines()
    assert rep[1].split("\t")[0] == "12" and rep[1].split("\t")[1] == "100.0"


@pytest.mark.gpu
def test_cli_type_then_merge_end_to_end():
    from metamlst_amd.cli import main
    db, idx = fx.ecoli_small(80)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][6], size=200_000)
    b, q = synth.sample_reads(g, 20000)
    d = tempfile.mkdtemp()
    write_fastq(d + "/iso7.fastq.gz", b, q, gz=True)
    assert main(["type", d + "/iso7.fastq.gz", "-d", db.path, "-o", d + "/out", "--quiet", "--log"]) == 0
    assert os.path.exists(d + "/out/iso7.nfo")
    assert main(["merge", d + "/out", "-d", db.path]) == 0
    rep = open(d + "/out/merged/ecoli_report.txt").read().splitlines()
    assert rep[0].startswith("ST\tConfidence") and rep[1].split("\t") == ["7", "100.0", "iso7"]


@pytest.mark.gpu
def test_gpu_fastq_parser_equals_host_parser():
    """mlst_submit_fastq (FASTQ text parsed on the GPU) gives the statistics of the host-parsed path: LF and CRLF,
    with and without a final newline, ragged lengths, N bases, several chunks."""
    from metamlst_amd.engine import Engine, MlstError
    from metamlst_amd.fastq import text_chunks
    import fixtures as fx2
    db, idx = fx2.ecoli_small(40)
    rng = np.random.default_rng(3)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][2], size=60_000)
    recs = []
    for k in range(5000):
        L = int(rng.choice([150, 150, 101, 75, 36, 250, 320, 1]))
        at = int(rng.integers(0, len(g) - L))
        r = bytearray(g[at:at + L].tobytes())
        if k % 7 == 0 and L > 5:
            r[int(rng.integers(L))] = ord("N")
        q = bytes((rng.integers(2, 42, size=L)).astype(np.uint8) + 33)
        recs.append((b"r%d some comment" % k, bytes(r), q))
    eng = Engine(0)
    eng.load_reference(idx)
    eng.submit_reads(*synth.ragged_reads([r for _, r, _ in recs], [q for _, _, q in recs]))
    want = eng.stats()
    want_items = fx2.sorted_items(eng.items(1 << 16))
    d = tempfile.mkdtemp()
    for eol, final in ((b"\n", True), (b"\r\n", True), (b"\n", False)):
        text = eol.join(b"@" + n + eol + r + eol + b"+" + eol + q for n, r, q in recs) + (eol if final else b"")
        eng.reset_sample()
        assert eng.submit_fastq(text) == len(recs)
        fx2.assert_stats_equal(eng.stats(), want)
        assert np.array_equal(fx2.sorted_items(eng.items(1 << 16)), want_items)
    # chunked through the file helper (chunks cut after whole records), read indices continue across chunks
    path = d + "/x.fastq"
    open(path, "wb").write(b"\n".join(b"@" + n + b"\n" + r + b"\n+\n" + q for n, r, q in recs) + b"\n")
    eng.reset_sample()
    n = sum(eng.submit_fastq(c) for c in text_chunks(path, chunk_bytes=200_000))
    assert n == len(recs)
    fx2.assert_stats_equal(eng.stats(), want)
    # malformed input is refused
    eng.reset_sample()
    with pytest.raises(MlstError, match="4-line records"):
        eng.submit_fastq(b"@a\nACGT\n+\n")
    with pytest.raises(MlstError, match="malformed FASTQ"):
        eng.submit_fastq(b"@a\nACGT\n+\nIII\n")
    with pytest.raises(MlstError, match="longer than 320"):
        eng.submit_fastq(b"@a\n" + b"A" * 400 + b"\n+\n" + b"I" * 400 + b"\n")
    assert eng.submit_fastq(b"") == 0


@pytest.mark.gpu
def test_bgzf_fastq_inflated_on_the_gpu_equals_the_text_path():
    """mlst_submit_fastq_bgzf: the compressed blocks are inflated by k_inflate and parsed on the GPU; chunks are cut between
    blocks, i.e. in the middle of records, which the next chunk completes.  Same statistics as the uncompressed text."""
    import zlib
    from bam_writer import _bgzf_block
    from metamlst_amd.engine import Engine, MlstError
    from metamlst_amd.fastq import bgzf_chunks
    from metamlst_amd.cli import main
    db, idx = fx.ecoli_small(60)
    rng = np.random.default_rng(9)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][4], size=80_000)
    recs = []
    for k in range(9000):
        L = int(rng.choice([150, 150, 101, 250, 60]))
        at = int(rng.integers(0, len(g) - L))
        recs.append(b"@read%d/1 x\n" % k + g[at:at + L].tobytes() + b"\n+\n" + bytes((rng.integers(2, 42, size=L)).astype(np.uint8) + 33) + b"\n")
    text = b"".join(recs)
    eng = Engine(0)
    eng.load_reference(idx)
    assert eng.submit_fastq(text) == len(recs)
    want = eng.stats()
    d = tempfile.mkdtemp()
    sizes = [65280, 1000, 30000, 17, 65280, 4096]
    blocks, at, k = [], 0, 0
    while at < len(text):
        n = sizes[k % len(sizes)]; k += 1
        blocks.append(_bgzf_block(text[at:at + n])); at += n
        if k % 5 == 0:
            blocks.append(_bgzf_block(b""))          # empty blocks are legal anywhere
    blocks.append(_bgzf_block(b""))
    path = d + "/s5.fastq.gz"
    open(path, "wb").write(b"".join(blocks))
    for chunk_bytes in (1 << 30, 90_000):
        eng.reset_sample()
        chunks = list(bgzf_chunks(path, chunk_bytes=chunk_bytes))
        n = sum(eng.submit_fastq_bgzf(c, final=last) for c, last in chunks)
        assert n == len(recs) and (chunk_bytes > 1 << 20 or len(chunks) > 5)
        fx.assert_stats_equal(eng.stats(), want)
    for chunk_bytes in (1 << 30, 70_001):       # raw reads of the file: the library takes the whole blocks of every buffer
        eng.reset_sample()
        assert eng.submit_fastq_bgzf_file(path, chunk_bytes=chunk_bytes) == len(recs)
        fx.assert_stats_equal(eng.stats(), want)
    # a corrupt block, a chunk that is not whole blocks, a file that ends inside a record
    eng.reset_sample()
    bad = bytearray(b"".join(blocks[:3]))
    bad[len(blocks[0]) + 40] ^= 0x5A
    with pytest.raises(MlstError, match="corrupt deflate|not a whole number"):
        eng.submit_fastq_bgzf(bytes(bad), final=True)
    eng.reset_sample()
    with pytest.raises(MlstError, match="not a whole BGZF block"):
        eng.submit_fastq_bgzf(b"".join(blocks[:2])[:-3], final=False)
    eng.reset_sample()
    with pytest.raises(MlstError, match="4-line records"):
        eng.submit_fastq_bgzf(_bgzf_block(text[:len(recs[0]) + 7]), final=True)
    # the CLI takes the bgzip'd file as it is
    assert main(["type", path, "-d", db.path, "-o", d + "/out", "--quiet"]) == 0
    assert os.path.exists(d + "/out/s5.nfo")


@pytest.mark.gpu
def test_cli_alignment_input_sam_and_bam_end_to_end():
    """`type --alignments` on a SAM and on the same records as BAM: reads tiled over the alleles of a known ST (perfect
    local alignments, AS = 2 x length) plus weaker secondary records on a neighbouring allele -> the planted ST."""
    import bam_writer
    from metamlst_amd.cli import main
    db, idx = fx.ecoli_small(80)
    st_row = 9
    want_alleles = db.profiles["ecoli"][st_row]
    recs, refs = [], []
    for a in range(idx.n_alleles):
        refs.append((idx.label(a), int(idx.off[a + 1] - idx.off[a])))
    k = 0
    for (gene, _), alno in zip(db.loci["ecoli"], want_alleles):
        l = idx.locus_index("ecoli", gene)
        b = int(idx.locus_begin[l]); nos = idx.allele_no[b:b + int(idx.locus_count[l])]
        a = b + int(np.nonzero(nos == alno)[0][0]); other = b + (int(np.nonzero(nos == alno)[0][0]) + 1) % int(idx.locus_count[l])
        seq = idx.sequence(a)
        for at in range(0, len(seq) - 100 + 1, 7):
            L = min(150, len(seq) - at)
            s = seq[at:at + L]
            tags = ["AS:i:%d" % (2 * L), "XS:i:%d" % (2 * L - 16), "XN:i:0", "XM:i:0", "XO:i:0", "XG:i:0", "NM:i:0", "YT:Z:UU"]
            recs.append(("q%d" % k, 0, idx.label(a), at + 1, 255, "%dM" % L, s, "I" * L, tags))
            tags2 = ["AS:i:%d" % (2 * L - 16), "XS:i:%d" % (2 * L), "XN:i:0", "XM:i:2", "XO:i:0", "XG:i:0", "NM:i:2", "YT:Z:UU"]
            recs.append(("q%d" % k, 256, idx.label(other), at + 1, 255, "%dM" % L, s, "I" * L, tags2))
            k += 1
    hdr = "@HD\tVN:1.0\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    d = tempfile.mkdtemp()
    with open(d + "/alnS.sam", "w") as f:
        f.write(hdr)
        for r in recs:
            f.write("\t".join([r[0], str(r[1]), r[2], str(r[3]), str(r[4]), r[5], "*", "0", "0", r[6], r[7]] + r[8]) + "\n")
    bam_writer.write_bam(d + "/alnB.bam", hdr, refs, recs)
    for name in ("alnS.sam", "alnB.bam"):
        assert main(["type", d + "/" + name, "--alignments", "-d", db.path, "-o", d + "/out", "--quiet"]) == 0
    assert open(d + "/out/alnS.nfo").read().replace("alnS", "X") == open(d + "/out/alnB.nfo").read().replace("alnB", "X")
    assert main(["merge", d + "/out", "-d", db.path]) == 0
    rep = sorted(open(d + "/out/merged/ecoli_report.txt").read().splitlines()[1:])
    assert [r.split("\t") for r in rep] == [[str(st_row + 1), "100.0", "alnB"], [str(st_row + 1), "100.0", "alnS"]]


def _ragged_records(g, rng, n=4000):
    recs = []
    for k in range(n):
        L = int(rng.choice([150, 150, 101, 75, 36, 160, 1]))
        at = int(rng.integers(0, len(g) - L))
        r = bytearray(g[at:at + L].tobytes())
        if k % 7 == 0 and L > 5:
            r[int(rng.integers(L))] = ord("N")
        if k % 11 == 0:
            r = bytearray(bytes(r).lower())
        q = bytes((rng.integers(2, 42, size=L)).astype(np.uint8) + 33)
        recs.append((b"r%d x" % k, bytes(r), q))
    return recs


def test_host_packer_makes_the_resident_read_format():
    """mlst_pack_fastq_host (the threaded host packer behind mlst_submit_packed_host; replaces, like mlst_submit_fastq, the
    user-run `bowtie2 -U <fastq>` of /root/reference/README.md:20): 2-bit rows in the group-transposed resident layout, raw
    Phred rows with bit 7 on non-ACGT bases, lengths with bit 15 on reads that hold one -- for LF / CRLF, with and without a
    final newline, ragged lengths, lower case, any number of threads."""
    import torch
    from metamlst_amd.engine import MlstError, pack_fastq_host
    rng = np.random.default_rng(5)
    g = np.frombuffer(bytes(rng.choice(list(b"ACGT"), size=30_000).astype(np.uint8)), np.uint8)
    recs = _ragged_records(g, rng, 1500)
    code = np.full(256, 4, np.uint8)
    for k, c in enumerate(b"ACGT"):
        code[c] = k; code[c + 32] = k
    for eol, final, threads in ((b"\n", True, 1), (b"\r\n", True, 3), (b"\n", False, 8)):
        text = eol.join(b"@" + n + eol + r + eol + b"+" + eol + q for n, r, q in recs) + (eol if final else b"")
        packed, qrows, lens, n, wpr, qstride = pack_fastq_host(text, 160, threads)
        assert n == len(recs) and wpr == 10 and qstride == 160
        rows = synth.tiled_to_rows(torch.from_numpy(packed.view(np.int32)), n, wpr).numpy().view(np.uint32)
        for k, (_, r, q) in enumerate(recs):
            c = code[np.frombuffer(r, np.uint8)]
            L = len(r)
            assert int(lens[k]) == (L | (0x8000 if (c == 4).any() else 0)), k
            want = np.zeros(wpr, np.uint32)
            for i in range(L):
                want[i >> 4] |= np.uint32((int(c[i]) & 3 if c[i] < 4 else 0) << (2 * (i & 15)))
            assert np.array_equal(rows[k], want), k
            wq = (np.frombuffer(q, np.uint8) - 33) | ((c == 4).astype(np.uint8) << 7)
            assert np.array_equal(qrows[k, :L], wq) and not qrows[k, L:].any(), k
    with pytest.raises(MlstError):
        pack_fastq_host(b"@a\nACGT\n+\n", 160)                      # not whole records
    with pytest.raises(MlstError):
        pack_fastq_host(b"@a\nACGT\n+\nIII\n", 160)                 # sequence and quality of different lengths
    with pytest.raises(MlstError):
        pack_fastq_host(b"@a\n" + b"A" * 200 + b"\n+\n" + b"I" * 200 + b"\n", 160)


@pytest.mark.gpu
def test_packed_host_submit_equals_the_text_path():
    """mlst_submit_packed_host (bases + lengths over the link, the sieve's candidates' Phred rows behind them) = mlst_submit_fastq
    on the same text: statistics, work items, allele choice and consensus; single reads with N and ragged lengths, pairs, two
    submissions in a row."""
    from metamlst_amd.engine import Engine, pack_fastq_host
    import fixtures as fx2
    db, idx = fx2.ecoli_small(40)
    rng = np.random.default_rng(8)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][2], size=60_000)
    recs = _ragged_records(g, rng, 6000)
    text = b"\n".join(b"@" + n + b"\n" + r + b"\n+\n" + q for n, r, q in recs) + b"\n"
    eng = Engine(0)
    eng.load_reference(idx)
    for paired in (False, True):
        eng.reset_sample()
        assert eng.submit_fastq(text, paired=paired) == len(recs)
        assert eng.submit_fastq(text, paired=paired) == len(recs)            # a second submission: read indices go on
        eng.typing_enqueue()
        want, want_ch, want_let = eng.typing_fetch()
        want_items = fx2.sorted_items(eng.items(1 << 16))
        eng.reset_sample()
        packed, qrows, lens, n, wpr, qstride = pack_fastq_host(text, 160)
        eng.submit_packed_host(packed, qrows, lens, n, wpr, qstride, paired=paired)
        eng.submit_packed_host(packed, qrows, lens, n, wpr, qstride, paired=paired)
        eng.typing_enqueue()
        got, got_ch, got_let = eng.typing_fetch()
        fx2.assert_stats_equal(got, want)
        assert np.array_equal(fx2.sorted_items(eng.items(1 << 16)), want_items)
        assert got_ch == want_ch and {a: bytes(v) for a, v in got_let.items()} == {a: bytes(v) for a, v in want_let.items()}
    eng.close()


@pytest.mark.gpu
def test_page_locked_buffers_feed_the_parser_like_ordinary_memory(tmp_path):
    """engine.pinned_array (mlst_alloc_host): FASTQ text submitted from page-locked memory -- whole, and as the pooled chunks
    fastq.text_chunks(reuse=True) hands out once the allocator is set, the way `cli type` reads files -- gives the statistics of the
    same text from ordinary memory; the memory goes back when the arrays are collected."""
    import gc
    from metamlst_amd import fastq
    from metamlst_amd.engine import Engine, pinned_array
    import fixtures as fx2
    db, idx = fx2.ecoli_small(40)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][1], size=50_000)
    b, q = synth.sample_reads(g, 4000, seed=9)
    text = b"".join(b"@r%d\n" % k + b[k].tobytes() + b"\n+\n" + q[k].tobytes() + b"\n" for k in range(len(b)))
    eng = Engine(0)
    eng.load_reference(idx)
    assert eng.submit_fastq(text) == len(b)
    want = eng.stats()
    pin = pinned_array(len(text) + 1000)
    pin[:len(text)] = np.frombuffer(text, np.uint8)
    eng.reset_sample()
    assert eng.submit_fastq(pin[:len(text)]) == len(b)
    fx2.assert_stats_equal(eng.stats(), want)
    path = tmp_path / "p.fastq"
    path.write_bytes(text)
    fastq.set_buffer_allocator(pinned_array)
    try:
        ring = []
        eng.reset_sample()
        n = sum(eng.submit_fastq(c) for c in fastq.prefetch(fastq.text_chunks(str(path), 300_000, reuse=True, ring=ring)))
        fastq.release_buffers(ring)
        assert n == len(b)
        fx2.assert_stats_equal(eng.stats(), want)
    finally:
        fastq.set_buffer_allocator(None)
    del pin, ring
    gc.collect()
    eng.close()


@pytest.mark.gpu
def test_bgzf_chunks_of_tens_of_thousands_of_blocks_are_cut_by_the_library():
    """Round 5: mlst_submit_fastq_bgzf sends a chunk through its three stages in PIECES it chooses itself -- 16,384 blocks into
    an empty pipeline, 8,192 at the end of a stream, up to 49,152 between (phase 1 by k_inflate_tok2 above 18,432) -- and pieces
    of one stream overlap: the parse of piece k - 1 runs beside the inflate of piece k, a partial record travels from piece to
    piece.  40,000 small BGZF blocks (blocks may be any size) whose boundaries fall inside records, as ONE final chunk, as
    two chunks, through the file reader, with the pipeline and the cuts switched off: the statistics of the text path every time,
    and the records a call reports add up to the file's."""
    import subprocess
    import sys
    from bam_writer import _bgzf_block
    from metamlst_amd.engine import Engine
    db, idx = fx.ecoli_small(60)
    rng = np.random.default_rng(19)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][2], size=60_000)
    recs = []
    for k in range(52_000):
        L = int(rng.choice([150, 150, 150, 75]))
        at = int(rng.integers(0, len(g) - L))
        recs.append(b"@r%d\n" % k + g[at:at + L].tobytes() + b"\n+\n" + bytes((rng.integers(2, 42, size=L)).astype(np.uint8) + 33) + b"\n")
    text = b"".join(recs)
    n_blocks = 40_000
    step = len(text) // n_blocks + 1
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as ex:
        blocks = list(ex.map(_bgzf_block, [text[at:at + step] for at in range(0, len(text), step)]))
    assert 39_000 < len(blocks) <= n_blocks
    whole = b"".join(blocks) + _bgzf_block(b"")
    eng = Engine(0)
    eng.load_reference(idx)
    assert eng.submit_fastq(text) == len(recs)
    want = eng.stats()
    # one final chunk: head of 16,384 + a middle piece + a tail of 8,192
    eng.reset_sample()
    assert eng.submit_fastq_bgzf(whole, final=True) == len(recs)
    fx.assert_stats_equal(eng.stats(), want)
    # two chunks cut between blocks (inside a record): the first meets an empty pipeline, the second is final
    cut = sum(len(b) for b in blocks[:27_001])
    eng.reset_sample()
    n = eng.submit_fastq_bgzf(whole[:cut], final=False)
    assert 0 < n < len(recs)                      # (the records of the first chunk's last piece arrive with the next call)
    n += eng.submit_fastq_bgzf(whole[cut:], final=True)
    assert n == len(recs)
    fx.assert_stats_equal(eng.stats(), want)
    # an open piece is finished by whatever looks at the sample next (its records are then in the statistics, not in a call's count)
    eng.reset_sample()
    n = eng.submit_fastq_bgzf(whole[:cut], final=False)
    partial = eng.stats()
    assert int(want.counters[0]) * 0.5 < int(partial.counters[0]) < int(want.counters[0])
    n += eng.submit_fastq_bgzf(whole[cut:], final=True)
    assert n < len(recs)
    fx.assert_stats_equal(eng.stats(), want)
    # the file reader (pieces cut anywhere, the library says how much it took)
    d = tempfile.mkdtemp()
    path = d + "/many.fastq.gz"
    open(path, "wb").write(whole)
    for chunk_bytes in (1 << 30, 5_000_001):
        eng.reset_sample()
        assert eng.submit_fastq_bgzf_file(path, chunk_bytes=chunk_bytes) == len(recs)
        fx.assert_stats_equal(eng.stats(), want)
    eng.close()
    # the switches: no cuts, no pipeline, phase 1 forced either way (read when the library is first used: a process each)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from metamlst_amd.engine import Engine; from metamlst_amd.index import load_index;"
            "idx = load_index(%r, cache=False); e = Engine(0); e.load_reference(idx); n = e.submit_fastq_bgzf_file(%r); s = e.stats();"
            "print(n, int(s.sum_score.sum()), int(s.n_hits.sum()), int(s.counters[0]), int(s.counters[1]))" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), db.path, path))
    ref = "%d %d %d %d %d" % (len(recs), int(want.sum_score.sum()), int(want.n_hits.sum()), int(want.counters[0]), int(want.counters[1]))
    for env in ({"MLST_BGZF_SPLIT": "0"}, {"MLST_BGZF_PIPE": "0"}, {"MLST_INFLATE_TOK": "1"}, {"MLST_INFLATE_TOK": "2"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        assert r.stdout.strip().splitlines()[-1] == ref, (env, r.stdout, ref)


@pytest.mark.gpu
def test_bgzf_chunks_of_tens_of_megabytes_are_walked_by_four_threads_and_false_block_starts_cost_nothing():
    """A chunk of 32 MB or more has its block headers walked by four threads, three of them from a block start they FIND
    behind their quarter mark; a list counts only where the chain of the one before lands on its first block.  Stored BGZF
    blocks (level 0: 36 MB for 36 MB of text) (a) as they are -- every thread finds a true start -- and (b) with the bytes of
    two chained BGZF headers in every record's name line, so that every thread but the first starts from a false one: the
    text path's statistics both times, as one chunk, through the file reader in pieces cut anywhere, and without the pipeline."""
    import struct
    import zlib
    from metamlst_amd.engine import Engine
    db, idx = fx.ecoli_small(60)
    rng = np.random.default_rng(23)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][1], size=60_000)
    fake = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x27\x00"      # a BGZF header that claims 40 bytes
    trap = fake + b"x" * 22 + fake + b"y" * 22                                          # ... and 40 bytes on the next one
    pool = []
    for k in range(3000):
        at = int(rng.integers(0, len(g) - 150))
        pool.append(g[at:at + 150].tobytes() + b"\n+\n" + bytes((rng.integers(2, 42, size=150)).astype(np.uint8) + 33) + b"\n")

    def stored_block(data: bytes) -> bytes:
        c = zlib.compressobj(0, zlib.DEFLATED, -15)
        comp = c.compress(data) + c.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp
                + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))

    eng = Engine(0)
    eng.load_reference(idx)
    for name_extra, n_rec in ((b"", 115_000), (trap, 92_000)):
        text = b"".join(b"@r%d " % k + name_extra + b"\n" + pool[k % len(pool)] for k in range(n_rec))
        whole = b"".join(stored_block(text[at:at + 65280]) for at in range(0, len(text), 65280)) + stored_block(b"")
        assert len(whole) > (34 << 20)
        eng.reset_sample()
        assert eng.submit_fastq(text) == n_rec
        want = eng.stats()
        eng.reset_sample()
        assert eng.submit_fastq_bgzf(whole, final=True) == n_rec
        fx.assert_stats_equal(eng.stats(), want)
        d = tempfile.mkdtemp()
        path = d + "/big.fastq.gz"
        open(path, "wb").write(whole)
        eng.reset_sample()
        assert eng.submit_fastq_bgzf_file(path, chunk_bytes=33_000_001) == n_rec      # (chunks end inside blocks: the library says how much it took)
        fx.assert_stats_equal(eng.stats(), want)
        os.environ["MLST_BGZF_PIPE"] = "0"
        try:
            e2 = Engine(0)
            e2.load_reference(idx)
            assert e2.submit_fastq_bgzf(whole, final=True) == n_rec
            fx.assert_stats_equal(e2.stats(), want)
            e2.close()
        finally:
            del os.environ["MLST_BGZF_PIPE"]
    eng.close()
