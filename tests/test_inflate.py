"""The deflate decoder of the BGZF path (csrc/inflate_dev.h) against zlib, on the host through its test hook, and the
BGZF chunker; the GPU kernel that runs the same code per block is checked in test_fastq_cli.py."""
import ctypes as C
import os
import tempfile
import zlib

import numpy as np
import pytest

from metamlst_amd import engine
from metamlst_amd.fastq import bgzf_chunks, is_bgzf
from bam_writer import _bgzf_block


def inflate(raw: bytes, cap: int):
    lib = engine.load_library()
    src = np.frombuffer(raw, np.uint8) if raw else np.zeros(1, np.uint8)
    out = np.zeros(max(cap, 1), np.uint8)
    n = C.c_uint64()
    rc = lib.mlst_selftest_inflate(src.ctypes.data_as(C.POINTER(C.c_uint8)), len(raw), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n))
    return rc, out[:int(n.value)].tobytes()


def inflate_canon(raw: bytes, cap: int):
    """csrc/inflate_canon.h (the decoder of k_inflate_tok2) on the host: -> (rc, bytes, left to the other kernel)"""
    lib = engine.load_library()
    src = np.frombuffer(raw, np.uint8) if raw else np.zeros(1, np.uint8)
    out = np.zeros(max(cap, 1), np.uint8)
    n, over = C.c_uint64(), C.c_int()
    rc = lib.mlst_selftest_inflate_canon(src.ctypes.data_as(C.POINTER(C.c_uint8)), len(raw), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n), C.byref(over))
    return rc, out[:int(n.value)].tobytes(), bool(over.value)


def deflate(data: bytes, level: int, strategy: int = zlib.Z_DEFAULT_STRATEGY) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    return c.compress(data) + c.flush()


def payloads():
    rng = np.random.default_rng(7)
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (k, bytes(rng.choice(list(b"ACGT"), 150).astype(np.uint8)), bytes((rng.integers(2, 41, 150) + 33).astype(np.uint8)))
                  for k in range(180))
    return [b"", b"A", b"abc" * 5000, bytes(rng.integers(0, 256, 40000, dtype=np.uint8)), fq[:65280], bytes(60000), b"ACGT" * 16000 + fq[:1000]]


@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                             (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)])
def test_decoder_equals_zlib(level, strategy):
    for data in payloads():
        rc, got = inflate(deflate(data, level, strategy), len(data))
        assert rc == 0 and got == data


@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                             (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)])
def test_canonical_limit_decoder_equals_zlib(level, strategy):
    """The decoder of k_inflate_tok2 (no look-up tables: 15 left-justified limits per code, 800 bytes of state per stream)
    and the replay of its tokens.  A block whose literal / length code uses more than 192 symbols is left to the other kernel
    (`over`): the fixed code (288 symbols) and random bytes do, FASTQ text never."""
    for k, data in enumerate(payloads()):
        rc, got, over = inflate_canon(deflate(data, level, strategy), len(data))
        assert rc == 0
        if over:
            assert strategy == zlib.Z_FIXED or k in (1, 3) or level == 0 or len(data) < 200, (k, level, strategy)      # (tiny inputs are sent with the fixed code)
        else:
            assert got == data, (k, level, strategy)
        if k == 4 and strategy != zlib.Z_FIXED:
            assert not over


def test_canonical_limit_decoder_rejects_damage():
    rng = np.random.default_rng(12)
    data = payloads()[4]
    raw = deflate(data, 6)
    assert inflate_canon(raw, len(data) - 1)[0] < 0
    assert inflate_canon(raw[:len(raw) // 2], len(data))[0] < 0
    for _ in range(300):
        bad = bytearray(raw)
        for _ in range(int(rng.integers(1, 6))):
            bad[int(rng.integers(len(bad)))] = int(rng.integers(256))
        rc, got, over = inflate_canon(bytes(bad), len(data))
        assert rc <= 0 and len(got) <= len(data)
        rc0, got0 = inflate(bytes(bad), len(data))
        if rc == 0 and not over and rc0 == 0:
            assert got == got0                           # (what both decoders accept, they decode alike)
    assert inflate_canon(b"\x07", 10)[0] < 0


def test_decoder_rejects_damage_without_reading_or_writing_out_of_bounds():
    rng = np.random.default_rng(11)
    data = payloads()[4]
    raw = deflate(data, 6)
    assert inflate(raw, len(data) - 1)[0] < 0            # output too small
    assert inflate(raw[:len(raw) // 2], len(data))[0] < 0   # input cut short
    for _ in range(300):                                  # random corruption: any return code, but it has to return
        bad = bytearray(raw)
        for _ in range(int(rng.integers(1, 6))):
            bad[int(rng.integers(len(bad)))] = int(rng.integers(256))
        rc, got = inflate(bytes(bad), len(data))
        assert rc <= 0 and len(got) <= len(data)
    assert inflate(b"\x07", 10)[0] < 0                   # block type 3


def test_bgzf_chunks_cut_between_blocks():
    rng = np.random.default_rng(3)
    text = bytes(rng.choice(list(b"ACGT\n"), 400000).astype(np.uint8))
    d = tempfile.mkdtemp()
    blocks = [_bgzf_block(text[i:i + 30000]) for i in range(0, len(text), 30000)] + [_bgzf_block(b"")]
    with open(d + "/t.gz", "wb") as f:
        f.write(b"".join(blocks))
    assert is_bgzf(d + "/t.gz")
    import gzip
    with gzip.open(d + "/plain.gz", "wb") as f:
        f.write(text)
    assert not is_bgzf(d + "/plain.gz")
    got = list(bgzf_chunks(d + "/t.gz", chunk_bytes=50000))
    assert [last for _, last in got] == [False] * (len(got) - 1) + [True] and len(got) > 3
    assert b"".join(c for c, _ in got) == b"".join(blocks)
    assert zlib.decompress(b"".join(c for c, _ in got), 31) == text[:30000]      # first member decodes
    with open(d + "/cut.gz", "wb") as f:
        f.write(b"".join(blocks)[:-5])
    with pytest.raises(ValueError, match="truncated"):
        list(bgzf_chunks(d + "/cut.gz"))


def _bgzf_raw(raw: bytes, data: bytes) -> bytes:
    """a BGZF block around an already deflated stream"""
    import struct
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(raw) + 25) + raw
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["2", "2c", "1"])
def test_device_decoder_equals_zlib_on_every_block_kind(mode, monkeypatch):
    """The device decoders on the GPU -- mode 2: the two kernels of csrc/inflate_lane.h (one lane per block -> tokens, one
    workgroup per block -> bytes by pointer jumping; blocks of more than 24,576 symbols, here the Huffman-only ones, fall to the
    other kernel); 2c: the same with phase 1 by csrc/inflate_canon.h (k_inflate_tok2: canonical limits, three waves per CU --
    the engine's choice for pieces of more than 24,576 blocks; blocks with the fixed code or more than 192 literal / length
    symbols fall to the other kernel); mode 1: the one-wave-per-block decoder of csrc/inflate_wave.h for every block -- on every
    payload x level x strategy in ONE buffer of BGZF blocks (one launch), blocks at odd byte offsets, compared with the
    bytes that went in.  Includes stored blocks (level 0 and incompressible data), fixed codes, long codes (Huffman
    only), overlapping matches (runs), matches 32 K back, empty blocks, blocks of odd and of full (65,536 bytes) length."""
    from metamlst_amd.engine import Engine, MlstError
    monkeypatch.setenv("MLST_INFLATE_MODE", mode[0])      # (read when an engine inflates for the first time)
    monkeypatch.setenv("MLST_INFLATE_TOK", "2" if mode == "2c" else "1")
    rng = np.random.default_rng(5)
    far = bytes(rng.integers(0, 256, 300, dtype=np.uint8))
    extra = [far + bytes(rng.integers(65, 70, 32300, dtype=np.uint8)) + far,        # a match at the far end of the window
             bytes(rng.integers(0, 4, 65000, dtype=np.uint8)),                       # low entropy: short codes
             b"".join(bytes([k]) * (k + 1) for k in range(256)) * 2]                 # runs of every length
    blocks, want = [], []
    for level, strategy in [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                            (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)]:
        for data in payloads() + extra:
            data = data[:65280]
            blocks.append(_bgzf_raw(deflate(data, level, strategy), data))
            want.append(data)
    full = (payloads()[4] * 2)[:65536]                                            # the format's largest block; and an odd length
    for data in (full, full[:65535], full[:1], full[:2], full[:3]):
        blocks.append(_bgzf_raw(deflate(data, 6), data))
        want.append(data)
    eng = Engine(0)
    got = eng.inflate_bgzf(b"".join(blocks))
    assert got == b"".join(want)
    # multi-block deflate streams inside one BGZF block (zlib starts a new deflate block every 16 K symbols at level 1)
    fq = payloads()[4]
    many = [_bgzf_raw(deflate(fq[:60000], 1), fq[:60000]) for _ in range(300)]
    assert eng.inflate_bgzf(b"".join(many)) == fq[:60000] * 300
    # damage: an error code, never a fault or a hang (the launch returns)
    raw = deflate(fq, 6)
    for k in range(40):
        bad = bytearray(raw)
        for _ in range(int(rng.integers(1, 6))):
            bad[int(rng.integers(len(bad)))] = int(rng.integers(256))
        try:
            out = eng.inflate_bgzf(_bgzf_raw(bytes(bad), fq))
            assert len(out) == len(fq)
        except MlstError:
            pass
    with pytest.raises(MlstError):
        eng.inflate_bgzf(_bgzf_raw(raw[:len(raw) // 2], fq))
    eng.close()


def test_block_headers_of_a_large_chunk_are_walked_by_four_threads():
    """mlst_submit_fastq_bgzf lists the blocks of a chunk of 32 MB or more with four threads (a header costs a cache miss and says
    where the next one is); a thread's list counts only where the chain of the one before it lands on the block start the thread
    found.  Host code: stored blocks (41 MB), (a) as they are: all four lists count; (b) with two chained BGZF headers
    planted in the data behind every quarter mark: the three threads that start there find them, their lists are dropped, the count
    is that of the serial walk; (c) a chunk cut inside its last block: not a whole block, as the serial walk says."""
    import struct
    lib = engine.load_library()

    def stored_block(data: bytes) -> bytes:
        c = zlib.compressobj(0, zlib.DEFLATED, -15)
        comp = c.compress(data) + c.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp
                + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))

    def walk(buf: bytes):
        a = np.frombuffer(buf, np.uint8)
        nb, tb, taken = C.c_uint64(), C.c_uint64(), C.c_int()
        rc = lib.mlst_debug_bgzf_walk(a.ctypes.data_as(C.POINTER(C.c_uint8)), len(buf), C.byref(nb), C.byref(tb), C.byref(taken))
        return rc, int(nb.value), int(tb.value), int(taken.value)

    rng = np.random.default_rng(31)
    payload = bytes(rng.integers(32, 127, 60_000, dtype=np.uint8))
    n = 700
    plain = b"".join(stored_block(payload[:59_000 + (k % 900)]) for k in range(n)) + stored_block(b"")
    assert len(plain) > (36 << 20)
    want_text = sum(59_000 + (k % 900) for k in range(n))
    assert walk(plain) == (0, n, want_text, 4)
    assert walk(plain[:20 << 20])[3] == 0                                  # (a small chunk: the serial walk alone)
    fake = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x27\x00"
    trap = fake + b"x" * 22 + fake + b"y" * 22
    # the traps go INTO the stored data right behind every quarter mark of the final buffer: build, find the marks, rebuild
    blocks = [bytearray(payload[:59_000 + (k % 900)]) for k in range(n)]
    sizes = [len(stored_block(bytes(b))) for b in blocks]
    starts = np.concatenate([[0], np.cumsum(sizes)])
    total = int(starts[-1]) + len(stored_block(b""))
    for q in (1, 2, 3):
        mark = total * q // 4
        k = int(np.searchsorted(starts, mark, side="right")) - 1          # the block that holds the mark
        inside = mark - int(starts[k]) - 23                                 # (18 bytes of header + 5 of the stored block's own in front of the data)
        at = max(inside, 0) + 40
        assert at + len(trap) < len(blocks[k]) - 100
        blocks[k][at:at + len(trap)] = trap
    trapped = b"".join(stored_block(bytes(b)) for b in blocks) + stored_block(b"")
    assert len(trapped) == total
    rc, nb, tb, taken = walk(trapped)
    assert (rc, nb, tb) == (0, n, want_text) and taken == 1
    assert walk(plain[:-40])[0] != 0 and walk(plain[:len(plain) - 28 - 30_000])[0] != 0
