"""Minimal BAM writer for the tests (SAM/BAM specification: BGZF blocks + binary records).  Test infrastructure only."""
import struct
import zlib

OPS = "MIDNSHP=X"
SEQ16 = "=ACMGRSVTWYHKDBN"


def _bgzf_block(data: bytes) -> bytes:
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def _aux(tag_text: str) -> bytes:
    tag, typ, val = tag_text.split(":", 2)
    if typ == "i":
        v = int(val)
        for code, lo, hi, fmt in (("C", 0, 255, "<B"), ("c", -128, 127, "<b"), ("S", 0, 65535, "<H"), ("s", -32768, 32767, "<h"),
                                  ("I", 0, 4294967295, "<I"), ("i", -2147483648, 2147483647, "<i")):
            if lo <= v <= hi:      # samtools picks the smallest type too
                return tag.encode() + code.encode() + struct.pack(fmt, v)
    if typ == "A":
        return tag.encode() + b"A" + val.encode()
    if typ == "Z":
        return tag.encode() + b"Z" + val.encode() + b"\0"
    if typ == "f":
        return tag.encode() + b"f" + struct.pack("<f", float(val))
    raise ValueError(tag_text)


def write_bam(path: str, header_text: str, refs: list, records: list):
    """refs = [(name, length)], records = [(qname, flag, rname, pos1, mapq, cigar, seq, qual, [tag text, ...])]."""
    rid = {n: i for i, (n, _) in enumerate(refs)}
    out = bytearray(b"BAM\1" + struct.pack("<i", len(header_text)) + header_text.encode() + struct.pack("<i", len(refs)))
    for n, ln in refs:
        out += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", ln)
    for qn, flag, rn, pos, mapq, cigar, seq, qual, tags in records:
        ops, n = [], 0
        for ch in ("" if cigar == "*" else cigar):
            if ch.isdigit():
                n = n * 10 + int(ch)
            else:
                ops.append((n << 4) | OPS.index(ch)); n = 0
        s = "" if seq == "*" else seq
        nib = [SEQ16.index(ch.upper()) if ch.upper() in SEQ16 else 15 for ch in s] + [0]
        packed = bytes((nib[2 * i] << 4) | nib[2 * i + 1] for i in range((len(s) + 1) // 2))
        q = bytes([0xFF] * len(s)) if qual == "*" else bytes(ord(ch) - 33 for ch in qual)
        body = (struct.pack("<iiBBHHHiiii", rid.get(rn, -1), pos - 1, len(qn) + 1, mapq, 4680, len(ops), flag, len(s), -1, -1, 0)
                + qn.encode() + b"\0" + struct.pack("<%dI" % len(ops), *ops) + packed + q + b"".join(_aux(t) for t in tags))
        out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as f:
        for at in range(0, len(out), 60000):
            f.write(_bgzf_block(bytes(out[at:at + 60000])))
        f.write(_bgzf_block(b""))
