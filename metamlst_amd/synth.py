"""Seeded synthetic inputs for the MLST-typing hot path.

The real metamlstDB_2022 is downloaded at first run by the reference
(metaMLST_functions.py:39-57) and is not available offline, so tests and the
benchmark use a schema-compatible SQLite database (tables as created at
metamlst-index.py:62-65), isolate genomes with the alleles of a chosen ST embedded,
and Illumina-like reads sampled from them.  Everything is a pure function of its seed.
"""
from __future__ import annotations

import sqlite3
from dataclasses import dataclass, field

import numpy as np

SEED = 20221
ECOLI_LOCI = [("adk", 536), ("fumC", 469), ("gyrB", 460), ("icd", 518),
              ("mdh", 452), ("purA", 478), ("recA", 510)]
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = ord("N")
for _a, _b in zip(b"ACGTacgt", b"TGCAtgca"):
    _COMP[_a] = _b


def create_schema(conn: sqlite3.Connection) -> None:
    """The four tables of the MetaMLST database, as created at metamlst-index.py:62-65."""
    c = conn.cursor()
    c.execute("CREATE TABLE IF NOT EXISTS organisms (organismkey varchar(255), label VARCHAR(255), PRIMARY KEY(organismkey))")
    c.execute("CREATE TABLE IF NOT EXISTS genes (geneName varchar(255), bacterium VARCHAR(255), PRIMARY KEY(geneName,bacterium))")
    c.execute("CREATE TABLE IF NOT EXISTS alleles (recID INTEGER PRIMARY KEY AUTOINCREMENT,bacterium varchar(255), gene VARCHAR(255), sequence TEXT, alignedSequence TEXT, alleleVariant INT)")
    c.execute("CREATE TABLE IF NOT EXISTS profiles (recID INTEGER PRIMARY KEY AUTOINCREMENT, profileCode INTEGER, bacterium VARCHAR(255), alleleCode INTEGER)")
    conn.commit()


def gen_locus_alleles(rng: np.random.Generator, length: int, n_alleles: int, max_div: float = 0.03,
                      snp_lo: int = 1, snp_hi: int = 8, indel_every: int = 0, root: np.ndarray | None = None) -> list[np.ndarray]:
    """Alleles of one locus: a random root (or the one given) and a random tree with snp_lo..snp_hi SNPs per edge,
    every allele within max_div of the root, all distinct.  indel_every > 0 additionally gives
    every indel_every-th allele a 1-3 bp deletion (exercises the banded Smith-Waterman path)."""
    root = rng.integers(0, 4, size=length, dtype=np.uint8) if root is None else np.asarray(root, dtype=np.uint8).copy()
    length = len(root)
    alleles = [root]
    ndiff = [0]
    seen = {root.tobytes()}
    budget = max(snp_lo, int(max_div * length))
    guard = 0
    while len(alleles) < n_alleles:
        guard += 1
        if guard > 200 * n_alleles:
            raise RuntimeError("allele generator cannot reach the requested count")
        p = int(rng.integers(len(alleles)))
        k = int(rng.integers(snp_lo, snp_hi + 1))
        if ndiff[p] + k > budget:
            continue
        child = alleles[p].copy()
        pos = rng.choice(len(child), size=k, replace=False)
        child[pos] = (child[pos] + rng.integers(1, 4, size=k, dtype=np.uint8)) % 4
        if indel_every and len(alleles) % indel_every == 0 and len(child) == length and length > 130:
            dl = int(rng.integers(1, 4))
            at = int(rng.integers(60, len(child) - 60))
            child = np.delete(child, np.arange(at, at + dl))
        b = child.tobytes()
        if b in seen:
            continue
        seen.add(b)
        alleles.append(child)
        ndiff.append(ndiff[p] + k)
    return alleles


@dataclass
class SynthDB:
    path: str
    species: list[str]
    loci: dict[str, list[tuple[str, int]]]                 # species -> [(gene, length)]
    n_alleles: dict[tuple[str, str], int] = field(default_factory=dict)
    profiles: dict[str, np.ndarray] = field(default_factory=dict)   # species -> int array [n_st, n_loci] (allele numbers)
    duplicates: dict = field(default_factory=dict)          # (species, gene) -> (species, gene) it is a near copy of (make_skewed_db)


def make_db(path: str, species_loci: dict[str, list[tuple[str, int]]], alleles_per_locus,
            n_profiles: int, seed: int = SEED, indel_every: int = 0, max_div: float = 0.03, roots: dict | None = None) -> SynthDB:
    """Write a schema-compatible SQLite database.  Allele numbers run 1..alleles_per_locus (an int, or a dict
    {(species, gene): count}); sequence type k (1-based) is a random allele tuple (profiles rows point at alleles.recID).
    roots = {(species, gene): (other species, other gene)}: the locus grows from a mutated copy of the other locus' root
    (near-duplicate loci across species: their seeds collide in the index)."""
    root_of: dict = {}
    rng = np.random.default_rng(seed)
    conn = sqlite3.connect(path)
    create_schema(conn)
    cur = conn.cursor()
    db = SynthDB(path=path, species=list(species_loci), loci=species_loci)
    for sp, loci in species_loci.items():
        cur.execute("INSERT INTO organisms (organismkey,label) VALUES (?,?)", (sp, "Synthetic " + sp))
        recid = {}
        for gene, length in loci:
            cur.execute("INSERT INTO genes (geneName,bacterium) VALUES (?,?)", (gene, sp))
            n_here = alleles_per_locus[(sp, gene)] if isinstance(alleles_per_locus, dict) else alleles_per_locus
            root = None
            if roots and (sp, gene) in roots:
                root = root_of[roots[(sp, gene)]].copy()
                pos = rng.choice(len(root), size=max(1, len(root) // 100), replace=False)      # ~1 % apart from the locus it copies
                root[pos] = (root[pos] + rng.integers(1, 4, size=len(pos), dtype=np.uint8)) % 4
            alleles = gen_locus_alleles(rng, length, n_here, max_div=max_div, indel_every=indel_every, root=root)
            root_of[(sp, gene)] = alleles[0]
            db.n_alleles[(sp, gene)] = len(alleles)
            for k, a in enumerate(alleles, start=1):
                s = _ACGT[a].tobytes().decode()
                cur.execute("INSERT INTO alleles (bacterium,gene,sequence,alignedSequence,alleleVariant) VALUES (?,?,?,?,?)",
                            (sp, gene, s, s, k))
                recid[(gene, k)] = cur.lastrowid
        prof = np.stack([rng.integers(1, db.n_alleles[(sp, g)] + 1, size=n_profiles) for g, _ in loci], axis=1)
        # distinct profiles only
        _, first = np.unique(prof, axis=0, return_index=True)
        prof = prof[np.sort(first)]
        db.profiles[sp] = prof
        rows = []
        for st, tup in enumerate(prof, start=1):
            for (gene, _), al in zip(loci, tup):
                rows.append((st, sp, recid[(gene, int(al))]))
        cur.executemany("INSERT INTO profiles (profileCode,bacterium,alleleCode) VALUES (?,?,?)", rows)
    conn.commit()
    conn.close()
    return db


def make_ecoli_db(path: str, alleles_per_locus: int = 1430, n_profiles: int = 5000, seed: int = SEED,
                  indel_every: int = 0) -> SynthDB:
    """DB-ecoli of SURVEY.md 8(d): 7 loci with Achtman-scheme lengths, ~10 k alleles."""
    return make_db(path, {"ecoli": list(ECOLI_LOCI)}, alleles_per_locus, n_profiles, seed, indel_every)


def make_full_db(path: str, n_species: int = 150, alleles_per_locus: int = 300, n_profiles: int = 200,
                 seed: int = SEED) -> SynthDB:
    """DB-full of SURVEY.md 8(d): stand-in for metamlstDB_2022 (n_species x 7 loci, lengths U[400,600])."""
    rng = np.random.default_rng(seed + 1)
    sl = {}
    for s in range(n_species):
        sl["sp%03d" % s] = [("g%d" % g, int(rng.integers(400, 601))) for g in range(7)]
    return make_db(path, sl, alleles_per_locus, n_profiles, seed)


def make_skewed_db(path: str, n_species: int = 6, seed: int = SEED, n_profiles: int = 50, lo: int = 10, hi: int = 10_000,
                   n_duplicates: int = 3) -> SynthDB:
    """A PubMLST-shaped database (the real one, metaMLST_functions.py:39-57 with the schema of
    metamlst-index.py:62-65, has loci with tens to thousands of alleles): alleles per locus log-uniform in lo..hi, locus
    lengths U[300, 700], and n_duplicates loci that are near copies (~1 % apart) of a locus of ANOTHER species, so that
    most of their seeds have postings in two loci and MLST_MAX_POSTINGS / the vote bins meet real collisions."""
    rng = np.random.default_rng(seed + 2)
    sl, counts = {}, {}
    for s in range(n_species):
        sp = "sk%03d" % s
        sl[sp] = [("g%d" % g, int(rng.integers(300, 701))) for g in range(7)]
        for g, _ in sl[sp]:
            counts[(sp, g)] = int(round(float(np.exp(rng.uniform(np.log(lo), np.log(hi))))))
    roots = {}
    for k in range(min(n_duplicates, n_species - 1)):
        src, dst = "sk%03d" % k, "sk%03d" % (k + 1)
        g = "g%d" % int(rng.integers(0, 7))
        roots[(dst, g)] = (src, g)
        sl[dst] = [(gg, (dict(sl[src])[g] if gg == g else ln)) for gg, ln in sl[dst]]
    db = make_db(path, sl, counts, n_profiles, seed, roots=roots)
    db.duplicates = dict(roots)
    return db


def allele_sequence(db_path: str, species: str, gene: str, allele: int) -> str:
    conn = sqlite3.connect(db_path)
    row = conn.execute("SELECT sequence FROM alleles WHERE bacterium=? AND gene=? AND alleleVariant=?",
                       (species, gene, allele)).fetchone()
    conn.close()
    return row[0]


def make_genome(db: SynthDB, species: str, allele_tuple, size: int = 4_600_000, seed: int = SEED,
                mutate: dict[str, list[tuple[int, str]]] | None = None) -> tuple[np.ndarray, dict[str, int]]:
    """An isolate genome: uniform-random ACGT with the given alleles embedded at spread-out
    positions on the forward strand.  mutate = {gene: [(pos, base), ...]} plants SNPs in the
    embedded copy (a novel allele).  Returns (ASCII uint8 array, {gene: start})."""
    rng = np.random.default_rng(seed + 7)
    g = _ACGT[rng.integers(0, 4, size=size, dtype=np.uint8)]
    loci = db.loci[species]
    starts = {}
    slot = size // (len(loci) + 1)
    for k, ((gene, _), al) in enumerate(zip(loci, allele_tuple)):
        seq = np.frombuffer(allele_sequence(db.path, species, gene, int(al)).encode(), dtype=np.uint8).copy()
        if mutate and gene in mutate:
            for pos, base in mutate[gene]:
                seq[pos] = ord(base)
        at = slot * (k + 1) + int(rng.integers(0, 1000))
        g[at:at + len(seq)] = seq
        starts[gene] = at
    return g, starts


def sample_reads(genome: np.ndarray, n_reads: int, read_len: int = 150, seed: int = SEED,
                 err_rate: float = 0.001, q_good: int = 40, q_bad: int = 15,
                 region: tuple[int, int] | None = None) -> tuple[np.ndarray, np.ndarray]:
    """Single-end reads uniform over both strands; Phred q_good everywhere except substitution
    errors (rate err_rate) reported at q_bad.  Returns (bases[n, L], quals[n, L]) as ASCII
    (quals Phred+33).  region=(lo, hi) restricts read starts (used to build on-locus fixtures)."""
    rng = np.random.default_rng(seed + 13)
    lo, hi = (0, len(genome) - read_len) if region is None else (max(0, region[0]), min(region[1], len(genome) - read_len))
    start = rng.integers(lo, hi + 1, size=n_reads)
    idx = start[:, None] + np.arange(read_len)[None, :]
    bases = genome[idx]
    rev = rng.random(n_reads) < 0.5
    bases[rev] = _COMP[bases[rev][:, ::-1]]
    quals = np.full((n_reads, read_len), q_good + 33, dtype=np.uint8)
    if err_rate > 0:
        err = rng.random((n_reads, read_len)) < err_rate
        ne = int(err.sum())
        if ne:
            code = np.searchsorted(_ACGT, bases[err])
            bases[err] = _ACGT[(code + rng.integers(1, 4, size=ne)) % 4]
            quals[err] = q_bad + 33
    return bases, quals


def sample_pairs(genome: np.ndarray, n_pairs: int, read_len: int = 150, insert_mean: float = 300.0,
                 insert_sd: float = 30.0, seed: int = SEED, err_rate: float = 0.001,
                 region: tuple[int, int] | None = None) -> tuple[np.ndarray, np.ndarray]:
    """Paired-end 2 x read_len, insert ~ N(mean, sd) (mates may overlap).  Mates are interleaved:
    rows 2k and 2k+1.  cfg5 of SURVEY.md 8(d)."""
    rng = np.random.default_rng(seed + 17)
    ins = np.clip(rng.normal(insert_mean, insert_sd, size=n_pairs).astype(np.int64), read_len, None)
    lo, hi = (0, len(genome) - int(ins.max())) if region is None else (max(0, region[0]), min(region[1], len(genome) - int(ins.max())))
    start = rng.integers(lo, hi + 1, size=n_pairs)
    i1 = start[:, None] + np.arange(read_len)[None, :]
    i2 = (start + ins - read_len)[:, None] + np.arange(read_len)[None, :]
    m1 = genome[i1]
    m2 = _COMP[genome[i2][:, ::-1]]
    flip = rng.random(n_pairs) < 0.5
    m1f, m2f = m1.copy(), m2.copy()
    m1f[flip], m2f[flip] = m2[flip], m1[flip]
    bases = np.empty((2 * n_pairs, read_len), dtype=np.uint8)
    bases[0::2], bases[1::2] = m1f, m2f
    quals = np.full(bases.shape, 40 + 33, dtype=np.uint8)
    if err_rate > 0:
        err = rng.random(bases.shape) < err_rate
        ne = int(err.sum())
        if ne:
            code = np.searchsorted(_ACGT, bases[err])
            bases[err] = _ACGT[(code + rng.integers(1, 4, size=ne)) % 4]
            quals[err] = 15 + 33
    return bases, quals


def flatten_reads(bases: np.ndarray, quals: np.ndarray) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """[n, L] arrays -> (bases concat, quals concat, uint64 offsets[n+1]) for mlst_submit_reads."""
    n, L = bases.shape
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    return np.ascontiguousarray(bases).reshape(-1), np.ascontiguousarray(quals).reshape(-1), off


def ragged_reads(reads: list[bytes], quals: list[bytes]) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Variable-length reads -> concatenated arrays + offsets."""
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    b = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    q = np.frombuffer(b"".join(quals), dtype=np.uint8).copy() if quals else np.zeros(0, np.uint8)
    return b, q, off


# ---------------------------------------------------------------------------------------------------------------------
# Resident batches made on the GPU (bench.py, the BASELINE-sized GPU tests).  torch is imported by the callers: this
# module stays importable without it.

def synth_reads_gpu(eng, torch, device, genome: np.ndarray, n_reads: int, L: int, seed: int, chunk: int = 1 << 19):
    """Reads of SURVEY.md 8(d) made on the GPU: uniform starts over `genome`, both strands, Phred 40 except 0.1 %
    substitution errors at Phred 15; packed with the engine's own pack kernel (mlst_pack_reads_device).
    Returns (packed int32 tensor in the resident group-transposed layout, qrows uint8, lens int16, wpr, qstride)."""
    wpr = (L + 15) // 16
    wpr += wpr & 1
    qstride = (L + 7) & ~7
    g = torch.from_numpy(genome).to(device)
    comp = torch.full((256,), ord("N"), dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    code = torch.zeros(256, dtype=torch.int64, device=device)
    for k, a in enumerate(b"ACGT"):
        code[a] = k
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    packed = torch.zeros((n_reads + 63) // 64 * 64 * wpr + 4, dtype=torch.int32, device=device)   # whole groups of 64 rows (mlst.h)
    qrows = torch.zeros(n_reads * qstride, dtype=torch.uint8, device=device)
    lens = torch.zeros(n_reads + 2, dtype=torch.int16, device=device)
    ar = torch.arange(L, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    for c0 in range(0, n_reads, chunk):
        n = min(chunk, n_reads - c0)
        start = torch.randint(0, len(genome) - L + 1, (n,), generator=gen, device=device)
        b = g[start[:, None] + ar[None, :]]
        rev = torch.rand(n, generator=gen, device=device) < 0.5
        b = torch.where(rev[:, None], comp[b.flip(1).long()], b)
        err = torch.rand((n, L), generator=gen, device=device) < 0.001
        sub = acgt[(code[b.long()] + torch.randint(1, 4, (n, L), generator=gen, device=device)) % 4]
        b = torch.where(err, sub, b).contiguous()
        q = torch.where(err, torch.tensor(15 + 33, dtype=torch.uint8, device=device),
                        torch.tensor(40 + 33, dtype=torch.uint8, device=device)).contiguous()
        off = (torch.arange(n + 1, device=device, dtype=torch.int64) * L).contiguous()
        torch.cuda.synchronize(device)
        eng.pack_reads_device(b.data_ptr(), q.data_ptr(), off.data_ptr(), n, packed.data_ptr() + c0 * wpr * 4,
                              qrows.data_ptr() + c0 * qstride, lens.data_ptr() + c0 * 2, wpr, qstride)
        eng.synchronize()
        del b, q, err, sub, start, rev, off
    return packed, qrows, lens, wpr, qstride


def tiled_to_rows(packed, n_reads: int, wpr: int):
    """Resident 2-bit rows (groups of 64 reads, transposed in 8-byte units, include/mlst.h) -> plain [n_reads, wpr] rows."""
    g = (n_reads + 63) // 64
    return packed[:g * 64 * wpr].view(g, wpr // 2, 64, 2).permute(0, 2, 1, 3).reshape(g * 64, wpr)[:n_reads]


def rows_to_tiled(rows, torch):
    """Plain [n, wpr] rows -> the resident group-transposed layout (+4 words of slack)."""
    n, wpr = rows.shape
    g = (n + 63) // 64
    pad = torch.zeros((g * 64, wpr), dtype=rows.dtype, device=rows.device)
    pad[:n] = rows
    t = pad.view(g, 64, wpr // 2, 2).permute(0, 2, 1, 3).contiguous().view(-1)
    return torch.cat([t, torch.zeros(4, dtype=rows.dtype, device=rows.device)])


def _told_apart(seqs: dict[int, bytes], allele: int, margin: int) -> bool:
    """True when `allele` differs from every other allele of its locus (of the same length) in a column at least `margin`
    columns away from both ends."""
    me = np.frombuffer(seqs[allele], np.uint8)
    inner = me[margin:len(me) - margin]
    for other, sq in seqs.items():
        if other == allele or len(sq) != len(me):
            continue
        if np.array_equal(np.frombuffer(sq, np.uint8)[margin:len(me) - margin], inner):
            return False
    return True


def metagenome_plan(sdb: SynthDB, n_genomes: int, seed: int = 5, margin: int = 8):
    """cfg3 of SURVEY.md 8(d): which species are in the mixture, their log-normal abundances and the ST planted in
    each.  -> [(species, abundance fraction, st_row)]; the same for every batch and rank (pure function of the seed and the
    database).  The planted ST of species k is the first profile row from k on whose seven alleles are told apart from every
    other allele of their locus by an interior column: a local aligner soft-clips a mismatch in the outermost ~4 columns, so
    two alleles that differ only there get the same records, the lower allele number wins the tie (metamlst.py:244) and the
    uncovered end column is filled from it -- the reference itself cannot recover such a planted ST (whole-batch check against
    the oracle: profiles/round2/check_batch.json)."""
    rng = np.random.default_rng(seed)
    chosen = list(rng.choice(len(sdb.species), size=min(n_genomes, len(sdb.species)), replace=False))
    ab = rng.lognormal(0.0, 1.0, size=len(chosen))
    ab /= ab.sum()
    conn = sqlite3.connect(sdb.path)
    plan = []
    for k, (si, fr) in enumerate(zip(chosen, ab)):
        sp = sdb.species[int(si)]
        prof = sdb.profiles[sp]
        seqs: dict[str, dict[int, bytes]] = {g: {} for g, _ in sdb.loci[sp]}
        for gene, no, sq in conn.execute("SELECT gene, alleleVariant, sequence FROM alleles WHERE bacterium=?", (sp,)):
            seqs[gene][int(no)] = sq.encode()
        row = k % len(prof)
        for r in range(len(prof)):
            cand = (k + r) % len(prof)
            if all(_told_apart(seqs[g], int(a), margin) for (g, _), a in zip(sdb.loci[sp], prof[cand])):
                row = cand
                break
        plan.append((sp, float(fr), row))
    conn.close()
    return plan


def make_metagenome_gpu(eng, torch, device, sdb: SynthDB, plan, n_reads: int, genome_size: int, seed: int, read_len: int = 150,
                        genomes: dict | None = None):
    """One resident batch of the mixed metagenome: every genome of `plan` contributes its share of n_reads, the reads
    are shuffled.  `genomes` caches the isolate genomes between batches.  -> (packed, qrows, lens, wpr, qstride, n_total)."""
    rows_all, q_all, l_all = [], [], []
    wpr = qstride = None
    for k, (sp, frac, st_row) in enumerate(plan):
        if genomes is not None and sp in genomes:
            g = genomes[sp]
        else:
            g, _ = make_genome(sdb, sp, sdb.profiles[sp][st_row], size=genome_size, seed=1000 + k)
            if genomes is not None:
                genomes[sp] = g
        n = max(1000, int(n_reads * frac))
        p, q, l, wpr, qstride = synth_reads_gpu(eng, torch, device, g, n, read_len, seed=seed * 1000 + k)
        rows_all.append(tiled_to_rows(p, n, wpr).clone())
        q_all.append(q[:n * qstride])
        l_all.append(l[:n])
        del p
    n_total = sum(int(x.shape[0]) for x in l_all)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    perm = torch.randperm(n_total, device=device, generator=gen)
    packed = rows_to_tiled(torch.cat(rows_all)[perm].contiguous(), torch)
    del rows_all
    qrows = torch.cat(q_all).view(n_total, qstride)[perm].contiguous().view(-1)
    del q_all
    lens = torch.cat([torch.cat(l_all)[perm], torch.zeros(2, dtype=torch.int16, device=device)])
    return packed, qrows, lens, wpr, qstride, n_total


def resident_to_host_reads(packed, qrows, n_reads: int, wpr: int, qstride: int, first: int, count: int, read_len: int = 150):
    """A slice of a resident batch back as ASCII (bases[count, L], quals Phred+33 [count, L]) numpy arrays on the host
    (for the CPU oracle: the parity checks of bench.py and the tests run it on the very reads the engine saw)."""
    pk = tiled_to_rows(packed, n_reads, wpr)[first:first + count].cpu().numpy().view(np.uint32).reshape(count, wpr)
    qr = qrows[first * qstride:(first + count) * qstride].cpu().numpy().reshape(count, qstride)
    codes = np.zeros((count, wpr * 16), np.uint8)
    for k in range(16):
        codes[:, k::16] = (pk >> np.uint32(2 * k)) & np.uint32(3)
    bases = _ACGT[codes[:, :read_len]]
    bases[(qr[:, :read_len] & 0x80) != 0] = ord("N")
    quals = ((qr[:, :read_len] & 0x7F) + 33).astype(np.uint8)
    return bases, quals


def resident_to_fastq_text(torch, packed, qrows, n_reads: int, wpr: int, qstride: int, first: int, count: int, read_len: int = 150):
    """A slice of a resident batch as FASTQ text (flat uint8 tensor on the GPU): 4-line records named '@r' + nine digits.
    Record = 12 + L + 3 + L + 1 bytes: '@r#########' LF, L bases LF, '+' LF, L qualities LF."""
    dev, L = packed.device, read_len
    pk = tiled_to_rows(packed, n_reads, wpr)[first:first + count]
    sh = (torch.arange(16, device=dev, dtype=torch.int32) * 2)[None, None, :]
    codes = ((pk[:, :, None] >> sh) & 3).reshape(count, wpr * 16)[:, :L].long()
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    qr = qrows[first * qstride:(first + count) * qstride].view(count, qstride)[:, :L]
    rec = torch.empty((count, 16 + 2 * L), dtype=torch.uint8, device=dev)
    rec[:, 0] = ord("@")
    rec[:, 1] = ord("r")
    ids = torch.arange(first, first + count, device=dev, dtype=torch.int64)
    for d in range(9):
        rec[:, 2 + d] = ((ids // (10 ** (8 - d))) % 10 + 48).to(torch.uint8)
    rec[:, 11] = 10
    rec[:, 12:12 + L] = torch.where((qr & 0x80) != 0, torch.tensor(ord("N"), dtype=torch.uint8, device=dev), acgt[codes])
    rec[:, 12 + L] = 10
    rec[:, 13 + L] = ord("+")
    rec[:, 14 + L] = 10
    rec[:, 15 + L:15 + 2 * L] = (qr & 0x7F) + 33
    rec[:, 15 + 2 * L] = 10
    return rec.view(-1)
