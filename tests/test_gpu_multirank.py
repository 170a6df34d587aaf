"""REAL engine processes: `cli type --gpus N` / `bench.py --gpus N` start their ranks
themselves; on the one-GPU box all of them sit on device 0 and the collectives go through gloo (MLST_ONE_GPU /
MLST_BACKEND; RCCL needs several GPUs).  What rank 0 writes must be byte for byte what one process writes.  Five ranks,
not eight: the GPU pool admits at most six processes on one card, and the test runner itself is one of them."""
import glob
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import fixtures as fx
from metamlst_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_fastq(path, bases, quals):
    n, L = bases.shape
    with open(path, "wb") as f:
        for k in range(n):
            f.write(b"@r%d\n" % k + bases[k].tobytes() + b"\n+\n" + quals[k].tobytes() + b"\n")


def run_cli(args, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-m", "metamlst_amd.cli"] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_cli_type_two_ranks_equals_one_process():
    d = tempfile.mkdtemp(prefix="mlst_mr_")
    db = synth.make_full_db(os.path.join(d, "m.db"), n_species=3, alleles_per_locus=25, n_profiles=10)
    parts_b, parts_q = [], []
    for k, sp in enumerate(db.species[:2]):
        g, _ = synth.make_genome(db, sp, db.profiles[sp][k], size=150_000, seed=50 + k)
        b, q = synth.sample_reads(g, 30_000, seed=60 + k)
        parts_b.append(b); parts_q.append(q)
    bases, quals = np.concatenate(parts_b), np.concatenate(parts_q)
    perm = np.random.default_rng(1).permutation(len(bases))
    fq = os.path.join(d, "sample.fastq")
    write_fastq(fq, bases[perm], quals[perm])
    chunk = {"MLST_FASTQ_CHUNK": str(3 << 20)}          # ~19 MB of text -> seven chunks
    out1, out2 = os.path.join(d, "one"), os.path.join(d, "two")
    run_cli(["type", fq, "-d", db.path, "-o", out1, "--log", "--quiet"], chunk)
    run_cli(["type", fq, "-d", db.path, "-o", out2, "--log", "--quiet", "--gpus", "2"], dict(chunk, MLST_ONE_GPU="1", MLST_BACKEND="gloo"))
    nfo1 = open(os.path.join(out1, "sample.nfo"), "rb").read()
    nfo2 = open(os.path.join(out2, "sample.nfo"), "rb").read()
    assert nfo1 and nfo1 == nfo2
    assert nfo1.count(b"\r\n") == 2                      # both species written
    log1 = open(glob.glob(os.path.join(out1, "sample_*.out"))[0], "rb").read()
    log2 = open(glob.glob(os.path.join(out2, "sample_*.out"))[0], "rb").read()
    assert log1 == log2                                  # every allele's hits / score of the --log table


def two_species_sample(d, n_each=30_000, seed=0, db=None):
    if db is None:
        db = synth.make_full_db(os.path.join(d, "m.db"), n_species=3, alleles_per_locus=25, n_profiles=10)
    parts_b, parts_q = [], []
    for k, sp in enumerate(db.species[:2]):
        g, _ = synth.make_genome(db, sp, db.profiles[sp][k], size=150_000, seed=50 + k + seed)
        b, q = synth.sample_reads(g, n_each, seed=60 + k + seed)
        parts_b.append(b); parts_q.append(q)
    bases, quals = np.concatenate(parts_b), np.concatenate(parts_q)
    perm = np.random.default_rng(1 + seed).permutation(len(bases))
    return db, bases[perm], quals[perm]


def bgzip(src, dst, block=65280, level=6):
    import struct
    import zlib
    raw = open(src, "rb").read()
    with open(dst, "wb") as f:
        for at in list(range(0, len(raw), block)) + [None]:
            data = raw[at:at + block] if at is not None else b""
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            comp = c.compress(data) + c.flush()
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp
                    + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


ONE_GPU = {"MLST_ONE_GPU": "1", "MLST_BACKEND": "gloo"}


def outputs(out_dir, name="sample"):
    nfo = open(os.path.join(out_dir, name + ".nfo"), "rb").read()
    logs = glob.glob(os.path.join(out_dir, name + "_*.out"))
    return nfo, (open(logs[0], "rb").read() if logs else None)


def test_cli_type_five_ranks_on_byte_ranges_plain_and_bgzip():
    """Rank r reads the r-th byte range of the file (plain: resynchronised on a record boundary; bgzip: whole BGZF blocks,
    the boundary blocks inflated on the host): .nfo and --log table byte-identical to one process."""
    d = tempfile.mkdtemp(prefix="mlst_mr6_")
    db, bases, quals = two_species_sample(d)
    fq = os.path.join(d, "sample.fastq")
    write_fastq(fq, bases, quals)
    bz = os.path.join(d, "bz", "sample.fastq.gz")
    os.mkdir(os.path.dirname(bz))
    bgzip(fq, bz)
    chunk = {"MLST_FASTQ_CHUNK": str(1 << 20)}
    one = os.path.join(d, "one")
    run_cli(["type", fq, "-d", db.path, "-o", one, "--log", "--quiet"], chunk)
    want = outputs(one)
    assert want[0].count(b"\r\n") == 2
    for tag, path in (("plain", fq), ("bgzip", bz)):
        out = os.path.join(d, "five_" + tag)
        run_cli(["type", path, "-d", db.path, "-o", out, "--log", "--quiet", "--gpus", "5"], dict(chunk, **ONE_GPU))
        got = outputs(out)
        assert got[0] == want[0], tag
        # the --log table names the input file in its first line; everything else must agree
        assert got[1].split(b"\n", 1)[1] == want[1].split(b"\n", 1)[1] or got[1] == want[1], tag
    # the single-process bgzip path (GPU inflate of the whole file) agrees as well
    out = os.path.join(d, "one_bgzip")
    run_cli(["type", bz, "-d", db.path, "-o", out, "--quiet"], chunk)
    assert outputs(out)[0] == want[0]


def test_cli_types_five_samples_on_two_ranks_like_five_runs():
    """Multi-sample mode (BASELINE configs[3], metamlst-merge.py:93-107 reads a folder of .nfo files): whole samples are
    dealt to the ranks, rank 0 gathers the .nfo lines -- byte for byte the five sequential runs."""
    d = tempfile.mkdtemp(prefix="mlst_ms_")
    folder = os.path.join(d, "reads")
    os.mkdir(folder)
    db = None
    names = []
    for k in range(5):
        db, bases, quals = two_species_sample(d, n_each=6_000 + 3_000 * k, seed=10 * k, db=db)
        names.append("s%d" % k)
        write_fastq(os.path.join(folder, names[-1] + ".fastq"), bases, quals)
    seq = os.path.join(d, "seq")
    for n in names:
        run_cli(["type", os.path.join(folder, n + ".fastq"), "-d", db.path, "-o", seq, "--quiet"], {})
    par = os.path.join(d, "par")
    run_cli(["type", folder, "-d", db.path, "-o", par, "--quiet", "--gpus", "2"], ONE_GPU)
    solo = os.path.join(d, "solo")
    run_cli(["type"] + [os.path.join(folder, n + ".fastq") for n in names] + ["-d", db.path, "-o", solo, "--quiet"], {})
    for n in names:
        want = open(os.path.join(seq, n + ".nfo"), "rb").read()
        assert want.count(b"\r\n") == 2
        assert open(os.path.join(par, n + ".nfo"), "rb").read() == want, n
        assert open(os.path.join(solo, n + ".nfo"), "rb").read() == want, n
    assert sorted(os.listdir(par)) == sorted(n + ".nfo" for n in names)


def test_cli_mate_files_count_a_pair_once_per_locus():
    """`cli type R1 -2 R2`: mates are interleaved on the GPU (mlst_submit_fastq_pair) and submitted as
    pairs, so sequenceBank (metamlst.py:127, Q3) holds one length per pair and locus: the coverage column the CLI prints
    (metamlst.py:228-230) equals the one computed from the API with paired=True, also on two ranks; with /1 /2 read names
    (two QNAMEs in bowtie2 -U's SAM) it equals the unpaired one."""
    from metamlst_amd import db as mdb
    from metamlst_amd.engine import Engine
    from metamlst_amd.typing import TypingArgs, type_sample
    db, idx = fx.ecoli_small(80)
    g, _ = synth.make_genome(db, "ecoli", db.profiles["ecoli"][2], size=100_000)
    b, q = synth.sample_pairs(g, n_pairs=int(100_000 * 20 / 300))
    d = tempfile.mkdtemp(prefix="mlst_pe_")
    database = mdb.metaMLST_db(db.path)
    eng = Engine(0)
    eng.load_reference(idx)
    fb, fq_, off = synth.flatten_reads(b, q)
    want = {}
    for paired in (True, False):
        eng.reset_sample()
        eng.submit_reads(fb, fq_, off, paired=paired)
        res = type_sample(idx, eng.stats(), eng.pileup, database, "x", TypingArgs(), out_dir=None)
        want[paired] = sorted((gene, str(v[3])) for r in res for gene, v in r.closest.items())
    eng.close()
    assert want[True] != want[False] and len(want[True]) == 7

    def coverage_column(stdout):
        rows = [ln.split() for ln in stdout.splitlines() if ln.startswith("  ") and not ln.startswith("  ->") and len(ln.split()) == 5]
        return sorted((r[0], r[1]) for r in rows)

    for style, shared in (((b" 1:N:0", b" 2:N:0"), True), ((b"/1", b"/2"), False)):
        r1, r2 = os.path.join(d, "s_%d_R1.fastq" % shared), os.path.join(d, "s_%d_R2.fastq" % shared)
        for path, rows, suf in ((r1, slice(0, None, 2), style[0]), (r2, slice(1, None, 2), style[1])):
            with open(path, "wb") as f:
                for k, (bb, qq) in enumerate(zip(b[rows], q[rows])):
                    f.write(b"@p%d%s\n" % (k, suf) + bb.tobytes() + b"\n+\n" + qq.tobytes() + b"\n")
        nfos = []
        for gpus, env in ((1, {}), (2, ONE_GPU)):
            out = os.path.join(d, "out_%d_%d" % (shared, gpus))
            stdout = run_cli(["type", r1, "-2", r2, "-d", db.path, "-o", out] + (["--gpus", "2"] if gpus == 2 else []),
                             dict(env, MLST_FASTQ_CHUNK=str(2 << 20)))
            assert coverage_column(stdout) == want[shared], (shared, gpus, stdout[-1500:])
            nfos.append(outputs(out, "s_%d_R1" % shared)[0])
        assert nfos[0] == nfos[1] and nfos[0]


def test_bench_five_ranks_on_one_gpu_reports_the_world():
    """`bench.py --gpus 5` on a tiny workload: self-launched ranks, both collective forms, one JSON line with the world
    size the backend reported and every planted ST called."""
    import json
    env = dict(os.environ, MLST_BENCH_ONE_GPU="1", MLST_BENCH_BACKEND="gloo")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "5", "--reads", "400000", "--species", "12", "--genomes", "4",
                        "--genome-size", "200000", "--steps", "3", "--warmup", "1", "--min-seconds", "0.05", "--pipeline", "2", "--cpu-seconds", "0",
                        "--no-secondary"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 5 and line["world_size_reported_by_backend"] == 5
    assert line["concordance"]["st_match"] and line["value"] > 0 and line["scaling"] == "weak"
    # the streamed exchange: pileup counts of the loci with a chosen allele only (4 of 12 species have a genome in the sample)
    ex = line["config"]["exchange_per_step"]
    assert line["config"]["collectives"].startswith("streamed") and ex["counts_layout"] == "compact"
    assert ex["counts_columns_needed"] * 16 <= ex["counts_bytes"] < ex["counts_bytes_fixed_layout"] and ex["steps_repeated_for_capacity"] == 0
    # ... the statistics of the loci that had hits in the steps before (4 x 7 of 84 loci), every engine slot on its own process
    # group, and the kernel figures behind `roofline` taken with the whole device at N > 1 as at N = 1
    assert ex["statistics_loci_listed"] == 28 and ex["statistics_bytes"] < ex["statistics_bytes_fixed_layout"] // 2
    assert ex["steps_repeated_for_statistics"] == 0 and ex["process_groups"] == 2
    assert line["roofline"]["measured_on"] == "whole device" 
