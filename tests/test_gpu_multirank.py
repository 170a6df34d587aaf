"""Two REAL engine processes: `cli type --gpus 2` starts two ranks itself; on the one-GPU box both
sit on device 0 and the collectives go through gloo (MLST_ONE_GPU / MLST_BACKEND; RCCL needs two GPUs).  The FASTQ is cut
into several chunks dealt to the ranks; what rank 0 writes must be byte for byte what one process writes."""
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
