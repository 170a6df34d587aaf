"""Command-line entry points with the reference's flags.

    python -m metamlst_amd.cli type  SAMPLE.fastq[.gz] [-2 MATES.fastq] -d DB [-o out] [--penalty ...]   (metamlst.py:34-49)
    python -m metamlst_amd.cli merge FOLDER -d DB [-z 5] [--filter ...] [--meta ...] [--idField ...]       (metamlst-merge.py:35-49)
    python -m metamlst_amd.cli index -d DB [-s seqs.fasta,...] [-t typings.txt,...] [-q dump.fa] [--list]   (metamlst-index.py:24-33)

`type` takes reads instead of a bowtie2 BAM: the alignment happens on the GPU.  Everything it
writes (<out>/<sample>.nfo, optional --log file) has the reference's format; `merge` writes
merged/<species>_ST.txt and _report.txt.  --presorted / --debug / --version of the reference have
no meaning here and are accepted and ignored."""
from __future__ import annotations

import argparse
import os
import sys
import time

from . import db as mdb
from .engine import Engine, default_params
from .fastq import is_bgzf, mates_share_names, pair_chunks, prefetch, text_chunks, tile_fasta
from .index import load_index
from .merge import EngineMatcher, merge_folder
from .typing import TypingArgs, log_table, sample_name, type_sample


def _type_parser(sub):
    p = sub.add_parser("type", help="reconstruct the MLST loci of one sample from its reads (counterpart of metamlst.py)")
    p.add_argument("READS", nargs="+",
                   help="FASTQ file (plain, .gz or bgzip; `a.fq,b.fq` = two files of one sample, as bowtie2 -U takes them); with "
                        "--alignments: a SAM (plain or .gz) or BAM file.  Several files, or a folder of FASTQ files: every one is a "
                        "sample of its own, typed one after the other (with --gpus N: whole samples dealt to the GPUs, rank 0 "
                        "gathers the .nfo lines) -- the many-samples-into-one-folder use that metamlst-merge.py reads")
    p.add_argument("--alignments", action="store_true",
                   help="READS is a SAM / BAM made by `bowtie2 --very-sensitive-local -a --no-unal` against the database's "
                        "alleles (the reference's own input): hit accumulation as metamlst.py:101-130, pileup on the GPU")
    p.add_argument("--contigs", action="store_true",
                   help="READS is a FASTA of contigs or an assembled genome (the input of the reference's mlst.py): it is cut into "
                        "overlapping windows (--tile LEN,STEP) that go through the same path as reads")
    p.add_argument("--tile", default="150,25", metavar="LEN,STEP")
    p.add_argument("-2", dest="mates",
                   help="second FASTQ of a paired-end sample: record k of it is the mate of record k of READS.  Mates are aligned as "
                        "unpaired reads (the documented pipeline is bowtie2 -U r1,r2); when the two files give a pair ONE read name "
                        "(checked on the first record) the pair counts once per locus in the coverage figures, as in the "
                        "reference's sequenceBank (metamlst.py:127)")
    p.add_argument("-o", metavar="OUTPUT FOLDER", default="./out")
    p.add_argument("-d", "--database", metavar="DB PATH", required=True)
    p.add_argument("--filter", metavar="species1,species2...")
    p.add_argument("--penalty", default=100, type=int)
    p.add_argument("--minscore", default=80, type=int)
    p.add_argument("--max_xM", default=5, type=int)
    p.add_argument("--min_read_len", default=50, type=int)
    p.add_argument("--min_accuracy", default=0.90, type=float)
    p.add_argument("--nloci", default=100, type=int)
    p.add_argument("--log", action="store_true")
    p.add_argument("-a", action="store_true", help="Write known sequences")
    p.add_argument("--quiet", action="store_true")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--presorted", action="store_true")
    p.add_argument("--device", default=0, type=int)
    p.add_argument("--gpus", default=1, type=int,
                   help="type the sample on N GPUs of this node: one process per GPU, FASTQ chunks dealt to the ranks, the statistics "
                        "and pileup counts all-reduced over RCCL (plain or .gz FASTQ; the .nfo is byte-identical to --gpus 1)")
    p.add_argument("--max-retained", default=0, type=int, metavar="READS", help="capacity of the on-locus read store (default 4 M)")
    p.add_argument("--max-items", default=0, type=int, metavar="ITEMS", help="capacity of the (read, locus, strand) work-item list (default 8 M)")
    p.add_argument("--max-pair-results", default=0, type=int, metavar="PAIRS", help="capacity of the (item, allele) result arena (default 256 M)")
    p.add_argument("--depth-cap", default=0, type=int, metavar="N",
                   help="an ORDER-FREE approximation of pysam's pileup(max_depth) (metaMLST_functions.py:255-259 runs with 8000): a consensus column "
                        "sees the first N alignment records that span it, in read-index order.  NOT bit-identical to pysam on deep samples: htslib drops "
                        "whole reads at their start position in the coordinate-sorted BAM (DESIGN.md section 6); 0 (default) = all records")
    return p


def _merge_parser(sub):
    p = sub.add_parser("merge", help="detect the ST of every sample in a folder of .nfo files (counterpart of metamlst-merge.py)")
    p.add_argument("folder")
    p.add_argument("-d", "--database", metavar="DB PATH", required=True)
    p.add_argument("--filter", metavar="species1,species2...")
    p.add_argument("-z", metavar="ED", default=5, type=int)
    p.add_argument("--meta", metavar="METADATA_PATH")
    p.add_argument("--idField", default=0, type=int)
    p.add_argument("--outseqformat", choices=["A", "A+", "B", "B+", "C", "C+"])
    p.add_argument("-j", metavar="subjectID,diet,age...")
    p.add_argument("--jgroup", action="store_true")
    p.add_argument("--device", default=0, type=int)
    return p


def _index_parser(sub):
    p = sub.add_parser("index", help="build / extend a MetaMLST SQLite database (ingest half of metamlst-index.py; no bowtie2 index is needed)")
    p.add_argument("-t", "--typings")
    p.add_argument("-s", "--sequences")
    p.add_argument("-q", "--dump_db")
    p.add_argument("-i", "--buildindex", help="accepted and ignored: the GPU index is built from the database when it is loaded")
    p.add_argument("-d", "--database", metavar="DB PATH", required=True)
    p.add_argument("--list", action="store_true")
    p.add_argument("--filter", default=None)
    return p


def run_index(a) -> int:
    from . import dbbuild
    conn = dbbuild.open_db(a.database)
    if a.list:      # metamlst-index.py:80-86
        for key, label in mdb.db_getOrganisms(conn).items():
            print(key.ljust(30) + " " * 5 + label.ljust(30))
        return 0
    if a.sequences:
        for f, r in dbbuild.add_sequences(conn, a.sequences.split(",")).items():
            print("ADDING SEQUENCES %s Added %d seqs (%d skipped)" % (f, r["added"], len(r["skipped"])))
    if a.typings:
        for f, r in dbbuild.add_typings(conn, a.typings.split(",")).items():
            print("%d/%d PROFILES LOADED from %s" % (r["loaded"], r["lines"], f))
    if a.dump_db:
        print("%d sequences written to %s" % (dbbuild.dump_db_to_fasta(conn, a.dump_db, a.filter), a.dump_db))
    conn.commit()
    conn.close()
    return 0


FASTQ_SUFFIXES = (".fastq", ".fq", ".fastq.gz", ".fq.gz", ".fastq.bgz", ".fq.bgz")


def expand_samples(reads: list[str]) -> list[list[str]]:
    """READS arguments -> one list of files per sample (a folder contributes its FASTQ files in name order)."""
    out = []
    for r in reads:
        if os.path.isdir(r):
            out += [[os.path.join(r, f)] for f in sorted(os.listdir(r)) if f.endswith(FASTQ_SUFFIXES)]
        else:
            out.append(r.split(",") if "," in r and not os.path.exists(r) else [r])
    return out


def run_type(a, argv=None) -> int:
    samples = expand_samples(a.READS)
    if not samples:
        print("no FASTQ file found in " + ", ".join(a.READS))
        return 1
    many = len(samples) > 1
    if many and (a.alignments or a.contigs or a.mates):
        print("several samples at once: FASTQ input only (use `r1.fq,r2.fq` for a sample made of two files)")
        return 1
    a.READS, extra_files = samples[0][0], samples[0][1:]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world == 1:
        if a.alignments or a.contigs:
            print("--gpus applies to FASTQ input")
            return 1
        from .multigpu import launch_ranks
        return launch_ranks(a.gpus, list(argv if argv is not None else sys.argv[1:]))      # before this process touches a GPU
    rank, device = 0, None
    if world > 1:
        from .multigpu import init_from_env
        rank, world, device = init_from_env()
        a.device = device.index
    try:
        database = mdb.metaMLST_db(a.database)
        idx = load_index(a.database, a.filter.split(",") if a.filter else None)
    except Exception as e:   # metamlst.py:73-75
        print("Failed to connect to the database: please check your database file! (%s)" % e)
        return 1
    prm = default_params()
    prm.minscore, prm.max_xm, prm.min_read_len = a.minscore, a.max_xM, a.min_read_len
    prm.max_retained_reads, prm.max_items, prm.max_pair_results = a.max_retained, a.max_items, a.max_pair_results
    if a.depth_cap and world > 1 and not many:
        print("--depth-cap orders the records of the whole sample by read index: one sample on several GPUs cannot apply it (use --gpus 1)")
        return 1
    eng = Engine(a.device, prm)
    # the built host index is kept next to the database (as the reference keeps <idx>.1.bt2: metamlst-index.py:224-225) unless a
    # species filter made this index a one-off or MLST_INDEX_CACHE=0
    ref_cache = (a.database + ".mlstref") if (not a.filter and os.environ.get("MLST_INDEX_CACHE", "1") != "0") else ""
    eng.load_reference(idx, cache_path=ref_cache)
    from . import fastq as _fq
    from .engine import pinned_array
    _fq.set_buffer_allocator(pinned_array)      # file chunks are read into page-locked buffers
    if a.depth_cap:
        eng.set_depth_cap(a.depth_cap)
    engines = [eng]
    if many:      # the pipelined loop (metamlst_amd/pipeline.py): a few engines take turns on this rank's samples
        n_mine = (len(samples) + world - 1) // world
        # six engines: a sample of a few million reads is a chain of short device stages (copy, inflate, parse, pass 1, allele choice,
        # pile-up: 10-15 ms end to end, most of it latency) -- 16 bgzip'd samples of 2 M reads: 4 / 6 / 8 engines = 265 / 300 / 290 Mreads/s
        # (profiles/round5/inflate.md 6)
        for _ in range(max(0, min(int(os.environ.get("MLST_PIPELINE_DEPTH", "6")), n_mine) - 1)):
            e2 = Engine(a.device, prm)
            e2.load_reference(idx)      # (the host index is cached inside the library: an upload, not a build)
            if a.depth_cap:
                e2.set_depth_cap(a.depth_cap)
            engines.append(e2)
    targs = TypingArgs(penalty=a.penalty, minscore=a.minscore, max_xM=a.max_xM, min_read_len=a.min_read_len,
                       min_accuracy=a.min_accuracy, nloci=a.nloci, a=a.a, quiet=a.quiet, filter=a.filter, log=a.log)
    chunk_bytes = int(os.environ.get("MLST_FASTQ_CHUNK", str(256 << 20)))
    if many:
        from .multigpu import type_many_samples
        rc = type_many_samples(engines, idx, database, targs, samples, rank, world, a.o, a.log, chunk_bytes,
                               printer=None if a.quiet else (lambda results: _print_results(a, results)))
        database.closeConnection()
        return rc
    if a.alignments:
        from .samin import AlignmentSample
        smp = AlignmentSample(idx, targs).add_file(a.READS)
        return _finish_type(a, idx, database, targs, smp.stats(), lambda chosen: smp.pileup(eng, chosen))
    if a.contigs:
        read_len, stride = (int(x) for x in a.tile.split(","))
        for chunk in tile_fasta(a.READS, read_len, stride, a.min_read_len):
            eng.submit_fastq(chunk, paired=False)
        return _finish_type_device(a, eng, idx, database, targs)
    # Mates are unpaired reads for the aligner (bowtie2 -U r1,r2).  What a shared read name changes is sequenceBank
    # (metamlst.py:127: one entry per QNAME and locus): pairs whose files name both mates alike are submitted as pairs.
    paired = bool(a.mates) and mates_share_names(a.READS, a.mates)
    paths = [a.READS] + extra_files + ([a.mates] if a.mates else [])
    if world > 1:      # this rank's share of the files, then the two all-reduces; rank 0 writes (metamlst_amd/multigpu.py)
        from .multigpu import submit_fastq_shard, type_sharded
        submit_fastq_shard(eng, paths, rank, world, chunk_bytes, paired=paired)
        fileName = sample_name(a.READS)
        if rank == 0 and not os.path.isdir(a.o):
            os.mkdir(a.o)
        log_path = (a.o + "/" + fileName + "_" + str(int(time.time())) + ".out") if a.log else None
        results = type_sharded(eng, idx, database, targs, rank, world, device, fileName, a.o, log_path, a.READS)
        if rank == 0 and not a.quiet:
            _print_results(a, results)
        database.closeConnection()
        import torch.distributed as dist
        dist.destroy_process_group()
        return 0
    # FASTQ text goes to the GPU as is and is parsed there (mlst_submit_fastq); a reader thread stays two chunks ahead
    submit_sample_files(eng, paths, paired, chunk_bytes)
    return _finish_type_device(a, eng, idx, database, targs)


def _finish_type_device(a, eng, idx, database, targs) -> int:
    """Allele choice (metamlst.py:133-151, 244), pile-up and majority consensus on the device, queued behind pass 1
    (mlst_typing_enqueue): one host synchronisation per sample instead of three (statistics, host choice, pile-up)."""
    eng.typing_enqueue(penalty=targs.penalty)
    st, chosen, letters = eng.typing_fetch()
    return _finish_type(a, idx, database, targs, st, None, typed=(chosen, letters))


class _FileReader:
    """prefetch(text_chunks(path, reuse=True)) whose buffers go back to the pool when the consumer says it is done (close)."""

    def __init__(self, path: str, chunk_bytes: int, lo: int = 0, hi=None):
        self.ring: list = []
        self.it = prefetch(text_chunks(path, chunk_bytes, lo, hi, reuse=True, ring=self.ring))

    def __iter__(self):
        return self.it

    def close(self) -> None:
        from .fastq import release_buffers
        release_buffers(self.ring)


def open_sample_reader(paths, paired: bool, chunk_bytes: int):
    """The reader thread of a sample's FIRST file, started now (None when that file is not plain or gzip FASTQ text): a caller
    with many samples opens sample k + 1 before it feeds sample k, and k + 1's first chunks are read meanwhile."""
    if paired or not paths or is_bgzf(paths[0]):
        return None
    return _FileReader(paths[0], chunk_bytes)


def submit_sample_files(eng, paths, paired: bool, chunk_bytes: int, first_reader=None) -> None:
    """All reads of one sample's FASTQ file(s) into one engine (first_reader: open_sample_reader(paths, ...), if opened ahead)."""
    if paired:
        from .fastq import release_buffers
        ring: list = []
        for c1, c2 in prefetch(pair_chunks(paths[0], paths[1], chunk_bytes // 2, reuse=True, ring=ring)):
            eng.submit_fastq_pair(c1, c2)
        release_buffers(ring)
        return
    for k, path in enumerate(paths):
        if is_bgzf(path):      # bgzip'd FASTQ: the compressed blocks go to the GPU and are inflated there
            eng.submit_fastq_bgzf_file(path, paired=False)
            continue
        reader = first_reader if (k == 0 and first_reader is not None) else _FileReader(path, chunk_bytes)
        for chunk in reader:
            eng.submit_fastq(chunk, paired=False)      # (returns when the chunk has left the host buffer)
        reader.close()


def _finish_type(a, idx, database, targs, st, pileup_fn, typed=None) -> int:
    fileName = sample_name(a.READS)
    if not os.path.isdir(a.o):
        os.mkdir(a.o)
    if a.log:   # metamlst.py:159-172
        with open(a.o + "/" + fileName + "_" + str(int(time.time())) + ".out", "w", newline="") as f:
            f.write(log_table(idx, st, targs, a.READS))
    results = type_sample(idx, st, pileup_fn, database, fileName, targs, out_dir=a.o, typed=typed)
    if not a.quiet:
        _print_results(a, results)
    database.closeConnection()
    return 0


def _print_results(a, results) -> None:
    for r in results:
        print(" %-18s Detected Loci: %s" % (r.species, ", ".join(r.detected)))
        if r.missing:
            print(" " * 20 + "Missing Loci : " + ", ".join(r.missing))
        for g, (avg, hits, alleles, cov) in sorted(r.closest.items()):
            print("  %-7s%15s%7s%6s  %s" % (g, cov, avg, hits, ",".join(alleles[:5]) + ("... (%d more)" % len(alleles) if len(alleles) > 5 else "")))
        for l in r.loci_report:
            print("  %-7s%-7s%7s%7s%7s%15s%10s" % (l["locus"], l["ref"], l["length"], l["ns"], l["snps"], l["confidence"], l["notes"]))
        print("  -> " + ("Reconstruction Successful [WRITE]" if r.written else
                         ("Accuracy lower than %s%% [SKIP]" % round(a.min_accuracy * 100, 2) if r.passed_nloci else "not enough loci [SKIP]")))


def run_merge(a) -> int:
    database = mdb.metaMLST_db(a.database)
    idx = load_index(a.database)
    eng = Engine(a.device)
    eng.load_reference(idx)
    tables = merge_folder(a.folder, database, EngineMatcher(eng, idx), z=a.z, filter=a.filter, meta=a.meta, idField=a.idField,
                          cache=mdb.DbCache(database.conn, idx), outseqformat=a.outseqformat, j=a.j, jgroup=a.jgroup)
    for sp, t in tables.items():
        print("%s: %d sample(s) typed, %d new profile(s)" % (sp, len(t["isolates"]), sum(1 for v in t["encounteredProfiles"].values() if v[2] in (1, 2))))
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="metamlst_amd", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    _type_parser(sub)
    _merge_parser(sub)
    _index_parser(sub)
    a = ap.parse_args(argv)
    if a.cmd == "type":
        return run_type(a, argv)
    return {"merge": run_merge, "index": run_index}[a.cmd](a)


if __name__ == "__main__":
    sys.exit(main())
