"""Database builder: FASTA sequences + typing tables -> the four MetaMLST tables.

Counterpart of the ingest half of metamlst-index.py (:90-217).  The other half
of that script -- dumping a FASTA and running bowtie2-build -- has no counterpart: the GPU index is
built by mlst_load_reference straight from the `alleles` table.  Same acceptance rules and quirks
as the reference (tests/golden/dbbuild holds tables produced by running the reference script)."""
from __future__ import annotations

import re
import sqlite3

from .synth import create_schema

MLST_KEYWORDS = ["clonal_complex", "species", "mlst_clade"]                                  # metaMLST_functions.py:409
_SKIP_COLUMNS = ["clonal_complex", "clonal-complex", "species", "mlst_clade", "Lineage", "comments", "CC", "mlst-clade"]   # metamlst-index.py:176
_NAME = re.compile("^([a-zA-Z0-9-])*$")
_NUM = re.compile("^([0-9])*$")


def read_fasta(path: str):
    """(id, sequence) per record; id = header up to the first blank, sequence lines joined (as SeqIO.parse)."""
    name, chunks = None, []
    for line in open(path):
        line = line.rstrip("\r\n")
        if line.startswith(">"):
            if name is not None:
                yield name, "".join(chunks)
            name, chunks = (line[1:].split() or [""])[0], []
        elif name is not None:
            chunks.append(line.strip())
    if name is not None:
        yield name, "".join(chunks)


def open_db(path: str) -> sqlite3.Connection:
    conn = sqlite3.connect(path)
    conn.row_factory = sqlite3.Row
    create_schema(conn)                      # metamlst-index.py:62-65
    return conn


def add_sequences(conn: sqlite3.Connection, fasta_files: list[str]) -> dict:
    """metamlst-index.py:92-137.  A record is taken iff its id is organism_gene_allele with
    [A-Za-z0-9-]* organism and gene and a numeric allele, and (organism, gene, allele) is not in the
    database yet.  Returns {file: {"added": n, "skipped": [ids]}}."""
    cursor = conn.cursor()
    report = {}
    for file in [f.strip() for f in fasta_files]:
        alleleList, geneList, skipped = [], {}, []
        for rec_id, sequence in read_fasta(file):
            splitLine = rec_id.split("_")
            if len(splitLine) != 3:
                skipped.append(rec_id)
                continue
            organism, gene, allele = splitLine
            if not (_NAME.match(organism) and _NAME.match(gene) and _NUM.match(allele)):
                skipped.append(rec_id)
                continue
            present = cursor.execute("SELECT 1 FROM alleles WHERE bacterium = ? AND gene = ? and alleleVariant = ?",
                                     (organism, gene, allele)).fetchall()
            if present:
                skipped.append(rec_id)
                continue
            geneList.setdefault(organism, [])
            if gene not in geneList[organism]:
                geneList[organism].append(gene)
            alleleList.append((gene, organism, allele, str(sequence)))
        cursor.executemany("INSERT OR IGNORE INTO genes (geneNAme, bacterium) VALUES (?,?)",
                           [(gen, org) for org, genes in geneList.items() for gen in genes])
        cursor.executemany("INSERT INTO alleles (gene, bacterium,alleleVariant,sequence) VALUES (?,?,?,?)", alleleList)
        report[file] = {"added": len(alleleList), "skipped": skipped}
    conn.commit()
    return report


def add_typings(conn: sqlite3.Connection, typing_files: list[str], logfile: str | None = "metamlst_logfile.log") -> dict:
    """metamlst-index.py:139-215.  A '#organism|Label' line starts an organism (its profiles are deleted
    first); the next line names the loci; every further line is `ST allele ...`.  A profile is loaded iff
    every locus allele exists in `alleles`; annotation columns are ignored; the others are logged."""
    cursor = conn.cursor()
    report = {}
    for file in [f.strip() for f in typing_files]:
        intest = 1
        profilesQuery, profilesLoaded = [], 0
        lines = open(file, "r").readlines()
        leng = len(lines) - 2
        problematicList: dict = {}
        organism = organismLabel = None
        genes: list = []
        for line in lines:
            if line.startswith("@"):
                continue
            if line.startswith("#"):
                organism = line.strip().split("|")[0].replace("#", "").replace("_", "")
                organismLabel = line.strip().split("|")[1] if len(line.strip().split("|")) == 2 else organism
                cursor.execute("INSERT OR IGNORE INTO organisms (organismkey,label) VALUES (?,?)", (organism, organismLabel))
                cursor.execute("DELETE FROM profiles WHERE bacterium = ?", (organism,))
                continue
            data = line.split()
            recID_Cache = dict((row["gene"] + "_" + str(row["alleleVariant"]), row["recID"]) for row in
                               cursor.execute("SELECT gene,alleleVariant,recID FROM alleles WHERE bacterium = ?", (organism,)))
            problematic = False
            if intest:
                intest = 0
                genes = data[1::]
            else:
                recIDs = []
                for key, variant in enumerate(data[1::]):
                    if key < len(genes):
                        if (genes[key] + "_" + str(variant)) in recID_Cache:
                            recIDs.append(recID_Cache[genes[key] + "_" + str(variant)])
                        elif genes[key] in _SKIP_COLUMNS:
                            continue
                        else:
                            problematicList.setdefault(str(data[0]), []).append(organism + "_" + genes[key] + "_" + variant)
                            problematic = True
                if not problematic:
                    profilesLoaded += 1
                    for element in recIDs:
                        profilesQuery.append((organism, data[0], element))
        if profilesLoaded > 0:
            cursor.execute("INSERT OR IGNORE INTO organisms (organismkey,label) VALUES (?,?)", (organism, organismLabel))
        cursor.executemany("INSERT INTO profiles (bacterium, profileCode, alleleCode) VALUES (?,?,?)", profilesQuery)
        if problematicList and logfile:
            with open(logfile, "a", newline="") as logf:      # metamlst-index.py:210-215, text as written by the reference
                logf.write("The following STs for " + organism + " were skipped as one or more of the alleles comprising the profile could not be found in your DB:\r\n")
                for key, element in problematicList.items():
                    logf.write("ST-" + " " + key + "\t".join(element) + " was missing \r\n")
                logf.write(("-" * 120) + "r\n")
        report[file] = {"loaded": profilesLoaded, "lines": leng, "problematic": problematicList}
    conn.commit()
    return report


def dump_db_to_fasta(conn: sqlite3.Connection, path: str, filterb: str | None = None) -> int:
    """metaMLST_functions.py:149-161: every allele with a sequence, id bacterium_gene_alleleVariant."""
    q = "SELECT bacterium,gene,alleleVariant,sequence FROM alleles WHERE sequence <> ''" + (" AND bacterium = ?" if filterb else "")
    rows = conn.execute(q, (filterb,) if filterb else ()).fetchall()
    with open(path, "w") as f:
        for row in rows:
            f.write(">%s_%s_%s\n%s\n" % (row["bacterium"], row["gene"], row["alleleVariant"], row["sequence"]))
    return len(rows)
