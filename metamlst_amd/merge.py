"""Host side of the ST call: counterpart of metamlst-merge.py:93-341.

Parses the .nfo lines written by the typing run, matches every reconstructed locus to the
database (exact match, recurring new sequence, or new allele accepted when some database
allele of the locus is within `z` mismatches), resolves the ST with defineProfile and writes
merged/<species>_ST.txt and merged/<species>_report.txt.

The only heavy step -- the stringDiff scan over every allele of a locus
(metamlst-merge.py:177-181) -- is delegated to `matcher`, which in the product is the GPU
Hamming kernel (Engine.hamming_le).  Sequence outputs (--outseqformat A / A+ / B / B+ / C / C+, -j, --jgroup;
metamlst-merge.py:345-494) are written by write_sequences; format A needs MUSCLE only when the sequences of a locus
differ in length, exactly as in the reference.
"""
from __future__ import annotations

import itertools
import os
import shutil
import subprocess

from . import db as mdb


def parse_nfo_folder(folder: str, filter: str | None = None) -> dict:
    """metamlst-merge.py:93-107.  cel[organism] = [({label: (SEQ.upper(), acc, snp%)}, sample), ...].
    Note the species filter here is a SUBSTRING test on the raw option string (Q12)."""
    cel: dict = {}
    for file in sorted(os.listdir(folder)):
        if file.split(".")[-1] != "nfo":
            continue
        for line in open(folder + "/" + file, "r"):
            organism = line.split()[0]
            sampleName = line.split()[1]
            genes = line.split()[2::]
            if filter and organism not in filter:
                continue
            if organism not in cel:
                cel[organism] = []
            cel[organism].append((dict((x.split("::")[0], (x.split("::")[1].upper(), x.split("::")[2], x.split("::")[3]))
                                       for x in genes), sampleName))
    return cel


class EngineMatcher:
    """`matcher` backed by the GPU engine: is any loaded allele of (species, gene) within z of seq?"""

    def __init__(self, engine, index):
        self.engine, self.index = engine, index

    def __call__(self, bacterium: str, geneName: str, geneSeq: str, z: int) -> bool:
        locus = self.index.locus_index(bacterium, geneName)
        first, _ = self.engine.hamming_le(locus, geneSeq.encode(), z)
        return first >= 0


class SpeciesSession:
    """metamlst-merge.py:112-240 for one organism, split the way the reference runs it: the
    prologue (:119-142, all known profiles of the organism) happens once per merge run, then
    add_sample() is the body of the per-sample loop (:144-240).  State (new allele / new profile
    numbering, recurring sequences, isolates) carries across samples exactly as in the script."""

    def __init__(self, database: mdb.metaMLST_db, bacterium: str, z: int | None, matcher, cache: mdb.DbCache | None = None):
        self.database, self.bacterium, self.z, self.matcher = database, bacterium, z, matcher
        self.cache = cache
        cursor = database.cursor
        self.oldProfiles: dict = {}
        self.genesBase: dict = {}
        self.encounteredProfiles: dict = {}
        self.isolates: list = []
        self.newSequences: dict = {}
        self.lastProfile = 100000                                                # merge:134
        self.lastGenes = dict((row["gene"], 100000) for row in cursor.execute(
            "SELECT gene, MAX(alleleVariant) as maxGene FROM alleles WHERE bacterium = ? GROUP BY gene", (bacterium,)))   # merge:136
        self.has_empty = set(row["gene"] for row in cursor.execute(
            "SELECT DISTINCT gene FROM alleles WHERE bacterium = ? AND sequence = ''", (bacterium,)))
        for row in cursor.execute("SELECT profileCode,gene,alleleVariant FROM profiles,alleles WHERE alleleCode = alleles.recID "
                                  "AND alleles.bacterium = ?", (bacterium,)):     # merge:140-142
            if row["profileCode"] not in self.oldProfiles:
                self.oldProfiles[row["profileCode"]] = [0, {}]
            self.oldProfiles[row["profileCode"]][1][row["gene"]] = row["alleleVariant"]

    # the three per-sample database questions, SQL (reference) or cached (same answers)
    def _exists(self, seq):
        return self.cache.sequenceExists(self.bacterium, seq) if self.cache else mdb.sequenceExists(self.database.conn, self.bacterium, seq)

    def _locate(self, seq):
        return self.cache.sequenceLocate(self.bacterium, seq) if self.cache else mdb.sequenceLocate(self.database.conn, self.bacterium, seq)

    def _define(self, labels):
        return self.cache.defineProfile(labels) if self.cache else mdb.defineProfile(self.database.conn, labels)

    def add_sample(self, bacteriumLine: dict, sampleRecord: str):
        """One iteration of the loop at merge:144.  Returns the ST appended to `isolates` (or None
        when the sample's profile is rejected)."""
        bacterium, z = self.bacterium, self.z
        profileLine = {}
        newAlleles = []
        flagRecurrent = False
        sum_of_accuracies = 0.0
        for geneLabel, (geneSeq, geneAccur, percent_snps) in bacteriumLine.items():
            geneOrganism, geneName, geneAllele = geneLabel.split("_")
            sum_of_accuracies += float(geneAccur)
            if geneSeq == "" or self._exists(geneSeq):                            # merge:157
                if geneSeq != "":
                    geneAllele = self._locate(geneSeq)
                profileLine[geneName] = (geneAllele, 0)
            elif geneSeq in self.genesBase:                                       # merge:164
                profileLine[geneName] = (self.genesBase[geneSeq].split("_")[2], 2)
                flagRecurrent = True
            else:                                                                 # merge:168-196
                geneCategoryCode = 1
                if z is not None:
                    geneCategoryCode = 3
                    # stringDiff(geneSeq, '') == 0 <= z: an allele row with an empty sequence accepts anything;
                    # every other row of the locus is scanned on the GPU (mlst_hamming_le)
                    if (geneName in self.has_empty and 0 <= z) or self.matcher(bacterium, geneName, geneSeq, z):
                        geneCategoryCode = 1
                geneNewAlleleNumber = str(self.lastGenes[geneName] + 1)
                self.lastGenes[geneName] += 1
                geneNewLabel = geneOrganism + "_" + geneName + "_" + geneNewAlleleNumber
                self.genesBase[geneSeq] = geneNewLabel
                profileLine[geneName] = (geneNewAlleleNumber, geneCategoryCode)
                newAlleles.append(geneName)
                self.newSequences.setdefault(geneName, []).append((geneNewLabel, geneSeq))

        meanAccuracy = sum_of_accuracies / float(len(bacteriumLine))              # merge:199
        if len(newAlleles) == 0:
            if not flagRecurrent:
                tryDefine = self._define([bacterium + "_" + k + "_" + v[0] for k, v in profileLine.items()])
                if tryDefine and tryDefine[0][1] == 100:                          # merge:207
                    self.oldProfiles[tryDefine[0][0]][0] += 1
                    self.isolates.append((tryDefine[0][0], meanAccuracy, sampleRecord))
                    return tryDefine[0][0]
            foundExistant = 0
            for key, (element, abundance, isNewProfile) in self.encounteredProfiles.items():
                if [k + str(v[0]) for k, v in sorted(profileLine.items())] == [k + str(v[0]) for k, v in sorted(element.items())]:
                    foundExistant = key
            if foundExistant:
                self.encounteredProfiles[foundExistant][1] += 1
                self.isolates.append((foundExistant, meanAccuracy, sampleRecord))
                return foundExistant
            self.lastProfile += 1
            self.encounteredProfiles[self.lastProfile] = [profileLine, 1, 2]
            self.isolates.append((self.lastProfile, meanAccuracy, sampleRecord))
            return self.lastProfile
        self.lastProfile += 1                                                     # merge:229
        profileCategoryCode = 1
        if z is not None:
            for k, (v, cat) in profileLine.items():
                if cat == 3:
                    profileCategoryCode = 3
                    break
        self.encounteredProfiles[self.lastProfile] = [profileLine, 1, profileCategoryCode]
        if profileCategoryCode != 3:
            self.isolates.append((self.lastProfile, meanAccuracy, sampleRecord))
            return self.lastProfile
        return None

    def tables(self) -> dict:
        return dict(oldProfiles=self.oldProfiles, encounteredProfiles=self.encounteredProfiles, isolates=self.isolates,
                    lastGenes=self.lastGenes, newSequences=self.newSequences)


def parse_nfo_line(line: str):
    """One line of a .nfo file -> (organism, ({label: (SEQ.upper(), acc, snp%)}, sample)); merge:99-107."""
    organism = line.split()[0]
    sampleName = line.split()[1]
    genes = line.split()[2::]
    return organism, (dict((x.split("::")[0], (x.split("::")[1].upper(), x.split("::")[2], x.split("::")[3])) for x in genes), sampleName)


def call_species(database: mdb.metaMLST_db, bacterium: str, bactRecord: list, z: int | None, matcher,
                 cache: mdb.DbCache | None = None) -> dict:
    """metamlst-merge.py:112-240 for one organism.  Returns the tables the writers need."""
    sess = SpeciesSession(database, bacterium, z, matcher, cache)
    for bacteriumLine, sampleRecord in bactRecord:                                # merge:144
        sess.add_sample(bacteriumLine, sampleRecord)
    return sess.tables()


def write_species(folder: str, bacterium: str, tables: dict, meta: str | None = None, idField: int = 0) -> None:
    """metamlst-merge.py:119,253-292 (<sp>_ST.txt) and :298-341 (<sp>_report.txt)."""
    oldProfiles, encounteredProfiles = tables["oldProfiles"], tables["encounteredProfiles"]
    isolates, lastGenes = tables["isolates"], tables["lastGenes"]
    with open(folder + "/merged/" + bacterium + "_ST.txt", "w", newline="") as profil:
        profil.write("ST\t" + "\t".join([x for x in sorted(lastGenes.keys())]) + "\r\n")
        for profileCode, (hits, profile) in oldProfiles.items():
            profil.write(str(profileCode) + "\t" + "\t".join([str(v) for k, v in sorted(profile.items())]) + "\r\n")
        for profileID, (profile, hits, profileCategoryCode) in encounteredProfiles.items():
            if profileCategoryCode not in [1, 2]:
                continue
            profil.write(str(profileID) + "\t" + "\t".join([str(v[0]) for k, v in sorted(profile.items())]) + "\n")
    identifiers = {}
    p1line = False
    keys = []
    metadataJoinField = "sampleID"                                               # merge:130
    if meta:
        for line in open(meta):
            if line == "":
                continue
            if not p1line:
                p1line = True
                keys = [str(x).strip() for x in line.split("\t")]
                metadataJoinField = keys[idField]
            else:
                l = line.strip().split("\t")
                if len(l) == len(keys):
                    identifiers[l[idField]] = dict((keys[i], l[i]) for i in range(0, len(keys)))
    with open(folder + "/merged/" + bacterium + "_report.txt", "w", newline="") as isolafil:
        isolafil.write("ST\tConfidence\t" + "\t".join(keys) + "\n")
        STmapper: dict = {}          # merge:321: which samples (or their metadata rows) carry each ST
        for profileST, meanAccur, sampleName in isolates:
            if profileST not in STmapper:
                STmapper[profileST] = []
            if sampleName.endswith(".fna"):
                sampleName = sampleName.split(".")[0]
            if sampleName in identifiers:
                strl = [identifiers[sampleName][ky] for ky in keys]
                isolafil.write(str(profileST) + "\t" + str(round(meanAccur, 2)) + "\t" + "\t".join(strl) + "\n")
                STmapper[profileST].append(identifiers[sampleName])
            else:
                isolafil.write(str(profileST) + "\t" + str(round(meanAccur, 2)) + "\t" + str(sampleName) + "\n")
                STmapper[profileST].append({"sampleID": sampleName})
    return STmapper, metadataJoinField


def _fasta(records, path):
    """Bio.SeqIO.write(records, path, "fasta") [Biopython NOT IN TREE]: '>id description', sequence in lines of 60."""
    with open(path, "w") as f:
        for rid, seq in records:
            f.write(">" + rid + "\n")
            for at in range(0, len(seq), 60):
                f.write(seq[at:at + 60] + "\n")


def _muscle(seqs):
    """MuscleCommandline('muscle')(stdin=fasta) of merge:402-405; only reached when a locus has sequences of different lengths."""
    exe = shutil.which("muscle")
    if exe is None:
        raise RuntimeError("the sequences of a locus differ in length: MUSCLE is needed to align them "
                           "(metamlst-merge.py:402-403) and is not installed")
    fa = "".join(">%s\n%s\n" % (i, q) for i, q in seqs)
    out = subprocess.run([exe], input=fa.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout.decode()
    res, name = {}, None
    for line in out.splitlines():
        if line.startswith(">"):
            name = line[1:].split()[0]; res[name] = ""
        elif name is not None:
            res[name] += line.strip()
    return res


def write_sequences(folder: str, bacterium: str, tables: dict, database: mdb.metaMLST_db, outseqformat: str,
                    STmapper: dict, metadataJoinField: str = "sampleID", j: str | None = None, jgroup: bool = False) -> None:
    """metamlst-merge.py:345-494: merged/<sp>_sequences.fna (A, A+, B, B+) or merged/<sp>_sequences.txt (C)."""
    oldProfiles, encounteredProfiles = tables["oldProfiles"], tables["encounteredProfiles"]
    lastGenes, newSequences = tables["lastGenes"], tables["newSequences"]          # newSequences[gene] = [(label, seq), ...]
    base = folder + "/merged/" + bacterium + "_sequences"
    if outseqformat == "B":
        _fasta(sorted(itertools.chain(*newSequences.values()), key=lambda x: x[0]), base + ".fna")
    seqTable: dict = {}
    preaLignTable: dict = {}
    for row in database.cursor.execute("SELECT gene,alleleVariant,sequence FROM alleles WHERE bacterium = ? "
                                       "ORDER BY bacterium,gene,alleleVariant", (bacterium,)):
        preaLignTable.setdefault(row["gene"], []).append((bacterium + "_" + row["gene"] + "_" + str(row["alleleVariant"]), row["sequence"]))
    for seqGene, seqList in newSequences.items():
        preaLignTable.setdefault(seqGene, []).extend(seqList)
    if outseqformat == "B+":
        _fasta(sorted(itertools.chain(*preaLignTable.values()), key=lambda x: x[0]), base + ".fna")
    if outseqformat == "C":                                                       # 'C+' never enters this block (merge:368)
        with open(base + ".txt", "w", newline="") as seqfile:
            nalign_Table = dict(itertools.chain(*preaLignTable.values()))
            seqfile.write("ST\t" + "\t".join([str(x) for x in sorted(lastGenes.keys())]) + "\r\n")
            for profileCode, (hits, profile) in oldProfiles.items():
                if hits > 0 or outseqformat == "C+":
                    seqfile.write(str(profileCode) + "\t" + "\t".join([str(nalign_Table[bacterium + "_" + gen + "_" + str(alle)])
                                                                      for gen, alle in sorted(profile.items())]) + "\r\n")
            for profileCode, (profile, hits, isNewProfile) in encounteredProfiles.items():
                if isNewProfile == 3:
                    continue
                seqfile.write(str(profileCode) + "\t" + "\t".join([str(nalign_Table[bacterium + "_" + gen + "_" + str(alle[0])])
                                                                  for gen, alle in sorted(profile.items())]) + "\r\n")
    if outseqformat in ["A", "A+"]:
        for gene, seqs in preaLignTable.items():
            tld = []
            for _, q in seqs:
                if len(q) not in tld:
                    tld.append(len(q))
            if len(tld) > 1:
                seqTable.update(_muscle(seqs))
            else:
                for i, q in seqs:
                    seqTable[i] = str(q)
        phyloSeq = []

        def described(stSeq, profileCode, hits):
            # merge:431-447 / :469-484: one record per sample with the -j fields, or one per ST when --jgroup
            listofkeys = dict((k, []) for k in j.split(","))
            descriptionString = None
            if profileCode in STmapper:
                prog = 0
                for i in [x for x in STmapper[profileCode]]:
                    if jgroup:
                        descriptionString = "n=" + str(hits)
                        for (kl, v) in i.items():
                            if kl in listofkeys.keys():
                                listofkeys[kl].append(v)
                        descriptionString += "".join([kll + "{" + "|".join(ell) + "}" for kll, ell in listofkeys.items()])
                    else:
                        prog += 1
                        descriptionString = "-".join([kll + "{" + str(ell) + "}" for kll, ell in i.items() if kll in j.split(",")])
                        phyloSeq.append((bacterium + "_ST" + str(profileCode) + "_" + str(prog) + "_" + descriptionString, stSeq))
            if jgroup:
                phyloSeq.append((bacterium + "_ST" + str(profileCode) + "_" + descriptionString, stSeq))

        for profileCode, (hits, profile) in oldProfiles.items():
            stSeq = ""
            if hits > 0:
                for gen, alle in sorted(profile.items()):
                    stSeq += str(seqTable[bacterium + "_" + gen + "_" + str(alle)])
                if j:
                    described(stSeq, profileCode, hits)
                else:
                    for profileInstance in STmapper[profileCode]:
                        metadataPointer = metadataJoinField if metadataJoinField in profileInstance else "sampleID"
                        phyloSeq.append((bacterium + "_ST" + str(profileCode) + "_" + profileInstance[metadataPointer], stSeq))
            elif outseqformat == "A+":
                for gen, alle in sorted(profile.items()):
                    stSeq += str(seqTable[bacterium + "_" + gen + "_" + str(alle)])
                phyloSeq.append(("ST_" + str(profileCode), stSeq))
        for profileCode, (profile, hits, isNewProfile) in encounteredProfiles.items():
            if isNewProfile == 3:
                continue
            stSeq = ""
            for gen, alle in sorted(profile.items()):
                stSeq += str(seqTable[bacterium + "_" + gen + "_" + alle[0]])
            if j:
                described(stSeq, profileCode, hits)
            else:
                for profileInstance in STmapper[profileCode]:
                    metadataPointer = metadataJoinField if metadataJoinField in profileInstance else "sampleID"
                    phyloSeq.append((bacterium + "_ST" + str(profileCode) + "_" + profileInstance[metadataPointer], stSeq))
        _fasta(phyloSeq, base + ".fna")


def merge_folder(folder: str, database: mdb.metaMLST_db, matcher, z: int | None = 5, filter: str | None = None,
                 meta: str | None = None, idField: int = 0, cache: mdb.DbCache | None = None,
                 outseqformat: str | None = None, j: str | None = None, jgroup: bool = False) -> dict:
    """The whole metamlst-merge.py run for one folder of .nfo files.  Returns {species: tables}."""
    if not os.path.isdir(folder + "/merged"):
        os.makedirs(folder + "/merged")
    cel = parse_nfo_folder(folder, filter)
    out = {}
    for bacterium, bactRecord in cel.items():
        tables = call_species(database, bacterium, bactRecord, z, matcher, cache)
        STmapper, joinField = write_species(folder, bacterium, tables, meta, idField)
        if outseqformat:
            write_sequences(folder, bacterium, tables, database, outseqformat, STmapper, joinField, j, jgroup)
        out[bacterium] = tables
    return out
