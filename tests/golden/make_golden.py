"""Regenerates tests/golden/*.records.tsv from the CPU oracle.

The reference's own BWA path cannot run in this container (bwa submodule empty), so these
vectors are the ORACLE's output on the reference's own fixtures (tests/data/sim{1,2}_bcr.fq x
tests/data/tiny.fa.* copied into tests/golden/).  sim1 reads 0..1999 reproduce SURVEY.md Appendix F
(sha256 6d87c1f575dc9c2e7e450068210f41dff5bf3c528f590b970746d65292c6a662), an independent restatement.
Columns: read#, record#, flag, rid, 0-based pos, mapq, CIGAR, AS, NM, NA.
sim{1,2}_bcr.fq.gz are the reference's two fixture files IN FULL (10 000 reads each, gzip -9 -n of
tests/data/sim{1,2}_bcr.fq); sim{1,2}_full.records.tsv.gz hold the records of all of them ("always run the real fixture").
Options: mem_opt_init defaults, hardclip=false, keepSecFrac=0.9, maxSecondary=10, lrand48 from X0=0,
one draw per read, ordinal = read# within the file.
"""
import gzip
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import orc  # noqa: E402


def records_text(idx, opt, seqs, first=0):
    lines = []
    for i, s in enumerate(seqs):
        for j, r in enumerate(orc.align_sequence(opt, idx, s, ordinal=first + i)):
            lines.append("\t".join(map(str, [first + i, j, r["flag"], r["rid"], r["pos"], r["mapq"],
                                             orc.cigar_str(r["cigar"]), r["AS"], r["NM"], r["NA"]])))
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    idx = orc.Index.load(os.path.join(HERE, "tiny.fa"))
    opt = orc.default_opt()
    for fq, out in (("sim1_bcr.head3000.fq", "sim1_head3000.records.tsv"), ("sim2_bcr.head3000.fq", "sim2_head3000.records.tsv")):
        _, seqs = orc.read_fastq(os.path.join(HERE, fq))
        txt = records_text(idx, opt, seqs)
        open(os.path.join(HERE, out), "w").write(txt)
        print(out, hashlib.sha256(txt.encode()).hexdigest())
    for fq, out in (("sim1_bcr.fq.gz", "sim1_full.records.tsv.gz"), ("sim2_bcr.fq.gz", "sim2_full.records.tsv.gz")):
        _, seqs = orc.read_fastq(os.path.join(HERE, fq))
        txt = records_text(idx, opt, seqs)
        with gzip.GzipFile(os.path.join(HERE, out), "wb", 9, mtime=0) as f:
            f.write(txt.encode())
        print(out, len(seqs), "reads", txt.count("\n"), "records", hashlib.sha256(txt.encode()).hexdigest())
    _, seqs = orc.read_fastq(os.path.join(HERE, "sim1_bcr.head3000.fq"), 2000)
    print("appendixF", hashlib.sha256(records_text(idx, opt, seqs).encode()).hexdigest())
