"""Runs small alignment configurations in child processes under a hard timeout and reports which ones finish:
   python scripts/hang_probe.py "knob=val,knob=val" ...   (each argument = one configuration; fixture reads, oracle check)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
import seqlib_amd as sl
from oracle import orc
g = os.path.join(%r, "tests", "golden")
_, s1 = orc.read_fastq(os.path.join(g, "sim1_bcr.head3000.fq"), 1200)
seqs = s1 + ["A" * 150, "AC" * 75, "ACG" * 50]
idx = sl.BWAIndex(); idx.LoadIndex(os.path.join(g, "tiny.fa"))
al = sl.BWAAligner(idx)
for kv in filter(None, sys.argv[1].split(",")):
    k, v = kv.split("="); al.set(k, int(v))
got = al.alignSequences(seqs)
exp = orc.align_batch(orc.default_opt(), orc.Index.load(os.path.join(g, "tiny.fa")), seqs)
print("fields that differ:", [k for k in ("hit_off", "rid", "pos", "flag", "mapq", "score", "nm", "na", "n_cigar", "cig_off", "cigar") if not np.array_equal(got[k], exp[k])])
''' % (ROOT, ROOT)
for cfg in sys.argv[1:]:
    try:
        r = subprocess.run([sys.executable, "-c", CHILD, cfg], capture_output=True, text=True, timeout=90)
        print(cfg, "->", r.stdout.strip()[-200:], r.stderr.strip()[-300:] if r.returncode else "", flush=True)
    except subprocess.TimeoutExpired:
        print(cfg, "-> HANG (killed after 90 s)", flush=True)
