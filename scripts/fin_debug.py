import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc
import seqlib_amd
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
_, s1 = orc.read_fastq(G + "/sim1_bcr.head3000.fq")
idx = seqlib_amd.BWAIndex(); idx.LoadIndex(G + "/tiny.fa")
lvl = int(sys.argv[1]); n = int(sys.argv[2])
al = seqlib_amd.BWAAligner(idx); al.set("fin_debug", lvl)
t = time.time(); r = al.alignSequences(s1[:n]); print("level", lvl, "n", n, "ok", time.time() - t, r["n_hits"], flush=True)
