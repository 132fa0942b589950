"""A 45 kb contig whose last 135 bases are a (GTTAT)n tract of the synthetic E. coli-sized reference, through BWAAligner with the debug hooks.
   SLX_DEBUG_CYC=1 SLX_DEBUG_SUB=1 python scripts/tandem_probe.py"""
import os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seqlib_amd
from seqlib_amd import synth
cfg = synth.CONFIGS["C2"]
refs = synth.make_reference(cfg)
g = synth.genome_ascii_bytes(refs[0][1])
m = re.search(rb"(GTTAT){20,}", g)
end = min(m.end(), m.start() + 300)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 45052
contig = g[end - L:end]
idx = seqlib_amd.BWAIndex()
idx.ConstructIndex([(nm, synth.genome_ascii_bytes(gg)) for nm, gg in refs])
al = seqlib_amd.BWAAligner(idx)
for it in range(2):
    t0 = time.time()
    h = al.alignSequences([contig])
    print("align %.3f s hits %d stage %s" % (time.time() - t0, h["n_hits"], {k: round(v, 1) for k, v in al.stage_ms().items()}), flush=True)
