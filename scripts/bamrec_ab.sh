#!/bin/bash
# value_bamrecords experiments: tools/bamrec_bench (the C++ class end to end) on N reads of the C3 read set, over host-thread counts,
# chunk sizes and glibc malloc settings.  Usage: scripts/bamrec_ab.sh <out_dir> [n_reads]
OUT=$1; N=${2:-10000000}; R=$GRAFT_REPO_ROOT
mkdir -p $OUT /tmp/bamrec
cd $R
python3 - $N <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench as b
from seqlib_amd import synth
import seqlib_amd
n = int(sys.argv[1])
cfg = synth.CONFIGS["C3"]
refs = synth.make_reference(cfg)
reads = b.gen_reads(cfg, refs, n, 0)
np.ascontiguousarray(reads).tofile("/tmp/bamrec/reads.bin")
idx = seqlib_amd.BWAIndex()
idx.ConstructIndex([(nm, synth.genome_ascii_bytes(g)) for nm, g in refs])
idx.WriteIndex("/tmp/bamrec/c3")
print("prepared", n, "reads")
PY
run() {  # name, env...
  name=$1; shift
  env "$@" SEQLIB_AMD_TRACE=1 timeout ${BAMREC_TIMEOUT:-300} $R/seqlib_amd/bamrec_bench /tmp/bamrec/c3 /tmp/bamrec/reads.bin 150 $N > $OUT/$name.json 2> $OUT/$name.err
  echo "$name: $(cut -c1-120 $OUT/$name.json)"
}
if [ -n "$RUNS" ]; then
  # RUNS="name:VAR=val,VAR=val name2:..."  (custom settings instead of the default sweep)
  for spec in $RUNS; do name=${spec%%:*}; vars=${spec#*:}; run $name ${vars//,/ }; done
  exit 0
fi
run t16         SEQLIB_AMD_THREADS=16
run t16_pad     SEQLIB_AMD_THREADS=16 MALLOC_TOP_PAD_=268435456 MALLOC_TRIM_THRESHOLD_=4294967296
run t24_pad     SEQLIB_AMD_THREADS=24 MALLOC_TOP_PAD_=268435456 MALLOC_TRIM_THRESHOLD_=4294967296
run t32_pad     SEQLIB_AMD_THREADS=32 MALLOC_TOP_PAD_=268435456 MALLOC_TRIM_THRESHOLD_=4294967296
run t16_pad_c1  SEQLIB_AMD_THREADS=16 SEQLIB_AMD_CHUNK=1000000 MALLOC_TOP_PAD_=268435456 MALLOC_TRIM_THRESHOLD_=4294967296
run t16_pad_c4  SEQLIB_AMD_THREADS=16 SEQLIB_AMD_CHUNK=4000000 MALLOC_TOP_PAD_=268435456 MALLOC_TRIM_THRESHOLD_=4294967296
run t16_c500k   SEQLIB_AMD_THREADS=16 SEQLIB_AMD_CHUNK=500000
run dflt
tail -12 $OUT/t16_pad.err
