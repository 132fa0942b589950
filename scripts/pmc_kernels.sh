#!/bin/bash
# Issue counters of named kernels with one worker and nothing overlapping (one 8.3 M-read chunk of a config): VALU / SALU / LDS / VMEM
# instructions, wave cycles and wait cycles per kernel, summed over the launch.  One rocprofv3 --pmc pass with --kernel-trace only.
# Usage: scripts/pmc_kernels.sh <config> <kernel name pattern> [<pattern> ...]     e.g.  scripts/pmc_kernels.sh C3 k_cig k_ext_first
exec < /dev/null
ulimit -c 0
CFG=${1:-C3}; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_kernels; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SLX_KNOBS=workers=1${KNOBS:+,$KNOBS} SLX_BENCH_READS_CACHE=/tmp/slx_reads_cache
BENCH="python3 $R/bench.py --config $CFG --reads 8333333 --no-cpu-baseline --no-extras --verify 0 --steps 1 --warmup 0"
timeout 300 $BENCH > $OUT/fill.log 2>&1                 # un-profiled: generates and caches the read set (forked generators)
if [ $? -ne 0 ] || ! ls ${SLX_BENCH_READS_CACHE}.$CFG.* > /dev/null 2>&1; then echo "pmc_kernels.sh: the un-profiled fill run failed"; tail -5 $OUT/fill.log; exit 1; fi
# two passes: eight SQ counters do not always fit one
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  if [ ! -s "$f" ]; then echo "pmc_kernels.sh: pass $i left no counter_collection.csv"; tail -5 $OUT/pmc$i.log; exit 1; fi
done
timeout 60 python3 - $OUT "$@" <<'PY'
import collections, csv, glob, sys
pats = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/p[0-9]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        if any(p in k for p in pats):
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    print(k, {c: "%.3g" % x for c, x in sorted(v.items())})
PY
rm -rf $OUT/p[0-9]
