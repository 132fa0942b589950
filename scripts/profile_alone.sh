#!/bin/bash
# Per-kernel time with nothing overlapping: one worker, one chunk of reads, one timed step under rocprofv3 --kernel-trace --stats.
# Usage: scripts/profile_alone.sh <out_prefix> <config> <reads>   ->  <out_prefix>_kernel_stats.csv
OUT=$1; CFG=${2:-C3}; N=${3:-8333333}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export SLX_KNOBS=workers=1${KNOBS:+,$KNOBS}
# the read set is generated in an un-profiled run (forked generator workers) and only loaded under the profiler
export SLX_BENCH_READS_CACHE=/tmp/slx_reads_cache
timeout -s KILL 600 python3 $R/bench.py --config $CFG --reads $N --no-cpu-baseline --no-extras --verify 0 --steps 1 --warmup 0 > ${OUT}_unprofiled.log 2>&1
if [ $? -ne 0 ] || ! ls ${SLX_BENCH_READS_CACHE}.$CFG.* > /dev/null 2>&1; then echo "profile_alone.sh: the un-profiled fill run failed"; tail -5 ${OUT}_unprofiled.log; exit 1; fi
timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d ${OUT}_dir -o a -- python3 $R/bench.py --config $CFG --reads $N --no-cpu-baseline --no-extras --verify 0 --steps 1 --warmup 1 > ${OUT}.log 2>&1
cp $(find ${OUT}_dir -name "*kernel_stats.csv" | head -1) ${OUT}_kernel_stats.csv
rm -rf ${OUT}_dir
python3 - ${OUT}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-70s calls %4s  total %9.2f ms  avg %9.3f ms" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
