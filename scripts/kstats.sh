#!/bin/bash
# Per-kernel time of one bench command under rocprofv3 --kernel-trace --stats, top N rows on stdout.
# Usage: scripts/kstats.sh <N> <bench args...>      e.g.  scripts/kstats.sh 16 --config C5 --steps 2 --warmup 1
N=${1:-16}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o b -- python3 $R/bench.py --no-cpu-baseline --verify 0 "$@" > /tmp/kst.log 2>&1
grep '^{"metric"' /tmp/kst.log | python3 -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); print('value', d['value'], 'ms_per_step', d['ms_per_step'], d.get('step_split_ms',''), d.get('realign_extension_rounds',''))"
python3 - $N <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/kst/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[1])]:
    print("%-64s calls %4s total %9.1f ms avg %9.3f ms" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
