#!/bin/bash
# kernel times of the assembly half alone (no overlap of steps): rocprofv3 kernel stats of a short C5 run
# usage: scripts/ec_ab.sh <out-dir under gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SLX_FML_TIMES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o ec -- python3 $GRAFT_REPO_ROOT/bench.py --config C5 --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --verify 0 > $OUT/bench.json 2> $OUT/bench.err
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
fs = glob.glob(out + "/prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
with open(out + "/kernels.txt", "w") as f:
    for r in rows[:28]:
        f.write("%-60s %5s calls  %9.2f ms avg  %5s %%\n" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
print(open(out + "/kernels.txt").read())
PY
tail -c 1500 $OUT/bench.json
