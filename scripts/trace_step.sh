#!/bin/bash
# kernel timeline of one timed bench step: scripts/trace_step.sh <name> [ENV=..]...  ->  gpurun_out/trace_<name>.csv (name,start,end,queue,stream)
name=$1; shift
for kv in "$@"; do export "$kv"; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$name -o t -- python3 $R/bench.py --no-cpu-baseline --verify 0 --steps 1 --warmup 1 > $R/gpurun_out/trace_$name.log 2>&1
f=$(find $R/gpurun_out/trace_$name -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$R/gpurun_out/trace_$name.csv" <<'PY'
import csv, sys
out = open(sys.argv[2], "w")
for r in csv.DictReader(open(sys.argv[1])):
    out.write("%s,%s,%s,%s,%s\n" % (r["Kernel_Name"][:48].replace(",", ";"), r["Start_Timestamp"], r["End_Timestamp"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
PY
rm -rf $R/gpurun_out/trace_$name
grep -o '"value": [0-9.]*' $R/gpurun_out/trace_$name.log | head -1
