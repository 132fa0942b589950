#!/bin/bash
# every kernel and copy of ONE single-read call, in order (rocprofv3 kernel + memory-copy trace of scripts/percall_probe.py; the last call's rows)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/prof -o pc -- python3 $GRAFT_REPO_ROOT/scripts/percall_probe.py > $OUT/probe.txt 2>&1
tail -1 $OUT/probe.txt
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
k = list(csv.DictReader(open(glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True)[0])))
m = list(csv.DictReader(open(glob.glob(out + "/prof/**/*memory_copy_trace.csv", recursive=True)[0])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in k] + [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))) for r in m]
ev.sort()
# the last call: events after the last gap of > 200 us ... take the last 80 events and cut at the largest early gap
tail = ev[-120:]
cut = 0
for i in range(1, len(tail)):
    if tail[i][0] - tail[i - 1][1] > 150000: cut = i
tail = tail[cut:]
t0 = tail[0][0]
with open(out + "/one_call.txt", "w") as f:
    for a, b, n in tail:
        f.write("%8.1f us +%6.1f  %s\n" % ((a - t0) / 1e3, (b - a) / 1e3, n))
    f.write("events %d, span %.1f us, busy %.1f us\n" % (len(tail), (tail[-1][1] - t0) / 1e3, sum(b - a for a, b, _ in tail) / 1e3))
print(open(out + "/one_call.txt").read())
PY
