#!/bin/bash
# kernel trace of the bench (three workers overlapping) -> coarse timeline.  Usage: scripts/trace_round.sh <out_prefix> <config>
OUT=$1; CFG=${2:-C3}; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 900 rocprofv3 --kernel-trace --output-format csv -d ${OUT}_dir -o t -- python3 $R/bench.py --config $CFG --no-cpu-baseline --no-extras --verify 0 --steps 1 --warmup 1 > ${OUT}.log 2>&1
python3 $R/scripts/trace_timeline.py $(find ${OUT}_dir -name "*kernel_trace.csv" | head -1) ${BIN:-10} > ${OUT}_timeline.txt
rm -rf ${OUT}_dir
cat ${OUT}_timeline.txt
