#!/bin/bash
# Round profile: (1) rocprofv3 --kernel-trace --stats of the default bench command, (2) separate --pmc passes
# (kernel-trace only) for HBM traffic and issue counters of the same command.  Usage: scripts/profile_round.sh <out_dir>
OUT=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench_under_rocprof.log 2>&1
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --verify 0 --steps 2 > $OUT/pmc$i.log 2>&1
done
ls $OUT $OUT/stats | head -30
