#!/bin/bash
# Round profile: (1) rocprofv3 --kernel-trace --stats of the bench command, (2) separate --pmc passes (kernel-trace only) for HBM
# traffic and issue counters of the same command, summarised by scripts/pmc_summary.py.
# Usage: scripts/profile_round.sh <out_dir> <config> [bench args]  ->  <out_dir>/{bench_kernel_stats.csv, bench_under_rocprof.json, pmc_summary.json}
OUT=$1; CFG=${2:-C3}; shift; shift
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --config $CFG --no-cpu-baseline --no-extras --verify 0 --steps 2 --warmup 1 $*"
# the read set is generated once, un-profiled (forked generator workers), and the profiled runs only load it: under rocprofv3 the
# tool has initialised the GPU before main(), and a process in that state must not fork
export SLX_BENCH_READS_CACHE=/tmp/slx_reads_cache
timeout -s KILL 600 $BENCH > $OUT/bench_unprofiled.log 2>&1
if [ $? -ne 0 ] || ! ls ${SLX_BENCH_READS_CACHE}.$CFG.* > /dev/null 2>&1; then echo "profile_round.sh: the un-profiled fill run failed"; tail -5 $OUT/bench_unprofiled.log; exit 1; fi
timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH > $OUT/bench_under_rocprof.log 2>&1
grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
done
python3 $R/scripts/pmc_summary.py $OUT $CFG > $OUT/pmc_summary.json
rm -rf $OUT/stats $OUT/pmc[0-9] $OUT/pmc[0-9].log
ls -la $OUT
