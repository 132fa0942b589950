#!/bin/bash
# C4 (GRCh38-sized synthetic reference, u64 index, nothing cache-resident: the one config where "achieved HBM GB/s on the seeding kernel" is literal): kernel stats,
# the seeding kernels' traffic counters (separate --pmc passes), then a CLEAN bench line of the same workload whose roofline.traffic / physical_gbs / random_access
# are filled from that summary.
# Usage: scripts/profile_c4.sh <abs out_dir> [config (C4 | C4h)] [reads]   ->   <out_dir>/{c4_kernel_stats.csv, pmc_summary.json, bench_clean.json}
OUT=$1; CFG=${2:-C4}; READS=${3:-25000000}; R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SLX_BENCH_READS_CACHE=/tmp/slx_reads_cache
BENCH="python3 $R/bench.py --config $CFG --reads $READS --no-cpu-baseline --no-extras --verify 0 --steps 1 --warmup 1"
timeout -s KILL 900 $BENCH > $OUT/fill.log 2>&1          # un-profiled: generates and caches the read set (forked generators)
if [ $? -ne 0 ] || ! ls ${SLX_BENCH_READS_CACHE}.$CFG.* > /dev/null 2>&1; then echo "profile_c4.sh: the un-profiled fill run failed"; tail -5 $OUT/fill.log; exit 1; fi
timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH > $OUT/c4_under_rocprof.log 2>&1
grep '^{"metric"' $OUT/c4_under_rocprof.log > $OUT/bench_under_rocprof.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/c4_kernel_stats.csv
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 900 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
done
python3 $R/scripts/pmc_summary.py $OUT $CFG > $OUT/pmc_summary.json
rm -rf $OUT/stats $OUT/pmc[0-9] $OUT/pmc[0-9].log
# the clean line: same workload, three timed steps, CPU baseline (the checker loads the index the GPU wrote) for the algorithmic bytes, traffic from the summary above
SLX_PMC_SUMMARY=$OUT/pmc_summary.json timeout -s KILL 2400 python3 $R/bench.py --config $CFG --reads $READS --no-extras --steps 3 --warmup 1 > $OUT/bench_clean.json 2> $OUT/bench_clean.err
tail -c 300 $OUT/bench_clean.json
rm -f ${SLX_BENCH_READS_CACHE}.$CFG.*
ls -la $OUT
