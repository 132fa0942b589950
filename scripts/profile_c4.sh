#!/bin/bash
# C4-scale (GRCh38-sized, u64 index, HBM-random rank reads) bench line + the seeding kernel's traffic counters.
OUT=$1; R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --config C4h --reads 12000000 --no-cpu-baseline --no-extras --verify 0 --steps 1 --warmup 1"
timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $BENCH > $OUT/c4_under_rocprof.log 2>&1
grep '^{"metric"' $OUT/c4_under_rocprof.log > $OUT/bench_under_rocprof.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/c4_kernel_stats.csv
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 900 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
done
python3 $R/scripts/pmc_summary.py $OUT C4h > $OUT/pmc_summary.json
rm -rf $OUT/stats $OUT/pmc[0-9]
ls -la $OUT
