#!/bin/bash
# Final evidence of round 6 after the traffic fixes (k_cig_fast_coop, k_cig_lanes interleaved, k_first_lanes, the atomics): the whole -m gpu suite, smoke, the default
# bench line, then kernel stats + PMC passes of the C3 command.  Every part under its own timeout.  Usage: scripts/r06_final.sh <tag>
exec < /dev/null
ulimit -c 0
TAG=${1:-r06z}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; cd $R
timeout 1000 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 200 $OUT/bench_default.json
case " $SKIP " in *" prof "*) ;; *) timeout 420 bash scripts/profile_round.sh $OUT/c3 C3 > $OUT/profile_c3.log 2>&1; tail -3 $OUT/profile_c3.log ;; esac
rm -f /tmp/slx_reads_cache.*
ls $OUT
