#!/bin/bash
# Round-6 evidence in one GPU call (every part under its own timeout): the whole -m gpu suite, smoke, the default bench line (C3 + the bounded C5 leg + the C++ tools),
# the C2 and C5 lines, the default line again at the host budget of one rank of eight (SEQLIB_AMD_LOCAL_RANKS=8: 2 CPUs), rocprofv3 kernel stats + PMC passes of the
# C3 and C5 commands, and the nothing-overlapping per-kernel profile.  Usage: scripts/r06_evidence.sh <tag>   (writes gpurun_out/<tag>/...)   SKIP="tests prof" leaves parts out.
exec < /dev/null
ulimit -c 0
TAG=${1:-r06}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
case " $SKIP " in *" tests "*) ;; *)
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt ;; esac
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.json
timeout 400 python bench.py --config C2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -c 200 $OUT/bench_c2.json
timeout 600 python bench.py --config C5 --steps 12 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -c 200 $OUT/bench_c5.json
SEQLIB_AMD_LOCAL_RANKS=8 timeout 900 python bench.py > $OUT/bench_host_budget_2cpu.json 2> $OUT/bench_host_budget_2cpu.err; tail -c 200 $OUT/bench_host_budget_2cpu.json
case " $SKIP " in *" prof "*) ;; *)
bash scripts/profile_round.sh $OUT/c3 C3 > $OUT/profile_c3.log 2>&1
bash scripts/profile_alone.sh $OUT/alone_C3 C3 16666666 > $OUT/alone_C3.txt 2>&1
case " $SKIP " in *" c5prof "*) ;; *) bash scripts/profile_c5.sh $OUT/c5 > $OUT/profile_c5.log 2>&1 ;; esac ;; esac
rm -f /tmp/slx_reads_cache.*
ls $OUT
