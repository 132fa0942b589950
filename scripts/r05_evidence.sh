#!/bin/bash
# Round-5 evidence in one GPU call: the whole -m gpu suite, smoke, the default bench line (C3 + the bounded C5 leg), the C5 line with one whole window
# through both checkers, the C5 kernel stats + PMC passes.  Usage: scripts/r05_evidence.sh <tag>   (writes gpurun_out/<tag>/...)
exec < /dev/null
ulimit -c 0
TAG=${1:-r05}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
case " $SKIP " in *" tests "*) ;; *)
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt ;; esac
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.json
timeout 1200 python bench.py --config C5 --steps 12 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -c 300 $OUT/bench_c5.json
case " $SKIP " in *" prof "*) ;; *)
bash scripts/profile_c5.sh $OUT/c5 > $OUT/profile_c5.log 2>&1 ;; esac
ls $OUT
