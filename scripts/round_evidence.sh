#!/bin/bash
# Everything the round's documentation quotes, in one GPU call: parity suite, smoke, the default bench line (C3, with CPU baseline and
# extras), the C2 and C5 lines, the rocprofv3 kernel stats + PMC passes of the C3 and C5 commands, and the nothing-overlapping per-kernel
# profile.  SKIP="tests c5" leaves parts out.
# Usage: scripts/round_evidence.sh <tag>      (writes gpurun_out/<tag>/...)
exec < /dev/null
ulimit -c 0          # a GPU memory fault makes ROCr dump the whole HBM image: 288 GB onto a 79 GB disk
TAG=${1:-r04}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
case " $SKIP " in *" tests "*) ;; *)
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt ;; esac
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
timeout 600 python bench.py --config C2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -c 300 $OUT/bench_c2.json
case " $SKIP " in *" c5 "*) ;; *)
timeout 900 python bench.py --config C5 --steps 6 --warmup 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -c 400 $OUT/bench_c5.json
bash scripts/profile_c5.sh $OUT/c5 > $OUT/profile_c5.log 2>&1 ;; esac
bash scripts/profile_round.sh $OUT/c3 C3 > $OUT/profile_c3.log 2>&1
bash scripts/profile_alone.sh $OUT/alone_C3 C3 8333333 > $OUT/alone_C3.txt 2>&1
rm -f /tmp/slx_reads_cache.*          # 7.5 GB per C3 read set on a 79 GB disk
ls $OUT
