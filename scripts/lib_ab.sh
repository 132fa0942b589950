#!/bin/bash
# A/B of library variants (scripts/build_variant.sh) on the default bench: scripts/lib_ab.sh <config> <steps> base w5l5 ...   (base = the default library)
exec < /dev/null
ulimit -c 0
CFG=$1; STEPS=$2; shift; shift
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
export SLX_BENCH_READS_CACHE=/tmp/slx_reads_cache
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib=$R/seqlib_amd/variants/libseqlib_amd_$v.so
  SLX_LIB=$lib timeout -s KILL 600 python $R/bench.py --config $CFG --no-cpu-baseline --no-extras --steps $STEPS --warmup 1 --verify 2000 > $R/gpurun_out/lab_$v.json 2> $R/gpurun_out/lab_$v.err
  python - "$v" $R/gpurun_out/lab_$v.json <<'PY'
import json,sys
for l in open(sys.argv[2]):
    if l.startswith('{"metric"'):
        d=json.loads(l); s=d["stage_ms_per_step"]; p=d["probe_ms_per_step"]
        print("%-8s value %6.2f M/s  step %7.1f ms  probes seed %7.1f extend %7.1f cigar %6.1f  match %s" % (sys.argv[1], d["value"]/1e6, d["ms_per_step"], p.get("seed",0), p.get("extend",0), p.get("cigar",0), d["cigar_bit_match_rate"]))
PY
done
rm -f /tmp/slx_reads_cache.*
