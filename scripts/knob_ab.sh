#!/bin/bash
# A/B of SLX_KNOBS settings on one bench configuration (no CPU baseline, no extras, no verification): one JSON line per setting
# in <out_dir>/<name>.json and a summary on stdout.  The read set is generated once (SLX_BENCH_READS_CACHE).
# Usage: scripts/knob_ab.sh <out_dir> <config> "<name>=<knobs>" ...     e.g.  base= cu4=seed_free_cus=4 cu4p=seed_free_cus=4,stream_prio=1
OUT=$1; CFG=$2; shift; shift
mkdir -p $OUT
export SLX_BENCH_READS_CACHE=/tmp/slx_reads_cache
for spec in "$@"; do
  name=${spec%%=*}; knobs=${spec#*=}
  # a "lib=<variant>" token selects seqlib_amd/variants/libseqlib_amd_<variant>.so (scripts/build_variant.sh) instead of the default library
  lib=""; rest=""
  for kv in ${knobs//,/ }; do if [ "${kv%%=*}" = lib ]; then lib=${kv#lib=}; else rest="${rest:+$rest,}$kv"; fi; done
  if [ -n "$lib" ]; then export SLX_LIB=$GRAFT_REPO_ROOT/seqlib_amd/variants/libseqlib_amd_$lib.so; else unset SLX_LIB; fi
  SLX_KNOBS="$rest" timeout 400 python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --no-cpu-baseline --no-extras --verify 0 --steps ${STEPS:-4} --warmup 1 > $OUT/$name.json 2> $OUT/$name.err
  python3 - "$name" "$knobs" $OUT/$name.json <<'PY'
import json, sys
name, knobs, fn = sys.argv[1:4]
try:
    d = json.loads(open(fn).read().strip().splitlines()[-1])
    st = d.get("stage_ms_per_step", {})
    print("%-10s %-40s %8.2f M reads/s  %7.1f ms/step  seed %.0f chain %.0f extend %.0f finalize %.0f (summed stream ms)" % (
        name, knobs, d["value"] / 1e6, d["ms_per_step"], st.get("seed", 0), st.get("chain", 0), st.get("extend", 0), st.get("finalize", 0)))
except Exception as e:
    print("%-10s %-40s FAILED (%s)" % (name, knobs, e))
PY
done
