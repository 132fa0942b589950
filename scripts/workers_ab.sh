#!/bin/bash
# whole-pipeline throughput vs number of workers / hardware queues:  scripts/workers_ab.sh "name|ENV=.. KN=.." ...
mkdir -p gpurun_out
for spec in "$@"; do
    name=${spec%%|*}; rest=${spec#*|}
    for c in ${CFGS:-C2 C3}; do
        env $rest timeout -s KILL 400 python bench.py --config $c --no-cpu-baseline --no-extras --steps 2 --warmup 1 --verify 2000 > gpurun_out/wk_${name}_$c.json 2> gpurun_out/wk_${name}_$c.err
        python - "$name" "$c" gpurun_out/wk_${name}_$c.json <<'PY'
import json,sys
ok=False
for l in open(sys.argv[3]):
    if l.startswith('{"metric"'):
        d=json.loads(l); s=d["stage_ms_per_step"]; ok=True
        print("%-14s %-3s value %6.2f M/s  ms/step %7.1f  stream-ms: seed %7.1f chain %7.1f extend %6.1f finalize %6.1f  match %s" % (sys.argv[1], sys.argv[2], d["value"]/1e6, d["ms_per_step"], s["seed"], s["chain"], s["extend"], s["finalize"], d["cigar_bit_match_rate"]))
if not ok: print(sys.argv[1], sys.argv[2], "FAILED")
PY
    done
done
