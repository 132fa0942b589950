#!/bin/bash
# seeding A/B: alone-run (one worker, one chunk) and whole-pipeline numbers per library variant (scripts/build_variant.sh) and knob set (KN=...)
mkdir -p gpurun_out
for spec in "$@"; do
    name=${spec%%|*}; envs=${spec#*|}
    [ "$envs" = "$spec" ] && envs=""
    for cfg in C2:3333333 C3:8333332; do
        c=${cfg%%:*}; n=${cfg#*:}
        env $envs SLX_KNOBS="workers=1,${KN}" timeout -s KILL 300 python bench.py --config $c --reads $n --no-cpu-baseline --no-extras --steps 2 --warmup 1 --verify 0 > gpurun_out/ab_${name}_$c.json 2> gpurun_out/ab_${name}_$c.err
        python - "$name" "$c alone" gpurun_out/ab_${name}_$c.json <<'PY'
import json,sys
for l in open(sys.argv[3]):
    if l.startswith('{"metric"'):
        d=json.loads(l); s=d["stage_ms_per_step"]
        print("%-10s %-9s value %6.2f M/s  seed %7.1f chain %7.1f extend %6.1f finalize %6.1f total %7.1f" % (sys.argv[1], sys.argv[2], d["value"]/1e6, s["seed"], s["chain"], s["extend"], s["finalize"], s["total"]))
PY
    done
    for c in C2 C3; do
        env $envs SLX_KNOBS="${KN}" timeout -s KILL 300 python bench.py --config $c --no-cpu-baseline --no-extras --steps 2 --warmup 1 --verify 2000 > gpurun_out/ab_${name}_${c}_full.json 2> gpurun_out/ab_${name}_${c}_full.err
        python - "$name" "$c full" gpurun_out/ab_${name}_${c}_full.json <<'PY'
import json,sys
for l in open(sys.argv[3]):
    if l.startswith('{"metric"'):
        d=json.loads(l); s=d["stage_ms_per_step"]
        print("%-10s %-9s value %6.2f M/s  seed %7.1f chain %7.1f extend %6.1f finalize %6.1f total %7.1f match %s" % (sys.argv[1], sys.argv[2], d["value"]/1e6, s["seed"], s["chain"], s["extend"], s["finalize"], s["total"], d["cigar_bit_match_rate"]))
PY
    done
done
