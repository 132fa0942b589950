#!/bin/bash
# usage: scripts/run_variants.sh "name|ENV=.. ENV2=.." ...   -- one bench.py run per argument, one summary line each
mkdir -p gpurun_out
for spec in "$@"; do
    name=${spec%%|*}; envs=${spec#*|}
    [ "$envs" = "$spec" ] && envs=""
    env $envs timeout -s KILL 300 python bench.py --no-cpu-baseline --steps ${STEPS:-3} --warmup 1 --verify ${VERIFY:-2000} > gpurun_out/v_$name.json 2> gpurun_out/v_$name.err
    python scripts/bench_line.py $name < gpurun_out/v_$name.json
    grep -o '"cigar_bit_match_rate": [0-9.]*' gpurun_out/v_$name.json
done
