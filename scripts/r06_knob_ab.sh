#!/bin/bash
# A/B of one 0|1 knob (KNOB=..., default cig_lane_il) on C3: a parity subset first, then alternating bench runs.
KNOB=${KNOB:-cig_lane_il}
exec < /dev/null
ulimit -c 0
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${TAG:-r06p}; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${TESTS:-full_fixture or edge_cases or knobs_do_not or stage_by_stage or C1_plumbing or long_reads_seed or light_heavy or C3_chr20 or ecoli_block or option_fuzz_vs}" > $OUT/pytest_subset.txt 2>&1; tail -3 $OUT/pytest_subset.txt
for rep in 1 2; do
for v in 1 0; do
SLX_KNOBS=${KNOB}=$v timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 4 --warmup 1 > $OUT/ab_${v}_$rep.json 2> $OUT/ab_${v}_$rep.err
python - <<P
import json
try:
    d=json.loads(open("$OUT/ab_${v}_$rep.json").read().strip().splitlines()[-1])
    print("${KNOB}=$v", round(d["value"]/1e6,2), d["cigar_bit_match_rate"], {k:round(x) for k,x in d["probe_ms_per_step"].items()}, round(d["stage_ms_per_step"]["finalize"]))
except Exception as e:
    print("${KNOB}=$v failed", e)
P
done; done
