#!/bin/bash
# after sizing the arena for k_cig_lanes' interleaved blocks up front: parity subset, the first-call retry check, one C3 line
exec < /dev/null
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06x; mkdir -p $OUT; cd $R
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_fixture or edge_cases or knobs_do_not or stage_by_stage or C1_plumbing or light_heavy or C3_chr20 or ecoli_block" > $OUT/pytest_subset.txt 2>&1; tail -2 $OUT/pytest_subset.txt
timeout 150 python scripts/r06_retry_check.py > $OUT/retry_check.txt 2>&1; cat $OUT/retry_check.txt | tail -8
timeout 100 python bench.py --no-extras --no-cpu-baseline --steps 4 --warmup 1 > $OUT/c3.json 2> $OUT/c3.err; python -c "
import json; d=json.loads(open('$OUT/c3.json').read().strip().splitlines()[-1]); print('C3', round(d['value']/1e6,2), d['cigar_bit_match_rate'])"
