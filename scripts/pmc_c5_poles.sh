#!/bin/bash
# Issue counters of the contig-length kernels of the C5 realignment (8 windows, steps not pipelined): instructions and wait cycles per launch.
# Usage: scripts/pmc_c5_poles.sh
exec < /dev/null
ulimit -c 0
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_c5_poles; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --config C5 --windows 8 --no-pipeline --no-cpu-baseline --verify 0 --steps 1 --warmup 1"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout -s KILL 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
done
python3 - $OUT <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p[0-9]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
        if any(p in k for p in ("k_cig_band", "k_ext_block", "k_regs_wave_long", "k_seed12m", "k_extend_reg", "k_chain_coop")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, {c: "%.3g (x%d)" % (max(x), len(x)) for c, x in sorted(v.items())})
PY
rm -rf $OUT/p[0-9]
