#!/bin/bash
# two quick PMC passes (issue counters) over a one-worker bench step; prints per-kernel means.  usage: scripts/pmc_quick.sh [reads]
READS=${1:-10000000}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmcq
cd /tmp && export TMPDIR=/tmp
export SLX_KNOBS=${SLX_KNOBS:-workers=1}
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pass$i -o p -- python3 $R/bench.py --reads $READS --steps 1 --warmup 1 --no-cpu-baseline --verify 0 > $OUT.pass$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    if "rocprim" in k or "rocclr" in k: continue
    # the largest launch of each kernel (the timed step equals the warm-up step)
    print(k, {c: "%.3g" % max(v) for c, v in sorted(d.items())})
PY
rm -rf $OUT
