#!/bin/bash
# PMC passes over the bench (one worker so that kernels do not overlap).  Each --pmc set is its own run, with
# --kernel-trace only (gpurun refuses pmc + sys-trace).  Usage: scripts/pmc.sh <out_dir> [reads]
OUT=$1; READS=${2:-2000000}
cd /tmp && export TMPDIR=/tmp
export SLX_KNOBS=workers=1
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pass$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --reads $READS --steps 1 --warmup 1 --no-cpu-baseline --verify 0 > $OUT/pass$i.log 2>&1
done
ls -R $OUT | head -40
