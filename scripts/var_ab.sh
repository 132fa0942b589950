#!/bin/bash
# alone-run kernel times of build variants: PAT='k_cig' scripts/var_ab.sh name...   (seqlib_amd/variants/libseqlib_amd_<name>.so; "main" = the default library)
for v in "$@"; do
  echo "== $v"
  if [ "$v" = main ]; then unset SLX_LIB; else export SLX_LIB=$GRAFT_REPO_ROOT/seqlib_amd/variants/libseqlib_amd_$v.so; fi
  bash scripts/profile_alone.sh $GRAFT_REPO_ROOT/gpurun_out/var_$v ${CFG:-C3} ${NREADS:-8333333} 2>&1 | grep "${PAT:-k_seed}"
done
