#!/bin/bash
# Tuning builds: scripts/build_variant.sh NAME [-DFLAG=..]...  ->  seqlib_amd/variants/libseqlib_amd_NAME.so
# (select with SLX_LIB=<path> python bench.py; the default library is untouched).  Run seqlib_amd/build.py first.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p seqlib_amd/variants seqlib_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Iinclude -Iseqlib_amd/csrc "$@" \
    -c seqlib_amd/csrc/slx_align.hip -o seqlib_amd/build/slx_align_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC seqlib_amd/build/slx_index.cpp.o seqlib_amd/build/slx_index_gpu.hip.o seqlib_amd/build/slx_index_gpu64.hip.o \
    seqlib_amd/build/slx_align_$name.o seqlib_amd/build/slx_align_wide.hip.o seqlib_amd/build/slx_fml.hip.o seqlib_amd/build/slx_fml_asm.hip.o -o seqlib_amd/variants/libseqlib_amd_$name.so
echo seqlib_amd/variants/libseqlib_amd_$name.so
