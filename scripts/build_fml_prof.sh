#!/bin/bash
# Experiment build of the BFC correction kernels with cycle counters (FML_EC_PROF: dev_fml.h) -> seqlib_amd/variants/libseqlib_amd_fmlprof.so
# (select with SLX_LIB=<path>; prints "[fml ec prof] ..." on stderr after every correction launch).  Run seqlib_amd/build.py first.
set -e
cd "$(dirname "$0")/.."
mkdir -p seqlib_amd/variants seqlib_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Iinclude -Iseqlib_amd/csrc -DFML_EC_PROF \
    -c seqlib_amd/csrc/slx_fml.hip -o seqlib_amd/build/slx_fml_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC seqlib_amd/build/slx_index.cpp.o seqlib_amd/build/slx_index_gpu.hip.o seqlib_amd/build/slx_index_gpu64.hip.o \
    seqlib_amd/build/slx_align.hip.o seqlib_amd/build/slx_align_wide.hip.o seqlib_amd/build/slx_fml_prof.o seqlib_amd/build/slx_fml_asm.hip.o -o seqlib_amd/variants/libseqlib_amd_fmlprof.so
echo seqlib_amd/variants/libseqlib_amd_fmlprof.so
