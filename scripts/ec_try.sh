#!/bin/bash
# one GPU call for a change to the correction kernels: parity tests (both kernels), kernel times of a C5 run, nothing else
# usage: scripts/ec_try.sh <out-dir under gpurun_out>
D=gpurun_out/$1; mkdir -p $D
timeout 600 python -m pytest tests/test_gpu_fml.py tests/test_cpp_fml.py -m gpu -q -x > $D/pytest.txt 2>&1; tail -3 $D/pytest.txt

timeout 700 bash scripts/ec_ab.sh $1 > $D/k.txt; grep "k_fml_occ\|k_fml_ec" $D/k.txt; grep -i "fml times" $D/bench.err | tail -1
