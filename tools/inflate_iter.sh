#!/bin/bash
# dev loop for the inflate kernel on the GPU box: zlib equivalence tests first, then the two timing cases
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_front.py -x -q -k "inflate" > gpurun_out/iter_tests.txt 2>&1 || { tail -30 gpurun_out/iter_tests.txt; exit 1; }
tail -2 gpurun_out/iter_tests.txt
ALGO=0 timeout -k 10 200 python tools/inflate_bench.py ${1:-20000} 1 cigar 2>&1 | grep -v amdgpu.ids | tail -1
ALGO=0 timeout -k 10 200 python tools/inflate_bench.py ${1:-20000} 1 qual 2>&1 | grep -v amdgpu.ids | tail -1
ALGO=0 timeout -k 10 200 python tools/inflate_bench.py ${1:-20000} 1 ont 2>&1 | grep -v amdgpu.ids | tail -1
ALGO=0 timeout -k 10 200 python tools/inflate_bench.py ${1:-20000} 6 cigar 2>&1 | grep -v amdgpu.ids | tail -1
