#!/bin/bash
# GPU box: the workgroup inflate with 2 KB-root stretches (INQ_WG_CAP=2048: 16.3 KB of LDS per block) at four and at five waves per
# SIMD (eight and ten blocks per CU) against the shipped form (4 KB roots, eight per CU) - what the smaller stretches cost and what
# two more blocks in flight buy (profiles/r05_results/inflate_blocks_in_flight_per_cu.txt).
# usage: tools/inflate_cap_variants.sh [blocks] [level]   -> gpurun_out/inflate_cap/result.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/inflate_cap
mkdir -p $OUT
cd $ROOT
N=${1:-40000}; LV=${2:-6}
SRC="kernels.hip deep_select.hip capi.hip bgzf_inflate.hip bgzf_inflate_wg.hip bam_scan.hip span.hip outlier.hip"
build() { (cd inquistr_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $2 -shared -o /tmp/libinq_$1.so $SRC 2>&1 | grep -E "error|occupancy|LDS" ) ; }
build base "" &
build cap2048_w4 "-DINQ_WG_CAP=2048" &
build cap2048_w5 "-DINQ_WG_CAP=2048 -DINQ_WG_WAVES=5" &
wait
: > $OUT/result.txt
for k in cigar seq ont qual; do
  for v in base cap2048_w4 cap2048_w5; do
    echo -n "$v $k: " | tee -a $OUT/result.txt
    INQ_LIB=/tmp/libinq_$v.so ALGO=0 timeout -k 10 200 python3 tools/inflate_bench.py $N $LV $k 2>&1 | grep kernel | sort -t' ' -k10 -n | head -1 | sed 's/^blocks [0-9]* level [0-9]* [a-z]*: //' | tee -a $OUT/result.txt
  done
done
