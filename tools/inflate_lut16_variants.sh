#!/bin/bash
# GPU box: the workgroup inflate with 16-bit table entries (INQ_WG_LUT16=1: 17.9 KB of LDS per BGZF block instead of 20.4) at four waves
# per SIMD (eight blocks per CU as shipped: what the arithmetic on the match path costs) and at five (96 VGPRs: NINE blocks per CU)
# against the shipped form - the one way to a ninth block per CU that leaves the root array, the staged bits and the table's index
# width alone (profiles/r05_results/inflate_blocks_in_flight_per_cu.txt).
# usage: tools/inflate_lut16_variants.sh [blocks] [level] [test]   -> gpurun_out/inflate_lut16/result.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/inflate_lut16
mkdir -p $OUT
cd $ROOT
N=${1:-40000}; LV=${2:-6}
SRC="kernels.hip deep_select.hip capi.hip bgzf_inflate.hip bgzf_inflate_wg.hip bam_scan.hip span.hip outlier.hip"
build() { (cd inquistr_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $2 -shared -o /tmp/libinq_$1.so $SRC 2>&1 | grep -E "error" ) ; }
build base "" &
build lut16_w4 "-DINQ_WG_LUT16=1" &
build lut16_w5 "-DINQ_WG_LUT16=1 -DINQ_WG_WAVES=5" &
wait
: > $OUT/result.txt
for k in cigar seq ont qual; do
  for v in base lut16_w4 lut16_w5 base lut16_w5; do
    echo -n "$v $k: " | tee -a $OUT/result.txt
    INQ_LIB=/tmp/libinq_$v.so ALGO=0 timeout -k 10 200 python3 tools/inflate_bench.py $N $LV $k 2>&1 | grep kernel | sort -t' ' -k10 -n | head -1 | sed 's/^blocks [0-9]* level [0-9]* [a-z]*: //' | tee -a $OUT/result.txt
  done
done
if [ -n "$3" ]; then  # the front-end tests on the nine-per-CU build (this copy of the repo is scratch)
  cp /tmp/libinq_lut16_w5.so inquistr_amd/lib/libinquistr_hip.so
  timeout -k 10 600 python3 -m pytest tests/test_gpu_front.py -m gpu -x -q 2>&1 | tail -3 | tee -a $OUT/result.txt
fi
