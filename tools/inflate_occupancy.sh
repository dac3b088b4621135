#!/bin/bash
# GPU box: how much of the workgroup inflate's time do MORE independent blocks per CU hide?  (VERDICT r4 item 6 asks for two blocks
# interleaved per workgroup - the dependent chain of one hidden behind the other's.)  The cheap form of that question: the same
# kernel at 4 / 6 / 8 workgroups (= blocks in flight) per CU, by padding its LDS; if halving the blocks in flight costs far less
# than half the rate, the SIMDs' issue slots are the bound and a second chain per lane has nothing to hide behind.
# usage: tools/inflate_occupancy.sh [blocks] [level]   -> gpurun_out/inflate_occupancy/result.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/inflate_occupancy
mkdir -p $OUT
cd $ROOT
N=${1:-40000}; LV=${2:-6}
SRC="kernels.hip deep_select.hip capi.hip bgzf_inflate.hip bgzf_inflate_wg.hip bam_scan.hip span.hip outlier.hip"
build() {  # name, extra flags
  (cd inquistr_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $2 -shared -o /tmp/libinq_$1.so $SRC) || exit 1
}
build wg8 "" &
build wg6 "-DINQ_WG_PAD=6400" &      # 19.7 + 6.25 KB = 26 KB: six per CU
build wg4 "-DINQ_WG_PAD=19200" &     # 38.9 KB: four per CU
build wg3 "-DINQ_WG_PAD=32000" &     # 51.4 KB: three per CU
wait
: > $OUT/result.txt
for k in cigar seq ont qual; do
  for v in wg8 wg6 wg4 wg3; do
    echo -n "$v $k: " | tee -a $OUT/result.txt
    INQ_LIB=/tmp/libinq_$v.so ALGO=0 timeout -k 10 200 python3 tools/inflate_bench.py $N $LV $k 2>&1 | grep kernel | sort -t' ' -k10 -n | head -1 | sed 's/^blocks [0-9]* level [0-9]* [a-z]*: //' | tee -a $OUT/result.txt
  done
done
