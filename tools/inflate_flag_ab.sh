#!/bin/bash
# GPU box: the workgroup inflate built with a set of -D flags against the shipped build, the four kinds of tools/inflate_bench.py,
# base and variant twice each in turn; optionally the front-end tests on the variant.
# usage: tools/inflate_flag_ab.sh "<flags>" [blocks] [level] [test]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
FLAGS=$1; N=${2:-40000}; LV=${3:-6}
SRC="kernels.hip deep_select.hip capi.hip bgzf_inflate.hip bgzf_inflate_wg.hip bam_scan.hip span.hip outlier.hip"
build() { (cd inquistr_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $2 -shared -o /tmp/libinq_$1.so $SRC 2>&1 | grep -E "error" ) ; }
build base "" &
build variant "$FLAGS" &
wait
echo "variant = $FLAGS"
for k in cigar seq ont qual; do
  for v in base variant base variant; do
    echo -n "$v $k: "
    INQ_LIB=/tmp/libinq_$v.so ALGO=0 timeout -k 10 200 python3 tools/inflate_bench.py $N $LV $k 2>&1 | grep kernel | sort -t' ' -k10 -n | head -1 | sed 's/^blocks [0-9]* level [0-9]* [a-z]*: //'
  done
done
if [ -n "$4" ]; then
  cp /tmp/libinq_variant.so inquistr_amd/lib/libinquistr_hip.so
  timeout -k 10 600 python3 -m pytest tests/test_gpu_front.py -m gpu -x -q 2>&1 | tail -3
fi
