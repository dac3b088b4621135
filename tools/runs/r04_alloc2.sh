#!/bin/bash
# is hipMalloc slow (50 ms per GB) when the memory was released by a process a moment ago?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_alloc2
mkdir -p $OUT
cd $ROOT
for i in 1 2 3 4; do echo "== run $i (back to back)"; timeout -k 10 60 inquistr_amd/lib/alloc_probe 2>&1 | grep -E "hipMalloc (1024|2048|4096) MB|again"; done | tee $OUT/alloc_back_to_back.txt
sleep 5
echo "== after 5 s of rest"; timeout -k 10 60 inquistr_amd/lib/alloc_probe 2>&1 | grep -E "hipMalloc (1024|2048|4096) MB" | tee -a $OUT/alloc_back_to_back.txt
