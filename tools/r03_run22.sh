#!/bin/bash
# GPU box, round 3: the round's last kernel state - full GPU suite, default bench line, inflate per kind (options at their defaults),
# the front end's rocprof summary, a soak of damaged streams.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03z
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/gputest_full.log 2>&1; rc=$?; echo "pytest rc $rc" | tee -a $OUT/gputest_full.log; tail -3 $OUT/gputest_full.log
[ $rc = 0 ] || exit 1
for n in 20000 40000; do for lv in 1 6; do for k in cigar seq ont qual; do
  echo -n "$n blocks level $lv $k: " | tee -a $OUT/inflate_final_state.txt
  ALGO=0 timeout -k 10 300 python3 tools/inflate_bench.py $n $lv $k 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/comp .* kernel/kernel/' | tee -a $OUT/inflate_final_state.txt
done; done; done
timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"; tail -c 600 $OUT/bench_default.json
timeout -k 10 400 bash tools/profile_front.sh r03_front_final 50000 > $OUT/profile_front.log 2>&1; echo "profile_front rc $?"; tail -8 $OUT/profile_front.log
INQ_SOAK_SEED=91000 timeout -k 10 300 python3 tools/soak_inflate.py 60 3 > $OUT/soak_inflate.txt 2>&1; echo "soak_inflate rc $?" | tee -a $OUT/soak_inflate.txt; tail -3 $OUT/soak_inflate.txt
