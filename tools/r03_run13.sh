#!/bin/bash
# GPU box, round 3: final libraries - whole gpu suite, inflate per kind with the form chosen by the data, front profile, bench line.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03n
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/gputest.log; tail -4 $OUT/gputest.log
for lv in 1 6; do
  for k in cigar ont qual seq; do
    for pairs in 0 1 -1; do
      echo -n "level $lv $k pairs=$pairs: " | tee -a $OUT/inflate_forms.txt
      PAIRS=$pairs ALGO=0 timeout -k 10 200 python3 tools/inflate_bench.py 20000 $lv $k 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/comp .* kernel/kernel/' | tee -a $OUT/inflate_forms.txt
    done
  done
done
for k in cigar ont; do echo -n "40000 blocks level 6 $k: " | tee -a $OUT/inflate_forms.txt; ALGO=0 timeout -k 10 300 python3 tools/inflate_bench.py 40000 6 $k 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/comp .* kernel/kernel/' | tee -a $OUT/inflate_forms.txt; done
bash tools/profile_front.sh r03_front 50000 > gpurun_out/prof_r03_front.log 2>&1; echo "profile_front rc $?"; head -4 gpurun_out/prof_r03_front/kernel_stats.csv | cut -c1-160
timeout -k 10 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"; tail -c 300 $OUT/bench_default.json
