#!/bin/bash
# GPU box, round 3: the round's rocprofv3 profiles (default bench command; device front end through the CLI; inflate counters)
# and the headline result files.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export TMPDIR=/tmp
bash tools/profile_round.sh r03 > gpurun_out/prof_r03.log 2>&1; echo "profile_round rc $?"; tail -3 gpurun_out/prof_r03.log
python3 tools/summarize_profile.py r03 unphased100k 100000 2463229012 > gpurun_out/prof_r03_summary.log 2>&1; echo "summarize rc $?"; tail -12 gpurun_out/prof_r03_summary.log
mkdir -p gpurun_out/profiles_r03; cp profiles/r03_* profiles/pmc_latest.json gpurun_out/profiles_r03/ 2>/dev/null
bash tools/profile_front.sh r03_front 50000 > gpurun_out/prof_r03_front.log 2>&1; echo "profile_front rc $?"; tail -25 gpurun_out/prof_r03_front.log | cut -c1-200
for k in ont cigar; do bash tools/profile_inflate.sh 20000 6 $k > gpurun_out/prof_inflate_$k.txt 2>&1; echo "== inflate counters $k"; cat gpurun_out/prof_inflate_$k.txt | grep -v amdgpu.ids | tail -30; done
bash tools/collect_round.sh r03 > gpurun_out/collect_r03.log 2>&1; echo "collect rc $?"; ls gpurun_out/results_r03
for i in 1 2; do timeout -k 10 600 python3 bench.py > gpurun_out/results_r03/bench_default_run$i.json 2> gpurun_out/results_r03/bench_default_run$i.err; echo "bench $i rc $?"; done
