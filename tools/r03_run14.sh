#!/bin/bash
# GPU box, round 3: tokens by the data; two-rank rehearsal of bench.py --gpus; the whole gpu suite once more.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03o
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/gputest.log; tail -4 $OUT/gputest.log
for lv in 1 6; do for k in cigar ont; do
  for tk in 0 1 -1; do
    echo -n "level $lv $k tokens=$tk (pairs by the data): " | tee -a $OUT/inflate_tokens_auto.txt
    TOKENS=$tk ALGO=0 timeout -k 10 200 python3 tools/inflate_bench.py 20000 $lv $k 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/comp .* kernel/kernel/' | tee -a $OUT/inflate_tokens_auto.txt
  done
done; done
for mode in weak strong; do
  W=unphased100k; [ $mode = strong ] && W=shard500k
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --backend gloo --same-device --steps 6 --warmup 2 --workload $W --scaling $mode --loci-per-gpu 40000 > $OUT/bench_two_ranks_$mode.json 2> $OUT/bench_two_ranks_$mode.err; echo "two-rank bench ($mode) rc $?"; tail -c 400 $OUT/bench_two_ranks_$mode.json; echo
done
bash tools/profile_front.sh r03_front 50000 > gpurun_out/prof_r03_front.log 2>&1; echo "profile_front rc $?"; head -3 gpurun_out/prof_r03_front/kernel_stats.csv | cut -c1-140
