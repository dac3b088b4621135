#!/bin/bash
# What one rank of a core-starved node gets: the CLI restricted (taskset) to 2 / 3 / 4 / 6 CPUs of the GPU's NUMA node - every thread of
# the process (readers, loader, uploader, the caller's thread, the runtime's own) shares them - against the same reader counts with all
# 16 granted cores to run on (LOCAL_WORLD_SIZE).  usage (GPU box): bash tools/few_cores.sh [loci]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
D=/tmp/inq_rts; mkdir -p $D
[ -f $D/f.bam ] || timeout -k 10 400 python3 tools/make_synth_bam.py unphased100k ${1:-24000} $D/f native-seq 6 | tail -1
cat $D/f.bam > /dev/null; cat $D/f.bam > /dev/null
CPUS=$(python3 -c "
import os
print(','.join(map(str, sorted(os.sched_getaffinity(0)))))")
echo "granted CPUs: $CPUS"
run() {  # label, env..., -- taskset list or ''
  local label=$1; shift
  for r in 1 2 3; do
    sleep 1.2
    env "$@" INQ_FRONTEND=device INQ_TIMING=1 timeout -k 10 120 $TS inquistr_amd/lib/inquistr call $D/f.bam -R $D/f.bed -t 16 -u --sample-name S $ARGS 2> $D/err > $D/out.inq
    echo "$label run $r: $(grep -o '[0-9.]* s from the first.*' $D/err) | wall $(grep -o 'total [0-9.]*s' $D/err | tail -1)"
  done
}
for n in 2 3 4 6; do
  LIST=$(python3 -c "
import os
c = sorted(os.sched_getaffinity(0))
# the CPUs of the second half of the mask are the ones the product binds readers to on this box (the GPU's node): take from there
print(','.join(map(str, c[len(c) // 2 : len(c) // 2 + $n])))")
  ARGS="--ctx-option blocking_sync=0" TS="taskset -c $LIST" run "taskset $n cpus, spinning waits" X=1
  ARGS="" TS="taskset -c $LIST" run "taskset $n cpus, blocking waits (the default below 8 cores)" X=1
done
ARGS=""
ARGS="--ctx-option blocking_sync=0" TS="" run "all cores, 2 readers, spinning" LOCAL_WORLD_SIZE=8 LOCAL_RANK=0
ARGS="" TS="" run "all cores, 2 readers, blocking (auto)" LOCAL_WORLD_SIZE=8 LOCAL_RANK=0
TS="" run "all cores, 4 readers" LOCAL_WORLD_SIZE=4 LOCAL_RANK=0
ARGS="--ctx-option blocking_sync=1" TS="" run "all cores, 16 readers, blocking waits" X=1
ARGS="--ctx-option blocking_sync=0" TS="" run "all cores, 16 readers, spinning waits (the default)" X=1
