#!/bin/bash
# GPU box, round 4: where the 12.8 GB SEQ-bearing file's time goes before the span loop is changed - six runs with INQ_TIMING=2 stamps,
# the page cache's NUMA placement and the dirty / writeback state in front of every run (the 1.45 s outlier), stream costs.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_trace
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
LOCI=${1:-40000}
LEVEL=${2:-1}
timeout -k 10 60 inquistr_amd/lib/stream_probe 5 > $OUT/stream_probe.txt 2>&1; cat $OUT/stream_probe.txt
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; df -h /tmp | tail -1; grep -E "MemTotal|MemAvailable" /proc/meminfo
lscpu | grep -E "NUMA|Socket|Model name" ; cat /sys/class/drm/renderD*/device/numa_node 2>/dev/null | tr '\n' ' '; echo
D=/tmp/inq_r04; mkdir -p $D
( time timeout -k 10 600 python3 tools/make_synth_bam.py unphased100k $LOCI $D/big native-seq $LEVEL ) 2>&1 | grep -E "wrote|real"
CLI=inquistr_amd/lib/inquistr
meminfo() { grep -E "^(Dirty|Writeback|Cached):" /proc/meminfo | tr -s ' ' | tr '\n' ' '; echo; }
for i in 0 1 2 3 4 5; do
  echo "== run $i" | tee -a $OUT/runs.txt
  meminfo | tee -a $OUT/runs.txt
  inquistr_amd/lib/pagecache_nodes $D/big.bam 256 | tee -a $OUT/runs.txt
  s=$(date +%s.%N)
  INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 120 $CLI call $D/big.bam -R $D/big.bed -t 16 -u --sample-name S > $D/out$i.inq 2> $OUT/run$i.err || echo "run $i failed"
  e=$(date +%s.%N)
  echo "run $i wall $(echo "$e - $s" | bc -l) s" | tee -a $OUT/runs.txt
  grep -E "inq timing|context ready" $OUT/run$i.err | tee -a $OUT/runs.txt
  if [ $i -eq 1 ]; then sync; echo "(sync done)" | tee -a $OUT/runs.txt; fi
done
cmp $D/out0.inq $D/out5.inq && echo "outputs identical"
rm -rf $D
