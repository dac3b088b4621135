#!/bin/bash
# One entry for the round's runs on the GPU box:  gpurun --timeout 1200 -- 'bash tools/gpurun_round.sh <what> [args]'
#   tests                 the -m gpu suite
#   bench [args]          the default bench line (what the driver runs), timed, with a short digest
#   state                 inflate rates by kind of data + tests + bench
#   profiles              rocprofv3 summaries: hot kernel (tools/profile_round.sh) and device front end (tools/profile_front.sh)
#   loop [loci] [level]   span loop of a SEQ-bearing file: runs with the CLI's own clocks, /proc/vmstat and cpu.stat beside them
#   cig [runs]            the same on the 0.8 GB CIGAR-only file, back to back and with rests between the processes
#   soak                  damaged deflate streams, random BAMs through both front ends (new seeds)
#   probes                what allocations, streams, synchronisations cost; hipMalloc back to back
# Results go to gpurun_out/<what>/ (scratch); what is quoted in DESIGN.md is copied to profiles/r05_results/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
WHAT=${1:-tests}; shift
OUT=$ROOT/gpurun_out/round_$WHAT
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
D=/tmp/inq_round; mkdir -p $D
CLI=inquistr_amd/lib/inquistr

run_tests() { timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -3 $OUT/gpu_tests.txt; }
run_bench() {
  local s=$(date +%s)
  timeout -k 10 1100 python3 bench.py "$@" > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $? in $(( $(date +%s) - s )) s"
  tail -c 400 $OUT/bench_default.err
  python3 tools/bench_digest.py $OUT/bench_default.json
}
case $WHAT in
tests) run_tests ;;
bench) run_bench "$@" ;;
state)
  for n in 20000 40000; do for lv in 1 6; do for k in cigar seq ont qual; do
    echo -n "$n blocks level $lv $k: " | tee -a $OUT/inflate_state.txt
    timeout -k 10 200 python3 tools/inflate_bench.py $n $lv $k 2>&1 | grep kernel | tail -1 | tee -a $OUT/inflate_state.txt
  done; done; done
  run_tests; run_bench ;;
profiles)
  bash tools/profile_round.sh r05 2>&1 | tail -25
  bash tools/profile_front.sh r05_front 50000 2>&1 | tail -30 ;;
loop)
  [ -x inquistr_amd/lib/pagecache_nodes ] || gcc -O2 -o inquistr_amd/lib/pagecache_nodes tools/pagecache_nodes.c
  ( time timeout -k 10 600 python3 tools/make_synth_bam.py unphased100k ${1:-40000} $D/seq native-seq ${2:-6} ) 2>&1 | grep -E "wrote|real"
  inquistr_amd/lib/pagecache_nodes $D/seq.bam 256 | tee $OUT/pagecache.txt
  timeout -k 10 500 python3 tools/span_loop_runs.py $D/seq ${3:-14} --unphased --keep-slow $OUT/err - 2>&1 | tee $OUT/seq_runs.txt ;;
cig)
  ( time timeout -k 10 300 python3 tools/make_synth_bam.py unphased100k 100000 $D/cig native 6 ) 2>&1 | grep -E "wrote|real"
  echo "--- back to back" | tee $OUT/cig_runs.txt
  SHOW_CALLS=1 timeout -k 10 300 python3 tools/span_loop_runs.py $D/cig ${1:-12} --unphased - 2>&1 | tee -a $OUT/cig_runs.txt
  echo "--- 1.5 s between runs" | tee -a $OUT/cig_runs.txt
  PAUSE_S=1.5 timeout -k 10 300 python3 tools/span_loop_runs.py $D/cig ${1:-12} --unphased - 2>&1 | tee -a $OUT/cig_runs.txt ;;
soak)
  INQ_SOAK_SEED=${1:-740400} timeout -k 10 420 python3 tools/soak_inflate.py ${4:-250} 3 > $OUT/soak_inflate.txt 2>&1; echo "soak_inflate rc $?" | tee -a $OUT/soak_inflate.txt; tail -2 $OUT/soak_inflate.txt
  timeout -k 10 420 python3 tools/soak_e2e.py --cases ${5:-400} --frontend device --seed0 ${2:-818000} > $OUT/soak_e2e_device.txt 2>&1; echo "soak_e2e device rc $?" | tee -a $OUT/soak_e2e_device.txt; tail -2 $OUT/soak_e2e_device.txt
  timeout -k 10 120 python3 tools/soak_e2e.py --cases 80 --frontend host --seed0 ${3:-919000} > $OUT/soak_e2e_host.txt 2>&1; echo "soak_e2e host rc $?" | tee -a $OUT/soak_e2e_host.txt; tail -2 $OUT/soak_e2e_host.txt ;;
probes)
  for x in alloc_probe stream_probe; do [ -x inquistr_amd/lib/$x ] || /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -pthread -Wno-unused-value tools/$x.hip -o inquistr_amd/lib/$x; done
  timeout -k 10 60 inquistr_amd/lib/alloc_probe 2>&1 | tee $OUT/alloc_probe.txt
  timeout -k 10 60 inquistr_amd/lib/stream_probe 5 2>&1 | tee $OUT/stream_probe.txt
  for i in 1 2 3; do timeout -k 10 60 inquistr_amd/lib/alloc_probe 2>&1 | grep -E "hipMalloc (1024|4096) MB"; done | tee $OUT/alloc_back_to_back.txt
  timeout -k 10 60 inquistr_amd/lib/hip_startup_probe 2>&1 | tee $OUT/hip_startup.txt ;;
*) echo "unknown: $WHAT"; exit 2 ;;
esac
rm -rf $D
