#!/bin/bash
# GPU box: a SPARSE target list on a large BAM (targeted calling on a whole-genome file): every 100th / 10th locus of the 100 000-locus
# SEQ-bearing file (0.3 / 3.1 GB of its 31 GB are needed), product CLI (front end chosen by itself, and both forced) against CPU mode B.
# -> gpurun_out/sparse/result.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/sparse
mkdir -p $OUT; cd $ROOT
LOCI=${1:-100000}
python3 tools/make_synth_bam.py unphased100k $LOCI /tmp/sp native-seq 6 > $OUT/gen.log 2>&1 || { tail $OUT/gen.log; exit 1; }
make -C oracle ref_shaped_call > /dev/null
cat /tmp/sp.bam > /dev/null; cat /tmp/sp.bam > /dev/null
: > $OUT/result.txt
for step in 100 10 1; do
  awk -v s=$step 'NR % s == 1 || s == 1' /tmp/sp.bed > /tmp/sp_$step.bed
  n=$(wc -l < /tmp/sp_$step.bed)
  for fe in auto device host; do
    best=999
    for r in 1 2 3; do
      sleep 1.2
      t0=$(date +%s.%N)
      if [ $fe = auto ]; then inquistr_amd/lib/inquistr call /tmp/sp.bam -R /tmp/sp_$step.bed -t 16 -u --sample-name S > /tmp/sp_$fe.inq 2> /tmp/sp.err
      else INQ_FRONTEND=$fe inquistr_amd/lib/inquistr call /tmp/sp.bam -R /tmp/sp_$step.bed -t 16 -u --sample-name S > /tmp/sp_$fe.inq 2> /tmp/sp.err; fi
      rc=$?; t1=$(date +%s.%N)
      best=$(python3 -c "print(min($best, $t1 - $t0))")
    done
    echo "every ${step}th locus ($n loci) front end $fe: rc $rc best of 3 $best s" | tee -a $OUT/result.txt
  done
  t0=$(date +%s.%N); oracle/ref_shaped_call /tmp/sp.bam /tmp/sp_$step.bed B 16 1 5 3 S > /tmp/sp_B.inq; t1=$(date +%s.%N)
  echo "every ${step}th locus ($n loci) CPU mode B 16 threads: $(python3 -c "print($t1 - $t0)") s; identical to device/auto/host: $(cmp -s /tmp/sp_B.inq /tmp/sp_device.inq && echo yes || echo NO) $(cmp -s /tmp/sp_B.inq /tmp/sp_auto.inq && echo yes || echo NO) $(cmp -s /tmp/sp_B.inq /tmp/sp_host.inq && echo yes || echo NO)" | tee -a $OUT/result.txt
done
rm -f /tmp/sp.* /tmp/sp_*
