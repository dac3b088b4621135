#!/bin/bash
# GPU box, round 3: what the process's exit pays for (INQ_EXIT_PROBE), on a 4.0 GB CIGAR-only and a 4.8 GB SEQ-bearing file.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03p
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
CLI=$ROOT/inquistr_amd/lib/inquistr
python3 tools/make_synth_bam.py unphased100k 400000 /tmp/big native > $OUT/gen.txt 2>&1
python3 tools/make_synth_bam.py unphased100k 15000 /tmp/seq native-seq >> $OUT/gen.txt 2>&1
for f in big seq; do
  for probe in none 1 2; do
    for i in 1 2 3; do
      if [ $probe = none ]; then E="INQ_X=1"; else E="INQ_EXIT_PROBE=$probe"; fi
      t0=$(date +%s.%N); env $E INQ_FRONTEND=device INQ_TIMING=1 $CLI call /tmp/$f.bam -R /tmp/$f.bed -t 16 -u --sample-name S > /tmp/$f.inq 2> $OUT/${f}_probe${probe}_run$i.err; t1=$(date +%s.%N)
      tot=$(grep "timing\] device" $OUT/${f}_probe${probe}_run$i.err | sed 's/.*total \([0-9.]*\)s.*/\1/')
      python3 -c "print('$f probe=$probe run $i: process wall %.3f s, inside the call %s s, outside %.3f s' % ($t1 - $t0, '$tot', $t1 - $t0 - float('$tot')))" | tee -a $OUT/exit_probe.txt
      grep "exit probe" $OUT/${f}_probe${probe}_run$i.err | tee -a $OUT/exit_probe.txt
    done
  done
done
