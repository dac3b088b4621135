#!/bin/bash
# GPU box, round 3: span buffers page-locked in place (hipHostRegister) before their first upload - A/B on a 4.8 GB SEQ-bearing
# and a 1.0 GB CIGAR-only file, with and without the NUMA placement of the threads.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03q
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
CLI=$ROOT/inquistr_amd/lib/inquistr
python3 tools/make_synth_bam.py unphased100k 100000 /tmp/c0 native > $OUT/gen.txt 2>&1
python3 tools/make_synth_bam.py unphased100k 15000 /tmp/seq native-seq >> $OUT/gen.txt 2>&1
for f in seq c0; do
  for mode in base reg reg_nonuma base_nonuma; do
    case $mode in base) E="INQ_X=1";; reg) E="INQ_SPAN_REGISTER=1";; reg_nonuma) E="INQ_SPAN_REGISTER=1 INQ_NUMA_NODE=-1";; base_nonuma) E="INQ_NUMA_NODE=-1";; esac
    for i in 1 2 3 4; do
      t0=$(date +%s.%N); env $E INQ_FRONTEND=device INQ_TIMING=2 $CLI call /tmp/$f.bam -R /tmp/$f.bed -t 16 -u --sample-name S > /tmp/$f.inq 2> $OUT/${f}_${mode}_run$i.err; t1=$(date +%s.%N)
      python3 - $OUT/${f}_${mode}_run$i.err "$f $mode run $i: process wall $(python3 -c "print('%.3f' % ($t1 - $t0))") s" <<'PY' | tee -a $OUT/register_ab.txt
import re,sys,statistics as s
t=open(sys.argv[1]).read()
up=[float(m.group(1)) for m in re.finditer(r'upload ([\d.]+) ms for', t)]
sp=[(float(m.group(1)),float(m.group(2)),float(m.group(4))) for m in re.finditer(r'inq span\] @([\d.]+) waited ([\d.]+) ms \| loci \d+ comp ([\d.]+) MB.*\| wall ([\d.]+) ms', t)]
ctx=re.search(r'inq ctx\] @([\d.]+)', t)
tot=re.search(r'total ([\d.]+)s', t)
loop=sp[-1][0]-sp[0][0]+sp[0][2]
print('%s | ctx @%s ms, total %s s | uploads: first three %s, median of the rest %.2f ms | span loop %.1f ms, waited behind the first span %.1f ms' % (sys.argv[2], ctx.group(1) if ctx else '?', tot.group(1) if tot else '?', [round(x,1) for x in up[:3]], s.median(up[3:]) if len(up)>3 else -1, loop, sum(x[1] for x in sp[1:])))
PY
    done
  done
done
