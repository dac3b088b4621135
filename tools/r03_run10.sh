#!/bin/bash
# GPU box, round 3: NUMA preference of the allocating threads (12.8 GB SEQ-bearing file), inflate re-check, outlier timing.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03k
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for lv in 1 6; do
  for k in cigar ont qual seq; do
    for lib in libinq_nopair.so libinquistr_hip.so; do
      echo -n "level $lv $k $lib: " | tee -a $OUT/inflate_ab.txt
      INQ_LIB=$ROOT/inquistr_amd/lib/$lib ALGO=0 timeout -k 10 200 python3 tools/inflate_bench.py 20000 $lv $k 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/comp .* kernel/kernel/' | tee -a $OUT/inflate_ab.txt
    done
  done
done
CLI=$ROOT/inquistr_amd/lib/inquistr
python3 tools/make_synth_bam.py unphased100k 40000 /tmp/seq40k native-seq > $OUT/gen.txt 2>&1; cat $OUT/gen.txt
for mode in default cpus off; do
  case $mode in default) E="INQ_X=1";; cpus) E="INQ_NUMA_CPUS=1";; off) E="INQ_NUMA_NODE=-1";; esac
  for i in 1 2 3 4; do
    t0=$(date +%s.%N); env $E INQ_FRONTEND=device INQ_TIMING=2 $CLI call /tmp/seq40k.bam -R /tmp/seq40k.bed -t 16 -u --sample-name S > /tmp/seq40k.inq 2> $OUT/seq13GB_${mode}_run$i.err; t1=$(date +%s.%N)
    python3 -c "print('seq 12.8 GB numa=$mode run $i: process wall %.3f s' % ($t1 - $t0))" | tee -a $OUT/seq13GB_walls.txt
    grep "timing\] device" $OUT/seq13GB_${mode}_run$i.err | cut -c1-230 | tee -a $OUT/seq13GB_walls.txt
    grep "inq ctx\] @" $OUT/seq13GB_${mode}_run$i.err | cut -c1-120 | tee -a $OUT/seq13GB_walls.txt
    python3 - $OUT/seq13GB_${mode}_run$i.err <<'PY' | tee -a $OUT/seq13GB_walls.txt
import re,sys,statistics as s
t=open(sys.argv[1]).read()
up=[float(m.group(1)) for m in re.finditer(r'upload ([\d.]+) ms for', t)]
rd=[float(m.group(1)) for m in re.finditer(r'read\+tables ([\d.]+) ms', t)]
sp=[(float(m.group(1)),float(m.group(2)),float(m.group(3)),float(m.group(4))) for m in re.finditer(r'inq span\] @([\d.]+) waited ([\d.]+) ms \| loci \d+ comp ([\d.]+) MB.*\| wall ([\d.]+) ms', t)]
loop=sp[-1][0]-sp[0][0]+sp[0][3]; waited=sum(x[1] for x in sp[1:]); comp=sum(x[2] for x in sp)
print('   %d spans, %.1f MB: span loop %.1f ms = %.1f GB/s of compressed bytes; waiting for the loader behind the first span %.1f ms = %.1f %% of the loop; uploads median %.2f ms, reads median %.2f ms, span calls median %.2f ms' % (len(sp), comp, loop, comp/loop, waited, 100*waited/loop, s.median(up[3:]), s.median(rd[3:]), s.median(x[3] for x in sp[1:])))
PY
  done
done
rm -f /tmp/seq40k.bam
python3 tools/make_cohort.py 200000 200 /tmp/cohort.tsv
for i in 1 2 3 4 5; do
  t0=$(date +%s.%N); INQ_TIMING=1 $CLI outlier /tmp/cohort.tsv > /tmp/outl_$i.txt 2> $OUT/outlier_run$i.err; t1=$(date +%s.%N)
  python3 -c "print('outlier 139 MB run $i: process wall %.3f s' % ($t1 - $t0))" | tee -a $OUT/outlier_walls.txt
  grep "inq outlier" $OUT/outlier_run$i.err | tr '\n' ';' | tee -a $OUT/outlier_walls.txt; echo | tee -a $OUT/outlier_walls.txt
done
timeout -k 10 600 python3 -m pytest tests/test_gpu_outlier.py -m gpu -x -q 2>&1 | tail -2 | tee -a $OUT/outlier_walls.txt
ls $OUT | wc -l
