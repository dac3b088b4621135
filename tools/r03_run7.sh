#!/bin/bash
# GPU box, round 3: literal-first test order A/B; NUMA binding of the loader; locus-kernel launch size in the CLI (rocprofv3).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03h
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for lv in 1 6; do
  for k in cigar ont qual seq; do
    for lib in libinq_nopair.so libinquistr_hip.so libinq_litfirst.so; do
      echo -n "level $lv $k $lib: " | tee -a $OUT/inflate_ab.txt
      INQ_LIB=$ROOT/inquistr_amd/lib/$lib ALGO=0 timeout -k 10 200 python3 tools/inflate_bench.py 20000 $lv $k 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/comp .* kernel/kernel/' | tee -a $OUT/inflate_ab.txt
    done
  done
done
CLI=$ROOT/inquistr_amd/lib/inquistr
ls /sys/devices/system/node/ | tee $OUT/numa.txt; cat /sys/devices/system/node/node*/cpulist | tee -a $OUT/numa.txt; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' ' | tee -a $OUT/numa.txt
python3 tools/make_synth_bam.py unphased100k 15000 /tmp/seq native-seq > $OUT/gen.txt 2>&1
python3 tools/make_synth_bam.py unphased100k 400000 /tmp/big native >> $OUT/gen.txt 2>&1
for node in none 0 1; do
  for i in 1 2 3; do
    if [ $node = none ]; then E="INQ_X=1"; else E="INQ_NUMA_NODE=$node"; fi
    t0=$(date +%s.%N); env $E INQ_FRONTEND=device INQ_TIMING=2 $CLI call /tmp/seq.bam -R /tmp/seq.bed -t 16 -u --sample-name S > /tmp/seq.inq 2> $OUT/seq_numa${node}_run$i.err; t1=$(date +%s.%N)
    python3 -c "print('seq numa=$node run $i: process wall %.3f s' % ($t1 - $t0))" | tee -a $OUT/numa_walls.txt
    grep "timing\] device" $OUT/seq_numa${node}_run$i.err | cut -c1-230 | tee -a $OUT/numa_walls.txt
    python3 - $OUT/seq_numa${node}_run$i.err <<'PY' | tee -a $OUT/numa_walls.txt
import re,sys
up=[float(m.group(1)) for m in re.finditer(r'upload ([\d.]+) ms for', open(sys.argv[1]).read())]
rd=[float(m.group(1)) for m in re.finditer(r'read\+tables ([\d.]+) ms', open(sys.argv[1]).read())]
sp=[float(m.group(1)) for m in re.finditer(r'\| wall ([\d.]+) ms', open(sys.argv[1]).read())]
st=[float(m.group(1)) for m in re.finditer(r'inq span\] @([\d.]+)', open(sys.argv[1]).read())]
import statistics as s
print('   uploads median %.2f ms, reads median %.2f ms, span calls median %.2f ms, span loop %.1f ms for %d spans' % (s.median(up[3:]), s.median(rd[3:]), s.median(sp[1:]), st[-1]-st[0], len(st)))
PY
  done
done
for fl in 40000 100000; do
  for i in 1 2 3; do
    t0=$(date +%s.%N); INQ_FLUSH_LOCI=$fl INQ_FRONTEND=device INQ_TIMING=2 $CLI call /tmp/big.bam -R /tmp/big.bed -t 16 -u --sample-name S > /tmp/big.inq 2> $OUT/big_flush${fl}_run$i.err; t1=$(date +%s.%N)
    python3 -c "print('big flush=$fl run $i: process wall %.3f s' % ($t1 - $t0))" | tee -a $OUT/flush_walls.txt
    grep "timing\] device" $OUT/big_flush${fl}_run$i.err | cut -c1-230 | tee -a $OUT/flush_walls.txt
  done
  cd /tmp && INQ_FLUSH_LOCI=$fl INQ_FAST_EXIT=0 INQ_FRONTEND=device rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$fl -- $CLI call /tmp/big.bam -R /tmp/big.bed -t 16 -u --sample-name S > /tmp/big_prof.inq 2> $OUT/trace_$fl.log; cd $ROOT
  for f in $(find $OUT/trace_$fl -name '*kernel_stats.csv'); do cp $f $OUT/cli4GB_flush${fl}_kernel_stats.csv; done
  f=$(find $OUT/trace_$fl -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && (head -1 $f; grep -E 'locus_call' $f) > $OUT/cli4GB_flush${fl}_locus_call_trace.csv
  rm -rf $OUT/trace_$fl
done
ls $OUT | wc -l
