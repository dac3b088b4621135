#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace stats + separate PMC passes for the DEVICE
# FRONT END (BGZF inflate, record scan, join) on a synthetic BAM, through the product CLI.
# Usage: tools/profile_front.sh <tag> [loci]
set -o pipefail
TAG=${1:-r01_front}
LOCI=${2:-50000}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/make_synth_bam.py unphased100k $LOCI /tmp/front_prof native > $OUT/gen.log 2>&1 || { tail $OUT/gen.log; exit 1; }
export INQ_FRONTEND=device
export INQ_FAST_EXIT=0  # the CLI normally leaves through _Exit, which would skip the profiler's output
CLI="$ROOT/inquistr_amd/lib/inquistr call /tmp/front_prof.bam -R /tmp/front_prof.bed -t 16 -u --sample-name S"
$CLI > $OUT/device.inq 2> $OUT/device.err || { tail $OUT/device.err; exit 1; }
INQ_FRONTEND=host $CLI > $OUT/host.inq 2>/dev/null
cmp $OUT/device.inq $OUT/host.inq && echo "device == host front end: identical .inq" > $OUT/identical.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CLI > /dev/null 2> $OUT/trace.log || { tail -20 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CLI > /dev/null 2> $OUT/pmc_fetch.log || { tail -20 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CLI > /dev/null 2> $OUT/pmc_write.log || { tail -20 $OUT/pmc_write.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- $CLI > /dev/null 2> $OUT/pmc_sq.log || { tail -20 $OUT/pmc_sq.log; }
for f in $(find $OUT/trace -name '*kernel_stats.csv'); do cp $f $OUT/kernel_stats.csv; done
for d in pmc_fetch pmc_write pmc_sq; do
  f=$(find $OUT/$d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && (head -1 $f; grep -E 'bgzf_|inflate|chain_kernel|record_parse|cigar_gather|join_kernel|scan_' $f) > $OUT/$d.csv
done
ls -l /tmp/front_prof.bam > $OUT/bam_size.txt
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/device.inq $OUT/host.inq
ls -la $OUT
cat $OUT/kernel_stats.csv
