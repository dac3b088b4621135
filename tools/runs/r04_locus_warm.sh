#!/bin/bash
# GPU box, round 4: the locus kernels inside the CLI (0.66 - 0.73 of HBM in round 3's trace against 0.83 in bench.py's loop):
# the gather's stores non-temporal or not, one flush or two, the inflate of later spans beside the launch or not.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_locus
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
D=/tmp/inq_r04; mkdir -p $D
( time timeout -k 10 300 python3 tools/make_synth_bam.py unphased100k 100000 $D/cig native 6 ) 2>&1 | grep -E "wrote|real"
SHOW_CALLS=1 timeout -k 10 400 python3 tools/span_loop_runs.py $D/cig 4 --unphased INQ_INFLATE_AHEAD=0 INQ_INFLATE_AHEAD=0,INQ_GATHER_NT=1 INQ_INFLATE_AHEAD=0,INQ_FLUSH_LOCI=100000 INQ_INFLATE_AHEAD=0,INQ_FLUSH_LOCI=100000,INQ_GATHER_NT=1 - INQ_GATHER_NT=1 INQ_FLUSH_LOCI=100000,INQ_GATHER_NT=1 2>&1 | tee $OUT/locus_runs.txt
cd /tmp
for nt in 0 1; do
  INQ_GATHER_NT=$nt INQ_INFLATE_AHEAD=0 INQ_FRONTEND=device timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_nt$nt -o cli -- $ROOT/inquistr_amd/lib/inquistr call $D/cig.bam -R $D/cig.bed -t 16 -u --sample-name S > /dev/null 2> $OUT/prof_nt$nt.err || echo "rocprof nt=$nt failed"
done
find $OUT -name "*kernel_stats.csv" | head; for f in $(find $OUT -name "*kernel_stats.csv"); do echo $f; head -12 $f | cut -c1-200; done
rm -rf $D
