#!/bin/bash
# GPU box, round 3: can the device copy straight from a mapped file's page-cache pages?
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03fm
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
python3 tools/make_synth_bam.py unphased100k 12000 /tmp/fm native-seq > $OUT/gen.log 2>&1 || { tail $OUT/gen.log; exit 1; }
cat /tmp/fm.bam > /dev/null
timeout -k 10 120 $ROOT/inquistr_amd/lib/filemap_probe /tmp/fm.bam 2>&1 | tee $OUT/filemap_probe.txt
