#!/bin/bash
# GPU box, round 3: where a served call's 89 ms go (server-side stage lines, INQ_TIMING=2) - 1.0 GB CIGAR-only file, 100 000 loci.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03sv2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
python3 tools/make_synth_bam.py unphased100k 100000 /tmp/sv native > $OUT/gen.log 2>&1 || { tail $OUT/gen.log; exit 1; }
CLI=$ROOT/inquistr_amd/lib/inquistr
INQ_TIMING=2 INQ_FRONTEND=device $CLI serve --socket /tmp/sv.sock --idle-exit 60 2> $OUT/server.err &
SP=$!
for i in $(seq 1 100); do [ -S /tmp/sv.sock ] && break; sleep 0.05; done
for i in 1 2 3 4; do
  s=$(date +%s%N); INQ_SERVER=/tmp/sv.sock $CLI call /tmp/sv.bam -R /tmp/sv.bed -t 16 -u --sample-name S > /tmp/sv_$i.inq; e=$(date +%s%N)
  echo "served call $i: $(( (e - s) / 1000000 )) ms" | tee -a $OUT/served_walls.txt
done
$CLI serve --socket /tmp/sv.sock --quit
wait $SP
grep -E "inq prepare|device front end:|inq output" $OUT/server.err | tail -12
