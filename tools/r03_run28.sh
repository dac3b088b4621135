#!/bin/bash
# GPU box, round 3: the round's final rocprofv3 summaries - default bench command (kernel trace + PMC passes), SQ counters of the inflate.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export TMPDIR=/tmp
timeout -k 10 600 bash tools/profile_round.sh r03 > gpurun_out/profile_round_r03.log 2>&1; echo "profile_round rc $?"; tail -5 gpurun_out/profile_round_r03.log
timeout -k 10 250 bash tools/profile_inflate.sh 20000 6 cigar > gpurun_out/inflate_counters_cigar_only_level6.txt 2>&1; echo "rc $?"
timeout -k 10 250 bash tools/profile_inflate.sh 20000 6 ont > gpurun_out/inflate_counters_nanopore_like_level6.txt 2>&1; echo "rc $?"
tail -30 gpurun_out/inflate_counters_nanopore_like_level6.txt
