#!/bin/bash
# GPU box: the round's rocprofv3 summaries (hot kernel: kernel trace + PMC passes; device front end: the same on the 501 MB file)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export TMPDIR=/tmp
bash tools/profile_round.sh r04 2>&1 | tail -25
bash tools/profile_front.sh r04_front 50000 2>&1 | tail -30
