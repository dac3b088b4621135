#!/bin/bash
# GPU box, round 3: does any runtime environment knob shorten the HIP start-up of a one-file process?  (five starts each)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03env
mkdir -p $OUT
cd $ROOT
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/hsp tools/hip_startup_probe.hip 2>/dev/null || exit 1
run() {  # label, env assignments...
  label=$1; shift
  for i in 1 2 3 4 5; do
    s=$(date +%s%N); env "$@" /tmp/hsp > /tmp/hsp.out 2>&1; e=$(date +%s%N)
    echo "$label: process $(( (e - s) / 1000000 )) ms | $(tr '\n' ';' < /tmp/hsp.out | tr -s ' ')" | tee -a $OUT/startup_env.txt
  done
}
run "default" X=1
run "HIP_VISIBLE_DEVICES=0" HIP_VISIBLE_DEVICES=0
run "ROCR_VISIBLE_DEVICES=0" ROCR_VISIBLE_DEVICES=0
run "HSA_ENABLE_SDMA=0" HSA_ENABLE_SDMA=0
run "GPU_MAX_HW_QUEUES=1" GPU_MAX_HW_QUEUES=1
run "HSA_ENABLE_INTERRUPT=0" HSA_ENABLE_INTERRUPT=0
run "HIP_HOST_COHERENT=0" HIP_HOST_COHERENT=0
run "AMD_SERIALIZE_KERNEL=0,HSA_NO_SCRATCH_RECLAIM=1" HSA_NO_SCRATCH_RECLAIM=1
run "HSA_XNACK=0" HSA_XNACK=0
