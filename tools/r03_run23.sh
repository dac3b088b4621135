#!/bin/bash
# GPU box, round 3: finer clock probe of the workgroup inflate (scratch build): lane 0's own decode / job time against the waits.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03p3
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
cat > /tmp/probe.py <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.environ["ROOT"])
from inquistr_amd import hipcall
from tools import inflate_bench
kind, level, tokens = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
comp = inflate_bench.make_blocks(20000, level, kind)
blocks = hipcall.scan_bgzf(comp)
ctx = hipcall.Context(0, lib=hipcall.load(os.environ["ROOT"] + "/inquistr_amd/lib/libinq_dbg_new.so"))
ctx.set_option("inflate_algo", 0); ctx.set_option("inflate_tokens", tokens)
for rep in range(2):
    rc, out, st = ctx.bgzf_inflate(comp, blocks, check=False)
ms, _ = ctx.timing_read(2)
v = st[: len(st) // 8 * 8].reshape(-1, 8).astype(np.int64)
n = v[:, 0] >> 8 & 0xfff
print(f"{kind} level {level} tokens {tokens}: kernel {ms:.2f} ms; per block: rounds {(v[:,1]&0xff).mean():.1f} passes {v[:,2].mean():.1f} stretches {n.mean():.1f} sweeps {(v[:,0]>>20).mean():.1f}")
print(f"  kcyc: counting: lane 0 decoding {(v[:,3]>>16).mean():.0f} + everything else (barriers, exchanges, scan) {v[:,5].mean():.0f}")
print(f"  kcyc: commit: setup {(v[:,3]&0xffff).mean():.0f} + lane 0's own jobs {v[:,4].mean():.0f} + waiting for the slowest job {v[:,6].mean():.0f}; sweeps {(v[:,1]>>8).mean():.0f}; sweeps + gather {v[:,7].mean():.0f}")
PY
export ROOT
for k in "cigar 6 1" "cigar 1 1" "ont 6 0" "qual 6 0"; do
  INQ_INFLATE_DEBUG=8 timeout -k 10 200 python3 /tmp/probe.py $k 2>&1 | grep -v amdgpu.ids | tee -a $OUT/inflate_probe_fine.txt
done
