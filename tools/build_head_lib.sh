#!/bin/bash
# builds the committed (HEAD) csrc/ into inquistr_amd/lib/libinq_A.so, for tools/inflate_ab.sh against the working tree's build
set -e
cd "$(dirname "$0")/.."
rm -rf inquistr_amd/csrcA && mkdir inquistr_amd/csrcA
git archive HEAD inquistr_amd/csrc | tar -x --strip-components=2 -C inquistr_amd/csrcA
(cd inquistr_amd/csrcA && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -shared -o ../lib/libinq_A.so kernels.hip capi.hip bgzf_inflate.hip bgzf_inflate_wg.hip bam_scan.hip span.hip outlier.hip)
rm -rf inquistr_amd/csrcA
