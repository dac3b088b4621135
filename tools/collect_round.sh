#!/bin/bash
# Collects the round's headline measurements on the GPU box into gpurun_out/results_<tag>/ (copy to profiles/).
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/results_$TAG
mkdir -p $OUT
cd $ROOT
for w in unphased100k phased10k expansion50k longreads20k; do
  timeout -k 10 500 python bench.py --workload $w > $OUT/l0_$w.json 2> $OUT/l0_$w.err || echo "bench $w failed"
done
timeout -k 10 500 python bench.py --workload shard500k > $OUT/l0_shard500k_one_rank.json 2> $OUT/l0_shard500k.err || echo "bench shard failed"
timeout -k 10 300 python tools/l1_bench.py --loci 50000 > $OUT/l1_unphased100k_50kloci.txt 2>&1 || echo "l1 failed"
timeout -k 10 300 inquistr_amd/lib/hbm_read_peak > $OUT/hbm_read_peak.txt 2>&1 || echo "peak failed"
timeout -k 10 900 python bench.py --l2 --workload unphased100k --l2-loci 50000 > $OUT/l2_unphased100k_50kloci.json 2> $OUT/l2.err || echo "l2 failed"
timeout -k 10 600 python bench.py --l2 --workload phased10k --l2-loci 10000 > $OUT/l2_phased10k.json 2>> $OUT/l2.err || echo "l2b failed"
ls -la $OUT
