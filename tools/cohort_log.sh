ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
D=/tmp/inq_cohort; mkdir -p $D $D/out gpurun_out/s10
timeout -k 10 300 python3 tools/make_synth_bam.py unphased100k 6000 $D/s0 native-seq 6 | tail -1
for k in 1 2 3; do cp $D/s0.bam $D/s$k.bam; cp $D/s0.bam.bai $D/s$k.bam.bai 2>/dev/null || cp $D/s0.bai $D/s$k.bai; done
for k in 0 1 2 3; do cat $D/s$k.bam > /dev/null; cat $D/s$k.bam > /dev/null; done
sleep 1
INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 300 inquistr_amd/lib/inquistr cohort -R $D/s0.bed -t 16 -u --out-dir $D/out $D/s0.bam $D/s1.bam $D/s2.bam $D/s3.bam 2> gpurun_out/s10/cohort_timing.err
grep -c . gpurun_out/s10/cohort_timing.err
