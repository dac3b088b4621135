ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
D=/tmp/cigc; mkdir -p $D/out gpurun_out/s12
python3 tools/make_synth_bam.py unphased100k 100000 $D/c0 native 6 | tail -1
for k in 1 2; do cp $D/c0.bam $D/c$k.bam; cp $D/c0.bam.bai $D/c$k.bam.bai 2>/dev/null || cp $D/c0.bai $D/c$k.bai; done
for k in 0 1 2; do cat $D/c$k.bam > /dev/null; cat $D/c$k.bam > /dev/null; done
sleep 1
INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 120 inquistr_amd/lib/inquistr cohort -R $D/c0.bed -t 16 -u --out-dir $D/out $D/c0.bam $D/c1.bam $D/c2.bam 2> gpurun_out/s12/cig_cohort.err
grep -c . gpurun_out/s12/cig_cohort.err
