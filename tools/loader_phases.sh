ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
D=/tmp/inq_rts; mkdir -p $D
[ -f $D/f.bam ] || timeout -k 10 400 python3 tools/make_synth_bam.py unphased100k 24000 $D/f native-seq 6 | tail -1
cat $D/f.bam > /dev/null; cat $D/f.bam > /dev/null
for lws in 1 4 8; do
    sleep 1.2
    LOCAL_WORLD_SIZE=$lws LOCAL_RANK=0 INQ_FRONTEND=device INQ_TIMING=2 timeout -k 10 120 inquistr_amd/lib/inquistr call $D/f.bam -R $D/f.bed -t 16 -u --sample-name S 2> $D/err > $D/out.inq
    echo "== LOCAL_WORLD_SIZE=$lws: $(grep -o 'from the first.*' $D/err)"
    grep "read+tables" $D/err | sed -n '6,12p' | grep -o "plan.*"
    grep "upload [0-9.]* ms" $D/err | sed -n '8,11p' | grep -o "slot.*"
done
