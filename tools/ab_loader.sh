ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
D=/tmp/inq_rts; mkdir -p $D
[ -f $D/f.bam ] || timeout -k 10 400 python3 tools/make_synth_bam.py unphased100k 24000 $D/f native-seq 6 | tail -1
cat $D/f.bam > /dev/null; cat $D/f.bam > /dev/null
for lws in 1 4 8; do
 for rep in 1 2 3 4; do
  for which in old new; do
    CLI=inquistr_amd/lib/inquistr; [ $which = old ] && CLI=inquistr_amd/lib/old/inquistr
    sleep 1.2
    LOCAL_WORLD_SIZE=$lws LOCAL_RANK=0 INQ_FRONTEND=device INQ_TIMING=1 timeout -k 10 120 $CLI call $D/f.bam -R $D/f.bed -t 16 -u --sample-name S 2> $D/err > $D/out_$which.inq
    echo "lws=$lws $which run $rep: $(grep -o '[0-9.]* s from the first.*' $D/err)"
  done
 done
done
cmp $D/out_old.inq $D/out_new.inq && echo "same .inq"
