ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
D=/tmp/inq_rts; mkdir -p $D
[ -f $D/f.bam ] || timeout -k 10 400 python3 tools/make_synth_bam.py unphased100k 24000 $D/f native-seq 6 | tail -1
cat $D/f.bam > /dev/null; cat $D/f.bam > /dev/null
for cfg in "taskset -c 128-131|" "taskset -c 128-131|--ctx-option blocking_sync=0" "|" "|--ctx-option blocking_sync=1"; do
  TS=${cfg%%|*}; ARGS=${cfg##*|}
  for r in 1 2; do
    sleep 1.2
    INQ_FRONTEND=device INQ_TIMING=1 timeout -k 10 120 $TS inquistr_amd/lib/inquistr call $D/f.bam -R $D/f.bed -t 16 -u --sample-name S $ARGS 2> $D/err > $D/out.inq
    echo "[$TS $ARGS] $(grep -o '[0-9.]* s from the first.*' $D/err)"
    grep -o "cpu seconds so far.*uploader [0-9.]*" $D/err
  done
done
