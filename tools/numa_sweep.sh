ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
D=/tmp/inq_rts; mkdir -p $D
[ -f $D/f.bam ] || timeout -k 10 400 python3 tools/make_synth_bam.py unphased100k 24000 $D/f native-seq 6 | tail -1
cat $D/f.bam > /dev/null; cat $D/f.bam > /dev/null
gcc -O2 -o /tmp/pcn tools/pagecache_nodes.c && /tmp/pcn $D/f.bam 256 | tail -2
for lws in 4 8; do
for mode in "default" "INQ_NUMA_CPUS=0" "INQ_NUMA_NODE=-1" "INQ_NUMA_NODE=0" "INQ_NUMA_NODE=1"; do
  for r in 1 2 3; do
    sleep 1.2
    if [ "$mode" = default ]; then E=""; else E="$mode"; fi
    env $E LOCAL_WORLD_SIZE=$lws LOCAL_RANK=0 INQ_FRONTEND=device INQ_TIMING=1 timeout -k 10 120 inquistr_amd/lib/inquistr call $D/f.bam -R $D/f.bed -t 16 -u --sample-name S 2> $D/err > $D/out.inq
    echo "lws=$lws $mode run $r: $(grep -o 'from the first.*' $D/err)"
  done
done
done
