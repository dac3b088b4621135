#!/bin/bash
# times several builds of the library on the inflate timing cases: tools/inflate_variants.sh lib1.so lib2.so ...
for k in cigar qual ont; do
  for lib in "$@"; do
    echo -n "$(basename $lib) "
    INQ_LIB=$lib ALGO=0 timeout -k 10 200 python tools/inflate_bench.py 20000 1 $k 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/comp .* kernel/kernel/'
  done
done
