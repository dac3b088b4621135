#!/bin/bash
# A/B of two builds of the library on the inflate timing cases: tools/inflate_ab.sh libA.so libB.so [blocks]
for k in cigar qual seq ont; do
  for lib in "$1" "$2"; do
    echo -n "$(basename $lib) "
    INQ_LIB=$lib ALGO=0 timeout -k 10 200 python tools/inflate_bench.py ${3:-20000} 1 $k 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
