#!/bin/bash
out=gpurun_out/r05y_quad_ablation.txt
: > $out
for v in ns4 nostore $EXTRA_VARIANTS; do
  cp tools/scratch/variants/lib_$v.so hamiltorch_amd/libhamiltorch_amd.so
  python tools/history/r05y.py $v >> $out 2>&1
done
cat $out
