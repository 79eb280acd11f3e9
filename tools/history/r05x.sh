#!/bin/bash
# r05x: cfg2's fused quad kernel with a deeper record look-ahead (NS = 6, 8: vmcnt(11) / vmcnt(15) instead of vmcnt(7)) and with the sample-row
# store removed (timing only): does the wave wait for store acknowledgements?  Variants built by the commands in profiles/r05x_*.txt's header.
out=gpurun_out/r05x_quad_lookahead.txt
: > $out
for v in ns4 ns6 ns8 nostore ns4; do
  cp tools/scratch/variants/lib_$v.so hamiltorch_amd/libhamiltorch_amd.so
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --no-secondary --no-api --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print('$v', 'value %.4g' % j['value'], 'ms_per_step %.5f' % j['ms_per_step'], 'kernel_ms %.5f' % r['kernel_ms'], 'frac %.3f' % r['frac'])" >> $out
  done
done
cat $out
