#!/bin/bash
# usage (on the GPU box): tools/ab_coop.sh ROUNDS  -- interleaved timing of the cooperative split against the library build before it
rounds=${1:-3}
run() { # lib, args
  line=$(RT_LIB=$PWD/$1 timeout -k 10 200 python bench.py $2 --no-cpu-baseline --no-compare 2>&1 | tail -1)
  echo "$1 | $2 => $(echo "$line" | python3 -c "import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms', 'kernel', d['roofline']['kernel_ms'])
except Exception as e: print('FAILED', e)")"
}
for r in $(seq $rounds); do
  for w in "" "--workload bunny1080" "--workload dragon4k"; do
    run raytrace_amd/variants/librtamd_head.so "$w"
    run raytrace_amd/librtamd.so "$w --param split_coop=0"
    run raytrace_amd/librtamd.so "$w --param split_coop=1"
    run raytrace_amd/librtamd.so "$w --param split_coop=1 --param split_units=128"
    run raytrace_amd/librtamd.so "$w --param split_coop=1 --param split_units=512"
  done
done
