#!/bin/bash
# interleaved: library before the split kernels / with them at split_units 0 / at split_units N (default 32)
n=${1:-32}
run() { line=$(RT_LIB=$PWD/$1 timeout -k 10 200 python bench.py $2 --no-cpu-baseline --no-compare 2>&1 | tail -1); echo "$1 | $2 => $(echo "$line" | python3 -c "import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms')
except Exception as e: print('FAILED', e)")"; }
for r in 1 2; do
for w in "--steps 200 --warmup 20" "--workload bunny1080 --steps 200 --warmup 20" "--workload dragon4k --steps 100 --warmup 10" "--workload ao --rays 16777216 --steps 5 --warmup 2"; do
  run raytrace_amd/variants/librtamd_stopev.so "$w"
  run raytrace_amd/variants/librtamd_split.so "$w --param split_units=0"
  run raytrace_amd/variants/librtamd_split.so "$w --param split_units=$n"
done; done
