#!/bin/bash
# usage (on the GPU box): tools/ab_params.sh ROUNDS "bench args" "k=v k=v" "k=v" ...   (optionally LIBS="lib1 lib2")
# Interleaved timing of scene parameter sets (each set: space-separated k=v, given to bench.py as --param) on one workload.
rounds=$1; wl=$2; shift 2
libs=${LIBS:-raytrace_amd/librtamd.so}
for r in $(seq $rounds); do
  for set in "$@"; do
    for lib in $libs; do
      params=""; for kv in $set; do params="$params --param $kv"; done
      line=$(RT_LIB=$PWD/$lib timeout -k 10 200 python bench.py $wl $params --no-cpu-baseline --no-compare 2>&1 | tail -1)
      echo "$(basename $lib) | $wl | $set => $(echo "$line" | python3 -c "import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms', 'kernel', d['roofline']['kernel_ms'])
except Exception as e: print('FAILED', e)")"
    done
  done
done
