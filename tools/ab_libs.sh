#!/bin/bash
# usage (on the GPU box): tools/ab_libs.sh ROUNDS "lib1 lib2 ..." "bench args 1" "bench args 2" ...
# Interleaved A/B of library builds: every round runs every (library, args) pair once; prints all lines.
rounds=$1; libs=$2; shift 2
for r in $(seq $rounds); do
  for args in "$@"; do
    for lib in $libs; do
      line=$(RT_LIB=$PWD/$lib timeout -k 10 200 python bench.py $args --no-cpu-baseline --no-compare 2>&1 | tail -1)
      echo "$lib | $args => $(echo "$line" | python3 -c "import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms', 'kernel', d['roofline']['kernel_ms'])
except Exception as e: print('FAILED', e)")"
    done
  done
done
