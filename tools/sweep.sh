#!/bin/bash
# usage: tools/sweep.sh "common bench args" "variant args 1" "variant args 2" ...   (runs on the GPU box)
common=$1; shift
for v in "$@"; do
  line=$(timeout -k 10 150 python bench.py $common $v --no-cpu-baseline --no-compare 2>&1 | tail -1)
  echo "$v => $(echo "$line" | python3 -c "import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms', 'kernel', d['roofline']['kernel_ms'])
except Exception as e: print('FAILED', e)")"
done
