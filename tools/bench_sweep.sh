#!/bin/bash
# usage: tools/bench_sweep.sh "args for run 1" "args for run 2" ...   (each run is one bench.py call)
set -o pipefail
for args in "$@"; do
  timeout -k 10 300 python3 -u bench.py --no-cpu-baseline --no-compare $args 2>&1 | python3 tools/short.py "$args" || exit 1
done
