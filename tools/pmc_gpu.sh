#!/bin/bash
# Quick PMC passes on the default bench workload.  usage: BENCH_ARGS="..." tools/pmc_gpu.sh TAG "CTR CTR ..." "CTR ..." ...
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$i" -- python3 "$ROOT/bench.py" --steps 30 --warmup 5 --no-cpu-baseline --no-compare $BENCH_ARGS > "$OUT/pmc_$i.log" 2>&1 || { echo "pass $pass failed"; tail -3 "$OUT/pmc_$i.log"; exit 1; }
done
python3 "$ROOT/tools/summarize_profile.py" "$OUT" | grep -v "^==" | grep -v "^$"
find "$OUT" -name "*.db" -delete; find "$OUT" -size +2M -delete
