#!/bin/bash
# kernel-trace of one bench configuration: tools/trace_gpu.sh TAG [bench args]  -> per-kernel mean durations
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 50 --warmup 5 --no-cpu-baseline "$@" > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
python3 "$ROOT/tools/summarize_profile.py" "$OUT" | grep -v "^$" | head -12
find "$OUT" -name "*.db" -delete; find "$OUT" -size +4M -delete
