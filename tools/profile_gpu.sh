#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace statistics and, in separate passes, the PMC
# counters behind the roofline numbers, all on the default bench.py workload.
# usage: tools/profile_gpu.sh TAG [bench args...]
set -o pipefail
TAG=${1:-run}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --no-cpu-baseline --no-compare $*"
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
            "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" \
            "GRBM_GUI_ACTIVE"; do
  name=$(echo "$pass" | tr ' ' '+' | cut -c1-40)
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_$name.log" 2>&1 || { echo "pass $pass failed"; tail -3 "$OUT/pmc_$name.log"; exit 1; }
done
python3 "$ROOT/tools/summarize_profile.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep only small files for the merge back
find "$OUT" -name "*.db" -delete
find "$OUT" -size +4M -delete
