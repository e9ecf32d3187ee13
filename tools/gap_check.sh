#!/bin/bash
# usage (GPU box): tools/gap_check.sh "bench args"  -> kernel durations and launch gaps of a short profiled run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gap && timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gap -o t -- python3 bench.py $1 --steps 100 --warmup 10 --no-cpu-baseline --no-compare > gpurun_out/gap.log 2>&1 || exit 1
python3 tools/launch_gaps.py $(find gpurun_out/gap -name "*kernel_trace.csv" | head -1)
cut -d, -f1-4,6,7 $(find gpurun_out/gap -name "*kernel_stats.csv" | head -1) | cut -c1-170
