set -e
run() { echo "== $*" ; timeout -k 10 120 python bench.py --workload tiles4k --share-of 8 --no-cpu-baseline --steps 200 --warmup 40 "$@" 2>gpurun_out/r3m/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:d.get(k) for k in ('ms_per_step','trace_only_ms','gather_and_scatter_ms','host_enqueue_us_per_step','frames_in_flight','launch_orders_built_on','pipelines_tried','root_head_start')}, d['config']['rays_per_gpu'])
" || tail -5 gpurun_out/r3m/err.txt; }
run --gather bgra8
run --gather bgra8
run --gather hits8
run --gather bgra8 --share-of 2
run --gather bgra8 --share-of 4
