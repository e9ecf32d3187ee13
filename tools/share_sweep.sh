set -e
mkdir -p gpurun_out/r3m
run() { echo "== $*" ; timeout -k 10 120 python bench.py --workload tiles4k --share-of 8 --no-cpu-baseline --steps 200 --warmup 40 "$@" | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:d.get(k) for k in ('ms_per_step','trace_only_ms','gather_and_scatter_ms','host_enqueue_us_per_step','same_steps_one_frame_at_a_time')})
" ; }
run --pipeline 2
run --pipeline 2 --scene-per-stream 1
run --pipeline 3 --scene-per-stream 1
run --pipeline 4 --scene-per-stream 1
run --pipeline 2 --param feedback_period=100000
run --pipeline 3 --scene-per-stream 1 --tile 256
run --pipeline 2 --scene-per-stream 1 --gather hits16
