set -o pipefail
V=$PWD/raytrace_amd/variants/librtamd_tsplit.so
RT_LIB=$V timeout -k 10 400 python -u -m pytest tests/test_gpu_trace_pass.py tests/test_gpu_tiles.py tests/test_gpu_tiled_frame.py -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3 || exit 1
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('frames_in_flight'), d.get('trace_only_ms'), (d.get('rt_tiled_object') or {}).get('ms_per_step'))"; }
for r in 1 2; do
for lib in raytrace_amd/librtamd.so raytrace_amd/variants/librtamd_tsplit.so; do
  RT_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --workload mirror5 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | show "$lib mirror5" || exit 1
  RT_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --workload tiles4k --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | show "$lib tiles4k" || exit 1
  RT_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --workload tiles4k --share-of 8 --steps 200 --warmup 40 --no-cpu-baseline 2>/dev/null | show "$lib share8" || exit 1
  RT_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --workload tiles4k --share-of 8 --pipeline 0 --steps 200 --warmup 40 --no-cpu-baseline 2>/dev/null | show "$lib share8-one-at-a-time" || exit 1
done; done
