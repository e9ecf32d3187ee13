show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['rays_per_gpu'], 'rays:', d['ms_per_step'], 'ms per frame, trace only', d.get('trace_only_ms'))"; }
for share in 8 4; do for su in 64 128 256 512 1024; do
  timeout -k 10 200 python3 bench.py --workload tiles4k --share-of $share --pipeline 0 --param split_units=$su --steps 100 --warmup 30 --no-cpu-baseline 2>/dev/null | show "share-of-$share split_units=$su" || exit 1
done; done
