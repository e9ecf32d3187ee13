# one frame at a time (no pipelining), the full trace step with and without split units, by launch size
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['rays_per_gpu'], 'rays:', d['ms_per_step'], 'ms per frame, trace only', d.get('trace_only_ms'))"; }
for share in 8 4 2 1; do for ts in 0 1; do
  extra=""; [ $share -gt 1 ] && extra="--share-of $share"
  timeout -k 10 200 python3 bench.py --workload tiles4k $extra --pipeline 0 --param trace_split=$ts --steps 100 --warmup 30 --no-cpu-baseline 2>/dev/null | show "share-of-$share trace_split=$ts" || exit 1
done; done
