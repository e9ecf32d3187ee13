#!/bin/bash
# Final bench lines of every workload (runs on the GPU box): gpurun_out/final/bench_<name>.json
mkdir -p gpurun_out/final
for w in dragon1080 bunny1080 dragon4k tiles4k mirror5 ao; do
  extra=""; [ $w = ao ] && extra="--steps 10 --warmup 2"; [ $w = tiles4k ] && extra="--steps 50"
  timeout -k 10 400 python bench.py --workload $w $extra > gpurun_out/final/bench_$w.json 2> gpurun_out/final/bench_$w.err || { echo "$w failed"; tail -3 gpurun_out/final/bench_$w.err; exit 1; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/final/bench_$w.json')); r=d['roofline']; print('$w', d['value'], d['ms_per_step'], 'kernel', r['kernel_ms'], 'bound', r['bound'], r['frac'], {k:v['frac'] for k,v in r['resources'].items()}, 'cpu', (d['cpu_baseline'] or {}).get('id_mismatches_vs_gpu'))"
done
