set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_tiles.py tests/test_layout_and_symbols.py -x -q --timeout 120 --timeout-method thread > gpurun_out/s7_rehearse_tests.log 2>&1 || { grep -n "^E " gpurun_out/s7_rehearse_tests.log | head -20; exit 1; }
for g in hits16 bgra8; do
  timeout -k 10 300 python bench.py --workload tiles4k --steps 50 --gather $g --no-cpu-baseline 2> gpurun_out/s7_t4k_$g.err | python3 tools/short.py "N=1 $g" || { tail -5 gpurun_out/s7_t4k_$g.err; exit 1; }
done
for g in hits16 bgra8; do for pl in 2 0; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --share-gpu --gather $g --pipeline $pl --no-cpu-baseline > gpurun_out/s7_n2_${g}_$pl.json 2> gpurun_out/s7_n2_${g}_$pl.err || { tail -8 gpurun_out/s7_n2_${g}_$pl.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/s7_n2_${g}_$pl.json').read().strip().splitlines()[-1]); print('N=2 gloo $g pipeline $pl:', d['value'], d['ms_per_step'], {k:d.get(k) for k in ('trace_only_ms','gather_and_scatter_ms','gather_bytes_into_rank0','framebuffer_equals_one_gpu','frames_in_flight','streams','same_steps_on_one_stream','same_frame_gathered_as')}, d['one_gpu_same_frame'])"
done; done
