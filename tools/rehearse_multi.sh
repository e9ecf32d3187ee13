# Rehearsal of the N > 1 flow of `bench.py --workload tiles4k` on ONE GPU: two processes share GPU 0, the records cross through
# host memory (gloo) instead of RCCL, everything else (plan from measured costs, compact outputs, frames in flight, scatter,
# re-deal) is the code an N-GPU job runs.  Proves byte-equality of the gathered framebuffer; its times mean nothing.
set -o pipefail
mkdir -p gpurun_out/rehearse
for g in bgra8 hits8; do for pl in 4 0; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --share-gpu --gather $g --pipeline $pl --redeal 8 --no-cpu-baseline > gpurun_out/rehearse/n2_${g}_$pl.json 2> gpurun_out/rehearse/n2_${g}_$pl.err || { tail -12 gpurun_out/rehearse/n2_${g}_$pl.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/rehearse/n2_${g}_$pl.json').read().strip().splitlines()[-1]); print('N=2 gloo $g pipeline $pl:', d['value'], d['ms_per_step'], {k:d.get(k) for k in ('trace_only_ms','gather_and_scatter_ms','gather_bytes_into_rank0','framebuffer_equals_one_gpu','frames_in_flight','streams','root_head_start','redeal','tile_assignment')}, d['one_gpu_same_frame'])"
done; done
