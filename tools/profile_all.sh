set -o pipefail
tools/profile_gpu.sh dragon1080 > gpurun_out/s7_prof_dragon1080.log 2>&1 || exit 1
tools/profile_gpu.sh bunny1080 --workload bunny1080 > gpurun_out/s7_prof_bunny1080.log 2>&1 || exit 1
tools/profile_gpu.sh dragon4k --workload dragon4k > gpurun_out/s7_prof_dragon4k.log 2>&1 || exit 1
tools/profile_gpu.sh ao64m --workload ao --steps 6 --warmup 1 > gpurun_out/s7_prof_ao64m.log 2>&1 || exit 1
echo all-done
