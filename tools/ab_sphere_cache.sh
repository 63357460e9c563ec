#!/bin/bash
# A/B of the LDS sphere cache on the GPU box (RT_NO_SPHERE_CACHE=1 turns it off): workload, Mrays/s, ms per frame, and the traits line of the launch
run() { timeout -k 10 200 python bench.py --workload $1 --no-cpu-baseline --no-target-scene --no-extra-configs --steps 5 --warmup 2 > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
  grep traits gpurun_out/ab.err | head -1
  python -c "import json; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$2', d['config']['workload'], d['value'], d['ms_per_step'])"; }
export RT_DEBUG_TRAITS=1
for i in 1 2; do
  for w in ${AB_WORKLOADS:-weekend-1080p-256spp weekend-400x225-16spp}; do
    run $w cache_on
    RT_NO_SPHERE_CACHE=1 run $w cache_off
  done
done
