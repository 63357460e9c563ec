#!/bin/bash
# A/B of several builds of librt_hip on the GPU box: tools/ab_libs.sh <lib>... (paths relative to the repo root)
# Prints workload, Mrays/s and ms per frame for each library.
set -e
for lib in "$@"; do
  for w in ${AB_WORKLOADS:-weekend-1080p-256spp dreadnaught-1080p-128spp meshdemo-4k-64spp dootdoot-4k-64spp}; do
    RT_HIP_LIB=$lib timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 2 --no-target-scene --no-extra-configs > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
    python -c "import json; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$lib', d['config']['workload'], d['value'], d['ms_per_step'])"
  done
done
