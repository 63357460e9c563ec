#!/bin/bash
# A/B of one environment switch of librt_hip on the GPU box: tools/ab_env.sh VAR v1 v2 ... [-- workload ...]
# Prints value, workload, Mrays/s and ms per frame.
set -e
var=$1; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
wl=("$@"); [ ${#wl[@]} -eq 0 ] && wl=(weekend-1080p-256spp dreadnaught-1080p-128spp dootdoot-4k-64spp)
mkdir -p gpurun_out
for v in "${vals[@]}"; do
  for w in "${wl[@]}"; do
    export $var=$v
    timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-target-scene --no-extra-configs --steps 3 --warmup 1 > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
    python -c "import json; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$var=$v', d['config']['workload'], d['value'], d['ms_per_step'])"
  done
done
