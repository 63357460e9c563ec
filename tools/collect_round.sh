#!/bin/bash
# Everything profiles/ holds for one round, in one GPU call: tools/collect_round.sh <tag>   (e.g. r02_v20)
#   gpurun_out/<tag>/stats      rocprofv3 --kernel-trace --stats of the default bench command
#   gpurun_out/<tag>/pmc*       PMC passes (one rocprofv3 --pmc run each, never combined with a trace) -> <tag>_pmc_<workload>.json
#   gpurun_out/<tag>/bench_*    the bench line of every BASELINE config
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-configs > $OUT/stats.log 2>&1
echo "stats ok"
for w in weekend-1080p-256spp dreadnaught-1080p-128spp meshdemo-4k-64spp dootdoot-4k-64spp; do
  rm -rf $OUT/pmc_$w
  tools/pmc_passes.sh $OUT/pmc_$w "sq1 sq2 sq3 tcp tcc1 tcc2 ta ta2 grbm" --workload $w --no-target-scene --no-extra-configs
  python tools/pmc_summary.py $OUT/pmc_$w render_kernel > $OUT/pmc_$w.json
done
for w in weekend-1080p-256spp weekend-400x225-16spp dreadnaught-1080p-128spp dootdoot-4k-64spp meshdemo-4k-64spp meshdemo-4k-1024spp; do
  timeout -k 10 500 python bench.py --workload $w --steps 3 --warmup 1 --no-target-scene --no-extra-configs > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python -c "import json; d=json.loads(open('$OUT/bench_$w.json').read().strip().splitlines()[-1]); print(d['config']['workload'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['gpu_over_cpu'], d['roofline']['traversal_mismatch'])"
done
