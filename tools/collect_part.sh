#!/bin/bash
# One part of tools/collect_round.sh (a whole round does not fit one 20-minute GPU call): tools/collect_part.sh <tag> stats|pmc|bench [workload ...]
set -e
TAG=$1; PART=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
case $PART in
  stats)
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-configs > $OUT/stats.log 2>&1
    echo "stats ok";;
  pmc)
    for w in "$@"; do
      rm -rf $OUT/pmc_$w
      tools/pmc_passes.sh $OUT/pmc_$w "sq1 sq2 sq3 tcp tcc1 tcc2 ta ta2 grbm" --workload $w --no-target-scene --no-extra-configs
      python tools/pmc_summary.py $OUT/pmc_$w render_kernel > $OUT/pmc_$w.json
    done;;
  bench)
    for w in "$@"; do
      timeout -k 10 500 python bench.py --workload $w --steps 3 --warmup 1 --no-target-scene --no-extra-configs > $OUT/bench_$w.json 2> $OUT/bench_$w.err
      python -c "import json; d=json.loads(open('$OUT/bench_$w.json').read().strip().splitlines()[-1]); print(d['config']['workload'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_traced'], d['cpu_baseline']['value'], d['gpu_over_cpu'], d['roofline']['traversal_mismatch'])"
    done;;
esac
