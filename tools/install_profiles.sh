#!/bin/bash
# Copies one collection (tools/collect_part.sh <tag> ...) from gpurun_out/<tag> into profiles/ and regenerates the counter summaries:
#   tools/install_profiles.sh <tag> [older tag whose files it replaces]
set -e
T=$1; OLD=$2; O=gpurun_out/$T
cd "$(dirname "$0")/.."
if [ -n "$OLD" ]; then rm -f profiles/${OLD}_* profiles/bench/bench_r03_default_${OLD#r03_}.json; fi
cp $O/stats/vm/*_kernel_stats.csv profiles/${T}_kernel_stats.csv
python - "$T" <<'PY'
import csv, glob, sys
t = sys.argv[1]
f = glob.glob(f'gpurun_out/{t}/stats/vm/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if any(k in r['Kernel_Name'] for k in ('render_kernel', 'classify_kernel', 'empty_tiles_kernel', 'order_kernel', 'fold_kernel', 'finish_kernel'))]
w = csv.DictWriter(open(f'profiles/{t}_kernel_trace_render.csv', 'w', newline=''), fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
PY
for w in weekend-1080p-256spp dreadnaught-1080p-128spp meshdemo-4k-64spp dootdoot-4k-64spp meshdemo-4k-1024spp; do [ -f $O/pmc_$w.json ] && cp $O/pmc_$w.json profiles/${T}_pmc_$w.json; done
for w in weekend-1080p-256spp weekend-400x225-16spp dreadnaught-1080p-128spp dootdoot-4k-64spp meshdemo-4k-64spp meshdemo-4k-1024spp; do cp $O/bench_$w.json profiles/bench/bench_r03_$w.json; done
[ -f $O/bench_default.json ] && cp $O/bench_default.json profiles/bench/bench_r03_default_${T#r03_}.json
python tools/pmc_counters.py $O $T > /dev/null
