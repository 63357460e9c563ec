set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/bench10 gpurun_out/prof_v10
timeout -k 10 200 python tools/emulate_ranks.py
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v10 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_v10/bench.log 2>&1
echo prof ok
rm -rf gpurun_out/pmc_v10; tools/pmc_passes.sh gpurun_out/pmc_v10 "sq1 sq2 sq3 tcp tcc1 tcc2 grbm"
python tools/pmc_summary.py gpurun_out/pmc_v10 render_kernel > gpurun_out/pmc_v10.json
for w in weekend-1080p-256spp weekend-400x225-16spp dreadnaught-1080p-128spp dootdoot-4k-64spp meshdemo-4k-1024spp; do
  timeout -k 10 500 python bench.py --workload $w --steps 3 --warmup 1 > gpurun_out/bench10/bench_r01_$w.json 2> gpurun_out/bench10/$w.err
  python -c "import json; d=json.loads(open('gpurun_out/bench10/bench_r01_$w.json').read().strip().splitlines()[-1]); print(d['config']['workload'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['gpu_over_cpu'])"
done
