#!/bin/bash
# One-off: the C++ host harness (rt_render = the reference's render job over the C ABI) must write the same bytes for every
# chunk grid and worker-thread count.   usage (GPU box): bash tools/harness_sweep.sh
set -e
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/harness
fail=0
for scene in "Random Spheres" "Mesh" "Fleet" "Planes"; do
  tag=$(echo "$scene" | tr ' ' '_')
  ./raytracer_amd/rt_render --scene "$scene" --width 333 --height 187 --spp 9 --depth 6 --chunks 1x1 --threads 1 --out gpurun_out/harness/$tag.ref.ppm > /dev/null
  for cfg in "2x2 1" "5x3 3" "8x8 4" "13x7 8" "32x32 16" "1x9 2"; do
    set -- $cfg
    ./raytracer_amd/rt_render --scene "$scene" --width 333 --height 187 --spp 9 --depth 6 --chunks $1 --threads $2 --out gpurun_out/harness/$tag.$1.$2.ppm > /dev/null
    if cmp -s gpurun_out/harness/$tag.ref.ppm gpurun_out/harness/$tag.$1.$2.ppm; then echo "$scene chunks $1 threads $2: identical"; else echo "$scene chunks $1 threads $2: DIFFERENT"; fail=1; fi
  done
done
rm -rf gpurun_out/harness
exit $fail
