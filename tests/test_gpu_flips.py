"""What the image tolerances of this suite rest on, as a checked statement.

Primary hits are bit-identical to the oracle (tests/test_gpu_parity.py, test_gpu_fuzz.py).  An IMAGE can still differ in single pixels:
sin / cos / pow / acos come from different libraries on the two sides (glibc against the ROCm device library, <= 2 ulp apart), their
results feed scatter directions, checker / colour-map lookups and lamp cones, and a last-ulp difference can flip a decision of a
SECONDARY ray - hit or miss, this texel or the next, inside or outside a cone.  A flipped sample moves its pixel by up to 1 / spp.
Model, from the one-off sweeps of round 2 (tools/fuzz_sweep.py: 3 000 scenes at depth 5 x 3 spp; tools/invariant_sweep.py: 1 000 cases
at depth 0-8 x 1-64 spp; re-run in round 3 after the sampler change with the same outcome: every bit-identity holds in all of them, the same 9 + 4 cases
exceed the per-test pixel bounds; their logs are not part of the repository): every sample and every level of recursion is one more chance, so

    fraction of pixels beyond 2e-5  <=  FLIP_RATE x max_reflections x samples_per_pixel,   FLIP_RATE = 1e-3,
    and 0 at max_reflections = 1, where no secondary ray exists (round 2's notes said "depth 1 and 2 are exact": depth 2 is not - this
    test found 14 and 3 differing pixels on two of the scenes: a secondary ray exists there, its children end in the sky).

Worst observed: 6.1e-4 per depth x sample (fuzz seed 1503 at depth 2 x 3 spp: 14 of 3 840 pixels); at depth 7 x 64 spp 2.7e-4 (seed 1305
of the second sweep: 12 % of the pixels, checker lookups of secondary hits).  This file re-runs exactly the cases the sweeps flagged - the nine scenes of the first (they exceeded the suite's
5e-3 bound by 5-7 pixels of 3 840) and the two of the second - and holds them to the model; the per-test outlier fractions elsewhere
(2e-3 ... 5e-3 at depth x spp <= 32) are below it (C1 at its own size, tests/test_gpu_parity.py: 0.5 % asserted where the model allows 12.8 %)."""
import numpy as np
import pytest

from raytracer_amd import RenderChunk, RenderSettings, Scene, sample_segment
from tests.oracle import ORC_RNG_COUNTER_STRATIFIED, OracleScene, use_lens_mode
from tests.test_gpu_fuzz import H, W, random_scene

pytestmark = pytest.mark.gpu

FLIP_RATE = 1e-3
ATOL = 2e-5


def flip_bound(depth, spp):
    return min(1.0, FLIP_RATE * depth * spp)


def differing(scene, oracle, st, chunk, rng_mode, seed):
    rgb, _, cn = scene.render_chunk(st, chunk, seed=seed)
    ref, _, rcn = oracle.render_chunk(st, chunk, rng_mode, seed, sum_segment=sample_segment())
    assert cn.primary == rcn.rays.primary
    d = np.abs(rgb.astype(np.float64) - ref).max(axis=-1)
    return int((d > ATOL).sum()), float(d.max())


@pytest.mark.parametrize("seed", [385, 811, 927, 1503, 1823, 1875, 2322, 2486, 2812,
                                  50077])      # (seeds 50000-50599, re-entry session of round 3, final library: the one failure of 600)
def test_scenes_the_fuzz_sweep_flagged(seed):
    """tools/fuzz_sweep.py's failures among seeds 0..2999 (tests/test_gpu_fuzz.py's gate: 80 x 48, depth 5, 3 spp, render seed = seed)."""
    b = random_scene(1000 + seed)
    scene, oracle = Scene(b, 0), OracleScene(b)
    whole = RenderChunk(0, 0, W, H)
    n, worst = differing(scene, oracle, RenderSettings(W, H, 1, 3), whole, ORC_RNG_COUNTER_STRATIFIED, seed)
    assert n == 0, (seed, n, worst)                          # depth 1: no secondary ray, no decision left to flip - every pixel within 2e-5
    n2, _ = differing(scene, oracle, RenderSettings(W, H, 2, 3), whole, ORC_RNG_COUNTER_STRATIFIED, seed)
    assert n2 <= flip_bound(2, 3) * W * H, (seed, n2)
    n, worst = differing(scene, oracle, RenderSettings(W, H, 5, 3), whole, ORC_RNG_COUNTER_STRATIFIED, seed)
    print(f"seed {seed}: {n} of {W * H} pixels beyond {ATOL} at depth 5 x 3 spp (bound {flip_bound(5, 3) * W * H:.0f}), largest difference {worst:.3f}; depth 2: {n2}")
    assert 0 < n <= flip_bound(5, 3) * W * H, (seed, n)     # (> 0: these are the scenes that DO flip; if one stops, the list above is stale)
    assert worst <= 1.0 / 3 * 2 + ATOL                       # at most two flipped samples of three in one pixel, each worth <= 1 / spp
    scene.close()


def sweep_case(seed):
    """The case tools/invariant_sweep.py derives from `seed` (same draws in the same order; the schedule options it draws never change a bit)."""
    r = np.random.default_rng(seed)
    b = random_scene(5000 + seed)
    w, h = int(r.integers(9, 150)), int(r.integers(9, 100))
    spp, depth = int(r.choice([1, 2, 3, 7, 8, 9, 16, 17, 24, 33, 64])), int(r.integers(0, 9))
    lens_mode = "reference" if r.integers(0, 2) else "stratified"
    if r.integers(0, 3) == 0: r.choice([-1, 1])
    if r.integers(0, 3) == 0: r.choice([1, 2, 4, 8])
    if r.integers(0, 3) == 0: r.choice([0.0, 1.0, 1e9])
    return b, w, h, spp, depth, lens_mode, int(r.integers(0, 2**31))


@pytest.mark.parametrize("seed,expect", [(1234, (147, 97, 64, 3, "stratified")), (1305, (93, 99, 64, 7, "stratified")),      # round 2's second sweep (seeds 1000-1399)
                                         (183, (145, 95, 33, 8, "stratified")), (198, (65, 81, 64, 7, "reference")),          # seeds 0-799, re-run in round 3 with the
                                         (256, (71, 74, 17, 8, "stratified")), (275, (40, 93, 64, 7, "reference")),          # 8 x 8 lens cells: the same four as in round 2
                                         (40108, (113, 70, 24, 8, "reference")), (40377, (73, 62, 33, 8, "reference"))])     # seeds 40000-40399 on the final library (empty tiles, single-round claims): the same with those switched off
def test_cases_the_invariant_sweep_flagged(seed, expect):
    b, w, h, spp, depth, lens_mode, render_seed = sweep_case(seed)
    assert (w, h, spp, depth, lens_mode) == expect
    scene, oracle = Scene(b, 0), OracleScene(b)
    rng_mode = use_lens_mode(scene, lens_mode)
    whole = RenderChunk(0, 0, w, h)
    n1, _ = differing(scene, oracle, RenderSettings(w, h, 1, spp), whole, rng_mode, render_seed)
    assert n1 <= flip_bound(1, spp) * w * h
    n, worst = differing(scene, oracle, RenderSettings(w, h, depth, spp), whole, rng_mode, render_seed)
    print(f"seed {seed}: {w}x{h}, depth {depth} x {spp} spp: {n} of {w * h} pixels beyond {ATOL} ({n / (w * h) / (depth * spp):.2e} per depth x sample; "
          f"bound {FLIP_RATE:.0e}), largest difference {worst:.4f}; depth 1: {n1}")
    assert n <= flip_bound(depth, spp) * w * h
    assert worst <= 8.0 / spp + ATOL                          # a handful of flipped samples of 64 in the worst pixel
    scene.close()
