"""One-off sweep of rt_scene_update_transforms (bit-exact gate) and rt_scene_rebuild_tlas (nearest-hit gate) over random
instance scenes of random sizes.   usage: python tools/refit_sweep.py <first seed> <count>"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import RenderChunk, RenderSettings, Scene
from tests.test_gpu_refit import random_instances, same_hits
from tests.test_gpu_rebuild import nearest_hit_gate

first, count = int(sys.argv[1]), int(sys.argv[2])
bad, t0, worst = [], time.perf_counter(), 0.0
w, h = 160, 90
st1, st, whole = RenderSettings(w, h, 8, 1), RenderSettings(w, h, 4, 2), RenderChunk(0, 0, w, h)
for k, seed in enumerate(range(first, first + count)):
    try:
        r = np.random.default_rng(seed)
        n = int(r.choice([1, 2, 3, 4, 7, 16, 50, 150, 400]))
        a, b = random_instances(10 * seed + 1, n=n), random_instances(10 * seed + 2, n=n)
        refit, rebuilt, fresh = Scene(a, 0), Scene(a, 0), Scene(b, 0)
        refit.update_transforms(b); rebuilt.rebuild_tlas(b)
        want = want_b = fresh.primary_hits(st1, whole)
        assert same_hits(refit.primary_hits(st1, whole), want), f"seed {seed} n {n}: refit hits"
        worst = max(worst, nearest_hit_gate(rebuilt.primary_hits(st1, whole), want, f"seed {seed} n {n}: rebuilt hits"))
        f = fresh.render_chunk(st, whole, seed=3)[0]
        assert np.array_equal(refit.render_chunk(st, whole, seed=3)[0], f), f"seed {seed} n {n}: refit image"
        assert (np.abs(rebuilt.render_chunk(st, whole, seed=3)[0] - f).max(axis=-1) > 2e-5).mean() <= 5e-3, f"seed {seed} n {n}: rebuilt image"
        rebuilt.rebuild_tlas(a); refit.update_transforms(a)          # and back
        back = Scene(a, 0)
        want = back.primary_hits(st1, whole)
        assert same_hits(refit.primary_hits(st1, whole), want), f"seed {seed} n {n}: refit back"
        worst = max(worst, nearest_hit_gate(rebuilt.primary_hits(st1, whole), want, f"seed {seed} n {n}: rebuilt back"))
        rebuilt.update_transforms(b)                                  # a refit of the device-built tree
        worst = max(worst, nearest_hit_gate(rebuilt.primary_hits(st1, whole), want_b, f"seed {seed} n {n}: refit of the built tree"))
        for s in (refit, rebuilt, fresh, back):
            s.close()
    except AssertionError as e:
        bad.append(seed); print("FAILED", str(e)[:300].replace("\n", " "), flush=True)
    except Exception:
        bad.append(seed); print("ERROR seed", seed, traceback.format_exc().splitlines()[-4:], flush=True)
    if (k + 1) % 20 == 0:
        print(f"{k + 1} cases, {len(bad)} failures, worst rebuilt difference {worst * 100:.4f} % of the pixels, {time.perf_counter() - t0:.1f} s", flush=True)
print("done:", count, "cases, failures:", bad, flush=True)
sys.exit(1 if bad else 0)
