import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tests.test_gpu_fuzz as t
from raytracer_amd import Scene, RenderSettings, RenderChunk, sample_segment
from tests.oracle import OracleScene, ORC_RNG_COUNTER_STRATIFIED
for seed in [int(a) for a in sys.argv[1:]]:
    b = t.random_scene(1000 + seed); s, o = Scene(b, 0), OracleScene(b)
    ch = RenderChunk(0, 0, t.W, t.H)
    for depth, spp in ((5, 3), (1, 3), (2, 3), (5, 1)):
        st = RenderSettings(t.W, t.H, depth, spp)
        rgb, _, cn = s.render_chunk(st, ch, seed=seed)
        ref, _, rcn = o.render_chunk(st, ch, ORC_RNG_COUNTER_STRATIFIED, seed, sum_segment=sample_segment())
        d = np.abs(rgb.astype(np.float64) - ref).max(axis=2)
        bad = np.argwhere(d > 2e-5)
        print(f"seed {seed} depth {depth} spp {spp}: mean {np.abs(rgb.astype(np.float64)-ref).mean():.2e} max {d.max():.3e} outliers {len(bad)} rays gpu {cn.primary}/{cn.secondary}/{cn.shadow} ref {rcn.rays.primary}/{rcn.rays.secondary}/{rcn.rays.shadow}")
        for y, x in bad[:6]:
            print(f"   px({x},{y}) gpu {rgb[y,x]} ref {ref[y,x]}")
