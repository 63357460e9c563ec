import sys, time; sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import SceneBuilder, CameraConfiguration, Scene, RenderSettings, RenderChunk
from tests.oracle import OracleScene
n = 4_000_000
W, H = 1920, 1080
b = SceneBuilder.sample("Point Cloud", CameraConfiguration(W, H, lens_radius=0.0), params={"Cloud Point Count": n})
t0 = time.time(); s = Scene(b, 0); print("scene create %.1fs" % (time.time() - t0), s.info().tlas_depth, s.info().device_bytes / 1e6, "MB")
st = RenderSettings(W, H, 5, 16)
for i in range(2):
    t0 = time.time(); rgb, _, cn = s.render_chunk(st, RenderChunk(0, 0, W, H), want_rgba=False); dt = time.time() - t0
    print("render %.1f ms, %.0f Mrays/s" % (dt * 1e3, cn.rays / dt / 1e6))
o = OracleScene(b)
ch = RenderChunk(900, 500, 64, 48); st1 = RenderSettings(W, H, 5, 1)
t, nrm, e, tri = s.primary_hits(st1, ch); rt, rn, re, rtri = o.primary_hits(st1, ch)
print("primary ids equal", np.array_equal(e, re), "t bit-identical", float((t == rt).mean()), "hit frac", float((re >= 0).mean()))
