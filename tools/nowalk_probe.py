import sys; sys.path.insert(0, '/root/repo')
from bench import WORKLOADS
from raytracer_amd import SceneBuilder, CameraConfiguration, Scene, RenderSettings, RenderChunk
for wl in ("weekend-1080p-256spp",):
    name, W, H, spp, depth, lens, params = WORKLOADS[wl]
    sc = Scene(SceneBuilder.sample(name, CameraConfiguration(W, H, lens_radius=lens), params=params), 0)
    sc.render_chunk(RenderSettings(W, H, depth, 8), RenderChunk(0, 0, W, H), want_rgba=False)
    a, b = sc.culled_pixels(both=True)
    print(wl, "left out", a, "without walk", b, f"{b / (W * H) * 100:.1f} %")
