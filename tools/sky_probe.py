"""What the samples cost whose camera rays miss the scene's root box: the frame against the chunk that bounds every pixel with a primary hit
(rounded out to whole 8 x 8 tiles; the chunk still holds sky around the silhouette, so the difference is a LOWER bound of what a packet test
of sky tiles against the root box could save).   usage: python tools/sky_probe.py <bench workload> ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import WORKLOADS
from raytracer_amd import SceneBuilder, CameraConfiguration, Scene, RenderSettings, RenderChunk
stream = torch.cuda.current_stream().cuda_stream
for wl in sys.argv[1:]:
    name, W, H, spp, depth, lens, params = WORKLOADS[wl]
    sc = Scene(SceneBuilder.sample(name, CameraConfiguration(W, H, lens_radius=lens), params=params), 0)
    st = RenderSettings(W, H, depth, spp)
    t, n, e, tri = sc.primary_hits(st, RenderChunk(0, 0, W, H))
    ys, xs = np.nonzero(e >= 0)
    x0, x1, y0, y1 = xs.min() // 8 * 8, min(W, (xs.max() // 8 + 1) * 8), ys.min() // 8 * 8, min(H, (ys.max() // 8 + 1) * 8)
    tiles_hit = len(set(zip((ys // 8).tolist(), (xs // 8).tolist())))
    out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0"); cn = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    def run(ch):
        best = None
        for it in range(4):
            cn.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
            sc.render_chunk_device(st, ch, 12345, out.data_ptr(), None, W, ch.left, ch.top, cn.data_ptr(), stream)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
            if it and (best is None or dt < best): best = dt
        return best, cn.tolist()
    full, cf = run(RenderChunk(0, 0, W, H)); part, cp = run(RenderChunk(int(x0), int(y0), int(x1 - x0), int(y1 - y0)))
    print(f"{wl}: pixels with a primary hit {len(xs) / (W * H) * 100:.1f} %, tiles with one {tiles_hit * 64 / (W * H) * 100:.1f} %, bounding chunk {(x1 - x0) * (y1 - y0) / (W * H) * 100:.1f} % of the frame; "
          f"frame {full:.2f} ms, bounding chunk {part:.2f} ms ({(1 - part / full) * 100:.1f} % less); rays {sum(cf[:3])} / {sum(cp[:3])}", flush=True)
    sc.close()
