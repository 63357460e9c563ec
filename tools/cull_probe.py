"""Empty-tile packet test on / off on whole frames: ms per frame, rays, pixels in empty tiles.   usage: python tools/cull_probe.py <bench workload> ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS
from raytracer_amd import SceneBuilder, CameraConfiguration, Scene, RenderSettings, RenderChunk
stream = torch.cuda.current_stream().cuda_stream
for wl in sys.argv[1:]:
    name, W, H, spp, depth, lens, params = WORKLOADS[wl]
    sc = Scene(SceneBuilder.sample(name, CameraConfiguration(W, H, lens_radius=lens), params=params), 0)
    st = RenderSettings(W, H, depth, spp)
    out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0"); cn = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    res = {}
    for cull in (1, 0, 1, 0):
        sc.set_option("cull_empty_tiles", cull)
        best = None
        for it in range(4):
            cn.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
            sc.render_chunk_device(st, RenderChunk(0, 0, W, H), 12345, out.data_ptr(), None, W, 0, 0, cn.data_ptr(), stream)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
            if it and (best is None or dt < best): best = dt
        res.setdefault(cull, []).append((best, cn.tolist(), sc.culled_pixels(), float(out.sum())))
    for cull in (1, 0):
        print(wl, "cull", cull, " ".join(f"{b:.3f} ms" for b, *_ in res[cull]), "rays", res[cull][0][1], "culled px", res[cull][0][2], f"{res[cull][0][2] / (W * H) * 100:.1f} %", "sum", res[cull][0][3], flush=True)
    sc.close()
