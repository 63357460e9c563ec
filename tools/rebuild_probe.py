"""Frame time of a bench workload on the host-built top-level tree (the reference's leaves) and on the device-built one
(rt_scene_rebuild_tlas in place).   usage: python tools/rebuild_probe.py <bench workload> [spp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS
from raytracer_amd import SceneBuilder, CameraConfiguration, Scene, RenderSettings, RenderChunk
name, W, H, spp, depth, lens, params = WORKLOADS[sys.argv[1]]
if len(sys.argv) > 2:
    spp = int(sys.argv[2])
b = SceneBuilder.sample(name, CameraConfiguration(W, H, lens_radius=lens), params=params)
sc = Scene(b, 0)
st = RenderSettings(W, H, depth, spp)
out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0"); cn = torch.zeros(4, dtype=torch.int64, device="cuda:0")
def frame():
    best = 1e9
    for _ in range(4):
        cn.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        sc.render_chunk_device(st, RenderChunk(0, 0, W, H), 12345, out.data_ptr(), None, W, 0, 0, cn.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best, int(cn[:3].sum().item()), out.clone()
t0, r0, i0 = frame()
t_rb = time.perf_counter(); sc.rebuild_tlas(b); t_rb = time.perf_counter() - t_rb
t1, r1, i1 = frame()
bad = float(((i0 - i1).abs().amax(dim=-1) > 2e-5).float().mean())
print(f"{sys.argv[1]} {spp} spp: host-built {t0 * 1e3:.2f} ms ({r0 / t0 / 1e6:.0f} Mrays/s), device-built {t1 * 1e3:.2f} ms ({r1 / t1 / 1e6:.0f} Mrays/s), "
      f"first rebuild call {t_rb * 1e3:.1f} ms, depth {sc.info().tlas_depth}, pixels off by > 2e-5: {bad * 100:.4f} %, rays {r0} / {r1}", flush=True)
