"""How much of the lane divergence is depth of field?  Renders a scene with the UI's lens radius and with a pinhole.
   usage: python tools/lens_probe.py [scene] [W H spp]   (RT_HIP_LIB=raytracer_amd/librt_hip_stats.so prints phase utilisation)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import SceneBuilder, CameraConfiguration, Scene, RenderSettings, RenderChunk
name = sys.argv[1] if len(sys.argv) > 1 else "Mesh"
W, H, spp = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (3840, 2160, 64)
stream = torch.cuda.current_stream().cuda_stream
for lens in (0.1, 0.0):
    b = SceneBuilder.sample(name, CameraConfiguration(W, H, lens_radius=lens))
    sc = Scene(b, 0); st = RenderSettings(W, H, 8, spp)
    out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0"); cn = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    for it in range(2):
        cn.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        sc.render_chunk_device(st, RenderChunk(0, 0, W, H), 12345, out.data_ptr(), None, W, 0, 0, cn.data_ptr(), stream)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rays = int(cn[:3].sum().item())
    print(f"{name} lens_radius {lens}: {dt * 1e3:.1f} ms, {rays / dt / 1e6:.0f} Mrays/s, rays {cn.tolist()}", flush=True)
    sc.close()
