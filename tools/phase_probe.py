"""Per-phase lane statistics of one frame (diagnostics build: `make -C raytracer_amd/csrc stats`).
   usage: RT_HIP_LIB=raytracer_amd/librt_hip_stats.so python tools/phase_probe.py <bench workload> [spp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS
from raytracer_amd import SceneBuilder, CameraConfiguration, Scene, RenderSettings, RenderChunk
name, W, H, spp, depth, lens, params = WORKLOADS[sys.argv[1]]
if len(sys.argv) > 2:
    spp = int(sys.argv[2])
sc = Scene(SceneBuilder.sample(name, CameraConfiguration(W, H, lens_radius=lens), params=params), 0)
st = RenderSettings(W, H, depth, spp)
out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0"); cn = torch.zeros(4, dtype=torch.int64, device="cuda:0")
t0 = time.perf_counter()
sc.render_chunk_device(st, RenderChunk(0, 0, W, H), 12345, out.data_ptr(), None, W, 0, 0, cn.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(sys.argv[1], spp, "spp:", f"{(time.perf_counter() - t0) * 1e3:.1f} ms", "rays", cn.tolist(), flush=True)
