"""One rank's share of an N-GPU tile-column split, three times: for `rocprofv3 --kernel-trace --stats -- python tools/rank_trace.py <workload> <N> [rank]`
(which kernels a rank's step consists of, and how long each takes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS
from raytracer_amd import CameraConfiguration, RenderSettings, Scene, SceneBuilder, tile_columns_width
wl, world = sys.argv[1], int(sys.argv[2]); r = int(sys.argv[3]) if len(sys.argv) > 3 else 0
name, W, H, spp, depth, lens, params = WORKLOADS[wl]
sc = Scene(SceneBuilder.sample(name, CameraConfiguration(W, H, lens_radius=lens), params=params), 0)
st = RenderSettings(W, H, depth, spp)
buf = torch.zeros((H, max(tile_columns_width(W, r, world), 1), 3), dtype=torch.float32, device="cuda:0")
for _ in range(3):
    sc.render_tile_columns_device(st, r, world, 12345, buf.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
sc.close()
