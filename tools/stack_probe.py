"""Smallest traversal stack that a frame of a bench workload does not overflow (RT_OPT_DEBUG_STACK_ENTRIES), next to the
bound the scene build sizes the LDS stacks for.   usage: python tools/stack_probe.py <bench workload> [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS
from raytracer_amd import SceneBuilder, CameraConfiguration, Scene, RenderSettings, RenderChunk
name, W, H, spp, depth, lens, params = WORKLOADS[sys.argv[1]]
if len(sys.argv) > 2:
    spp = int(sys.argv[2])
sc = Scene(SceneBuilder.sample(name, CameraConfiguration(W, H, lens_radius=lens), params=params), 0)
st = RenderSettings(W, H, depth, spp)
out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
inf = sc.info()
print(sys.argv[1], "info:", {k: getattr(inf, k) for k, _ in inf._fields_}, flush=True)
ok_at = None
for n in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 20, 24):
    sc.set_option("debug_stack_entries", 0)
    sc.set_option("debug_stack_entries", n)
    try:
        sc.render_chunk_device(st, RenderChunk(0, 0, W, H), 12345, out.data_ptr(), None, W, 0, 0, None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        # an overflow inside an enqueued launch surfaces at the next call on the scene
        sc.render_chunk_device(RenderSettings(W, H, depth, 1), RenderChunk(0, 0, 8, 8), 12345, out.data_ptr(), None, W, 0, 0, None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ok_at = n
        break
    except Exception as e:
        print("  entries", n, "->", str(e)[:60], flush=True)
print(sys.argv[1], "smallest stack without overflow <=", ok_at, flush=True)
