"""What one rank of an N-GPU tile split spends per frame, measured on ONE GPU: rank r of N renders strips r, r + N, ... of the frame
(rt_render_strips_device: classify + order + render + fold + finish, all of it) alone on the device, as it would on its own GPU.
Prints, per workload and N, the WHOLE step per rank (wall clock around the call + synchronise, best of 3, every rank), the render
kernel alone (HIP events, rt_scene_profile_*), hence the fixed cost per launch (step - kernel), and the strong-scaling efficiency the
slowest rank allows before the composite (one frame-sized exchange over xGMI, not emulated here).
   usage: python tools/emulate_ranks.py [--strips] [workload ...]     (bench.py workload names; default: the target scene, mesh_demo at 64 spp, the headline)
   The partition is bench.py's: 8-pixel tile columns dealt round-robin (rt_render_tile_columns_device); --strips: round 2's 8-row strips."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import WORKLOADS
from raytracer_amd import CameraConfiguration, RenderSettings, Scene, SceneBuilder, strip_rows, tile_columns_width

stream = torch.cuda.current_stream().cuda_stream
strips = "--strips" in sys.argv
for workload in ([a for a in sys.argv[1:] if a != "--strips"] or ["dreadnaught-1080p-128spp", "meshdemo-4k-64spp", "weekend-1080p-256spp"]):
    name, W, H, spp, depth, lens, params = WORKLOADS[workload]
    sc = Scene(SceneBuilder.sample(name, CameraConfiguration(W, H, lens_radius=lens), params=params), 0)
    st = RenderSettings(W, H, depth, spp)
    one = None
    for world in (1, 2, 4, 8):
        steps, kernels = [], []
        for r in range(world):
            rows = strip_rows(H, 8, r, world) if strips else H
            buf = torch.zeros((rows, W if strips else max(tile_columns_width(W, r, world), 1), 3), dtype=torch.float32, device="cuda:0")
            best, best_k = None, None
            for it in range(5):
                profiled = it == 4                # (the event pair around the kernel costs a few microseconds of its own: the step is timed without it)
                sc.profile_enable(profiled); sc.profile_read()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                if strips:
                    sc.render_strips_device(st, 8, r, world, 12345, buf.data_ptr(), None, None, stream)
                else:
                    sc.render_tile_columns_device(st, r, world, 12345, buf.data_ptr(), None, None, stream)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
                if it == 0:
                    continue                      # (first call of a shape: allocations)
                if profiled:
                    best_k = sc.profile_read()[0]
                elif best is None or dt < best:
                    best = dt
            steps.append(best); kernels.append(best_k)
        slow = max(steps)
        if os.environ.get("EMULATE_VERBOSE"):
            print("   per rank, kernel ms:", " ".join(f"{k:.2f}" for k in kernels), flush=True)
        if world == 1:
            one = slow
        fixed = [s - k for s, k in zip(steps, kernels)]
        print(f"{workload} {'strips' if strips else 'columns'} N={world}: step per rank {min(steps):.3f}..{slow:.3f} ms (ideal {one / world:.3f}), render kernel {min(kernels):.3f}..{max(kernels):.3f} ms, "
              f"everything else (classify, order, fold, finish, launches) {min(fixed):.3f}..{max(fixed):.3f} ms; efficiency before the composite {one / world / slow * 100:.1f} %",
              flush=True)
    sc.close()
