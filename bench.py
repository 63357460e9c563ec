#!/usr/bin/env python3
"""Benchmark of the hot path: Mrays/s of the render core on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1 is launched by torch.distributed.run, one rank per GPU over RCCL)

A "step" is one full render pass of the workload frame: every sample of every
pixel goes through camera-ray generation, two-level BVH traversal, scatter,
shadow rays and shading (src/render.rs:148-159 over the whole image), and for
N > 1 the composite of the tiles onto rank 0.  A ray = one Scene::hit_closest
call (implementation.rs:372), counted by the kernel itself.

Workload at N = 1 (BASELINE.json configs[1], the configuration the metric is
quoted on): raytracer-samples "Random Spheres" (the `weekend` scene), 1920x1080,
256 spp, max depth 8, lens radius 0.1, seed 12345.  `--workload` selects the
other BASELINE configs (they are parity/diagnostic cases, not the default line).

N > 1: the image is cut into 8-row strips (create_render_chunks(size,[1,n]),
viewport.rs:30-54), dealt round-robin to the ranks; every rank renders all its
strips in one launch into device memory and rank 0 gathers them over RCCL/xGMI
(one exchange per frame).  The frame is fixed, so scaling is "strong".

The JSON line also carries
  roofline     : the render kernel against the f32 VALU peak (the path is f32
                 vector arithmetic + latency bound; no MFMA: not a dense
                 contraction; scenes are cache resident so HBM is reported as a
                 second object, `roofline_hbm`).  Algorithmic flops/bytes per ray
                 follow SURVEY.md 8(d): per-ray node / primitive / transform
                 counts of an ordered, t_max-shrinking traversal, measured by the
                 oracle on a sample of the same frame, times fixed per-op costs.
  cpu_baseline : the oracle's restatement of the reference's multithreaded CPU
                 path (kind "port"; the Rust reference cannot be built: no rustc),
                 timed on this box's host cores on a bounded sample of the same
                 frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# per-op flop costs fixed from the reference arithmetic
F_AABB, F_SPH, F_TRI, F_PLANE, F_XFORM, F_SHADE = 27, 40, 90, 30, 45, 150
# bytes touched per visit in the HBM layout of device_scene.h
B_NODE, B_ENTITY, B_TRI, B_INST, B_RAY, B_PIXEL = 32, 48, 48, 48 + 48, 24, 12
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (scene, width, height, spp, depth, lens_radius, params)
    "weekend-1080p-256spp": ("Random Spheres", 1920, 1080, 256, 8, 0.1, None),          # configs[1]
    "weekend-400x225-16spp": ("Random Spheres", 400, 225, 16, 8, 0.1, None),            # configs[0]
    "dreadnaught-1080p-128spp": ("Dreadnaught", 1920, 1080, 128, 8, 0.1, None),         # configs[2]
    "meshdemo-4k-1024spp": ("Mesh", 3840, 2160, 1024, 8, 0.1, None),                    # configs[3]
    "dootdoot-4k-64spp": ("Doot Doot", 3840, 2160, 64, 8, 0.1, {"Count": 30, "Spacing": 30}),  # configs[4] geometry
    "dootdoot-4k-4096spp": ("Doot Doot", 3840, 2160, 4096, 8, 0.1, {"Count": 30, "Spacing": 30}),  # configs[4] in full (~30 s per frame on one GPU)
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="weekend-1080p-256spp", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="tiles", choices=["tiles", "spp"], help="N>1: tile partition + gather, or spp split + accumulator sum")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-chunk-stride", type=int, default=0, help="oracle sample: every k-th of 32x32 chunks")
    args = ap.parse_args()

    import numpy as np
    import torch
    from raytracer_amd import (CameraConfiguration, RenderChunk, RenderSettings, Scene, SceneBuilder,
                               create_render_chunks)
    from raytracer_amd._ffi import RtRayCounters

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ     # launched by torch.distributed.run
    if distributed:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world if distributed else 1
    device = local_rank if distributed else 0
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)

    scene_name, W, H, spp, depth, lens, params = WORKLOADS[args.workload]
    cfg = CameraConfiguration(W, H, lens_radius=lens)
    builder = SceneBuilder.sample(scene_name, cfg, params=params)
    scene = Scene(builder, device=device)
    settings = RenderSettings(W, H, depth, spp)
    seed = 12345
    info = scene.info()

    # ---- work split ---------------------------------------------------------
    from raytracer_amd.dist import InterleavedCompositor, reduce_accumulators, sample_range_of_rank
    from raytracer_amd import resolve_device
    counters = torch.zeros(4, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    whole = RenderChunk(0, 0, W, H)
    if args.mode == "tiles":
        comp = InterleavedCompositor(W, H, rank, n_gpus, dev, strip_height=8)
        parallelism = ("single GPU" if n_gpus == 1 else
                       f"8-row strips dealt round-robin over {n_gpus} GPUs, one launch per GPU, RCCL gather to rank 0")
    else:
        s_first, s_count = sample_range_of_rank(spp, rank, n_gpus)
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev) if rank == 0 else None
        parallelism = f"spp split: {n_gpus} ranks x {s_count} samples + accumulator gather/sum on rank 0"

    def step():
        if args.mode == "tiles":
            if n_gpus == 1:
                scene.render_chunk_device(settings, whole, seed, comp.packed.data_ptr(), None, W, 0, 0,
                                          counters.data_ptr(), stream)
            else:           # all of this rank's strips in one launch, straight into the send buffer
                scene.render_strips_device(settings, comp.strip_height, rank, n_gpus, seed, comp.packed.data_ptr(), None,
                                           counters.data_ptr(), stream)
                comp.composite()
        else:
            acc.zero_()
            if s_count:
                scene.accumulate_chunk_device(settings, whole, seed, s_first, s_count, acc.data_ptr(), counters.data_ptr(), stream)
            total = reduce_accumulators(acc, rank, n_gpus)
            if rank == 0:
                resolve_device(device, total.data_ptr(), W * H, spp, frame.data_ptr(), None, stream)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    counters.zero_()
    scene.profile_enable(True)
    scene.profile_read()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = scene.profile_read()
    scene.profile_enable(False)

    cn = counters.clone()
    local_rays = int(cn[:3].sum().item())          # this rank's rays: what its own kernel launches traced
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(cn, op=dist.ReduceOp.SUM)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    primary, secondary, shadow, samples = [int(v) for v in cn.tolist()]
    rays = primary + secondary + shadow
    rays_per_step = rays / args.steps

    result = None
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = rays / elapsed / 1e6
        result = {
            "metric": "Mrays/sec (whole node)", "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scene": scene_name, "width": W, "height": H,
                       "samples_per_pixel": spp, "max_reflections": depth, "lens_radius": lens, "seed": seed,
                       "entities": int(info.entity_count), "triangles": int(info.instanced_triangle_count),
                       "parallelism": parallelism},
            "msamples_per_s": round(samples / elapsed / 1e6, 3),
            "rays_per_step": {"primary": primary // args.steps, "secondary": secondary // args.steps,
                              "shadow": shadow // args.steps},
        }

    # ---- oracle legs on rank 0: (1) roofline statistics at any N ---------------------------------------
    if rank == 0 and not args.no_cpu_baseline:
        from tests.oracle import ORC_RNG_COUNTER, ORC_RNG_XORSHIFT_CHUNK, OracleScene
        oracle = OracleScene(builder)
        grid = (32, 32)
        n_chunks = grid[0] * grid[1]
        threads = min(16, os.cpu_count() or 1)
        # (1) per-ray work of the ordered, shrinking traversal (algorithmic minimum), low spp
        st_stats = RenderSettings(W, H, depth, min(spp, 4))
        stride_stats = 13
        cn_s, ms_s = oracle.render_chunks_mt(st_stats, grid, 5, stride_stats, (n_chunks - 6) // stride_stats, threads,
                                             rng_mode=ORC_RNG_COUNTER, seed=seed, stats=True)
        cpu_samples_per_s = cn_s.rays.samples / max(ms_s, 1e-3) * 1e3 * 2.0      # the statistics run traverses twice
        r_s = cn_s.rays.primary + cn_s.rays.secondary + cn_s.rays.shadow
        per_ray = {k: getattr(cn_s, "min_" + k) / r_s for k in ("nodes", "spheres", "planes", "tris", "xforms", "instances")}
        hits_per_ray = (cn_s.rays.primary + cn_s.rays.secondary) / r_s          # shading happens per path ray
        flops_per_ray = (per_ray["nodes"] * F_AABB + per_ray["spheres"] * F_SPH + per_ray["planes"] * F_PLANE +
                         per_ray["tris"] * F_TRI + per_ray["xforms"] * F_XFORM + hits_per_ray * F_SHADE)
        # on-chip gather demand per ray in the HBM layout (served by L1/L2 for these scenes)
        gather_bytes_per_ray = (per_ray["nodes"] * B_NODE + (per_ray["spheres"] + per_ray["planes"]) * B_ENTITY +
                                per_ray["tris"] * B_TRI + per_ray["instances"] * B_INST + B_RAY)
        kernel_s = kernel_ms / 1e3 / max(launches, 1)
        rays_per_launch = local_rays / max(launches, 1)     # rank 0's launches (N > 1: its share of the frame)
        tflops = flops_per_ray * rays_per_launch / kernel_s / 1e12
        # algorithmic HBM bytes of one launch: the scene once + one float3 segment sum per work item
        from raytracer_amd import sample_segment
        segs = (spp + sample_segment() - 1) // sample_segment()
        hbm_bytes_per_launch = (int(info.device_bytes) + W * H * segs * B_PIXEL // n_gpus) * (args.steps / max(launches, 1))
        gbs = hbm_bytes_per_launch / kernel_s / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # HBM bytes per launch from rocprofv3 --pmc passes
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get(args.workload)
            except Exception:
                traffic = None
        # VALU issue time of one launch from the committed PMC pass (profiles/pmc_traffic.json "valu_insts"):
        # every non-packed VALU wave-instruction holds its SIMD for 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz
        valu_issue_ms = None
        try:
            valu_insts = json.load(open(pmc_path)).get("valu_insts", {}).get(args.workload) if n_gpus == 1 else None
            if valu_insts:
                valu_issue_ms = valu_insts * 4.0 / (256 * 4) / 2.4e9 * 1e3
        except Exception:
            valu_issue_ms = None
        result["roofline"] = {
            "bound": "valu", "achieved": round(tflops, 4), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tflops / VALU_PEAK_TFLOPS, 5), "traffic": traffic if n_gpus == 1 else None,
            "kernel": "render_kernel", "kernel_ms_avg": round(kernel_s * 1e3, 4), "launches": int(launches),
            "flops_per_ray": round(flops_per_ray, 2),
            "per_ray": {k: round(v, 3) for k, v in per_ray.items()}, "traversal_mismatch": int(cn_s.min_mismatch),
            "valu_issue_ms": None if valu_issue_ms is None else round(valu_issue_ms, 3),
            "valu_issue_frac": None if valu_issue_ms is None else round(valu_issue_ms / (kernel_s * 1e3), 4),
        }
        result["roofline_hbm"] = {"bound": "hbm", "achieved": round(gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(gbs / HBM_PEAK_GBS, 6), "bytes_per_launch": int(hbm_bytes_per_launch),
                                  "scene_bytes": int(info.device_bytes),
                                  "on_chip_gather_bytes_per_ray": round(gather_bytes_per_ray, 2),
                                  "on_chip_gather_GBps": round(gather_bytes_per_ray * rays_per_launch / kernel_s / 1e9, 1)}
    # (2) the reference's multithreaded CPU path, bounded sample of the same frame: N = 1 only
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        # about 6e8 samples (10-20 s on 16 host threads), fewer on scenes where that would take over ~20 s
        target = min(6.0e8, max(2.0e6, cpu_samples_per_s * 20.0))
        stride = args.cpu_chunk_stride or max(1, int(round(W * H * spp / target)))
        if stride > 4:          # a sparse sample: many small chunks spread over the frame instead of a few large ones
            grid = (128, 128)
            n_chunks = grid[0] * grid[1]
            stride = max(1, int(round(W * H * spp / target)))
            if stride % 128 == 0 or stride % 2 == 0:
                stride += 1     # odd and not a multiple of the row length: the picks walk across rows and columns
        count = (n_chunks - 1) // stride
        cn_b, ms_b = oracle.render_chunks_mt(settings, grid, stride // 2, stride, count, threads, rng_mode=ORC_RNG_XORSHIFT_CHUNK,
                                             seed=seed)
        r_b = cn_b.rays.primary + cn_b.rays.secondary + cn_b.rays.shadow
        result["cpu_baseline"] = {
            "value": round(r_b / ms_b / 1e3, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{count} of {n_chunks} chunks ({grid[0]}x{grid[1]} grid, every {stride}th) of the same frame at full spp, "
                      f"{cn_b.rays.samples} samples, {ms_b / 1e3:.1f} s; oracle/rt_oracle.cpp -O2 -ffp-contract=off, "
                      f"{threads} std::threads pulling chunks (src/render.rs:186-224), per-chunk XorShift RNG",
            "msamples_per_s": round(cn_b.rays.samples / ms_b / 1e3, 3),
        }
        result["gpu_over_cpu"] = round(result["value"] / result["cpu_baseline"]["value"], 1)

    if rank == 0:
        print(json.dumps(result))
    scene.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
