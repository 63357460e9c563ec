#!/usr/bin/env python3
"""Benchmark of the hot path: Mrays/s of the render core on N MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Started directly (`python bench.py --gpus 8`) this process
launches `python -m torch.distributed.run --nproc-per-node N ... bench.py` as a child BEFORE any
GPU call and relays rank 0's JSON line; started by torch.distributed.run (RANK / WORLD_SIZE in
the environment) it is one of the ranks.  A world size that differs from --gpus, or more ranks
than visible GPUs, is an error - never a silent one-GPU run.

A "step" is one full render pass of the workload frame: every sample of every pixel goes through
camera-ray generation, two-level BVH traversal, scatter, shadow rays and shading
(src/render.rs:148-159 over the whole image), and for N > 1 the composite onto rank 0.  A ray =
one Scene::hit_closest call (implementation.rs:372), counted by the kernel itself.

Workload at N = 1 (BASELINE.json configs[1], the configuration the metric is quoted on):
raytracer-samples "Random Spheres" (the `weekend` scene), 1920x1080, 256 spp, max depth 8, lens
radius 0.1, seed 12345.  The default run also times the north_star's TARGET scene (configs[2], the
OBJ mesh BVH scene: Dreadnaught 1080p x 128 spp) and emits it under "target_scene" of the same JSON
line with its own roofline / cpu_baseline / gpu_over_cpu.  `--workload` selects any other BASELINE
config as the headline (parity / diagnostic cases, not the default line).

N > 1, --mode tiles: the image is cut into 8-row strips dealt round-robin to the ranks; every rank
renders all its strips in one launch into device memory and rank 0 gathers them over RCCL/xGMI
(one exchange per frame).  --mode spp: every rank renders the whole frame for its range of sample
indices into a float3 accumulator; rank 0 gathers and adds them in rank order.  The frame is fixed,
so scaling is "strong".

The default line also carries `configs`: every BASELINE config measured in the same run, each with its own steps / ms_per_step /
roofline / bounded cpu_baseline - C1 (configs[0], 400x225x16), C2 (the headline itself), C3 (the target scene), C4 (configs[3]:
mesh_demo 4K at the full 1024 spp, tile mode) and C5 (configs[4]: the 9.99 M-triangle instanced scene at 4K, spp mode: timed
at 64 spp, plus ONE step at the full 4096 spp).  Their step counts are fixed and small (a C5 frame at 4096 spp is 10 s on one
GPU) and do not follow --steps / --warmup; `--no-extra-configs` leaves them out.

The timed step writes the float3 frame (the V3 of cast_rays_into_scene); the RGBA8 conversion of src/rgba.rs is the same
finish_kernel with a second store (9 us per 1080p frame) and is exercised by the tests, not timed here (`writes`: "float3").

The JSON line also carries
  roofline     : the render kernel against the f32 VALU peak (the path is f32 vector arithmetic +
                 latency bound; no MFMA: not a dense contraction; scenes are cache resident so HBM
                 is reported as a second object, `roofline_hbm`).  Algorithmic flops/bytes per ray
                 follow SURVEY.md 8(d): per-ray node / primitive / transform counts of an ordered,
                 t_max-shrinking traversal, measured by the oracle on a sample of the same frame,
                 times fixed per-op costs.
                 `counter_frac` beside `frac`: the same fraction from the hardware counter instead of the model -
                 SQ_INSTS_VALU_FLOPS_FP32 (wave instructions) x 64 x lane utilisation per ray, from the committed rocprofv3 --pmc
                 pass of the workload (profiles/pmc_counters.json), x this run's rays per second / peak.
  cpu_baseline : the oracle's restatement of the reference's multithreaded CPU path (kind "port";
                 the Rust reference cannot be built: no rustc), timed on this box's host cores on a
                 bounded sample of the same frame: median of 3 runs with 16 threads (the reference
                 UI's maximum, src/settings.rs:201) and with all the threads this process may use.
"""
import argparse
import json
import math
import os
import socket
import subprocess
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
    "dreadnaught-1080p-128spp": ("Dreadnaught", 1920, 1080, 128, 8, 0.1, None),         # configs[2]: the target scene
    "meshdemo-4k-1024spp": ("Mesh", 3840, 2160, 1024, 8, 0.1, None),                    # configs[3]
    "meshdemo-4k-64spp": ("Mesh", 3840, 2160, 64, 8, 0.1, None),                        # configs[3] at 64 spp: quick A/B runs only
    "dootdoot-4k-64spp": ("Doot Doot", 3840, 2160, 64, 8, 0.1, {"Count": 30, "Spacing": 30}),  # configs[4] geometry
    "dootdoot-4k-4096spp": ("Doot Doot", 3840, 2160, 4096, 8, 0.1, {"Count": 30, "Spacing": 30}),  # configs[4] in full (~12 s per frame on one GPU)
}
# the BASELINE configs beside the headline and the target scene: (key, workload, mode, reduce, steps, warmup, cpu baseline?, same scene as)
EXTRA_CONFIGS = [
    ("C1", "weekend-400x225-16spp", "tiles", "ordered", 5, 1, True),
    ("C4", "meshdemo-4k-1024spp", "tiles", "ordered", 2, 1, True),
    ("C5", "dootdoot-4k-64spp", "spp", "allreduce", 2, 1, True),
    ("C5_full", "dootdoot-4k-4096spp", "spp", "allreduce", 1, 0, False),
]
# PMC passes are collected on the 64-spp frames of the 4K scenes: per-ray counters do not depend on the sample count
PMC_WORKLOAD = {"meshdemo-4k-1024spp": "meshdemo-4k-64spp", "dootdoot-4k-4096spp": "dootdoot-4k-64spp"}
HEADLINE = "weekend-1080p-256spp"
TARGET_SCENE = "dreadnaught-1080p-128spp"
SEED = 12345


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="tiles", choices=["tiles", "spp"], help="N>1: tile partition + gather, or spp split + accumulator sum")
    ap.add_argument("--partition", default="columns", choices=["columns", "strips"],
                    help="tile mode, N>1: 8-pixel tile columns (every rank holds a share of every image region) or 8-row strips, dealt round-robin")
    ap.add_argument("--reduce", default="ordered", choices=["ordered", "allreduce"],
                    help="--mode spp: gather + sum in rank order on rank 0 (deterministic), or one RCCL all-reduce")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs (roofline statistics and CPU baseline)")
    ap.add_argument("--no-target-scene", action="store_true", help="do not also time the target scene (configs[2])")
    ap.add_argument("--no-extra-configs", action="store_true", help="do not also time BASELINE configs C1, C4 and C5 (the `configs` object)")
    ap.add_argument("--launcher", action="store_true", help="go through torch.distributed.run even for --gpus 1 (rehearsal of the N > 1 path)")
    ap.add_argument("--cpu-seconds", type=float, default=4.0, help="CPU work per baseline run (3 runs per thread count)")
    return ap.parse_args(argv)


def launch_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child job (before this
    process touches a GPU), relay its output, exit with its status."""
    import torch                                    # device_count() does not initialise the GPU
    visible = torch.cuda.device_count()
    if args.gpus > visible:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {visible} GPU(s) are visible: refusing to report a {args.gpus}-GPU number\n")
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--launcher"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "16")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def measure(workload, args, ctx, mode=None, reduce=None, steps=None, warmup=None):
    """Times `steps` (default args.steps) render passes of one workload on this rank's GPU (all ranks together); returns the
    fields of the JSON line on rank 0 (None elsewhere) plus what the oracle legs need."""
    import types
    args = types.SimpleNamespace(**vars(args))
    if mode is not None: args.mode = mode
    if reduce is not None: args.reduce = reduce
    if steps is not None: args.steps = steps
    if warmup is not None: args.warmup = warmup
    import torch
    from raytracer_amd import CameraConfiguration, RenderChunk, RenderSettings, Scene, SceneBuilder, resolve_device
    from raytracer_amd.dist import ColumnCompositor, InterleavedCompositor, reduce_accumulators, sample_range_of_rank
    dist, rank, n_gpus, dev, device = ctx["dist"], ctx["rank"], ctx["n_gpus"], ctx["dev"], ctx["device"]

    scene_name, W, H, spp, depth, lens, params = WORKLOADS[workload]
    cfg = CameraConfiguration(W, H, lens_radius=lens)
    builder = SceneBuilder.sample(scene_name, cfg, params=params)
    scene = Scene(builder, device=device)
    settings = RenderSettings(W, H, depth, spp)
    info = scene.info()

    counters = torch.zeros(4, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    whole = RenderChunk(0, 0, W, H)
    if args.mode == "tiles":
        strips = getattr(args, "partition", "columns") == "strips"
        comp = InterleavedCompositor(W, H, rank, n_gpus, dev, strip_height=8) if strips else ColumnCompositor(W, H, rank, n_gpus, dev)
        parallelism = ("single GPU" if n_gpus == 1 else
                       (f"8-row strips" if strips else "8-pixel tile columns") + f" dealt round-robin over {n_gpus} GPUs, one launch per GPU, RCCL gather to rank 0")
    else:
        s_first, s_count = sample_range_of_rank(spp, rank, n_gpus)
        acc = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev) if rank == 0 else None
        parallelism = (f"spp split: {n_gpus} ranks x {s_count} samples + " +
                       ("accumulator gather + sum in rank order on rank 0" if args.reduce == "ordered" else "RCCL all-reduce of the accumulators"))

    def step():
        if args.mode == "tiles":
            if n_gpus == 1:
                scene.render_chunk_device(settings, whole, SEED, comp.packed.data_ptr(), None, W, 0, 0, counters.data_ptr(), stream)
            else:           # all of this rank's tile columns (or strips) in one launch, straight into the send buffer
                if strips:
                    scene.render_strips_device(settings, comp.strip_height, rank, n_gpus, SEED, comp.packed.data_ptr(), None, counters.data_ptr(), stream)
                elif comp.my_width:
                    scene.render_tile_columns_device(settings, rank, n_gpus, SEED, comp.packed.data_ptr(), None, counters.data_ptr(), stream)
                comp.composite()
        else:
            acc.zero_()
            if s_count:
                scene.accumulate_chunk_device(settings, whole, SEED, s_first, s_count, acc.data_ptr(), counters.data_ptr(), stream)
            total = reduce_accumulators(acc, rank, n_gpus, how=args.reduce)
            if rank == 0:
                resolve_device(device, total.data_ptr(), W * H, spp, frame.data_ptr(), None, stream)

    def barrier():
        if dist is not None:
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
    kernel_ms, launches = scene.profile_read()       # HIP events around every render_kernel launch, on its stream
    scene.profile_enable(False)

    # camera rays in EMPTY tiles (every camera ray of the tile misses every leaf box of the scene under a black sky: one packet test per tile - its beam walks the trees - instead of 64 x spp rays,
    # rt_hip.h RT_OPT_CULL_EMPTY_TILES): counted as the rays they are for the reference (hit_closest is called and finds no candidate) and for the CPU leg, never traced here
    local_culled = scene.culled_pixels() * (spp if args.mode == "tiles" else s_count)     # per step, this rank
    cn = torch.cat([counters, torch.tensor([local_culled], dtype=torch.int64, device=dev)])
    local_rays = int(cn[:3].sum().item())          # this rank's rays: what its own kernel launches traced (+ its empty tiles' camera rays)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(cn, op=dist.ReduceOp.SUM)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    primary, secondary, shadow, samples, culled = [int(v) for v in cn.tolist()]
    rays = primary + secondary + shadow
    scene.close()
    if rank != 0:
        return None
    line = {
        "metric": "Mrays/sec (whole node)", "value": round(rays / elapsed / 1e6, 3), "unit": "Mrays/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "scene": scene_name, "width": W, "height": H,
                   "samples_per_pixel": spp, "max_reflections": depth, "lens_radius": lens, "seed": SEED,
                   "entities": int(info.entity_count), "triangles": int(info.instanced_triangle_count),
                   "mode": args.mode, "parallelism": parallelism},
        "rccl_ranks": ctx["world"] if dist is not None else 0,
        "msamples_per_s": round(samples / elapsed / 1e6, 3), "writes": "float3",
        "rays_per_step": {"primary": primary // args.steps, "secondary": secondary // args.steps, "shadow": shadow // args.steps,
                          "primary_in_empty_tiles": culled},
    }
    return dict(line=line, builder=builder, settings=settings, info=info, kernel_ms=kernel_ms, launches=launches, local_rays=local_rays, local_culled=local_culled,
                W=W, H=H, spp=spp, depth=depth, workload=workload, steps=args.steps)


def host_threads():
    """Threads this process may run at once (the GPU box reports the whole machine in os.cpu_count())."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):             # the box's CPU share, when it states one
        try:
            allowed = min(allowed, int(os.environ[var]))
        except (KeyError, ValueError):
            pass
    return max(1, allowed)


def oracle_legs(m, args, n_gpus, cpu_leg=True, cpu_seconds=None):
    """Rank 0: roofline statistics of the render kernel (any N) and the CPU baseline (N = 1)."""
    cpu_seconds = args.cpu_seconds if cpu_seconds is None else cpu_seconds
    from raytracer_amd import RenderSettings, sample_segment
    from tests.oracle import ORC_RNG_COUNTER_STRATIFIED, ORC_RNG_XORSHIFT_CHUNK, OracleScene
    line, W, H, spp, depth, info = m["line"], m["W"], m["H"], m["spp"], m["depth"], m["info"]
    oracle = OracleScene(m["builder"])
    grid = (32, 32)
    n_chunks = grid[0] * grid[1]
    threads_all = host_threads()
    threads = min(16, threads_all)
    # (1) per-ray work of the ordered, shrinking traversal (algorithmic minimum), low spp
    st_stats = RenderSettings(W, H, depth, min(spp, 4))
    stride_stats = 13
    cn_s, ms_s = oracle.render_chunks_mt(st_stats, grid, 5, stride_stats, (n_chunks - 6) // stride_stats, threads,
                                         rng_mode=ORC_RNG_COUNTER_STRATIFIED, seed=SEED, stats=True)
    cpu_samples_per_s = cn_s.rays.samples / max(ms_s, 1e-3) * 1e3 * 2.0      # the statistics run traverses twice
    r_s = cn_s.rays.primary + cn_s.rays.secondary + cn_s.rays.shadow
    per_ray = {k: getattr(cn_s, "min_" + k) / r_s for k in ("nodes", "spheres", "planes", "tris", "xforms", "instances")}
    hits_per_ray = (cn_s.rays.primary + cn_s.rays.secondary) / r_s          # shading happens per path ray
    flops_per_ray = (per_ray["nodes"] * F_AABB + per_ray["spheres"] * F_SPH + per_ray["planes"] * F_PLANE +
                     per_ray["tris"] * F_TRI + per_ray["xforms"] * F_XFORM + hits_per_ray * F_SHADE)
    # on-chip gather demand per ray in the HBM layout (served by LDS/L1/L2 for these scenes)
    gather_bytes_per_ray = (per_ray["nodes"] * B_NODE + (per_ray["spheres"] + per_ray["planes"]) * B_ENTITY +
                            per_ray["tris"] * B_TRI + per_ray["instances"] * B_INST + B_RAY)
    launches = max(m["launches"], 1)
    kernel_s = m["kernel_ms"] / 1e3 / launches
    rays_per_launch = m["local_rays"] / launches           # rank 0's launches (N > 1: its share of the frame)
    tflops = flops_per_ray * rays_per_launch / kernel_s / 1e12
    # the same without the camera rays of empty tiles, which the model prices and the kernel never traces (priced here at their floor, one box test + the sky:
    # those that pass inner boxes before they miss cost the model more, so frac_traced errs high, never low)
    culled_per_launch = m.get("local_culled", 0) * m["steps"] / launches
    tflops_traced = (flops_per_ray * rays_per_launch - culled_per_launch * (F_AABB + F_SHADE)) / kernel_s / 1e12
    # algorithmic HBM bytes of one launch: the scene once + one float3 segment sum per work item
    segs = (spp + sample_segment() - 1) // sample_segment()
    hbm_bytes_per_launch = (int(info.device_bytes) + W * H * segs * B_PIXEL // n_gpus) * (m["steps"] / launches)
    gbs = hbm_bytes_per_launch / kernel_s / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # HBM bytes per launch from rocprofv3 --pmc passes
    if os.path.exists(pmc_path) and n_gpus == 1:
        try:
            traffic = json.load(open(pmc_path)).get(m["workload"])
        except Exception:
            traffic = None
    # the same fraction from the hardware counter of the committed PMC pass (executed lane-flops per ray x this run's ray rate)
    counter_frac = counter_flops_per_ray = None
    cnt_path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    pmc_key = m["workload"]
    try:              # a workload without counter passes of its own borrows those of the same scene at 64 spp
        if pmc_key not in json.load(open(cnt_path)):
            pmc_key = PMC_WORKLOAD.get(pmc_key, pmc_key)
    except Exception:
        pmc_key = PMC_WORKLOAD.get(pmc_key, pmc_key)
    if os.path.exists(cnt_path) and n_gpus == 1:
        try:
            c = json.load(open(cnt_path)).get(pmc_key)
            if c:
                counter_flops_per_ray = c["valu_flops_fp32_wave_instr"] * 64.0 * c["lane_util"] / c["rays_per_launch"]
                counter_frac = counter_flops_per_ray * rays_per_launch / kernel_s / 1e12 / VALU_PEAK_TFLOPS
        except Exception:
            counter_frac = counter_flops_per_ray = None
    traffic_key = PMC_WORKLOAD.get(m["workload"], m["workload"])
    if traffic is None and os.path.exists(pmc_path) and n_gpus == 1 and traffic_key != m["workload"]:
        try:          # per-launch bytes of the 64-spp frame, scaled to this launch's rays (the traffic is the segment sums and per-ray spills)
            t64 = json.load(open(pmc_path)).get(traffic_key)
            r64 = json.load(open(cnt_path)).get(traffic_key, {}).get("rays_per_launch")
            traffic = int(t64 * rays_per_launch / r64) if t64 and r64 else None
        except Exception:
            traffic = None
    line["roofline"] = {
        "bound": "valu", "achieved": round(tflops, 4), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": round(tflops / VALU_PEAK_TFLOPS, 5), "traffic": traffic,
        "frac_traced": round(tflops_traced / VALU_PEAK_TFLOPS, 5), "rays_in_empty_tiles_per_launch": int(culled_per_launch),
        "counter_frac": None if counter_frac is None else round(counter_frac, 5),
        "counter_flops_per_ray": None if counter_flops_per_ray is None else round(counter_flops_per_ray, 2),
        "kernel": "render_kernel", "kernel_ms_avg": round(kernel_s * 1e3, 4), "launches": int(m["launches"]),
        "flops_per_ray": round(flops_per_ray, 2),
        "per_ray": {k: round(v, 3) for k, v in per_ray.items()}, "traversal_mismatch": int(cn_s.min_mismatch),
    }
    line["roofline_hbm"] = {"bound": "hbm", "achieved": round(gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(gbs / HBM_PEAK_GBS, 6), "bytes_per_launch": int(hbm_bytes_per_launch),
                            "scene_bytes": int(info.device_bytes),
                            "on_chip_gather_bytes_per_ray": round(gather_bytes_per_ray, 2),
                            "on_chip_gather_GBps": round(gather_bytes_per_ray * rays_per_launch / kernel_s / 1e9, 1)}
    if n_gpus != 1 or not cpu_leg:
        return
    # (2) the reference's multithreaded CPU path on a bounded sample of the same frame: chunks spread evenly over the
    # frame, full spp, per-chunk XorShift RNG; about args.cpu_seconds of work per run, 3 runs per thread count, median.
    target = min(3.0e8, max(1.0e6, cpu_samples_per_s * cpu_seconds))
    stride = max(1, int(round(W * H * spp / target)))
    if stride > 4:          # a sparse sample: many small chunks spread over the frame instead of a few large ones
        grid = (128, 128)
        if grid[0] * grid[1] // stride < 200 and min(W, H) >= 1024:
            grid = (512, 512)       # at least some hundred chunks however small the share: 31 chunks of mesh_demo's frame gave 7.5, 21 or 77 Mrays/s depending on where they fell
        n_chunks = grid[0] * grid[1]
        # the same chunks from run to run: the stride follows the host's measured speed, and on a frame whose chunks differ a hundredfold in cost (glass against sky) a
        # slightly different stride draws a different sample - 7.5 against 21 Mrays/s on mesh_demo.  Quantised to a power of two times the golden ratio, made odd (the
        # picks then form a lattice over rows and columns - a stride of 2^k + 1 would walk ONE diagonal of a 2^m-wide grid), the sample only changes when the host is
        # twice as fast or slow.
        stride = int(round((1 << max(2, int(round(math.log2(stride / 1.618)))) ) * 1.618)) | 1
    count = (n_chunks - 1) // stride

    def timed(n_threads):
        runs = []
        for _ in range(3):
            cn_b, ms_b = oracle.render_chunks_mt(m["settings"], grid, stride // 2, stride, count, n_threads,
                                                 rng_mode=ORC_RNG_XORSHIFT_CHUNK, seed=SEED)
            r_b = cn_b.rays.primary + cn_b.rays.secondary + cn_b.rays.shadow
            runs.append((r_b / ms_b / 1e3, cn_b.rays.samples / ms_b / 1e3, ms_b, cn_b.rays.samples))
        runs.sort()
        return runs[1], [round(r[0], 3) for r in runs]

    legs = {}
    for n_threads in sorted({threads, threads_all if threads_all <= 64 else 64}):
        med, all_runs = timed(n_threads)
        legs[n_threads] = {"value": round(med[0], 3), "msamples_per_s": round(med[1], 3), "runs": all_runs, "seconds": round(med[2] / 1e3, 2),
                           "samples": int(med[3])}
    best = max(legs, key=lambda k: legs[k]["value"])
    line["cpu_baseline"] = {
        "value": legs[best]["value"], "unit": "Mrays/s", "cores": best, "kind": "port",
        "sample": f"{count} of {n_chunks} chunks ({grid[0]}x{grid[1]} grid, every {stride}th) of the same frame at full spp, "
                  f"{legs[best]['samples']} samples, {legs[best]['seconds']} s per run, median of 3; oracle/rt_oracle.cpp -O2 -ffp-contract=off, "
                  f"std::threads pulling chunks (src/render.rs:186-224), per-chunk XorShift RNG",
        "msamples_per_s": legs[best]["msamples_per_s"],
        "by_threads": {str(k): v for k, v in legs.items()},
        "host": {"os_cpu_count": os.cpu_count(), "usable_threads": threads_all},
    }
    line["gpu_over_cpu"] = round(line["value"] / line["cpu_baseline"]["value"], 1)


def main():
    args = parse_args()
    under_launcher = "RANK" in os.environ and "MASTER_ADDR" in os.environ     # one rank of torch.distributed.run
    if not under_launcher and (args.gpus > 1 or args.launcher):
        launch_ranks(args)                                                    # does not return

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1")) if under_launcher else 1
    rank = int(os.environ.get("RANK", "0")) if under_launcher else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if under_launcher else 0
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: refusing to report a {args.gpus}-GPU number\n")
        sys.exit(2)
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local_rank}, {torch.cuda.device_count()} visible)\n")
        sys.exit(2)
    dist = None
    torch.cuda.set_device(local_rank)
    if under_launcher:
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")      # RCCL's version banner must not share stdout with the JSON line
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    ctx = dict(dist=dist, rank=rank, world=world, n_gpus=world, device=local_rank, dev=torch.device("cuda", local_rank))

    m = measure(args.workload, args, ctx)
    target = None
    if not args.no_target_scene and args.workload == HEADLINE:
        target = measure(TARGET_SCENE, args, ctx)
    extras = []
    if not args.no_extra_configs and args.workload == HEADLINE:
        for key, workload, mode, reduce, steps, warmup, cpu_leg in EXTRA_CONFIGS:
            extras.append((key, cpu_leg, measure(workload, args, ctx, mode=mode, reduce=reduce, steps=steps, warmup=warmup)))
    if rank == 0:
        if not args.no_cpu_baseline:
            oracle_legs(m, args, world)
            if target is not None:
                oracle_legs(target, args, world)
            for key, cpu_leg, e in extras:
                oracle_legs(e, args, world, cpu_leg=cpu_leg, cpu_seconds=min(args.cpu_seconds, 2.0))
        result = m["line"]
        keep = ("value", "unit", "steps", "warmup", "ms_per_step", "config", "msamples_per_s", "rays_per_step", "roofline", "roofline_hbm", "cpu_baseline",
                "gpu_over_cpu")
        if target is not None:
            t = target["line"]
            result["target_scene"] = {k: t[k] for k in keep if k in t}
        if extras:
            # every BASELINE config in one line: C2 and C3 are the headline and the target scene above
            cfgs = {"C2": {"see": "the top level of this line", "workload": HEADLINE}}
            if target is not None:
                cfgs["C3"] = {"see": "target_scene", "workload": TARGET_SCENE}
            for key, cpu_leg, e in extras:
                cfgs[key] = {k: e["line"][k] for k in keep if k in e["line"]}
            c5, c5f = cfgs.get("C5"), cfgs.get("C5_full")
            if c5 and c5f and "cpu_baseline" in c5:        # Mrays/s of the CPU path do not depend on the sample count: one CPU leg for both
                c5f["gpu_over_cpu"] = round(c5f["value"] / c5["cpu_baseline"]["value"], 1)
                c5f["cpu_baseline"] = {"see": "configs.C5.cpu_baseline (same scene and frame; the CPU rate does not depend on the sample count)"}
            result["configs"] = cfgs
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
