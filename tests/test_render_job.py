"""The C++ host harness (raytracer_amd/rt_render = render_job.cpp over the C ABI):
the reference's RenderJob / worker-thread protocol (src/render.rs:18-224).

CPU part: the job machinery with a synthetic per-chunk pattern (every chunk rendered
exactly once, blitted at its place, by any number of threads).  GPU part: the frame it
assembles from rt_render_chunk calls of several threads equals the one-call frame.
"""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "raytracer_amd", "rt_render")


def run(args, tmp_path, expect_rc=0):
    out = str(tmp_path / "frame.ppm")
    p = subprocess.run([EXE] + args + ["--out", out], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == expect_rc, p.stderr
    if expect_rc != 0:
        return None, None, p.stderr
    info = json.loads(p.stdout.strip().splitlines()[-1])
    with open(out, "rb") as f:
        assert f.readline() == b"P6\n"
        w, h = map(int, f.readline().split())
        assert f.readline() == b"255\n"
        img = np.frombuffer(f.read(), dtype=np.uint8).reshape(h, w, 3)
    return info, img, p.stderr


@pytest.mark.parametrize("w,h,cx,cy,threads", [(130, 70, 3, 4, 3), (64, 64, 1, 1, 1), (97, 41, 8, 8, 5), (33, 17, 2, 16, 16)])
def test_job_machinery_self_test(tmp_path, w, h, cx, cy, threads):
    info, img, _ = run(["--self-test", "--width", str(w), "--height", str(h), "--chunks", f"{cx}x{cy}", "--threads", str(threads)], tmp_path)
    ax, ay = np.meshgrid(np.arange(w, dtype=np.uint32), np.arange(h, dtype=np.uint32))
    expect = np.stack([(ax * 7) & 255, (ay * 13) & 255, ((ax ^ ay) * 3) & 255], axis=2).astype(np.uint8)
    assert np.array_equal(img, expect)
    assert info["chunks"] == cx * cy and sum(info["chunks_per_thread"]) == cx * cy
    assert info["rays"]["samples"] == w * h          # every pixel exactly once
    assert len(info["chunks_per_thread"]) == threads


def test_lists_the_scene_factories():
    p = subprocess.run([EXE, "--list"], cwd=ROOT, capture_output=True, text=True, timeout=60)
    names = p.stdout.strip().splitlines()
    assert "Random Spheres" in names and "Mesh" in names and len(names) >= 12


def test_fails_loudly_without_a_device(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _, _, err = run(["--width", "32", "--height", "18"], tmp_path, expect_rc=1)
    assert "no HIP device" in err


def test_unknown_scene_is_an_error(tmp_path):
    _, _, err = run(["--scene", "No Such Scene"], tmp_path, expect_rc=1)
    assert "rt_render:" in err


@pytest.mark.gpu
def test_scene_replicas_render_the_same_frame(tmp_path):
    """--devices a,b,...: one scene replica per entry, worker threads bound round robin (here both on GPU 0)."""
    args = ["--scene", "Mesh", "--width", "160", "--height", "90", "--spp", "4", "--chunks", "5x3", "--threads", "4"]
    one, img1, _ = run(args + ["--devices", "0"], tmp_path)
    two, img2, _ = run(args + ["--devices", "0,0"], tmp_path)
    assert np.array_equal(img1, img2) and one["fnv1a"] == two["fnv1a"] and one["rays"] == two["rays"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,params", [("Random Spheres", None), ("Mesh", None)])
def test_threads_and_chunks_assemble_the_one_call_frame(tmp_path, name, params):
    from raytracer_amd import CameraConfiguration, RenderChunk, RenderSettings, Scene, SceneBuilder
    w, h, spp, depth = 200, 112, 4, 8
    info, img, _ = run(["--scene", name, "--width", str(w), "--height", str(h), "--spp", str(spp), "--depth", str(depth),
                        "--chunks", "4x4", "--threads", "3", "--seed", "12345"], tmp_path)
    b = SceneBuilder.sample(name, CameraConfiguration(w, h, lens_radius=0.1))
    scene = Scene(b, 0)
    _, rgba, cn = scene.render_chunk(RenderSettings(w, h, depth, spp), RenderChunk(0, 0, w, h), seed=12345)
    assert np.array_equal(img, rgba[:, :, :3])
    r = info["rays"]
    assert (r["primary"], r["secondary"], r["shadow"], r["samples"]) == (cn.primary, cn.secondary, cn.shadow, cn.samples)
    assert sum(info["chunks_per_thread"]) == 16


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["tiles", "spp"])
def test_multi_frame_call_of_the_harness_equals_the_chunk_job(tmp_path, mode):
    """--multi: the frame in ONE rt_render_frame_multi_host call over three replicas (all on GPU 0 here) instead of the chunk job."""
    args = ["--scene", "Mesh", "--width", "200", "--height", "112", "--spp", "16", "--seed", "12345"]
    job, img_job, _ = run(args + ["--chunks", "4x4", "--threads", "3", "--devices", "0"], tmp_path)
    one, img_one, _ = run(args + ["--devices", "0,0,0", "--multi", mode], tmp_path)
    assert one["replicas"] == 3 and one["multi"] == mode and one["rays"] == job["rays"]
    if mode == "tiles":
        assert np.array_equal(img_one, img_job) and one["fnv1a"] == job["fnv1a"]
    else:          # another association of the same segment sums: a byte may differ by one where 255 c sits on an integer
        assert np.abs(img_one.astype(np.int16) - img_job.astype(np.int16)).max() <= 1


@pytest.mark.gpu
def test_cancelling_a_job_ends_the_gpu_calls_that_are_running(tmp_path):
    """RenderJobWorkerHandle::stop() = the reference's cancellation (src/render.rs:131,149): the token AND the chunk calls running on the GPU.
    One 4K chunk of the 900-instance scene at 512 spp runs over a second when left alone."""
    p = subprocess.run([EXE, "--scene", "Doot Doot", "--params", "Count=30;Spacing=30", "--width", "3840", "--height", "2160", "--spp", "512", "--chunks", "1x1",
                        "--threads", "1", "--cancel-after-ms", "300"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    info = json.loads(p.stdout.strip().splitlines()[-1])
    print(info)
    assert info["cancelled"] is True and info["chunks_completed"] == 0
    assert info["stop_ms"] < 100.0                                  # cancel + drain + the D2H copy of the (unfinished) 4K frame + join
    assert 0 < info["blocks_done"] < info["blocks_total"]
