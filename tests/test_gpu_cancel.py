"""Cancellation and progress inside a call (rt_scene_cancel / rt_scene_progress).  The reference checks its cancellation token per work
item and per pixel and sends a snapshot of the chunk every 800 ms (src/render.rs:131,149,156-158); here the pixel loop of a chunk is one
GPU call that can run for seconds (BASELINE configs[4]: 10 s per frame), so the token has to reach into the running kernel: a wave reads
the scene's host-mapped cancel word at its block claims."""
import threading
import time

import numpy as np
import pytest

from raytracer_amd import CameraConfiguration, RenderChunk, RenderSettings, RtError, Scene, SceneBuilder, _ffi

pytestmark = pytest.mark.gpu


def test_cancel_reaches_into_a_running_call_and_the_scene_stays_usable():
    w, h = 3840, 2160                                             # BASELINE configs[4]: 900 x 11 101 instanced triangles at 4K
    builder = SceneBuilder.sample("Doot Doot", CameraConfiguration(w, h, lens_radius=0.1), params={"Count": 30, "Spacing": 30})
    scene = Scene(builder, 0)
    small = RenderChunk(1800, 1000, 64, 40)
    st_small = RenderSettings(w, h, 8, 8)
    want_small = scene.render_chunk(st_small, small, seed=12345)
    st = RenderSettings(w, h, 8, 512)                             # about 1.3 s of GPU time if left alone
    result = {}

    def call():
        t0 = time.perf_counter()
        try:
            scene.render_chunk(st, RenderChunk(0, 0, w, h), seed=12345, want_rgb=False)
            result["code"] = 0
        except RtError as e:
            result["code"] = e.code
        result["returned"] = time.perf_counter()
        result["seconds"] = result["returned"] - t0

    th = threading.Thread(target=call)
    th.start()
    seen = []
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < 0.25:                   # watch the progress for a quarter of a second ...
        seen.append(scene.progress())
        time.sleep(0.01)
    t_cancel = time.perf_counter()
    scene.cancel(True)                                            # ... then cancel from this thread
    th.join(timeout=30)
    assert not th.is_alive()
    latency = result["returned"] - t_cancel
    print(f"cancelled a {st.samples_per_pixel}-spp 4K frame of 900 x 11 101 triangles after {t_cancel - t_start:.3f} s: the call returned {latency * 1e3:.1f} ms later "
          f"(it ran {result['seconds']:.3f} s in all); progress seen: {seen[0]} ... {seen[-1]}")
    assert result["code"] == _ffi.RT_E_CANCELLED
    assert latency < 0.05
    big = (w // 8) * (h // 8) * (512 // 8)                         # 8x8 ray tiles x 8-sample units
    assert {t for _, t in seen} <= {320, big}                      # (the first polls may still see the small call before: 8 x 5 tiles x 8 units... x 1)
    done = [d for d, t in seen if t == big]
    assert len(done) > 5 and done[-1] > done[0] and done[-1] <= big
    # while the scene is cancelled every call is refused; re-armed, it renders the same pixels as before
    with pytest.raises(RtError) as e:
        scene.render_chunk(st_small, small, seed=12345)
    assert e.value.code == _ffi.RT_E_CANCELLED
    scene.cancel(False)
    again = scene.render_chunk(st_small, small, seed=12345)
    assert np.array_equal(again[0], want_small[0]) and np.array_equal(again[1], want_small[1]) and again[2].rays == want_small[2].rays
    scene.close()


def test_cancel_of_enqueued_work_and_progress_to_the_end():
    """Device-pointer calls only enqueue: work enqueued while the scene is cancelled is refused, work enqueued BEFORE drains at once when the
    flag is set; a call that is left alone reports all its blocks."""
    import torch
    w, h = 1920, 1080
    scene = Scene(SceneBuilder.sample("Random Spheres", CameraConfiguration(w, h, lens_radius=0.1)), 0)
    st, whole = RenderSettings(w, h, 8, 64), RenderChunk(0, 0, w, h)
    rgb = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    scene.render_chunk_device(st, whole, 12345, rgb.data_ptr(), None, w, 0, 0, None, stream)
    torch.cuda.synchronize()
    done, total = scene.progress()
    assert total == (w // 8) * (h // 8) * 8 and 0.9 * total <= done <= total      # (an estimate: the last polls of the waves)
    want = rgb.cpu().numpy()
    t0 = time.perf_counter()
    for _ in range(10):
        scene.render_chunk_device(st, whole, 12345, rgb.data_ptr(), None, w, 0, 0, None, stream)
    scene.cancel(True)
    torch.cuda.synchronize()
    drained = time.perf_counter() - t0
    assert drained < 0.06, drained                                 # ten frames of 7 ms each, had they run
    with pytest.raises(RtError) as e:
        scene.render_chunk_device(st, whole, 12345, rgb.data_ptr(), None, w, 0, 0, None, stream)
    assert e.value.code == _ffi.RT_E_CANCELLED
    scene.cancel(False)
    scene.render_chunk_device(st, whole, 12345, rgb.data_ptr(), None, w, 0, 0, None, stream)
    torch.cuda.synchronize()
    assert np.array_equal(rgb.cpu().numpy(), want)
    scene.close()
