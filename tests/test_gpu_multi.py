"""rt_render_frame_multi: one frame on N scene replicas, composited in the first replica's device memory, behind the C ABI (what a
single-process host - the reference's egui front end, src/render.rs:186-224 - calls instead of torch.distributed).  The box has one GPU,
so the replicas here share it (each its own RtScene, stream and host thread: the whole mechanism except the xGMI hop): tile mode must give
the frame of ONE rt_render_chunk_device call bit for bit, on the direct-write path and on the packed-copy path (RT_MULTI_NO_PEER, the
fallback where peer access is refused); spp mode must give the ordered sum of raytracer_amd/dist.py bit for bit and the one-call frame to
1e-5 (another association of the same segment sums)."""
import os

import numpy as np
import pytest

from raytracer_amd import (CameraConfiguration, RenderChunk, RenderSettings, RtError, Scene, SceneBuilder, _ffi, render_frame_multi,
                           resolve_device, sample_segment)

pytestmark = pytest.mark.gpu


def one_call(scene, st):
    import torch
    w, h = st.width, st.height
    rgb = torch.full((h, w, 3), -7.0, dtype=torch.float32, device="cuda:0")
    rgba = torch.full((h, w, 4), 9, dtype=torch.uint8, device="cuda:0")
    cn = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    scene.render_chunk_device(st, RenderChunk(0, 0, w, h), 12345, rgb.data_ptr(), rgba.data_ptr(), w, 0, 0, cn.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return rgb.cpu().numpy(), rgba.cpu().numpy(), cn.tolist()


def multi(scenes, st, mode):
    import torch
    w, h = st.width, st.height
    rgb = torch.full((h, w, 3), -7.0, dtype=torch.float32, device="cuda:0")
    rgba = torch.full((h, w, 4), 9, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    cn = render_frame_multi(scenes, st, mode, 12345, rgb.data_ptr(), rgba.data_ptr())
    return rgb.cpu().numpy(), rgba.cpu().numpy(), [cn.primary, cn.secondary, cn.shadow, cn.samples]


@pytest.mark.parametrize("no_peer", [False, True])
@pytest.mark.parametrize("name,w,h,spp,n", [("Random Spheres", 640, 360, 24, 2), ("Mesh", 333, 187, 16, 3), ("Dreadnaught", 256, 100, 9, 5)])
def test_tile_mode_is_the_one_call_frame_bit_for_bit(name, w, h, spp, n, no_peer):
    builder = SceneBuilder.sample(name, CameraConfiguration(w, h, lens_radius=0.1))
    if no_peer:
        os.environ["RT_MULTI_NO_PEER"] = "1"          # (read when the call builds its context for a new list of replicas)
    try:
        replicas = [Scene(builder, 0) for _ in range(n)]
        st = RenderSettings(w, h, 8, spp)
        want = one_call(replicas[0], st)
        for _ in range(2):                             # the second frame reuses the cached streams and buffers
            got = multi(replicas, st, _ffi.RT_MULTI_TILES)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2], (name, n, no_peer)
        # another frame size on the same replicas, and fewer replicas (the context follows the list)
        st2 = RenderSettings(w, h, 4, 3)
        want2 = one_call(replicas[0], st2)
        got2 = multi(replicas[:2], st2, _ffi.RT_MULTI_TILES)
        assert np.array_equal(got2[0], want2[0]) and got2[2] == want2[2]
    finally:
        os.environ.pop("RT_MULTI_NO_PEER", None)
    for s in replicas:
        s.close()


@pytest.mark.parametrize("name,w,h,spp,n", [("Random Spheres", 320, 180, 64, 2), ("Mesh", 200, 120, 40, 3), ("Random Spheres", 96, 54, 8, 4)])
def test_spp_mode_is_the_ordered_sum_of_the_replica_ranges(name, w, h, spp, n):
    import torch
    builder = SceneBuilder.sample(name, CameraConfiguration(w, h, lens_radius=0.1))
    replicas = [Scene(builder, 0) for _ in range(n)]
    st, whole = RenderSettings(w, h, 8, spp), RenderChunk(0, 0, w, h)
    got = multi(replicas, st, _ffi.RT_MULTI_SPP)
    # the same ranges through rt_accumulate_chunk_device, added in rank order (raytracer_amd/dist.py: reduce_accumulators "ordered")
    seg = sample_segment(); segs = (spp + seg - 1) // seg
    stream = torch.cuda.current_stream().cuda_stream
    total, cn = None, torch.zeros(4, dtype=torch.int64, device="cuda:0")
    for r in range(n):
        lo, hi = segs * r // n * seg, min(segs * (r + 1) // n * seg, spp)
        acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
        if hi > lo:
            replicas[0].accumulate_chunk_device(st, whole, 12345, lo, hi - lo, acc.data_ptr(), cn.data_ptr(), stream)
        torch.cuda.synchronize()
        total = acc if total is None else total + acc
    rgb = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    rgba = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda:0")
    resolve_device(0, total.data_ptr(), w * h, spp, rgb.data_ptr(), rgba.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(got[0], rgb.cpu().numpy()) and np.array_equal(got[1], rgba.cpu().numpy()) and got[2] == cn.tolist()
    want = one_call(replicas[0], st)
    assert got[2] == want[2]
    assert np.abs(got[0] - want[0]).max() <= 1e-5
    for s in replicas:
        s.close()


def test_multi_argument_errors():
    import torch
    w, h = 64, 36
    builder = SceneBuilder.sample("Random Spheres", CameraConfiguration(w, h))
    a, b = Scene(builder, 0), Scene(builder, 0)
    st = RenderSettings(w, h, 4, 2)
    rgb = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    with pytest.raises(RtError, match="same scene twice") as e:
        render_frame_multi([a, a], st, _ffi.RT_MULTI_TILES, 1, rgb.data_ptr(), None)
    assert e.value.code == _ffi.RT_E_INVALID
    with pytest.raises(RtError, match="mode") as e:
        render_frame_multi([a, b], st, 7, 1, rgb.data_ptr(), None)
    assert e.value.code == _ffi.RT_E_INVALID
    host_only = Scene(builder, _ffi.RT_DEVICE_HOST_ONLY)
    with pytest.raises(RtError) as e:
        render_frame_multi([a, host_only], st, _ffi.RT_MULTI_TILES, 1, rgb.data_ptr(), None)
    assert e.value.code == _ffi.RT_E_NO_DEVICE
    cn = render_frame_multi([a, b], st, _ffi.RT_MULTI_TILES, 1, rgb.data_ptr(), None)      # rgba is optional
    assert cn.samples == w * h * 2
    for s in (a, b, host_only):
        s.close()
