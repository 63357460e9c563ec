"""The allocation-free entry point (rt_render_chunk_device_noalloc) against the entry point it is a variant of
(rt_render_chunk_device): same pixels and ray counts, bit for bit, on two calls in a row, the second smaller -
every call on a GPU shares one static scratch array and finds the previous call's contents in it."""
import numpy as np
import pytest

from raytracer_amd import CameraConfiguration, RenderChunk, RenderSettings, RtError, Scene, SceneBuilder, _ffi

pytestmark = pytest.mark.gpu


def render(scene, fn, st, ch):
    import torch
    rgb = torch.full((ch.height, ch.width, 3), -7.0, dtype=torch.float32, device="cuda:0")
    rgba = torch.full((ch.height, ch.width, 4), 9, dtype=torch.uint8, device="cuda:0")
    cn = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    fn(st, ch, 12345, rgb.data_ptr(), rgba.data_ptr(), ch.width, 0, 0, cn.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return rgb.cpu().numpy(), rgba.cpu().numpy(), cn.tolist()


@pytest.mark.parametrize("name", ["Random Spheres", "Mesh"])
def test_noalloc_entry_point_gives_the_same_pixels(name):
    w, h = 640, 360
    scene = Scene(SceneBuilder.sample(name, CameraConfiguration(w, h, lens_radius=0.1)), 0)
    st = RenderSettings(w, h, 8, 40)                       # two groups of 16 samples + one segment of 8
    calls = [RenderChunk(0, 0, w, h),                      # classified tiles, lens-cell rounds, several units per launch
             RenderChunk(200, 100, 150, 77),               # smaller, ragged: the scratch still holds the first call's sums
             RenderChunk(17, 5, 9, 3)]                     # under-filled launch: single-sample work items
    for ch in calls:
        want = render(scene, scene.render_chunk_device, st, ch)
        got = render(scene, scene.render_chunk_device_noalloc, st, ch)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2], (name, ch.width, ch.height)
        assert want[2][3] == ch.width * ch.height * 40
    scene.close()


def test_noalloc_large_chunk_runs_with_fewer_units_per_launch():
    """1024 x 1024 pixels leave room for one unit of segment sums only: same bits, more launches."""
    w, h = 1024, 1024
    scene = Scene(SceneBuilder.sample("Random Spheres", CameraConfiguration(w, h, lens_radius=0.1)), 0)
    st, ch = RenderSettings(w, h, 6, 24), RenderChunk(0, 0, w, h)
    want = render(scene, scene.render_chunk_device, st, ch)
    got = render(scene, scene.render_chunk_device_noalloc, st, ch)
    assert np.array_equal(got[0], want[0]) and got[2] == want[2]
    scene.close()


def test_noalloc_rejects_chunks_beyond_the_scratch():
    import torch
    lib = _ffi.load_hip()
    limit = lib.rt_noalloc_max_pixels()
    assert 1024 * 1024 <= limit < 2048 * 1024
    w, h = 2048, 1024
    scene = Scene(SceneBuilder.sample("Random Spheres", CameraConfiguration(w, h)), 0)
    tiny = torch.zeros(16, dtype=torch.float32, device="cuda:0")
    with pytest.raises(RtError, match="exceeds the static scratch") as e:
        scene.render_chunk_device_noalloc(RenderSettings(w, h, 4, 1), RenderChunk(0, 0, w, h), 1, tiny.data_ptr(), None, w, 0, 0, None,
                                          torch.cuda.current_stream().cuda_stream)
    assert e.value.code == _ffi.RT_E_LIMIT
    scene.close()
