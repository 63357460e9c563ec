"""rt_render_tile_columns_device: the finer tile partition of a frame over N GPUs - rank r renders the 8-pixel-wide tile columns
r, r + N, ... over the whole height in ONE launch, packed side by side.  The N ranks are emulated one after the other on one GPU,
through the C ABI; reassembled (raytracer_amd.dist.ColumnCompositor's index) the columns must be the frame of ONE rt_render_chunk call
bit for bit - floats, bytes and ray counts - whatever the width (not a multiple of 8, narrower than the ranks' tile columns)."""
import numpy as np
import pytest

from raytracer_amd import (CameraConfiguration, RenderChunk, RenderSettings, RtError, Scene, SceneBuilder, tile_columns_width)
from raytracer_amd.dist import ColumnCompositor, columns_of_rank

pytestmark = pytest.mark.gpu


def render_ranks(scene, st, world, seed):
    import torch
    w, h = st.width, st.height
    stream = torch.cuda.current_stream().cuda_stream
    frame = np.full((h, w, 3), -3.0, dtype=np.float32)
    frame8 = np.full((h, w, 4), 7, dtype=np.uint8)
    rays = [0, 0, 0, 0]
    for r in range(world):
        pw = tile_columns_width(w, r, world)
        rgb = torch.full((h, max(pw, 1), 3), -5.0, dtype=torch.float32, device="cuda:0")
        rgba = torch.full((h, max(pw, 1), 4), 3, dtype=torch.uint8, device="cuda:0")
        cn = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        scene.render_tile_columns_device(st, r, world, seed, rgb.data_ptr(), rgba.data_ptr(), cn.data_ptr(), stream)
        torch.cuda.synchronize()
        cols = columns_of_rank(w, r, world)
        assert len(cols) == pw
        if pw:
            frame[:, cols] = rgb.cpu().numpy()
            frame8[:, cols] = rgba.cpu().numpy()
        rays = [a + b for a, b in zip(rays, cn.tolist())]
    return frame, frame8, rays


@pytest.mark.parametrize("name,w,h,spp,depth,lens", [("Random Spheres", 150, 70, 16, 6, 0.1), ("Mesh", 333, 61, 4, 8, 0.05), ("Dreadnaught", 64, 40, 9, 4, 0.0),
                                                      ("Simple", 5, 33, 2, 3, 0.0), ("Mesh", 26, 26, 64, 5, 0.2)])
def test_dealt_tile_columns_equal_the_one_call_frame(name, w, h, spp, depth, lens):
    scene = Scene(SceneBuilder.sample(name, CameraConfiguration(w, h, lens_radius=lens)), 0)
    st = RenderSettings(w, h, depth, spp)
    whole, whole8, cn = scene.render_chunk(st, RenderChunk(0, 0, w, h), seed=11)
    for world in (1, 2, 3, 8):
        frame, frame8, rays = render_ranks(scene, st, world, 11)
        assert np.array_equal(frame, whole) and np.array_equal(frame8, whole8), (name, world)
        assert rays == [cn.primary, cn.secondary, cn.shadow, cn.samples], (name, world)
    scene.close()


def test_tile_column_arguments():
    import torch
    scene = Scene(SceneBuilder.sample("Simple", CameraConfiguration(40, 24)), 0)
    st = RenderSettings(40, 24, 3, 2)
    buf = torch.zeros((24, 40, 3), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for first, period in ((0, 0), (2, 2), (5, 3)):
        with pytest.raises(RtError):
            scene.render_tile_columns_device(st, first, period, 1, buf.data_ptr(), None, None, stream)
        with pytest.raises(RtError):
            tile_columns_width(40, first, period)
    # a rank beyond the frame's tile columns owns nothing: the call succeeds and writes nothing
    buf.fill_(-1.0)
    scene.render_tile_columns_device(st, 7, 8, 1, buf.data_ptr(), None, None, stream)
    torch.cuda.synchronize()
    assert tile_columns_width(40, 7, 8) == 0 and float(buf.min()) == -1.0 and float(buf.max()) == -1.0
    scene.close()


def test_column_compositor_assembles_the_frame_on_the_device():
    """raytracer_amd.dist.ColumnCompositor (what bench.py --gpus N runs per rank) with device tensors: every rank renders into its own
    compositor's send buffer, rank 0's compositor assembles them (the gather itself is torch.distributed's: tests/test_dist_gloo.py)."""
    import torch
    w, h, world = 150, 70, 4
    scene = Scene(SceneBuilder.sample("Random Spheres", CameraConfiguration(w, h, lens_radius=0.1)), 0)
    st = RenderSettings(w, h, 6, 8)
    whole, _, _ = scene.render_chunk(st, RenderChunk(0, 0, w, h), seed=21)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    comps = [ColumnCompositor(w, h, r, world, dev) for r in range(world)]
    for r, c in enumerate(comps):
        if c.my_width:
            scene.render_tile_columns_device(st, r, world, 21, c.packed.data_ptr(), None, None, stream)
    torch.cuda.synchronize()
    frame = comps[0].assemble([c._packed_full for c in comps])
    assert frame.is_contiguous() and np.array_equal(frame.cpu().numpy(), whole)
    scene.close()
