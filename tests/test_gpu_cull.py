"""Empty tiles (RT_OPT_CULL_EMPTY_TILES, render.hip: tile_misses_root): a tile whose camera rays - every lens sample of every pixel - all miss
the scene's root box is not traced under a black sky.  The reference tests every such ray against the root (bvh.rs:210-236) and returns
color_sky_black's (0, 0, 0) (implementation.rs:517-519); the packet test replaces 64 x spp of those tests by one, and must never change a bit
of the image or a ray counter.  Checked here against the SAME library with the option off (every sample traced), which the parity suites hold
against the oracle - and against the oracle directly."""
import numpy as np
import pytest
import torch

from raytracer_amd import (CameraConfiguration, RenderChunk, RenderSettings, Scene, SceneBuilder, camera_new, sample_segment,
                           strip_rows, tile_columns_width)
from tests.oracle import ORC_RNG_COUNTER_STRATIFIED, OracleScene

pytestmark = pytest.mark.gpu


def render_both(scene, st, chunk, seed=7):
    out = []
    for cull in (1, 0):
        scene.set_option("cull_empty_tiles", cull)
        rgb, rgba, cn = scene.render_chunk(st, chunk, seed=seed)
        out.append((rgb, rgba, (cn.primary, cn.secondary, cn.shadow, cn.samples), scene.culled_pixels()))
    scene.set_option("cull_empty_tiles", 1)
    return out


def assert_same(on, off):
    assert np.array_equal(on[0].view(np.uint32), off[0].view(np.uint32)), "the image changed"      # bit for bit, the sign of zero included
    assert np.array_equal(on[1], off[1])
    assert on[2] == off[2], "the ray counters changed"
    assert off[3] == 0


@pytest.mark.parametrize("name,params", [("Dreadnaught", None), ("Interceptor Spin", {"Time Ms": 700.0}), ("Capsule", None), ("Spaceships", None),
                                         ("Fleet", None), ("Doot Doot", {"Count": 4, "Spacing": 40})])
def test_sample_scenes_under_a_black_sky(name, params):
    w, h = 384, 216
    b = SceneBuilder.sample(name, CameraConfiguration(w, h, lens_radius=0.1), params=params)
    scene = Scene(b, 0)
    scene.set_option("fine_split", 0)          # (small frames would otherwise run single-sample work items, whose tiles are not classified)
    st = RenderSettings(w, h, 4, 16)
    on, off = render_both(scene, st, RenderChunk(0, 0, w, h))
    assert_same(on, off)
    t, n, e, tri = scene.primary_hits(st, RenderChunk(0, 0, w, h))
    hit_tiles = {(y // 8, x // 8) for y, x in zip(*np.nonzero(e >= 0))}
    n_tiles = ((w + 7) // 8) * ((h + 7) // 8)
    print(f"{name}: {on[3]} of {w * h} pixels in empty tiles; {len(hit_tiles)} of {n_tiles} tiles hold a primary hit")
    assert on[3] <= (n_tiles - len(hit_tiles)) * 64
    if name in ("Dreadnaught", "Interceptor Spin"):      # a lone object in front of a black sky: most of the frame is empty
        assert on[3] > 0.3 * w * h
    scene.close()


def lone_objects(seed, w, h, sky=1, meshes=True):
    """A few spheres and meshes (no planes: their boxes are infinite) under a black sky, seen by a random camera - far, near, INSIDE the root box,
    wide and narrow, pinhole to a lens as large as the scene."""
    r = np.random.default_rng(seed)
    where = int(r.integers(0, 4))
    eye = (r.uniform(-1, 1, 3) * (2, 1, 2) + (0, 1, 0) if where == 0 else                    # inside or at the box
           r.uniform(-1, 1, 3) * 4 + (0, 2, 9) if where == 1 else                              # near
           r.uniform(-1, 1, 3) * 30 + (0, 10, 60) if where == 2 else                           # far
           r.uniform(-1, 1, 3) * (12, 1, 12) + (0, float(r.choice([-9.0, 9.0])), 0))           # above / below, looking along the axis
    to = r.uniform(-1, 1, 3) * float(r.choice([0.5, 4.0, 15.0]))                              # at the objects, or past them
    lens = float(r.choice([0.0, 0.05, 0.5, 3.0]))
    cam = camera_new(tuple(eye), tuple(to), float(r.choice([15.0, 50.0, 110.0])), w / h, lens, float(r.uniform(1.0, 30.0)))
    b = SceneBuilder(cam, sky)                                                                 # 1: RT_SKY_BLACK, 0: RT_SKY_DAY
    mats = [b.add_lambertian(float(r.uniform(0.0, 0.5))), b.add_specular(0.8, 0.1), b.add_dielectric(0.6, 0.2, 1.5)]
    tex = b.add_color_texture((0.7, 0.6, 0.5))
    verts = r.uniform(-1, 1, (12, 3)).astype(np.float32)
    mesh = b.add_mesh(verts[r.integers(0, 12, (20, 3))])
    for k in range(int(r.integers(1, 7))):
        tr = [tuple(r.uniform(-1, 1, 3) * (3, 1, 3) + (0, 1, 0))]
        ro = [((0.0, 1.0, 0.0), float(r.uniform(-3, 3)))] if r.random() < 0.5 else []
        if r.random() < 0.6 or not meshes:
            b.add_sphere_entity(float(r.uniform(0.2, 1.0)), mats[int(r.integers(0, 3))], tex, translations=tr, rotations=ro, id=k)
        else:
            b.add_mesh_entity(mesh, mats[int(r.integers(0, 3))], tex, translations=tr, rotations=ro, id=k)
    b.add_point_light((3.0, 8.0, 4.0), 60.0)
    return b


@pytest.mark.parametrize("seed", range(48))
def test_random_cameras_and_lenses(seed):
    w, h = 200 + 8 * (seed % 5) + (seed % 3), 136 + (seed % 7)      # ragged right and bottom tiles
    b = lone_objects(5000 + seed, w, h)
    scene = Scene(b, 0)
    scene.set_option("fine_split", 0)
    spp = [64, 16, 9][seed % 3]                                      # 64: the 8 x 8 lens cells reach into every corner of the lens square
    st = RenderSettings(w, h, 3, spp)
    on, off = render_both(scene, st, RenderChunk(0, 0, w, h), seed=seed)
    assert_same(on, off)
    scene.close()


@pytest.mark.parametrize("seed", range(32))
def test_tiles_without_a_hit_under_a_day_sky(seed):
    """A sky whose colour depends on the ray (color_sky_day, implementation.rs:502-515): tiles whose beam is clear of every leaf box are still rendered - lens draw,
    camera ray, sky colour per sample - but their camera rays skip the walk (render.hip: blk_nohit).  Same bits, same counters as with the option off."""
    w, h = 200 + 8 * (seed % 5) + (seed % 3), 136 + (seed % 7)
    scene = Scene(lone_objects(5000 + seed, w, h, sky=0, meshes=False), 0)
    scene.set_option("fine_split", 0)
    st = RenderSettings(w, h, 3, [64, 16, 9][seed % 3])
    on, off = render_both(scene, st, RenderChunk(0, 0, w, h), seed=seed)
    assert_same(on, off)
    assert on[3] == 0                       # (nothing is left out under a day sky)
    scene.close()


def test_a_good_share_of_the_day_sky_cases_skips_walks():
    n = 0
    for seed in range(32):
        w, h = 200, 136
        scene = Scene(lone_objects(5000 + seed, w, h, sky=0, meshes=False), 0)
        scene.set_option("fine_split", 0)
        scene.render_chunk(RenderSettings(w, h, 1, 8), RenderChunk(0, 0, w, h), want_rgba=False)
        left_out, without_walk = scene.culled_pixels(both=True)
        assert left_out == 0
        n += without_walk > 0
        scene.close()
    print(f"{n} of 32 day-sky cases have tiles rendered without a walk")
    assert n >= 10


@pytest.mark.parametrize("name", ["Random Spheres", "Simple", "Mesh", "Capsule"])
def test_sample_scenes_with_the_option_on_and_off(name):
    w, h = 384, 216
    scene = Scene(SceneBuilder.sample(name, CameraConfiguration(w, h, lens_radius=0.1)), 0)
    scene.set_option("fine_split", 0)
    on, off = render_both(scene, RenderSettings(w, h, 5, 16), RenderChunk(0, 0, w, h))
    assert_same(on, off)
    scene.close()


def test_a_good_share_of_the_random_cases_is_culled_at_all():
    """(the test above proves nothing where no tile is empty)"""
    culled = 0
    for seed in range(48):
        w, h = 200, 136
        scene = Scene(lone_objects(5000 + seed, w, h), 0)
        scene.set_option("fine_split", 0)
        scene.render_chunk(RenderSettings(w, h, 1, 8), RenderChunk(0, 0, w, h), want_rgba=False)
        culled += scene.culled_pixels() > 0
        scene.close()
    print(f"{culled} of 48 random cases have empty tiles")
    assert culled >= 16


def test_against_the_oracle_and_in_every_partition():
    w, h = 320, 184
    b = SceneBuilder.sample("Dreadnaught", CameraConfiguration(w, h, lens_radius=0.1))
    scene, oracle = Scene(b, 0), OracleScene(b)
    scene.set_option("fine_split", 0)
    st = RenderSettings(w, h, 4, 16)
    whole = RenderChunk(0, 0, w, h)
    (rgb, rgba, cn, culled), _ = render_both(scene, st, whole, seed=3)
    assert culled > 0.3 * w * h
    ref, _, rcn = oracle.render_chunk(st, whole, ORC_RNG_COUNTER_STRATIFIED, 3, sum_segment=sample_segment())
    assert cn[0] == rcn.rays.primary and cn[3] == w * h * 16
    empty = (ref == 0).all(axis=2)
    assert np.array_equal(rgb[empty], ref[empty])
    assert np.abs(rgb - ref).max() < 2e-3 and np.abs(rgb - ref).mean() < 1e-6
    # an off-centre chunk, strips and tile columns of 3 ranks, two sample ranges: the same pixels and counters with the option on
    dev = "cuda:0"
    stream = torch.cuda.current_stream().cuda_stream
    ch = RenderChunk(40, 16, 264, 160)
    (c_rgb, _, c_cn, c_culled), c_off = render_both(scene, st, ch, seed=3)
    assert_same((c_rgb, _, c_cn, c_culled), c_off)
    assert np.array_equal(c_rgb, rgb[16:176, 40:304])
    total = np.zeros(4, dtype=np.int64)
    frame = np.zeros_like(rgb)
    for r in range(3):
        rows = strip_rows(h, 8, r, 3)
        buf = torch.zeros((rows, w, 3), dtype=torch.float32, device=dev); cnt = torch.zeros(4, dtype=torch.int64, device=dev)
        scene.render_strips_device(st, 8, r, 3, 3, buf.data_ptr(), None, cnt.data_ptr(), stream)
        torch.cuda.synchronize()
        got, k = buf.cpu().numpy(), 0
        for top in range(r * 8, h, 24):
            n = min(8, h - top); frame[top:top + n] = got[k:k + n]; k += n
        total += cnt.cpu().numpy()
    assert np.array_equal(frame, rgb) and tuple(total) == cn
    total[:] = 0
    for r in range(3):
        pw = tile_columns_width(w, r, 3)
        buf = torch.zeros((h, pw, 3), dtype=torch.float32, device=dev); cnt = torch.zeros(4, dtype=torch.int64, device=dev)
        scene.render_tile_columns_device(st, r, 3, 3, buf.data_ptr(), None, cnt.data_ptr(), stream)
        torch.cuda.synchronize()
        got = buf.cpu().numpy()
        for j, tx in enumerate(range(r, (w + 7) // 8, 3)):
            frame[:, tx * 8:tx * 8 + 8] = got[:, j * 8:j * 8 + 8]
        total += cnt.cpu().numpy()
    assert np.array_equal(frame, rgb) and tuple(total) == cn
    scene.close()


def test_single_round_claims_do_not_change_the_frame(tmp_path):
    """Launches that hand out few blocks per resident wave claim single ROUNDS of a block (render.hip: RT_SINGLE_ROUNDS_BELOW; decided on the device, instantiations
    without depth of field).  The switch is read once per process, so the C++ harness renders the frame three times in processes of their own: never, always, as decided."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "raytracer_amd", "rt_render")
    seen = {}
    for mode in ("0", "1", None):
        env = dict(os.environ)
        env.pop("RT_SINGLE_ROUNDS", None)
        if mode is not None:
            env["RT_SINGLE_ROUNDS"] = mode
        p = subprocess.run([exe, "--scene", "Dreadnaught", "--width", "640", "--height", "360", "--spp", "32", "--depth", "4", "--chunks", "2x1", "--threads", "1",
                            "--out", str(tmp_path / f"f{mode}.ppm")], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        info = json.loads(p.stdout.strip().splitlines()[-1])
        seen[mode] = (info["fnv1a"], info["rays"]["primary"], info["rays"]["shadow"], info["rays"]["samples"])
    assert seen["0"] == seen["1"] == seen[None], seen
