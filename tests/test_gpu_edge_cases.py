"""GPU parity on the edges of the path: hand-built scenes through the generic builder
(Scene::add_*), ragged sample counts and chunks, depth limits, argument errors.
Same tolerances as tests/test_gpu_parity.py; every comparison is against the oracle."""
import numpy as np
import pytest

from raytracer_amd import RenderChunk, RenderSettings, RtError, Scene, SceneBuilder, _ffi, camera_new, sample_segment
from tests.oracle import ORC_RNG_COUNTER_STRATIFIED, OracleScene, use_lens_mode
from tests.test_gpu_parity import check_image, check_primary

pytestmark = pytest.mark.gpu

W, H = 96, 54


def camera(lens_radius=0.0, look_from=(0.0, 1.5, 6.0)):
    return camera_new(look_from, (0, 0.5, 0), 45.0, W / H, lens_radius, 6.0)


def both(b):
    return Scene(b, 0), OracleScene(b)


def compare(b, depth=5, spp=2, seed=9, chunk=None, lens_mode="stratified", **tol):
    scene, oracle = both(b)
    st, ch = RenderSettings(W, H, depth, spp), chunk or RenderChunk(0, 0, W, H)
    rng_mode = use_lens_mode(scene, lens_mode)
    rgb, rgba, cn = scene.render_chunk(st, ch, seed=seed)
    ref, ref_rgba, rcn = oracle.render_chunk(st, ch, rng_mode, seed, sum_segment=sample_segment())
    check_image(rgb, ref, **tol)
    assert (cn.primary, cn.samples) == (rcn.rays.primary, rcn.rays.samples)
    return rgb, cn, rcn


def basic_scene(sky=_ffi.RT_SKY_DAY, light=True, lens_radius=0.0):
    b = SceneBuilder(camera(lens_radius), sky)
    m, t = b.add_lambertian(), b.add_color_texture((0.8, 0.3, 0.2))
    g = b.add_color_texture((0.5, 0.5, 0.5))
    b.add_sphere_entity(1.0, m, t, translations=[(0, 1, 0)], id=7)
    b.add_sphere_entity(100.0, m, g, translations=[(0, -100, 0)])
    if light:
        b.add_directional_light((-1, -3, -1), 0.8)
    return b


@pytest.mark.parametrize("kinds", ["s", "m", "p", "sm", "sp", "mp", "smp"])
@pytest.mark.parametrize("xforms", [False, True])
@pytest.mark.parametrize("rich", [False, True])
def test_every_instantiation_of_the_render_kernel(kinds, xforms, rich, lens_mode):
    """The render kernel is compiled once per combination of scene traits (kinds of hitable: spheres, mesh instances,
    planes; general transforms; lamp lights / textures beyond colours and checkers) and a scene runs the
    instantiation without the code it does not need; each combination must give the oracle's picture.
    xforms = False: every entity is placed by exactly one translation."""
    b = SceneBuilder(camera(lens_radius=0.05), _ffi.RT_SKY_DAY)
    matte, shiny, glass = b.add_lambertian(0.3), b.add_specular(0.8, 0.1), b.add_dielectric(1.0, 0.1, 1.5)
    red, grey, white = b.add_color_texture((0.8, 0.3, 0.2)), b.add_color_texture((0.5, 0.5, 0.5)), b.add_color_texture((1, 1, 1))
    fancy = b.add_xyz_test_texture(2.0) if rich else b.add_checker_texture(4.0, red, white)
    turn = [((0.3, 0.9, 0.1), 0.5)] if xforms else []
    if "s" in kinds:
        b.add_sphere_entity(1.0, matte, red, translations=[(0, 1, 0)], id=7)
        b.add_sphere_entity(0.6, shiny, white, translations=[(-1.8, 0.6, 0.5)], rotations=turn, id=8)
        b.add_sphere_entity(0.7, glass, white, translations=[(1.7, 0.7, 0.8)], id=9)
        b.add_sphere_entity(100.0, matte, grey, translations=[(0, -100, 0)])
    if "m" in kinds:
        tri = np.array([[[-1, 0, 0], [1, 0, 0], [0, 1.5, 0]], [[-1, 0, 0], [0, 1.5, 0], [0, 0.5, -1]], [[1, 0, 0], [0, 0.5, -1], [0, 1.5, 0]]], dtype=np.float32)
        mesh = b.add_mesh(tri)
        b.add_mesh_entity(mesh, matte, fancy, translations=[(0.3, 0.0, 2.2)], rotations=turn, id=10)
        b.add_mesh_entity(mesh, glass if "s" not in kinds else shiny, white, translations=[(-1.2, 0.0, 2.8)], id=11)
    if "p" in kinds:
        b.add_plane_entity((0, 0, 1), matte, fancy, radius=1.5, translations=[(-3.0, 1.0, -2.0)], rotations=turn, id=12)
        if "s" not in kinds:
            b.add_plane_entity((0, 1, 0), matte, grey, translations=[(0, -0.01, 0)], id=13)
    if rich:
        b.add_lamp_light((2, 6, 4), (-2, -6, -4), 60.0, 40.0)
    b.add_directional_light((-1, -3, -1), 0.8)
    b.add_point_light((3, 4, 3), 20.0)
    scene, oracle = both(b)
    exact, cover = check_primary(scene, oracle, RenderSettings(W, H, 8, 1), RenderChunk(0, 0, W, H))
    assert exact == 1.0 and cover > 0.1
    compare(b, depth=6, spp=9, lens_mode=lens_mode)


def test_scene_without_lights():
    # implementation.rs:608-621: the light loop is empty, colour = reflections only (none here): black objects on sky
    rgb, cn, _ = compare(basic_scene(light=False))
    assert cn.shadow == 0 and rgb.max() > 0.5 and rgb.min() == 0.0


def test_depth_zero_is_sky_only():
    # cast_ray_recursive at limit 0 returns the sky without casting (implementation.rs:562-564)
    rgb, cn, rcn = compare(basic_scene(), depth=0)
    assert cn.primary == 0 and cn.shadow == 0 and rcn.rays.primary == 0
    assert rgb.min() > 0.4


def test_depth_one_and_maximum_depth():
    b = SceneBuilder(camera(), _ffi.RT_SKY_DAY)
    mirror, white = b.add_specular(0.95, 0.0), b.add_color_texture((1, 1, 1))
    glass, tint = b.add_dielectric(1.0, 0.1, 1.5), b.add_color_texture((0.9, 1.0, 0.9))
    b.add_plane_entity((1, 0, 0), mirror, white, translations=[(-2, 0, 0)])
    b.add_plane_entity((-1, 0, 0), mirror, white, translations=[(2, 0, 0)])
    b.add_sphere_entity(0.8, glass, tint, translations=[(0, 0.8, 0)])
    b.add_sphere_entity(100.0, b.add_lambertian(), b.add_color_texture((0.4, 0.4, 0.6)), translations=[(0, -100, 0)])
    b.add_point_light((0, 5, 3), 60.0)
    compare(b, depth=1)
    _, cn, rcn = compare(b, depth=31, spp=1)          # RT_MAX_REFLECTIONS: reflection/refraction budget split all the way down
    assert cn.secondary > 3 * cn.primary
    assert abs(cn.secondary - rcn.rays.secondary) <= max(4, 2e-3 * rcn.rays.secondary)
    scene, _ = both(b)
    with pytest.raises(RtError, match="max_reflections"):
        scene.render_chunk(RenderSettings(W, H, 32, 1), RenderChunk(0, 0, W, H))


@pytest.mark.parametrize("spp", [1, 7, 8, 9, 17, 40])
def test_ragged_sample_counts(spp, lens_mode):
    # the last sample segment is partial unless spp is a multiple of rt_sample_segment(); 40 = two groups of 16 + one of 8
    compare(basic_scene(lens_radius=0.2), spp=spp, lens_mode=lens_mode)


@pytest.mark.parametrize("chunk", [(0, 0, 1, 1), (W - 1, H - 1, 1, 1), (5, 0, 1, H), (0, 13, W, 1), (3, 2, 9, 10), (W - 13, H - 11, 13, 11)])
def test_ragged_chunks_are_cut_from_the_same_frame(chunk):
    b = basic_scene(lens_radius=0.2)
    scene, _ = both(b)
    st = RenderSettings(W, H, 4, 3)
    whole, whole8, _ = scene.render_chunk(st, RenderChunk(0, 0, W, H), seed=4)
    l, t, w, h = chunk
    part, part8, cn = scene.render_chunk(st, RenderChunk(l, t, w, h), seed=4)
    assert np.array_equal(part, whole[t:t + h, l:l + w]) and np.array_equal(part8, whole8[t:t + h, l:l + w])
    assert cn.samples == w * h * 3


def test_argument_errors():
    scene, _ = both(basic_scene())
    ok = RenderSettings(W, H, 3, 1)
    for st, ch, what in [(ok, RenderChunk(0, 0, 0, 4), "empty chunk"), (ok, RenderChunk(W - 3, 0, 4, 4), "outside"),
                         (ok, RenderChunk(0, H, 1, 1), "outside"), (RenderSettings(W, H, 3, 0), RenderChunk(0, 0, 4, 4), "must be > 0"),
                         (RenderSettings(70000, H, 3, 1), RenderChunk(0, 0, 4, 4), "65535")]:
        with pytest.raises(RtError, match=what):
            scene.render_chunk(st, ch)
    with pytest.raises(RtError, match="out of range"):
        Scene(basic_scene(), 99)
    # more than 2^28 pixels in one call (the work items of a launch are counted in 32 bits): refused before anything is touched
    import torch
    tiny = torch.zeros(16, dtype=torch.float32, device="cuda:0")
    with pytest.raises(RtError, match="2\\^28 pixels"):
        scene.render_chunk_device(RenderSettings(40000, 40000, 3, 1), RenderChunk(0, 0, 40000, 40000), 1, tiny.data_ptr(), None, 40000, 0, 0,
                                  None, torch.cuda.current_stream().cuda_stream)
    again, _, _ = scene.render_chunk(ok, RenderChunk(0, 0, 8, 8))       # the scene is still usable after errors
    assert again.shape == (8, 8, 3)


def test_stacked_transforms_on_every_hitable():
    # Entity::hit (implementation.rs:285-310): translations summed, rotations applied in order about arbitrary
    # axes, including exact identities (theta 0 and the axis-aligned case the kernel skips)
    b = SceneBuilder(camera(), _ffi.RT_SKY_DAY)
    m = b.add_lambertian(0.1)
    red, blue, chk = b.add_color_texture((0.9, 0.2, 0.2)), b.add_color_texture((0.2, 0.2, 0.9)), None
    chk = b.add_checker_texture(3.0, red, blue)
    tri = np.array([[[-1, 0, 0], [1, 0, 0], [0, 1.5, 0]], [[-1, 0, 0], [0, 1.5, 0], [0, 0.5, -1]]], dtype=np.float32)
    mesh = b.add_mesh(tri, uvs=[[(0, 0), (1, 0), (0.5, 1)], [(0, 0), (0.5, 1), (1, 1)]])
    axis = (0.3, 0.9, 0.1)
    b.add_mesh_entity(mesh, m, b.add_uv_test_texture(), translations=[(1.5, 0, 0), (0, 0.2, 0.3)],
                      rotations=[((0, 1, 0), 0.7), (axis, -0.4), ((1, 0, 0), 0.0)], id=1)
    b.add_mesh_entity(mesh, m, red, translations=[(-1.5, 0.1, 0)], rotations=[((0, 0, 1), 0.0), ((0, 1, 0), 2.5)], id=2)
    b.add_sphere_entity(0.5, m, b.add_xyz_test_texture(2.0), translations=[(0, 0.5, 1), (0.1, 0, 0), (0, 0.2, 0)], rotations=[(axis, 1.1)], id=3)
    b.add_plane_entity((0, 1, 0), m, chk, translations=[(0, -0.2, 0), (0.5, 0, 0)], rotations=[((0, 1, 0), 0.6)], id=4)
    b.add_plane_entity((0, 0, 1), m, blue, radius=1.2, translations=[(0, 1.0, -2.5)], rotations=[((1, 0, 0), -0.3)], id=5)
    b.add_lamp_light((2, 6, 4), (-2, -6, -4), 90.0, 40.0)
    b.add_point_light((-3, 4, 2), 30.0)
    scene, oracle = both(b)
    exact, cover = check_primary(scene, oracle, RenderSettings(W, H, 8, 1), RenderChunk(0, 0, W, H))
    assert exact > 0.999 and cover > 0.5
    compare(b, depth=6, spp=3)


def test_degenerate_geometry_does_not_disturb_the_rest():
    # a zero-area triangle (normal = 0: |denominator| < 1e-6 rejects it, plane.rs:7-10) and a zero-radius sphere
    b = SceneBuilder(camera(), _ffi.RT_SKY_BLACK)
    m, t = b.add_lambertian(), b.add_color_texture((0.7, 0.7, 0.2))
    tri = np.array([[[0, 0, 0], [1, 1, 1], [2, 2, 2]], [[-1, 0, 0], [1, 0, 0], [0, 1.5, 0]]], dtype=np.float32)
    b.add_mesh_entity(b.add_mesh(tri), m, t)
    b.add_sphere_entity(0.0, m, t, translations=[(0, 0.5, 2)])
    b.add_point_light((0, 3, 5), 40.0)
    scene, oracle = both(b)
    check_primary(scene, oracle, RenderSettings(W, H, 4, 1), RenderChunk(0, 0, W, H))
    rgb, _, _ = compare(b, depth=3, spp=1)
    assert np.isfinite(rgb).all()


def test_axis_aligned_rays_on_box_planes():
    """Rays with exactly zero direction components whose origin lies exactly on a box plane: (plane - o) * (1/0) =
    0 * inf = NaN, which implementation.rs:102-115's comparisons ignore; so must v_max3 / v_min3."""
    import os
    cam = camera_new((1.0, 0.0, 5.0), (1.0, 0.0, 0.0), 40.0, 64 / 32, 0.0, 5.0)       # u = 0.5 (column 32) and v = 0.5 (row 16): d.x == 0, d.y == 0 exactly
    b = SceneBuilder(cam, _ffi.RT_SKY_DAY)
    m, tx = b.add_lambertian(), b.add_color_texture((0.3, 0.6, 0.9))
    cube, _ = b.load_obj(os.path.join(_ffi.ASSET_MESH_DIR, "simple", "cube.obj"))        # spans [-1, 1]^3: x = 1 is a face plane
    b.add_mesh_entity(cube, m, tx)
    b.add_mesh_entity(cube, m, tx, translations=[(2.0, 0.0, 0.0)])                        # shares that plane from the other side
    b.add_sphere_entity(1.0, m, tx, translations=[(1.0, 2.0, 0.0)])
    b.add_point_light((1, 4, 6), 50.0)
    scene, oracle = both(b)
    w, h = 64, 32
    st, ch = RenderSettings(w, h, 3, 1), RenderChunk(0, 0, w, h)
    t, n, e, tri = scene.primary_hits(st, ch)
    rt, rn, re, rtri = oracle.primary_hits(st, ch)
    assert np.array_equal(e, re) and np.array_equal(tri, rtri) and np.array_equal(t, rt) and np.array_equal(n, rn)
    assert (re >= 0).mean() > 0.1 and (re[:, 32] >= 0).any() and (re[16, :] >= 0).any()     # the zero-component rays hit things
    rgb, _, cn = scene.render_chunk(st, ch, seed=2)
    ref, _, rcn = oracle.render_chunk(st, ch, ORC_RNG_COUNTER_STRATIFIED, 2, sum_segment=sample_segment())
    assert np.abs(rgb - ref).max() < 2e-5 and cn.rays == rcn.rays.primary + rcn.rays.secondary + rcn.rays.shadow


def test_point_cloud_large_tlas():
    # scene_point_cloud.rs: 200 000 spheres -> a deep TLAS (stack depth, tie ranks at scale)
    from raytracer_amd import CameraConfiguration
    b = SceneBuilder.sample("Point Cloud", CameraConfiguration(W, H, lens_radius=0.0), params={"Cloud Point Count": 200000})
    scene, oracle = both(b)
    assert scene.info().tlas_depth >= 17
    exact, _ = check_primary(scene, oracle, RenderSettings(W, H, 4, 1), RenderChunk(0, 0, W, H))
    assert exact > 0.999


def test_traversal_stack_overflow_is_an_error_not_a_wrong_image():
    """The per-lane traversal stack is sized from the trees at scene creation; if it is ever too small the kernel must not
    drop the subtree silently: a sticky per-scene error word turns the call (and every later one) into RT_E_DEPTH.
    Forced here by running the kernels with a 1-entry stack (RT_OPT_DEBUG_STACK_ENTRIES)."""
    import torch
    from raytracer_amd import CameraConfiguration
    b = SceneBuilder.sample("Random Spheres", CameraConfiguration(W, H, lens_radius=0.1))
    scene, _ = both(b)
    st, ch = RenderSettings(W, H, 4, 2), RenderChunk(0, 0, W, H)
    good, _, _ = scene.render_chunk(st, ch, seed=1)
    scene.set_option("debug_stack_entries", 1)
    with pytest.raises(RtError, match="traversal stack") as e:
        scene.render_chunk(st, ch, seed=1)
    assert e.value.code == _ffi.RT_E_DEPTH
    with pytest.raises(RtError, match="traversal stack"):          # sticky: the blocking entry points ...
        scene.primary_hits(st, ch)
    out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    with pytest.raises(RtError, match="traversal stack"):          # ... and the enqueue-only ones refuse further work
        scene.render_chunk_device(st, ch, 1, out.data_ptr(), None, W, 0, 0, None, torch.cuda.current_stream().cuda_stream)
    scene.set_option("debug_stack_entries", 0)                     # as built again; re-arms the error word
    again, _, _ = scene.render_chunk(st, ch, seed=1)
    assert np.array_equal(again, good)
    # an overflow inside an enqueued launch surfaces at the next call on the scene
    scene.set_option("debug_stack_entries", 1)
    scene.render_chunk_device(st, ch, 1, out.data_ptr(), None, W, 0, 0, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    with pytest.raises(RtError, match="traversal stack"):
        scene.render_chunk_device(st, ch, 1, out.data_ptr(), None, W, 0, 0, None, torch.cuda.current_stream().cuda_stream)
    scene.set_option("debug_stack_entries", 0)


def test_strip_height_must_hold_whole_ray_tiles():
    """rt_render_strips_device maps an 8x8 ray tile through the strip interleave by its first row, so a strip height that is
    not a multiple of 8 would scatter rows 4..7 of a tile to wrong image rows: refused (RT_E_INVALID), 16 works."""
    import torch
    from raytracer_amd import strip_rows
    scene, _ = both(basic_scene())
    st = RenderSettings(W, H, 3, 2)
    whole, _, _ = scene.render_chunk(st, RenderChunk(0, 0, W, H), seed=3)
    stream = torch.cuda.current_stream().cuda_stream
    buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    for bad in (4, 12, 7):
        with pytest.raises(RtError, match="multiple of 8") as e:
            scene.render_strips_device(st, bad, 0, 2, 3, buf.data_ptr(), None, None, stream)
        assert e.value.code == _ffi.RT_E_INVALID
    frame = np.zeros_like(whole)
    for r in range(2):
        rows = strip_rows(H, 16, r, 2)
        scene.render_strips_device(st, 16, r, 2, 3, buf.data_ptr(), None, None, stream)
        torch.cuda.synchronize()
        idx = [y for top in range(r * 16, H, 32) for y in range(top, min(top + 16, H))]
        assert len(idx) == rows
        frame[idx] = buf[:rows].cpu().numpy()
    assert np.array_equal(frame, whole)


def test_scene_options_are_validated():
    scene, _ = both(basic_scene())
    for name, bad in (("lens_sampler", 2), ("fine_split", 3), ("round_units", 3), ("lens_coc_px", -1.0), ("debug_stack_entries", -5)):
        with pytest.raises(RtError) as e:
            scene.set_option(name, bad)
        assert e.value.code == _ffi.RT_E_INVALID
    with pytest.raises(RtError, match="unknown option"):
        scene._check(scene._lib.rt_scene_set_option(scene._h, 77, 0.0))


@pytest.mark.timeout(120)
@pytest.mark.parametrize("what", ["nan position", "inf position", "nan radius", "negative radius", "huge radius", "nan angle", "1e20 position"])
def test_non_finite_scene_data_neither_hangs_nor_poisons_the_frame(what):
    """The reference does not validate scene data and neither does rt_scene_create; whatever the values, the build returns, every loop of the
    kernel ends and the pixels are finite (cast_ray clamps each sample, types.rs:46-50; a NaN compares false everywhere)."""
    from raytracer_amd import camera_new
    mut = {"nan position": {"pos": (float("nan"), 0.0, 0.0)}, "inf position": {"pos": (float("inf"), 0.0, 0.0)}, "nan radius": {"radius": float("nan")},
           "negative radius": {"radius": -1.0}, "huge radius": {"radius": 1e30}, "nan angle": {"theta": float("nan")}, "1e20 position": {"pos": (1e20, 0.0, 0.0)}}[what]
    b = SceneBuilder(camera_new((0.0, 2.0, 10.0), (0, 0, 0), 45.0, 16 / 9, 0.05, 10.0), _ffi.RT_SKY_DAY)
    m, g, t = b.add_lambertian(0.3), b.add_dielectric(1.0, 0.0, 1.5), b.add_color_texture((0.5, 0.5, 0.5))
    mesh = b.add_mesh(np.array([[[-1, 0, 0], [1, 0, 0], [0, 1.5, 0]]], dtype=np.float32))
    for k in range(40):
        pos = mut.get("pos") if k == 7 and "pos" in mut else (k * 0.5 - 10.0, 0.0, 0.0)
        if k % 2:
            b.add_sphere_entity(mut.get("radius", 0.5) if k == 9 else 0.5, g if k % 3 == 0 else m, t, translations=[pos], id=k)
        else:
            b.add_mesh_entity(mesh, m, t, translations=[pos], rotations=[((0.0, 1.0, 0.0), mut.get("theta", 0.3) if k == 8 else 0.3)], id=k)
    b.add_point_light((0, 10, 0), 100.0)
    scene = Scene(b, 0)
    rgb, rgba, cn = scene.render_chunk(RenderSettings(160, 90, 8, 8), RenderChunk(0, 0, 160, 90), seed=1)
    assert np.isfinite(rgb).all() and np.abs(rgb).max() <= 1.0 and cn.primary == 160 * 90 * 8
    scene.close()
