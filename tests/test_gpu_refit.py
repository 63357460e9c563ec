"""animated scenes.  rt_scene_update_transforms moves the entities of an existing scene and refits the TLAS
on the device; the hits must be those of a fresh rt_scene_create of the moved scene (and of the oracle), bit for bit."""
import time

import numpy as np
import pytest

from raytracer_amd import CameraConfiguration, RenderChunk, RenderSettings, RtError, Scene, SceneBuilder, _ffi, camera_new
from tests.oracle import OracleScene

pytestmark = pytest.mark.gpu


def same_hits(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def test_interceptor_spin_refit_equals_a_fresh_scene_on_8_angles():
    """raytracer-samples/src/scene_interceptor_spin.rs:34-40: the ship turns with the wall clock ("Time Ms" pins it here)."""
    w, h = 320, 180
    cfg = CameraConfiguration(w, h, lens_radius=0.1)
    st1, st, whole = RenderSettings(w, h, 8, 1), RenderSettings(w, h, 6, 4), RenderChunk(0, 0, w, h)
    spun = Scene(SceneBuilder.sample("Interceptor Spin", cfg, params={"Time Ms": 0}), 0)
    seen = []
    for k in range(1, 9):
        b = SceneBuilder.sample("Interceptor Spin", cfg, params={"Time Ms": 1250.0 * k - 300.0})
        spun.update_transforms(b)
        fresh = Scene(b, 0)
        hits, want = spun.primary_hits(st1, whole), fresh.primary_hits(st1, whole)
        assert same_hits(hits, want), k
        assert same_hits(hits, OracleScene(b).primary_hits(st1, whole)), k
        a, a8, ca = spun.render_chunk(st, whole, seed=7)
        f, f8, cf = fresh.render_chunk(st, whole, seed=7)
        assert np.array_equal(a, f) and np.array_equal(a8, f8) and ca.rays == cf.rays, k
        seen.append(hits[0].copy())
        fresh.close()
    assert not np.array_equal(seen[0], seen[3])          # the ship did turn
    spun.close()


def random_instances(seed, n=300):
    """n mesh / sphere entities under 1-2 translations and 0-3 rotations each."""
    r = np.random.default_rng(seed)
    b = SceneBuilder(camera_new((0.0, 6.0, 30.0), (0, 0, 0), 45.0, 16 / 9, 0.05, 30.0), _ffi.RT_SKY_DAY)
    m, t = b.add_lambertian(0.2), b.add_color_texture((0.7, 0.5, 0.3))
    tri = np.array([[[-1, 0, 0], [1, 0, 0], [0, 1.5, 0]], [[-1, 0, 0], [0, 1.5, 0], [0, 0.5, -1]], [[1, 0, 0], [0, 0.5, -1], [0, 1.5, 0]]], dtype=np.float32)
    mesh = b.add_mesh(tri)
    shape = np.random.default_rng(99)                      # the same entities, kinds and transform counts for every seed ...
    for k in range(n):
        kind, n_tr, n_rot = int(shape.integers(0, 2)), int(shape.integers(1, 3)), int(shape.integers(0, 4))
        tr = [tuple(r.uniform(-12, 12, 3)) for _ in range(n_tr)]             # ... under transforms drawn from `seed`
        ro = []
        for _ in range(n_rot):
            axis = r.normal(size=3); axis /= np.linalg.norm(axis)
            ro.append((tuple(axis), float(r.uniform(-3, 3))))
        if kind == 0:
            b.add_sphere_entity(float(shape.uniform(0.2, 0.8)), m, t, translations=tr, rotations=ro, id=k)
        else:
            b.add_mesh_entity(mesh, m, t, translations=tr, rotations=ro, id=k)
    b.add_directional_light((-1, -3, -1), 0.8)
    return b


def test_refit_of_many_moving_instances():
    w, h = 320, 180
    st1, st, whole = RenderSettings(w, h, 8, 1), RenderSettings(w, h, 4, 2), RenderChunk(0, 0, w, h)
    moving = Scene(random_instances(1), 0)
    for seed in (2, 3, 4):
        b = random_instances(seed)
        moving.update_transforms(b)
        fresh = Scene(b, 0)
        hits = moving.primary_hits(st1, whole)
        assert same_hits(hits, fresh.primary_hits(st1, whole)) and same_hits(hits, OracleScene(b).primary_hits(st1, whole)), seed
        assert (hits[2] >= 0).mean() > 0.05
        assert np.array_equal(moving.render_chunk(st, whole, seed=3)[0], fresh.render_chunk(st, whole, seed=3)[0]), seed
        fresh.close()
    moving.close()


def test_refit_latency_against_a_rebuild():
    """Frame-to-frame cost of moving 900 instances of the 11 101-triangle skeleton: refit against rt_scene_create."""
    cfg = CameraConfiguration(640, 360, lens_radius=0.1)
    a = SceneBuilder.sample("Doot Doot", cfg, params={"Count": 30, "Spacing": 30})
    b = SceneBuilder.sample("Doot Doot", cfg, params={"Count": 30, "Spacing": 31})
    t0 = time.perf_counter(); scene = Scene(a, 0); t_create = time.perf_counter() - t0
    scene.update_transforms(b)                               # (first call: kernels load)
    t0 = time.perf_counter()
    for _ in range(10):
        scene.update_transforms(a); scene.update_transforms(b)
    t_refit = (time.perf_counter() - t0) / 20
    st1, whole = RenderSettings(640, 360, 8, 1), RenderChunk(0, 0, 640, 360)
    fresh = Scene(b, 0)
    assert same_hits(scene.primary_hits(st1, whole), fresh.primary_hits(st1, whole))
    print(f"900 x 11 101 triangles: rt_scene_create {t_create * 1e3:.1f} ms, rt_scene_update_transforms {t_refit * 1e3:.3f} ms")
    assert t_refit < t_create      # (measured 0.15 ms against 11 ms; a loose bound: wall-clock on a shared box)
    scene.close(); fresh.close()


def test_update_transforms_rejects_other_shapes():
    scene = Scene(random_instances(1, n=20), 0)
    with pytest.raises(RtError, match="same|sizes") as e:
        scene.update_transforms(random_instances(1, n=21))
    assert e.value.code == _ffi.RT_E_INVALID
    scene.close()


def test_refit_of_a_hitable_with_an_infinite_box_that_is_not_a_plane():
    """implementation.rs:264 skips an entity's rotations when its hitable's box is infinite on all six sides - any hitable, not only a plane
    (the host build: box_is_infinite).  A sphere of infinite radius under a rotation must therefore come out of the refit with the infinite
    box a fresh rt_scene_create gives it, not with NaNs (the refit used to decide by the kind of hitable)."""
    def build(angle):
        b = SceneBuilder(camera_new((0.0, 2.0, 9.0), (0, 0, 0), 45.0, 16 / 9, 0.0, 9.0), _ffi.RT_SKY_DAY)
        m, t = b.add_lambertian(0.0), b.add_color_texture((0.6, 0.6, 0.3))
        for k in range(6):
            b.add_sphere_entity(0.5, m, t, translations=[(k - 2.5, 0.3 * k, -k)], rotations=[((0.0, 1.0, 0.0), angle + k)], id=k)
        b.add_sphere_entity(float("inf"), m, t, translations=[(0.0, 1.0, 0.0)], rotations=[((0.0, 0.0, 1.0), angle)], id=9)
        b.add_directional_light((-1, -3, -1), 0.8)
        return b
    w, h = 160, 90
    st1, st, whole = RenderSettings(w, h, 8, 1), RenderSettings(w, h, 4, 2), RenderChunk(0, 0, w, h)
    scene = Scene(build(0.3), 0)
    moved = build(1.1)
    scene.update_transforms(moved)
    fresh = Scene(moved, 0)
    assert same_hits(scene.primary_hits(st1, whole), fresh.primary_hits(st1, whole))
    assert (scene.primary_hits(st1, whole)[2] >= 0).mean() > 0.02
    assert np.array_equal(scene.render_chunk(st, whole, seed=2)[0], fresh.render_chunk(st, whole, seed=2)[0])
    scene.close(); fresh.close()
