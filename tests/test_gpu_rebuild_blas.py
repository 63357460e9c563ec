"""SURVEY.md 8 f4, the rest: rt_scene_rebuild_blas takes new vertex positions for ONE mesh of an existing scene and builds a new tree
over its triangles on the device (a linear BVH over Morton codes, raytracer_amd/csrc/tlas_build.h), then refits the boxes of the mesh's
instances and the TLAS above them.  The reference builds a mesh's BVH on the CPU whenever a MeshObject is made (MeshBvhRoot::new,
mesh.rs:84-90).  The tree is not the reference's, so this is the nearest-hit gate of tests/test_gpu_rebuild.py: t / normal / ids must be
those of a fresh rt_scene_create of the deformed scene (= the reference's trees) and of the oracle, except where a ray grazes the
boundary of a leaf box - bound 1e-4 of the pixels, and there t may only differ by 1e-5 relative.  Exact ties in t keep the visit ranks
of the tree rt_scene_create built."""
import os
import time

import numpy as np
import pytest

from raytracer_amd import CameraConfiguration, RenderChunk, RenderSettings, RtError, Scene, SceneBuilder, _ffi, camera_new
from tests.oracle import OracleScene
from tests.test_gpu_rebuild import nearest_hit_gate

pytestmark = pytest.mark.gpu


def obj_triangles(path):
    """(N, 3, 3) positions of an OBJ's triangles in the order the ingest emits them (through the product's own loader)."""
    b = SceneBuilder(camera_new((0.0, 0.0, 5.0), (0, 0, 0), 45.0, 1.0, 0.0, 5.0), _ffi.RT_SKY_DAY)
    mesh, _ = b.load_obj(path)
    m = b.desc().contents.meshes[mesh]
    out = np.zeros((m.tri_count, 3, 3), dtype=np.float32)
    for i in range(m.tri_count):
        t = m.tris[i]
        out[i] = [[t.a.x, t.a.y, t.a.z], [t.b.x, t.b.y, t.b.z], [t.c.x, t.c.y, t.c.z]]
    return out


def scene_with(tris, n_instances=5, seed=3, extra_mesh=None):
    """A ground sphere, `n_instances` rotated instances of the mesh `tris`, one of a second (never rebuilt) mesh, a lamp and a point light."""
    r = np.random.default_rng(seed)
    b = SceneBuilder(camera_new((0.0, 3.0, 12.0), (0, 0.8, 0), 45.0, 16 / 9, 0.05, 12.0), _ffi.RT_SKY_DAY)
    matte, metal = b.add_lambertian(0.0), b.add_specular(0.7, 0.0)
    red, grey = b.add_color_texture((0.8, 0.3, 0.2)), b.add_color_texture((0.6, 0.6, 0.6))
    b.add_sphere_entity(1000.0, matte, grey, translations=[(0.0, -1001.0, 0.0)], id=100)
    mesh = b.add_mesh(tris)
    for k in range(n_instances):
        axis = r.normal(size=3); axis /= np.linalg.norm(axis)
        b.add_mesh_entity(mesh, metal if k % 2 else matte, red, translations=[tuple(r.uniform(-4, 4, 3) * (1, 0.2, 1) + (0, 0.8, 0))],
                          rotations=[(tuple(axis), float(r.uniform(-3, 3)))], id=k)
    if extra_mesh is not None:
        other = b.add_mesh(extra_mesh)
        b.add_mesh_entity(other, matte, grey, translations=[(0.0, 0.5, -3.0)], id=50)
    b.add_point_light((4.0, 8.0, 6.0), 60.0)
    b.add_directional_light((-1, -3, -1), 0.5)
    return b


def wobble(tris, phase):
    """Every vertex displaced along x and y by a wave in z: shared vertices stay shared, the mesh changes its bounds."""
    out = tris.copy()
    out[..., 0] += 0.15 * np.sin(3.0 * tris[..., 2] + phase).astype(np.float32)
    out[..., 1] += 0.10 * np.cos(2.0 * tris[..., 0] + phase).astype(np.float32)
    return out


def first_existing(*names):
    for root, _, files in os.walk(_ffi.ASSET_MESH_DIR):
        for f in files:
            if f.lower() in names:
                return os.path.join(root, f)
    raise FileNotFoundError(names)


def test_a_deforming_mesh_finds_the_reference_hits():
    suzanne = obj_triangles(first_existing("suzanne.obj"))
    cube = obj_triangles(first_existing("cube.obj")) * 0.7
    assert len(suzanne) == 968
    w, h = 320, 180
    st1, st, whole = RenderSettings(w, h, 8, 1), RenderSettings(w, h, 5, 2), RenderChunk(0, 0, w, h)
    live = Scene(scene_with(suzanne, extra_mesh=cube), 0)
    worst = 0.0
    for phase in (0.7, 1.9, 3.1, 0.7):
        b = scene_with(wobble(suzanne, phase), extra_mesh=cube)
        live.rebuild_blas(b, 0)
        fresh = Scene(b, 0)
        hits = live.primary_hits(st1, whole)
        worst = max(worst, nearest_hit_gate(hits, fresh.primary_hits(st1, whole), ("fresh", phase)))
        worst = max(worst, nearest_hit_gate(hits, OracleScene(b).primary_hits(st1, whole), ("oracle", phase)))
        assert (hits[3] >= 0).mean() > 0.03                      # the mesh is in view
        a, _, ca = live.render_chunk(st, whole, seed=3)
        f, _, cf = fresh.render_chunk(st, whole, seed=3)
        assert (np.abs(a - f).max(axis=-1) > 2e-5).mean() <= 2e-3 and ca.primary == cf.primary
        fresh.close()
    print(f"rebuilt BLAS, 968 triangles x 5 instances: primary hits differ from the reference trees' on {worst * 100:.4f} % of the pixels at most")
    # the other rebuilds still work on top: the instances move (device-built TLAS), then the refit of that tree
    moved = scene_with(wobble(suzanne, 0.7), seed=9, extra_mesh=cube)
    live.rebuild_tlas(moved)
    fresh = Scene(moved, 0)
    nearest_hit_gate(live.primary_hits(st1, whole), fresh.primary_hits(st1, whole), "rebuild_tlas after rebuild_blas")
    live.rebuild_blas(scene_with(wobble(suzanne, 2.2), seed=9, extra_mesh=cube), 0)
    fresh2 = Scene(scene_with(wobble(suzanne, 2.2), seed=9, extra_mesh=cube), 0)
    nearest_hit_gate(live.primary_hits(st1, whole), fresh2.primary_hits(st1, whole), "rebuild_blas on a device-built TLAS")
    for s in (live, fresh, fresh2):
        s.close()


def test_the_skeleton_of_config_5_rebuilt_in_place():
    """BASELINE configs[4]: 900 instances of SKELETON.obj (11 101 triangles, coincident faces whose exact ties the visit ranks decide).  The mesh
    rebuilt on the device with its own triangles must give the frame of the host-built scene: every primary hit of the 4K frame."""
    w, h = 3840, 2160
    cfg = CameraConfiguration(w, h, lens_radius=0.1)
    b = SceneBuilder.sample("Doot Doot", cfg, params={"Count": 30, "Spacing": 30})
    scene, built = Scene(b, 0), Scene(b, 0)
    assert scene.info().triangle_count == 11101
    t0 = time.perf_counter(); scene.rebuild_blas(b, 0); t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(10):
        scene.rebuild_blas(b, 0)
    t_again = (time.perf_counter() - t0) / 10
    st1 = RenderSettings(w, h, 8, 1)
    differing = 0
    for top in range(0, h, 540):                                  # (four bands: the host arrays of a whole 4K frame are 200 MB)
        band = RenderChunk(0, top, w, 540)
        got, want = scene.primary_hits(st1, band), built.primary_hits(st1, band)
        differing += int(((got[0] != want[0]) | (got[2] != want[2]) | (got[3] != want[3]) | np.any(got[1] != want[1], axis=-1)).sum())
    print(f"SKELETON.obj, 11 101 triangles: rt_scene_rebuild_blas {t_first * 1e3:.1f} ms (first call, allocates), {t_again * 1e3:.3f} ms after; "
          f"{differing} of {w * h} primary hits of the 4K frame differ from the host-built tree's; BLAS depth {scene.info().blas_depth}")
    assert differing <= 1e-4 * w * h
    small = RenderSettings(w, h, 8, 1)
    ch = RenderChunk(1700, 900, 480, 270)
    nearest_hit_gate(scene.primary_hits(small, ch), OracleScene(b).primary_hits(small, ch), "oracle, 480 x 270 of the 4K frame")
    a, _, ca = scene.render_chunk(RenderSettings(w, h, 8, 4), ch, seed=12345)
    f, _, cf = built.render_chunk(RenderSettings(w, h, 8, 4), ch, seed=12345)
    assert (np.abs(a - f).max(axis=-1) > 2e-5).mean() <= 2e-3 and ca.primary == cf.primary
    assert t_again < 0.05                                         # (measured well under a millisecond of GPU work; a loose wall-clock bound)
    scene.close(); built.close()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 17])
def test_rebuild_of_tiny_meshes(n):
    r = np.random.default_rng(n)
    tris = r.uniform(-1, 1, (n, 3, 3)).astype(np.float32)
    w, h = 128, 72
    st1, whole = RenderSettings(w, h, 8, 1), RenderChunk(0, 0, w, h)
    scene = Scene(scene_with(tris, n_instances=3), 0)
    moved = (tris + r.uniform(-0.2, 0.2, tris.shape)).astype(np.float32)
    b = scene_with(moved, n_instances=3)
    scene.rebuild_blas(b, 0)
    fresh = Scene(b, 0)
    nearest_hit_gate(scene.primary_hits(st1, whole), fresh.primary_hits(st1, whole), n)
    scene.close(); fresh.close()


def test_rebuild_blas_refusals_leave_the_scene_as_it_was():
    suzanne = obj_triangles(first_existing("suzanne.obj"))
    w, h = 200, 112
    st, whole = RenderSettings(w, h, 5, 2), RenderChunk(0, 0, w, h)
    a = scene_with(suzanne)
    scene = Scene(a, 0)
    before = scene.render_chunk(st, whole, seed=4)
    with pytest.raises(RtError, match="triangle count") as e:
        scene.rebuild_blas(scene_with(suzanne[:100]), 0)
    assert e.value.code == _ffi.RT_E_INVALID
    with pytest.raises(RtError, match="no such mesh"):
        scene.rebuild_blas(a, 7, builder_mesh_index=0)
    scene.set_option("debug_rebuild_stack_limit", 3)
    with pytest.raises(RtError, match="too deep") as e:
        scene.rebuild_blas(scene_with(wobble(suzanne, 1.0)), 0)
    assert e.value.code == _ffi.RT_E_DEPTH
    after = scene.render_chunk(st, whole, seed=4)
    assert np.array_equal(before[0], after[0]) and before[2].rays == after[2].rays
    scene.set_option("debug_rebuild_stack_limit", 0)
    scene.rebuild_blas(scene_with(wobble(suzanne, 1.0)), 0)
    assert not np.array_equal(before[0], scene.render_chunk(st, whole, seed=4)[0])
    scene.close()
