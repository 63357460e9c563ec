"""Randomised parity: seeded random scenes through the generic builder, GPU against the oracle.

What it is after: the rules that make the GPU's shrinking, near-first traversal return the reference's
winner (first candidate in the reference's visit order on exact ties in t - implementation.rs:415,
mesh.rs:96-98), under arbitrary mixes of hitables, transforms and DUPLICATED geometry (coincident
spheres, coplanar duplicate triangles, shared mesh instances at the same place), which produce exact ties.
A quarter of the rotations use NON-UNIT axes, outside rotate_about_axis' stated contract (types.rs:190): the
reference's entity boxes then do not bound what Entity::hit finds, and the product must still return the
reference's answer (it stops shrinking the interval for TLAS boxes in such scenes - loose_boxes).
Found by this test, kept as seeds 46 and 60.
"""
import numpy as np
import pytest

from raytracer_amd import RenderChunk, RenderSettings, Scene, SceneBuilder, _ffi, camera_new, sample_segment
from tests.oracle import ORC_RNG_COUNTER_STRATIFIED, OracleScene
from tests.test_gpu_parity import check_image

pytestmark = pytest.mark.gpu

W, H = 80, 48


def random_scene(seed):
    r = np.random.default_rng(seed)
    cam = camera_new(tuple(r.uniform(-1, 1, 3) + (0, 2, 8)), (0, 0.5, 0), 50.0, W / H, float(r.choice([0.0, 0.15])), 8.0)
    b = SceneBuilder(cam, int(r.integers(0, 2)))
    mats = [b.add_lambertian(0.0), b.add_lambertian(float(r.uniform(0.05, 0.5))), b.add_specular(float(r.uniform(0.3, 1.0)), float(r.choice([0.0, 0.2]))),
            b.add_dielectric(float(r.uniform(0.2, 1.0)), float(r.uniform(0, 0.5)), float(r.choice([1.5, 0.8, 1.33])))]
    cols = [b.add_color_texture(tuple(r.uniform(0.1, 1.0, 3))) for _ in range(4)]
    texs = cols + [b.add_checker_texture(float(r.uniform(1, 8)), cols[0], cols[1]), b.add_uv_test_texture(), b.add_xyz_test_texture(3.0)]

    def xform():
        tr = [tuple(r.uniform(-3, 3, 3) * (1, 0.5, 1) + (0, 0.7, 0)) for _ in range(int(r.integers(0, 3)))]
        ro = []
        for _ in range(int(r.integers(0, 3))):
            axis = [(1, 0, 0), (0, 1, 0), (0, 0, 1), tuple(r.uniform(-1, 1, 3))][int(r.integers(0, 4))]
            ro.append((axis, float(r.choice([0.0, float(r.uniform(-3, 3))]))))
        return tr, ro

    n_meshes = int(r.integers(1, 4))
    meshes = []
    for _ in range(n_meshes):
        nt = int(r.integers(1, 40))
        verts = r.uniform(-1, 1, (max(4, nt // 2), 3)).astype(np.float32)          # shared vertices: edges and corners coincide
        idx = r.integers(0, len(verts), (nt, 3))
        tri = verts[idx]
        if nt > 3:
            tri[-1] = tri[0]                                                         # an exact duplicate triangle
        meshes.append(b.add_mesh(tri, uvs=[[tuple(r.uniform(0, 1, 2)) for _ in range(3)] for _ in range(nt)]))
    placed = []
    for k in range(int(r.integers(3, 40))):
        kind = int(r.integers(0, 10))
        m, t = mats[int(r.integers(0, 4))], texs[int(r.integers(0, len(texs)))]
        tr, ro = xform()
        if placed and kind == 9:                                                      # coincident copy of an earlier entity, other material
            kind, rad, mesh, tr, ro = placed[int(r.integers(0, len(placed)))]
        else:
            rad, mesh = float(r.uniform(0.1, 1.2)), meshes[int(r.integers(0, n_meshes))]
            placed.append((kind, rad, mesh, tr, ro))
        if kind <= 4 or kind == 9:
            b.add_sphere_entity(rad, m, t, translations=tr, rotations=ro, id=k)
        elif kind <= 7:
            b.add_mesh_entity(mesh, m, t, translations=tr, rotations=ro, id=k)
        else:
            b.add_plane_entity(tuple(r.uniform(-1, 1, 3) + (0, 1.5, 0)), m, t, radius=None if r.random() < 0.5 else float(r.uniform(1, 4)),
                               translations=tr or [(0, -1, 0)], rotations=ro)
    for _ in range(int(r.integers(0, 3))):
        which = int(r.integers(0, 3))
        if which == 0:
            b.add_point_light(tuple(r.uniform(-4, 4, 3) + (0, 6, 0)), float(r.uniform(20, 80)))
        elif which == 1:
            o = r.uniform(-4, 4, 3) + (0, 6, 0)
            b.add_lamp_light(tuple(o), tuple(-o), float(r.uniform(40, 120)), float(r.uniform(10, 60)))
        else:
            b.add_directional_light(tuple(r.uniform(-1, 1, 3) - (0, 2, 0)), float(r.uniform(0.3, 1.0)))
    return b


@pytest.mark.parametrize("seed", range(160))
def test_random_scene_matches_oracle(seed):
    b = random_scene(1000 + seed)
    scene, oracle = Scene(b, 0), OracleScene(b)
    whole = RenderChunk(0, 0, W, H)
    # gate 1: primary hits, ids equal everywhere, t and normal bit-identical
    st1 = RenderSettings(W, H, 8, 1)
    t, n, e, tri = scene.primary_hits(st1, whole)
    rt, rn, re, rtri = oracle.primary_hits(st1, whole)
    assert np.array_equal(e, re) and np.array_equal(tri, rtri)
    assert np.array_equal(t, rt) and np.array_equal(n, rn)
    # the image: recursion, shadows through glass, every texture kind
    st = RenderSettings(W, H, 5, 3)
    rgb, _, cn = scene.render_chunk(st, whole, seed=seed)
    ref, _, rcn = oracle.render_chunk(st, whole, ORC_RNG_COUNTER_STRATIFIED, seed, sum_segment=sample_segment())
    # a decision flip of one secondary ray (last-ulp libm difference) changes a whole pixel: allowed as an
    # outlier, but on an 80x48 image one such pixel moves the mean by 1e-4, hence the looser mean bound
    check_image(rgb, ref, outlier_frac=5e-3, mean_atol=3e-4)
    assert cn.primary == rcn.rays.primary
    assert abs(cn.rays - (rcn.rays.primary + rcn.rays.secondary + rcn.rays.shadow)) <= max(8, 5e-3 * cn.rays)


def test_shadow_ray_through_distant_glass_terminates():
    """Seed 1245 (found by tools/fuzz_sweep.py): a shadow ray meets a refractive triangle 195 732 units away, where t + BIAS rounds back to
    t; the reference's loop (implementation.rs:535-549) then finds the same occluder for ever.  Kernel and oracle step one ulp instead."""
    test_random_scene_matches_oracle(245)
