"""One-off sweep of the BLAS side: random triangle soups of 1...4000 triangles with shared vertices, exact duplicates, degenerate and
axis-aligned coincident faces (the SKELETON.obj case of DESIGN.md 3), instanced 1-6 times under unit-axis rotations; primary hits
(t, normal, entity, triangle) must equal the oracle's bit for bit, and a short render must count the same primary rays.
   usage: python tools/mesh_sweep.py <first seed> <count>"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import RenderChunk, RenderSettings, Scene, SceneBuilder, _ffi, camera_new
from tests.oracle import OracleScene


def soup(seed):
    r = np.random.default_rng(seed)
    b = SceneBuilder(camera_new(tuple(r.uniform(-1, 1, 3) * 2 + (0, 1, 9)), (0, 0, 0), 45.0, 16 / 9, 0.0, 9.0), _ffi.RT_SKY_DAY)
    m, t = b.add_lambertian(0.0), b.add_color_texture((0.6, 0.5, 0.4))
    nt = int(r.choice([1, 2, 3, 9, 40, 200, 1000, 4000]))
    nv = max(3, int(nt * r.uniform(0.3, 1.5)))
    verts = r.uniform(-1, 1, (nv, 3)).astype(np.float32)
    if r.integers(0, 2):
        verts = np.round(verts * 8) / 8                      # a lattice: coplanar and coincident faces, exact ties in t
    tri = verts[r.integers(0, nv, (nt, 3))].copy()
    k = nt // 5
    if k:
        tri[-k:] = tri[:k]                                   # exact duplicates
    if nt > 10:
        tri[5, 2] = tri[5, 1]                                # a degenerate triangle
        tri[6] = tri[6][[0, 2, 1]]                           # ... and one with the other winding
    mesh = b.add_mesh(tri)
    for i in range(int(r.integers(1, 7))):
        ro = []
        for _ in range(int(r.integers(0, 3))):
            axis = r.normal(size=3); axis /= np.linalg.norm(axis)
            ro.append((tuple(axis), float(r.uniform(-3, 3))))
        same = i > 0 and r.integers(0, 4) == 0                # sometimes two instances in the very same place: ties between entities
        b.add_mesh_entity(mesh, m, t, translations=[(0.0, 0.0, 0.0) if same else tuple(r.uniform(-2.5, 2.5, 3))], rotations=[] if same else ro, id=i)
    b.add_directional_light((-1, -3, -1), 0.8)
    return b, nt


first, count = int(sys.argv[1]), int(sys.argv[2])
bad, t0 = [], time.perf_counter()
w, h = 160, 90
st1, st, whole = RenderSettings(w, h, 8, 1), RenderSettings(w, h, 3, 2), RenderChunk(0, 0, w, h)
for k, seed in enumerate(range(first, first + count)):
    try:
        b, nt = soup(seed)
        scene, oracle = Scene(b, 0), OracleScene(b)
        got, want = scene.primary_hits(st1, whole), oracle.primary_hits(st1, whole)
        for name, x, y in zip(("t", "normal", "entity", "triangle"), got, want):
            assert np.array_equal(x, y), f"seed {seed} ({nt} triangles): {name} differs in {int((x != y).sum())} values"
        _, _, cn = scene.render_chunk(st, whole, seed=seed)
        _, _, rcn = oracle.render_chunk(st, whole, 2, seed, sum_segment=8)
        assert cn.primary == rcn.rays.primary and abs(cn.shadow - rcn.rays.shadow) <= max(4, 2e-3 * cn.shadow), f"seed {seed}: ray counts {cn.primary}/{cn.shadow} vs {rcn.rays.primary}/{rcn.rays.shadow}"
        scene.close()
    except AssertionError as e:
        bad.append(seed); print("FAILED", str(e)[:300].replace("\n", " "), flush=True)
    except Exception:
        bad.append(seed); print("ERROR seed", seed, traceback.format_exc().splitlines()[-4:], flush=True)
    if (k + 1) % 20 == 0:
        print(f"{k + 1} cases, {len(bad)} failures, {time.perf_counter() - t0:.1f} s", flush=True)
print("done:", count, "cases, failures:", bad, flush=True)
sys.exit(1 if bad else 0)
