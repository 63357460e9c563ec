"""One-off sweep behind tests/test_gpu_cull.py: N random scenes under a black or a day sky (spheres and mesh instances under stacked translations and rotations about
random UNIT axes, random cameras from inside the scene to far away, pinhole to huge lenses, random image sizes and chunks): the frame with the empty-tile
packet test on must equal the frame with it off bit for bit, counters included.   usage: python tools/cull_sweep.py [n_cases] [first_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import RenderChunk, RenderSettings, Scene, SceneBuilder, camera_new

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad, culled_cases, culled_px, total_px = [], 0, 0, 0
for case in range(n_cases):
    r = np.random.default_rng(900000 + seed0 + case)
    w, h = int(r.integers(128, 420)), int(r.integers(128, 300))
    size = float(r.choice([0.05, 1.0, 40.0, 3000.0]))                       # scene scale: margins are relative
    centre = r.uniform(-1, 1, 3) * size * float(r.choice([0.0, 3.0, 100.0]))   # ... also far from the origin
    where = int(r.integers(0, 4))
    eye = centre + r.uniform(-1, 1, 3) * size * [1.5, 6.0, 60.0, 0.3][where]
    to = centre + r.uniform(-1, 1, 3) * size * float(r.choice([0.3, 3.0, 12.0]))
    lens = size * float(r.choice([0.0, 0.01, 0.2, 2.0]))
    focus = float(np.linalg.norm(eye - to)) * float(r.choice([0.1, 1.0, 4.0])) + 1e-3 * size
    cam = camera_new(tuple(eye), tuple(to), float(r.choice([5.0, 30.0, 70.0, 140.0])), w / h, lens, focus)
    b = SceneBuilder(cam, int(r.random() < 0.6))             # black sky: empty tiles are left out; day sky: their camera rays skip the walk
    meshless = r.random() < 0.4                                  # (a day sky skips walks in scenes without mesh instances only)
    mats = [b.add_lambertian(float(r.uniform(0.0, 0.5))), b.add_specular(0.8, 0.1), b.add_dielectric(0.6, 0.2, 1.5)]
    tex = b.add_color_texture((0.7, 0.6, 0.5))
    meshes = []
    for _ in range(int(r.integers(1, 4))):
        nv = int(r.integers(4, 60))
        verts = (r.uniform(-1, 1, (nv, 3)) * size * float(r.choice([0.2, 1.0]))).astype(np.float32)
        meshes.append(b.add_mesh(verts[r.integers(0, nv, (int(r.integers(1, 300)), 3))]))
    for k in range(int(r.integers(1, 30))):
        tr = [tuple(centre / 2 + r.uniform(-1, 1, 3) * size * 1.5) for _ in range(int(r.integers(1, 3)))]
        ro = []
        for _ in range(int(r.integers(0, 3))):
            axis = r.normal(size=3); axis /= np.linalg.norm(axis)
            axis = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0), tuple(float(x) for x in axis.astype(np.float32))][int(r.integers(0, 4))]
            ro.append((axis, float(r.uniform(-3.2, 3.2))))
        if r.random() < 0.5 or meshless:
            b.add_sphere_entity(float(r.uniform(0.05, 0.6)) * size, mats[int(r.integers(0, 3))], tex, translations=tr, rotations=ro, id=k)
        else:
            b.add_mesh_entity(meshes[int(r.integers(0, len(meshes)))], mats[int(r.integers(0, 3))], tex, translations=tr, rotations=ro, id=k)
    b.add_point_light(tuple(centre + (0.3 * size, 4 * size, 0.4 * size)), 60.0 * size * size)
    scene = Scene(b, 0)
    if scene.info().loose_boxes if hasattr(scene.info(), "loose_boxes") else False:
        scene.close(); continue
    scene.set_option("fine_split", 0)
    spp = int(r.choice([1, 8, 16, 64]))
    st = RenderSettings(w, h, int(r.integers(1, 4)), spp)
    if r.random() < 0.3:
        x0, y0 = int(r.integers(0, w // 3)), int(r.integers(0, h // 3))
        ch = RenderChunk(x0, y0, int(r.integers(w // 2, w - x0 + 1)), int(r.integers(h // 2, h - y0 + 1)))
    else:
        ch = RenderChunk(0, 0, w, h)
    res = []
    for cull in (1, 0):
        scene.set_option("cull_empty_tiles", cull)
        rgb, _, cn = scene.render_chunk(st, ch, seed=case, want_rgba=False)
        res.append((rgb, (cn.primary, cn.secondary, cn.shadow, cn.samples), scene.culled_pixels()))
    same = np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32)) and res[0][1] == res[1][1]
    culled_cases += res[0][2] > 0; culled_px += res[0][2]; total_px += ch.width * ch.height
    if not same:
        bad.append(case)
        print(f"case {case}: DIFFERENT ({int((res[0][0] != res[1][0]).any(axis=2).sum())} pixels, counters {res[0][1]} / {res[1][1]}, culled {res[0][2]})", flush=True)
    if case % 100 == 99:
        print(f"{case + 1} cases: {len(bad)} differ, {culled_cases} with empty tiles, {culled_px / max(total_px, 1) * 100:.1f} % of all pixels in empty tiles", flush=True)
    scene.close()
print(f"cull sweep: {n_cases} cases, {len(bad)} differ {bad[:20]}, {culled_cases} with empty tiles, {culled_px / max(total_px, 1) * 100:.1f} % of all pixels in empty tiles")
sys.exit(1 if bad else 0)
