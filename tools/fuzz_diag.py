"""Delta-debugging of a failing tests/test_gpu_fuzz.py scene: drops entities while the GPU / oracle
primary-hit mismatch persists and prints the minimal scene.  usage: tools/fuzz_diag.py <seed>..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import tests.test_gpu_fuzz as t
from raytracer_amd import Scene, RenderSettings, RenderChunk, SceneBuilder, _ffi
from tests.oracle import OracleScene


def subset(d, keep):
    b = SceneBuilder(d.camera, d.sky)
    lib = b._lib
    for i in range(d.material_count): lib.rt_builder_add_material(b._h, C.byref(d.materials[i]))
    for i in range(d.texture_count): lib.rt_builder_add_texture(b._h, C.byref(d.textures[i]))
    for i in range(d.light_count): lib.rt_builder_add_light(b._h, C.byref(d.lights[i]))
    for i in range(d.mesh_count): lib.rt_builder_add_mesh(b._h, d.meshes[i].tris, d.meshes[i].tri_count)
    for i in keep:
        e = d.entities[i]
        sp = C.byref(d.spheres[e.hitable_index]) if e.hitable_kind == 0 else None
        pl = C.byref(d.planes[e.hitable_index]) if e.hitable_kind == 1 else None
        mo = C.byref(d.mesh_objects[e.hitable_index]) if e.hitable_kind == 2 else None
        tr = (_ffi.RtV3 * max(1, e.translation_count))(*[d.translations[e.translation_first + k] for k in range(e.translation_count)])
        ro = (_ffi.RtRotation * max(1, e.rotation_count))(*[d.rotations[e.rotation_first + k] for k in range(e.rotation_count)])
        lib.rt_builder_add_entity(b._h, e.hitable_kind, sp, pl, mo, e.id, tr, e.translation_count, ro, e.rotation_count)
    return b


def mismatches(b):
    s, o = Scene(b, 0), OracleScene(b)
    st = RenderSettings(t.W, t.H, 8, 1); ch = RenderChunk(0, 0, t.W, t.H)
    g = s.primary_hits(st, ch); r = o.primary_hits(st, ch)
    bad = np.argwhere((g[2] != r[2]) | (g[3] != r[3]))
    return bad, g, r, s


for seed in [int(a) for a in sys.argv[1:]]:
    full = t.random_scene(1000 + seed)
    d = full.desc().contents
    keep = list(range(d.entity_count))
    bad, g, r, s = mismatches(full)
    print("seed", seed, "entities", d.entity_count, "mismatches", len(bad))
    changed = True
    while changed and len(keep) > 1:
        changed = False
        for i in list(keep):
            trial = [k for k in keep if k != i]
            if len(mismatches(subset(d, trial))[0]):
                keep = trial; changed = True
    b = subset(d, keep)
    bad, g, r, s = mismatches(b)
    info = s.info()
    print(" minimal entity set", keep, "mismatches", len(bad), "tlas depth", info.tlas_depth, "nodes", info.tlas_node_count, "order", s.bvh_order().tolist())
    for i in keep:
        e = d.entities[i]
        tr = [tuple(round(v, 4) for v in (d.translations[e.translation_first + k].x, d.translations[e.translation_first + k].y, d.translations[e.translation_first + k].z)) for k in range(e.translation_count)]
        ro = [((round(q.axis.x, 3), round(q.axis.y, 3), round(q.axis.z, 3)), round(q.theta, 4)) for q in [d.rotations[e.rotation_first + k] for k in range(e.rotation_count)]]
        extra = f"r={d.spheres[e.hitable_index].radius:.4f}" if e.hitable_kind == 0 else (f"mesh={d.mesh_objects[e.hitable_index].mesh_index} tris={d.meshes[d.mesh_objects[e.hitable_index].mesh_index].tri_count}" if e.hitable_kind == 2 else f"plane radius={d.planes[e.hitable_index].radius if d.planes[e.hitable_index].has_radius else None}")
        print(f"  ent{i} kind={e.hitable_kind} {extra} tr={tr} ro={ro}")
    for y, x in bad[:4]:
        print(f"  px({x},{y}) gpu ent#{g[2][y,x]} tri {g[3][y,x]} t {float(g[0][y,x])!r} | ref ent#{r[2][y,x]} tri {r[3][y,x]} t {float(r[0][y,x])!r}")
