#!/usr/bin/env python3
"""Dumps the rays on which the oracle's ordered, t_max-shrinking traversal (the algorithm the GPU
kernel uses) disagrees with the reference-order traversal (oracle stats mode).  CPU only.

    python tools/mismatch_diag.py "Doot Doot" 3840 2160 --params Count=30 Spacing=30
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from raytracer_amd import CameraConfiguration, RenderSettings, SceneBuilder  # noqa: E402
from tests.oracle import ORC_RNG_COUNTER_STRATIFIED, OracleScene, lib  # noqa: E402


class OrcMismatch(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("d", C.c_float * 3), ("t_min", C.c_float), ("t_max", C.c_float),
                ("ref_have", C.c_int32), ("min_have", C.c_int32), ("ref_t", C.c_float), ("min_t", C.c_float),
                ("ref_ent", C.c_int32), ("ref_tri", C.c_int32), ("min_ent", C.c_int32), ("min_tri", C.c_int32),
                ("ref_spans", C.c_float * 6), ("min_spans", C.c_float * 6)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("width", type=int)
    ap.add_argument("height", type=int)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--lens", type=float, default=0.1)
    ap.add_argument("--params", nargs="*", default=[])
    ap.add_argument("--stride", type=int, default=13)
    ap.add_argument("--show", type=int, default=20)
    ap.add_argument("--slack", type=float, default=0.0, help="culling slack factor of the shrinking walk (1.0 = none)")
    a = ap.parse_args()
    params = {k: float(v) for k, v in (p.split("=") for p in a.params)} or None
    cfg = CameraConfiguration(a.width, a.height, lens_radius=a.lens)
    builder = SceneBuilder.sample(a.scene, cfg, params=params)
    oracle = OracleScene(builder)
    L = lib()
    L.orc_mismatch_capture.argtypes = [C.c_uint32]
    L.orc_mismatch_read.argtypes = [C.POINTER(OrcMismatch), C.c_uint32]
    L.orc_mismatch_read.restype = C.c_uint32
    L.orc_set_cull_slack.argtypes = [C.c_float]
    if a.slack:
        L.orc_set_cull_slack(a.slack)
    L.orc_mismatch_capture(100000)
    st = RenderSettings(a.width, a.height, a.depth, a.spp)
    grid = (32, 32)
    cn, ms = oracle.render_chunks_mt(st, grid, 5, a.stride, (grid[0] * grid[1] - 6) // a.stride, os.cpu_count() or 1,
                                     rng_mode=ORC_RNG_COUNTER_STRATIFIED, seed=12345, stats=True)
    rays = cn.rays.primary + cn.rays.secondary + cn.rays.shadow
    buf = (OrcMismatch * 100000)()
    n = L.orc_mismatch_read(buf, 100000)
    print(f"rays {rays}  mismatches {cn.min_mismatch}  captured {n}  ({ms / 1e3:.1f} s)  per ray: nodes {cn.min_nodes / rays:.3f} "
          f"tris {cn.min_tris / rays:.3f} spheres {cn.min_spheres / rays:.3f} instances {cn.min_instances / rays:.3f}")
    kinds = {}
    for r in buf[:n]:
        ulp = lambda t: float(np.spacing(np.float32(abs(t))))
        if r.ref_have and not r.min_have:
            kind = "shrinking walk found nothing"
        elif not r.ref_have:
            kind = "reference found nothing"
        elif r.ref_t == r.min_t:
            kind = "tie in t, other winner"
        elif r.min_t < r.ref_t:
            kind = "shrinking walk found a NEARER hit"
        else:
            # the reference winner was culled: which of its boxes lies beyond the hit the walk held?
            s = r.ref_spans
            if s[0] > r.ref_t or s[1] < r.ref_t:
                kind = "ref hit outside its ENTITY box"
            elif r.ref_tri >= 0 and (s[2] > r.ref_t or s[3] < r.ref_t):
                kind = "ref hit outside its mesh ROOT box"
            elif r.ref_tri >= 0 and (s[4] > r.ref_t or s[5] < r.ref_t):
                kind = "ref hit outside its TRIANGLE box"
            else:
                kind = "ref hit inside all its boxes"
            kind += f" (dt = {(r.min_t - r.ref_t) / ulp(r.ref_t):.0f} ulp)" if abs(r.min_t - r.ref_t) < 64 * ulp(r.ref_t) else " (far)"
        kinds[kind] = kinds.get(kind, 0) + 1
    for k, v in sorted(kinds.items(), key=lambda kv: -kv[1]):
        print(f"{v:8d}  {k}")
    for r in buf[:min(n, a.show)]:
        print(f"ref t={r.ref_t:.9g} e={r.ref_ent} tri={r.ref_tri} spans={[f'{v:.9g}' for v in r.ref_spans]}\n"
              f"min t={r.min_t:.9g} e={r.min_ent} tri={r.min_tri} spans={[f'{v:.9g}' for v in r.min_spans]}  tmin={r.t_min:.4g} tmax={r.t_max:.6g}")


if __name__ == "__main__":
    main()
