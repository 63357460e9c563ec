#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

    python tests/golden/make_golden.py

The reference (Rust) cannot be run in the build container and its own tests hold
no vectors for intersection / shading, so these fixtures are
outputs of the ORACLE (oracle/rt_oracle.cpp), committed so that (a) a change of
the oracle's results is caught on the CPU, and (b) the GPU path is compared with
fixed numbers and not only with a library built from the same tree.
Inputs are the sample scenes as built by librt_samples.so from assets/meshes.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from raytracer_amd import CameraConfiguration, RenderChunk, RenderSettings, SceneBuilder, create_render_chunks  # noqa: E402
from tests.oracle import LENS_MODES, ORC_RNG_COUNTER, ORC_RNG_XORSHIFT_CHUNK, OracleScene  # noqa: E402

PRIMARY = [("Random Spheres", 64, 36), ("Mesh", 64, 36), ("Simple", 48, 27), ("Mirrors", 48, 27), ("Dreadnaught", 48, 27)]
IMAGES = [("Random Spheres", 48, 27, 4, 8), ("Mesh", 48, 27, 2, 8), ("Simple", 32, 18, 2, 5), ("Mirrors", 32, 18, 2, 10),
          ("Dreadnaught", 32, 18, 1, 5)]
# enough samples for complete groups of 16 / 8 lens samples, where the reference's sampler (ORC_RNG_COUNTER) and the product's
# stratified one (ORC_RNG_COUNTER_STRATIFIED) differ: both are pinned.  (IMAGES have fewer than 8 samples: the two modes coincide.)
IMAGES_DOF = [("Random Spheres", 48, 27, 24, 8), ("Mesh", 48, 27, 19, 8)]
SEED = 12345


def key(name):
    return name.lower().replace(" ", "_")


def main():
    out = {}
    for name, w, h in PRIMARY:
        b = SceneBuilder.sample(name, CameraConfiguration(w, h))
        t, n, e, tri = OracleScene(b).primary_hits(RenderSettings(w, h, 8, 1), RenderChunk(0, 0, w, h))
        k = key(name)
        out[f"primary/{k}/t"], out[f"primary/{k}/normal"], out[f"primary/{k}/entity"], out[f"primary/{k}/tri"] = t, n, e, tri
    for name, w, h, spp, depth in IMAGES:
        b = SceneBuilder.sample(name, CameraConfiguration(w, h))
        o = OracleScene(b)
        st = RenderSettings(w, h, depth, spp)
        rgb, rgba, cn = o.render_chunk(st, RenderChunk(0, 0, w, h), ORC_RNG_COUNTER, SEED)
        k = key(name)
        out[f"image/{k}/rgb"], out[f"image/{k}/rgba"] = rgb, rgba
        out[f"image/{k}/rays"] = np.array([cn.rays.primary, cn.rays.secondary, cn.rays.shadow, cn.rays.samples], dtype=np.int64)
        # reference RNG semantics: one XorShift stream per chunk (2x2 chunks, src/render.rs:142)
        ref = np.zeros_like(rgb)
        for c in create_render_chunks((w, h), (2, 2)):
            part, _, _ = o.render_chunk(st, c, ORC_RNG_XORSHIFT_CHUNK, SEED)
            ref[c.top:c.top + c.height, c.left:c.left + c.width] = part
        out[f"image/{k}/rgb_xorshift_2x2"] = ref
    for name, w, h, spp, depth in IMAGES_DOF:
        o = OracleScene(SceneBuilder.sample(name, CameraConfiguration(w, h, lens_radius=0.1)))
        for mode, (rng_mode, _) in LENS_MODES.items():
            rgb, _, cn = o.render_chunk(RenderSettings(w, h, depth, spp), RenderChunk(0, 0, w, h), rng_mode, SEED, sum_segment=8)
            out[f"image_dof/{key(name)}/{mode}/rgb"] = rgb
            out[f"image_dof/{key(name)}/{mode}/rays"] = np.array([cn.rays.primary, cn.rays.secondary, cn.rays.shadow, cn.rays.samples], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
