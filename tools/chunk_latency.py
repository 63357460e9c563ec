#!/usr/bin/env python3
"""Wall time of rt_render_chunk per chunk for the reference's chunk protocol (one host call per chunk):
tools/chunk_latency.py [grid] -> ms per chunk against the kernel time of the whole frame / chunks."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import CameraConfiguration, RenderChunk, RenderSettings, Scene, SceneBuilder, create_render_chunks

W, H, SPP = 1920, 1080, 256
g = int(sys.argv[1]) if len(sys.argv) > 1 else 8
b = SceneBuilder.sample("Random Spheres", CameraConfiguration(W, H, lens_radius=0.1))
s = Scene(b, 0)
st = RenderSettings(W, H, 8, SPP)
chunks = create_render_chunks((W, H), (g, g))
for c in chunks[:4]:
    s.render_chunk(st, c, want_rgb=False)
t0 = time.perf_counter()
for c in chunks:
    s.render_chunk(st, c, want_rgb=False)
dt = time.perf_counter() - t0
print(f"{g}x{g} chunks: {dt * 1e3:.1f} ms per frame, {dt * 1e3 / len(chunks):.3f} ms per chunk")
from concurrent.futures import ThreadPoolExecutor
for threads in (2, 4, 8):
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda c: s.render_chunk(st, c, want_rgb=False), chunks[:threads]))
        t0 = time.perf_counter()
        list(ex.map(lambda c: s.render_chunk(st, c, want_rgb=False), chunks))
        dt = time.perf_counter() - t0
    print(f"{g}x{g} chunks, {threads} host threads: {dt * 1e3:.1f} ms per frame")
t0 = time.perf_counter()
s.render_chunk(st, RenderChunk(0, 0, W, H), want_rgb=False)
print(f"one chunk: {(time.perf_counter() - t0) * 1e3:.1f} ms")
