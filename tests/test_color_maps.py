"""Colour-map ingest: raytracer-obj/src/color_map.rs:16-33,
builder.rs:121-131,159-172 and the ColorMap lookup texture.rs:82-99.

The reference decodes with the third-party `image` crate (0.24.7, not in the tree):
the decoder here is checked against PIL on the reference's own Dreadnaught PNGs
(8-bit RGBA, Adam7-interlaced) and on synthetic files of the other layouts.
"""
import os

import numpy as np
import pytest

from raytracer_amd import CameraConfiguration, RenderChunk, RenderSettings, SceneBuilder, decode_image
from raytracer_amd import _ffi
from tests.oracle import ORC_RNG_COUNTER_STRATIFIED, OracleScene

PIL = pytest.importorskip("PIL.Image")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DREAD = os.path.join(ROOT, "assets", "meshes", "Dreadnaught")
PNGS = ["Hgn_DreadnaughtA_EXPORT[1].png", "HGN_DreadnaughtB_EXPORT[1].png", "Kpr_Dreadnaught_Engine_EXPORT[3].png"]


@pytest.mark.parametrize("name", PNGS)
def test_reference_pngs_decode_like_pil(name):
    path = os.path.join(DREAD, name)
    got = decode_image(path)
    want = np.asarray(PIL.open(path).convert("RGBA"))
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("mode,size,kw", [
    ("RGB", (37, 21), {}), ("RGBA", (16, 9), {}), ("L", (33, 7), {}), ("LA", (5, 5), {}), ("P", (40, 13), {}),
    ("1", (19, 11), {}), ("RGB", (1, 1), {}), ("RGB", (9, 13), {"compress_level": 0}), ("L", (64, 3), {"bits": 4}), ("P", (31, 6), {"bits": 2}),
])
def test_synthetic_pngs(tmp_path, mode, size, kw):
    rng = np.random.default_rng(hash((mode, size)) & 0xffff)
    w, h = size
    if mode == "P":
        img = PIL.fromarray(rng.integers(0, 1 << kw.get("bits", 8), (h, w), dtype=np.uint8), "P")
        img.putpalette(list(rng.integers(0, 256, 768, dtype=np.uint8)))
    elif mode == "1":
        img = PIL.fromarray((rng.integers(0, 2, (h, w), dtype=np.uint8) * 255), "L").convert("1")
    elif mode == "L" and "bits" in kw:
        img = PIL.fromarray((rng.integers(0, 16, (h, w), dtype=np.uint8) * 17), "L")
    else:
        ch = {"RGB": 3, "RGBA": 4, "L": 1, "LA": 2}[mode]
        a = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        img = PIL.fromarray(a[:, :, 0] if ch == 1 else a, mode)
    path = str(tmp_path / "t.png")
    img.save(path, **kw)
    got = decode_image(path)
    want = np.asarray(PIL.open(path).convert("RGBA"))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("mode", ["RGB", "P", "RGBA"])
def test_bmp_files(tmp_path, mode):
    rng = np.random.default_rng(3)
    w, h = 23, 10                                   # 23*3 bytes: exercises the row padding
    if mode == "P":
        img = PIL.fromarray(rng.integers(0, 256, (h, w), dtype=np.uint8), "P")
        img.putpalette(list(rng.integers(0, 256, 768, dtype=np.uint8)))
    else:
        img = PIL.fromarray(rng.integers(0, 256, (h, w, len(mode)), dtype=np.uint8), mode)
    path = str(tmp_path / "t.bmp")
    img.save(path)
    got = decode_image(path)
    want = np.asarray(PIL.open(path).convert("RGB"))
    assert np.array_equal(got[:, :, :3], want)


def test_bad_files_are_errors(tmp_path):
    p = tmp_path / "lfs.bmp"
    p.write_text("version https://git-lfs.github.com/spec/v1\noid sha256:00\nsize 1\n")
    with pytest.raises(ValueError, match="git-lfs"):
        decode_image(str(p))
    q = tmp_path / "junk.png"
    q.write_bytes(b"not an image at all")
    with pytest.raises(ValueError, match="type unknown"):
        decode_image(str(q))
    with pytest.raises(ValueError, match="expected file"):
        decode_image(str(tmp_path / "missing.png"))
    good = os.path.join(DREAD, PNGS[2])
    data = bytearray(open(good, "rb").read())
    data[len(data) // 2] ^= 0x55                     # flips a bit inside an IDAT chunk
    r = tmp_path / "corrupt.png"
    r.write_bytes(bytes(data))
    with pytest.raises(ValueError, match="CRC"):
        decode_image(str(r))


def test_dreadnaught_scene_carries_its_colour_maps():
    b = SceneBuilder.sample("Dreadnaught", CameraConfiguration(64, 36))
    d = b.desc().contents
    tex = [d.textures[i] for i in range(d.texture_count)]
    maps = [t for t in tex if t.kind == _ffi.RT_TEX_COLOR_MAP]
    assert sorted((t.width, t.height) for t in maps) == [(256, 128), (512, 128), (1024, 512)]
    assert d.texel_count == 256 * 128 + 512 * 128 + 1024 * 512
    mesh_tex = [t for t in tex if t.kind == _ffi.RT_TEX_MESH]
    assert len(mesh_tex) == 3 and all(t.child0 >= 0 and tex[t.child0].kind == _ffi.RT_TEX_COLOR_MAP for t in mesh_tex)
    # texel values: channel / 255.0, row 0 = top (color_map.rs:9-14, 22-26)
    m = [t for t in maps if t.width == 256][0]
    want = np.asarray(PIL.open(os.path.join(DREAD, PNGS[2])).convert("RGB")).astype(np.float32) / np.float32(255.0)
    got = np.ctypeslib.as_array(d.texels, shape=(d.texel_count,)).view(np.float32).reshape(-1, 3)[m.texel_first:m.texel_first + 256 * 128]
    assert np.array_equal(got.reshape(128, 256, 3), want)


def test_lfs_pointer_maps_are_skipped_with_a_warning():
    b = SceneBuilder.sample("Fleet", CameraConfiguration(64, 36))      # Interceptor-T: .bmp files are git-lfs pointers upstream
    d = b.desc().contents
    assert d.texel_count == 0
    assert b"git-lfs" in _ffi.load_samples().rt_samples_last_warning()


@pytest.mark.gpu
def test_textured_dreadnaught_matches_oracle():
    from raytracer_amd import Scene, sample_segment
    from tests.test_gpu_parity import check_image
    w, h = 240, 135
    b = SceneBuilder.sample("Dreadnaught", CameraConfiguration(w, h, lens_radius=0.1))
    scene, oracle = Scene(b, 0), OracleScene(b)
    st, ch = RenderSettings(w, h, 6, 2), RenderChunk(0, 0, w, h)
    rgb, rgba, cn = scene.render_chunk(st, ch, seed=12345)
    ref, ref_rgba, rcn = oracle.render_chunk(st, ch, ORC_RNG_COUNTER_STRATIFIED, 12345, sum_segment=sample_segment())
    check_image(rgb, ref)
    assert cn.primary == rcn.rays.primary
    assert len(np.unique(rgba.reshape(-1, 4), axis=0)) > 50          # the hull is textured, not one flat colour
