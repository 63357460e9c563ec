"""Golden fixtures (tests/golden/oracle_golden.npz, made by make_golden.py).
CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them."""
import os

import numpy as np
import pytest

from raytracer_amd import CameraConfiguration, RenderChunk, RenderSettings, Scene, SceneBuilder, create_render_chunks
from tests.golden.make_golden import IMAGES, IMAGES_DOF, PRIMARY, SEED, key
from tests.oracle import LENS_MODES, ORC_RNG_COUNTER, ORC_RNG_XORSHIFT_CHUNK, OracleScene

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.npz"))


@pytest.mark.parametrize("name,w,h", PRIMARY)
def test_oracle_primary_hits_reproduce_golden(name, w, h):
    b = SceneBuilder.sample(name, CameraConfiguration(w, h))
    t, n, e, tri = OracleScene(b).primary_hits(RenderSettings(w, h, 8, 1), RenderChunk(0, 0, w, h))
    k = key(name)
    # only IEEE add/mul/div/sqrt on the path: bit-exact
    assert np.array_equal(t, GOLD[f"primary/{k}/t"]) and np.array_equal(n, GOLD[f"primary/{k}/normal"])
    assert np.array_equal(e, GOLD[f"primary/{k}/entity"]) and np.array_equal(tri, GOLD[f"primary/{k}/tri"])


@pytest.mark.parametrize("name,w,h,spp,depth", IMAGES)
def test_oracle_images_reproduce_golden(name, w, h, spp, depth):
    b = SceneBuilder.sample(name, CameraConfiguration(w, h))
    o = OracleScene(b)
    st = RenderSettings(w, h, depth, spp)
    k = key(name)
    rgb, rgba, cn = o.render_chunk(st, RenderChunk(0, 0, w, h), ORC_RNG_COUNTER, SEED)
    # libm transcendentals on the path: allow the last ulps
    np.testing.assert_allclose(rgb, GOLD[f"image/{k}/rgb"], rtol=0, atol=2e-6)
    assert [cn.rays.primary, cn.rays.secondary, cn.rays.shadow, cn.rays.samples] == GOLD[f"image/{k}/rays"].tolist()
    ref = np.zeros_like(rgb)
    for c in create_render_chunks((w, h), (2, 2)):
        part, _, _ = o.render_chunk(st, c, ORC_RNG_XORSHIFT_CHUNK, SEED)
        ref[c.top:c.top + c.height, c.left:c.left + c.width] = part
    np.testing.assert_allclose(ref, GOLD[f"image/{k}/rgb_xorshift_2x2"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name,w,h,spp,depth", IMAGES_DOF)
def test_oracle_depth_of_field_images_reproduce_golden(name, w, h, spp, depth, lens_mode):
    o = OracleScene(SceneBuilder.sample(name, CameraConfiguration(w, h, lens_radius=0.1)))
    rgb, _, cn = o.render_chunk(RenderSettings(w, h, depth, spp), RenderChunk(0, 0, w, h), LENS_MODES[lens_mode][0], SEED, sum_segment=8)
    np.testing.assert_allclose(rgb, GOLD[f"image_dof/{key(name)}/{lens_mode}/rgb"], rtol=0, atol=2e-6)
    assert [cn.rays.primary, cn.rays.secondary, cn.rays.shadow, cn.rays.samples] == GOLD[f"image_dof/{key(name)}/{lens_mode}/rays"].tolist()


def test_the_two_lens_samplers_differ_only_in_the_lens_draws():
    """Same pixels hit by the lens-centre ray, different depth-of-field noise: the fixtures of the two modes are close but not equal."""
    for name, *_ in IMAGES_DOF:
        a, b = GOLD[f"image_dof/{key(name)}/reference/rgb"], GOLD[f"image_dof/{key(name)}/stratified/rgb"]
        assert not np.array_equal(a, b) and np.abs(a - b).mean() < 0.05


def test_counter_rng_matches_reference_rng_where_no_randomness_reaches_the_pixel():
    """with spp = 1 there is no lens jitter, and default
    Lambertians / fuzz-0 metals / dielectrics use no random number that reaches the
    pixel, so the reference's per-chunk XorShift stream and the counter RNG must give
    the same image there (here: 'Dreadnaught', all-Lambertian, spp 1)."""
    k = "dreadnaught"
    assert np.array_equal(GOLD[f"image/{k}/rgb"], GOLD[f"image/{k}/rgb_xorshift_2x2"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h", PRIMARY)
def test_gpu_primary_hits_reproduce_golden(name, w, h):
    b = SceneBuilder.sample(name, CameraConfiguration(w, h))
    t, n, e, tri = Scene(b, 0).primary_hits(RenderSettings(w, h, 8, 1), RenderChunk(0, 0, w, h))
    k = key(name)
    assert np.array_equal(e, GOLD[f"primary/{k}/entity"]) and np.array_equal(tri, GOLD[f"primary/{k}/tri"])
    hit = e >= 0
    np.testing.assert_allclose(t[hit], GOLD[f"primary/{k}/t"][hit], rtol=1e-5, atol=0)       # north_star bound
    np.testing.assert_allclose(n[hit], GOLD[f"primary/{k}/normal"][hit], rtol=1e-5, atol=1e-6)
    assert np.array_equal(t, GOLD[f"primary/{k}/t"])                                          # and in fact bit-exact


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,spp,depth", IMAGES)
def test_gpu_images_reproduce_golden(name, w, h, spp, depth):
    b = SceneBuilder.sample(name, CameraConfiguration(w, h))
    rgb, rgba, cn = Scene(b, 0).render_chunk(RenderSettings(w, h, depth, spp), RenderChunk(0, 0, w, h), seed=SEED)
    k = key(name)
    d = np.abs(rgb - GOLD[f"image/{k}/rgb"])
    assert (d.max(axis=2) > 2e-5).mean() <= 2e-3 and d.mean() < 2e-5
    assert cn.primary == GOLD[f"image/{k}/rays"][0] and cn.samples == GOLD[f"image/{k}/rays"][3]


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,spp,depth", IMAGES_DOF)
def test_gpu_depth_of_field_images_reproduce_golden(name, w, h, spp, depth, lens_mode):
    scene = Scene(SceneBuilder.sample(name, CameraConfiguration(w, h, lens_radius=0.1)), 0)
    scene.set_option("lens_sampler", LENS_MODES[lens_mode][1])
    rgb, _, cn = scene.render_chunk(RenderSettings(w, h, depth, spp), RenderChunk(0, 0, w, h), seed=SEED)
    gold = GOLD[f"image_dof/{key(name)}/{lens_mode}/rgb"]
    d = np.abs(rgb - gold)
    assert (d.max(axis=2) > 2e-5).mean() <= 5e-3 and d.mean() < 5e-5
    assert cn.primary == GOLD[f"image_dof/{key(name)}/{lens_mode}/rays"][0]
