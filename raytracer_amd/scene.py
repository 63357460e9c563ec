"""Host-side mirror of the reference's scene / settings / render-chunk API.

Names follow the reference (raytracer-impl/src/implementation.rs,
viewport.rs; raytracer-samples/src/scene.rs): ``SceneBuilder`` plays the role
of ``Scene::new`` + ``add_*``, ``Scene`` is the built scene resident on one GPU
(``Scene::build_bvh`` happened), ``RenderSettings`` / ``RenderChunk`` /
``create_render_chunks`` are the reference's own.  All work happens behind the
C ABI of include/rt_hip.h; this module only marshals arguments.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from ._ffi import (RtCamera, RtCameraConfig, RtEntity, RtLight, RtMaterial, RtMeshObject, RtMeshTri, RtPlane,
                   RtRayCounters, RtRenderChunk, RtRotation, RtSceneDesc, RtSceneInfo, RtSettings, RtSphere,
                   RtTexture, RtV3)


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[rt error {code}] {message}")
        self.code = code


def _v3(v):
    return RtV3(float(v[0]), float(v[1]), float(v[2]))


def _f3(v):
    return (C.c_float * 3)(float(v[0]), float(v[1]), float(v[2]))


class CameraConfiguration:
    """CameraConfiguration (raytracer-samples/src/scene.rs:12-20); defaults are
    the UI's (src/settings.rs:88-105)."""

    def __init__(self, width, height, lens_radius=0.1, fov=45.0, angle_adjust_v=0.0, angle_adjust_h=0.0,
                 focus_dist_adjust=0.0):
        self.c = RtCameraConfig(float(width), float(height), lens_radius, fov, angle_adjust_v, angle_adjust_h,
                                focus_dist_adjust)

    def make_camera(self, look_to, default_look_from):
        out = RtCamera()
        rc = _ffi.load_samples().rt_make_camera(C.byref(self.c), _f3(look_to), _f3(default_look_from), C.byref(out))
        if rc:
            raise RtError(rc, _ffi.load_samples().rt_samples_last_error().decode())
        return out


def camera_new(look_from, look_at, v_fov, aspect_ratio, lens_radius, focus_dist):
    """Camera::new (implementation.rs:462-480)."""
    out = RtCamera()
    _ffi.load_samples().rt_camera_new(_f3(look_from), _f3(look_at), v_fov, aspect_ratio, lens_radius, focus_dist,
                                      C.byref(out))
    return out


class RenderSettings:
    """RenderSettings (implementation.rs:332-337)."""

    def __init__(self, width, height, max_reflections=5, samples_per_pixel=1):
        self.c = RtSettings(width, height, max_reflections, samples_per_pixel)

    width = property(lambda s: s.c.width)
    height = property(lambda s: s.c.height)
    max_reflections = property(lambda s: s.c.max_reflections)
    samples_per_pixel = property(lambda s: s.c.samples_per_pixel)


def RenderChunk(left, top, width, height):
    """RenderChunk (viewport.rs:5-11)."""
    return RtRenderChunk(left, top, width, height)


def create_render_chunks(size, count):
    """create_render_chunks(size, count) (viewport.rs:30-54), via the C ABI."""
    lib = _ffi.load_hip()
    n = max(int(count[0]) * int(count[1]), 1)
    out = (RtRenderChunk * n)()
    rc = lib.rt_create_render_chunks(size[0], size[1], count[0], count[1], out, n)
    if rc < 0:
        raise RtError(rc, lib.rt_last_error().decode())
    return [RtRenderChunk(c.left, c.top, c.width, c.height) for c in out[:rc]]


class SceneBuilder:
    """Scene::new + add_entity / add_material / add_texture / add_light
    (implementation.rs:340-370).  Owns an RtSceneBuilder of librt_samples.so."""

    def __init__(self, camera=None, sky=_ffi.RT_SKY_DAY, _handle=None):
        self._lib = _ffi.load_samples()
        if _handle is not None:
            self._h = _handle
        else:
            self._h = self._lib.rt_builder_new(C.byref(camera), sky)
        if not self._h:
            raise RtError(_ffi.RT_E_INVALID, self._lib.rt_samples_last_error().decode())

    @classmethod
    def sample(cls, name, camera_config, mesh_dir=None, params=None):
        """One of the reference's sample scenes (raytracer-samples/src/lib.rs:18-37)."""
        lib = _ffi.load_samples()
        mesh_dir = mesh_dir or _ffi.ASSET_MESH_DIR
        p = None
        if params:
            p = ";".join(f"{k}={v}" for k, v in params.items()).encode()
        h = lib.rt_samples_create(name.encode(), C.byref(camera_config.c), mesh_dir.encode(), p)
        if not h:
            raise RtError(_ffi.RT_E_INVALID, lib.rt_samples_last_error().decode())
        return cls(_handle=h)

    @staticmethod
    def sample_names():
        lib = _ffi.load_samples()
        return [lib.rt_samples_name(i).decode() for i in range(lib.rt_samples_count())]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_builder_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise RtError(rc, self._lib.rt_samples_last_error().decode())
        return rc

    # -- materials (materials.rs) --
    def add_lambertian(self, reflectivity=0.0):
        return self._check(self._lib.rt_builder_add_material(
            self._h, C.byref(RtMaterial(_ffi.RT_MAT_LAMBERTIAN, reflectivity, 0.0, 0.0, 0.0))))

    def add_specular(self, reflectiveness=1.0, fuzz=0.0):
        return self._check(self._lib.rt_builder_add_material(
            self._h, C.byref(RtMaterial(_ffi.RT_MAT_SPECULAR, reflectiveness, fuzz, 0.0, 0.0))))

    def add_dielectric(self, reflectivity=1.0, opacity=0.0, ref_index=1.5):
        return self._check(self._lib.rt_builder_add_material(
            self._h, C.byref(RtMaterial(_ffi.RT_MAT_DIELECTRIC, reflectivity, 0.0, opacity, ref_index))))

    # -- textures (texture.rs) --
    def _tex(self, **kw):
        t = RtTexture()
        t.child0 = t.child1 = -1
        for k, v in kw.items():
            setattr(t, k, v)
        return self._check(self._lib.rt_builder_add_texture(self._h, C.byref(t)))

    def add_color_texture(self, rgb):
        return self._tex(kind=_ffi.RT_TEX_COLOR, color=_v3(rgb))

    def add_checker_texture(self, size, odd_tex, even_tex):
        return self._tex(kind=_ffi.RT_TEX_CHECKER, size=size, child0=odd_tex, child1=even_tex)

    def add_uv_test_texture(self):
        return self._tex(kind=_ffi.RT_TEX_UV_TEST)

    def add_xyz_test_texture(self, max_value):
        return self._tex(kind=_ffi.RT_TEX_XYZ_TEST, size=max_value)

    def add_color_map(self, pixels):
        """ColorMap from an (H, W, 3) float array, row 0 = top (color_map.rs:16-33)."""
        px = np.ascontiguousarray(pixels, dtype=np.float32)
        h, w, _ = px.shape
        return self._check(self._lib.rt_builder_add_color_map(self._h, w, h, px.ctypes.data))

    def load_color_map(self, path):
        """load_color_map(path), raytracer-obj/src/builder.rs:159-172: PNG / BMP file -> ColorMap TexId."""
        return self._check(self._lib.rt_builder_load_color_map(self._h, os.fsencode(path)))

    def add_mesh_texture(self, ambient, diffuse, color_map_tex=-1):
        return self._tex(kind=_ffi.RT_TEX_MESH, color=_v3(diffuse), color2=_v3(ambient), child0=color_map_tex)

    # -- lights (lights.rs) --
    def add_point_light(self, origin, intensity=100.0, color=(1, 1, 1)):
        l = RtLight(_ffi.RT_LIGHT_POINT, _v3(origin), RtV3(), _v3(color), intensity, 0.0)
        return self._check(self._lib.rt_builder_add_light(self._h, C.byref(l)))

    def add_lamp_light(self, origin, direction, intensity=80.0, angle_deg=45.0, color=(1, 1, 1)):
        d = np.asarray(direction, dtype=np.float32)
        n = np.float32(np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]))
        if n != 0:
            d = d / n
        l = RtLight(_ffi.RT_LIGHT_LAMP, _v3(origin), _v3(d), _v3(color), intensity, angle_deg)
        return self._check(self._lib.rt_builder_add_light(self._h, C.byref(l)))

    def add_directional_light(self, direction, intensity=1.0, color=(1, 1, 1)):
        d = np.asarray(direction, dtype=np.float32)
        n = np.float32(np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]))
        if n != 0:
            d = d / n
        l = RtLight(_ffi.RT_LIGHT_DIRECTIONAL, RtV3(), _v3(d), _v3(color), intensity, 0.0)
        return self._check(self._lib.rt_builder_add_light(self._h, C.byref(l)))

    # -- geometry --
    def add_mesh(self, triangles, uvs=None, tex_keys=None):
        """Mesh{tris} from an (N, 3, 3) float array; returns the mesh index."""
        tri = np.asarray(triangles, dtype=np.float32)
        n = tri.shape[0]
        arr = (RtMeshTri * n)()
        for i in range(n):
            arr[i].a, arr[i].b, arr[i].c = _v3(tri[i, 0]), _v3(tri[i, 1]), _v3(tri[i, 2])
            if uvs is not None:
                for k, name in enumerate(("a_uv", "b_uv", "c_uv")):
                    getattr(arr[i], name).x = float(uvs[i][k][0])
                    getattr(arr[i], name).y = float(uvs[i][k][1])
            arr[i].tex_key = -1 if tex_keys is None else int(tex_keys[i])
        return self._check(self._lib.rt_builder_add_mesh(self._h, arr, n))

    def load_obj(self, path):
        """load_obj_builder(path).build_mesh() -> (mesh index, MeshTextureSet TexId)."""
        tex = C.c_int(-1)
        mesh = self._check(self._lib.rt_builder_load_obj(self._h, str(path).encode(), C.byref(tex)))
        return mesh, tex.value

    def _entity(self, kind, sphere, plane, mesh_object, translations, rotations, id):
        tr = (RtV3 * max(len(translations), 1))(*[_v3(t) for t in translations])
        ro = (RtRotation * max(len(rotations), 1))(*[RtRotation(_v3(a), float(th)) for a, th in rotations])
        return self._check(self._lib.rt_builder_add_entity(
            self._h, kind, sphere, plane, mesh_object, -1 if id is None else id, tr, len(translations), ro,
            len(rotations)))

    def add_sphere_entity(self, radius, mat, tex, translations=(), rotations=(), id=None):
        """Entity::new(Sphere::new(radius, mat, tex)).translate(..).rotate(..).id(..)"""
        return self._entity(_ffi.RT_HITABLE_SPHERE, C.byref(RtSphere(radius, mat, tex)), None, None, translations,
                            rotations, id)

    def add_plane_entity(self, normal, mat, tex, radius=None, translations=(), rotations=(), id=None):
        p = RtPlane()
        self._check(self._lib.rt_plane_new(_f3(normal), mat, tex, 0 if radius is None else 1,
                                           0.0 if radius is None else radius, C.byref(p)))
        return self._entity(_ffi.RT_HITABLE_PLANE, None, C.byref(p), None, translations, rotations, id)

    def add_mesh_entity(self, mesh, mat, tex, translations=(), rotations=(), id=None):
        return self._entity(_ffi.RT_HITABLE_MESH, None, None, C.byref(RtMeshObject(mesh, mat, tex)), translations,
                            rotations, id)

    def desc(self):
        """The flattened POD description (valid until the builder changes)."""
        return self._lib.rt_builder_desc(self._h)


class Scene:
    """A scene resident on one MI355X: Scene::build_bvh done (TLAS + one BLAS per
    mesh), flattened and uploaded (rt_scene_create)."""

    def __init__(self, builder, device=0):
        self._lib = _ffi.load_hip()
        self._builder = builder   # keeps the description alive during the call
        self._h = self._lib.rt_scene_create(builder.desc(), device)
        if not self._h:
            raise RtError(_ffi.RT_E_INVALID, self._lib.rt_last_error().decode())
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_scene_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise RtError(rc, self._lib.rt_last_error().decode())
        return rc

    OPTIONS = {"lens_sampler": _ffi.RT_OPT_LENS_SAMPLER, "fine_split": _ffi.RT_OPT_FINE_SPLIT, "round_units": _ffi.RT_OPT_ROUND_UNITS,
               "lens_coc_px": _ffi.RT_OPT_LENS_COC_PX, "cull_empty_tiles": _ffi.RT_OPT_CULL_EMPTY_TILES, "debug_stack_entries": _ffi.RT_OPT_DEBUG_STACK_ENTRIES,
               "debug_rebuild_stack_limit": _ffi.RT_OPT_DEBUG_REBUILD_STACK_LIMIT}

    def set_option(self, name, value):
        """rt_scene_set_option: "lens_sampler" 0 = the reference's independent lens draws / 1 = stratified (default);
        "fine_split" -1 / 0 / 1, "round_units" 1 / 2 / 4 / 8, "lens_coc_px" and "cull_empty_tiles" 0 / 1 only change the schedule."""
        self._check(self._lib.rt_scene_set_option(self._h, self.OPTIONS[name], float(value)))

    def update_transforms(self, builder, stream=None):
        """rt_scene_update_transforms: take the translations / rotations of `builder` (a SceneBuilder describing the same
        entities, moved) and refit the TLAS on the device."""
        d = builder.desc().contents
        self._check(self._lib.rt_scene_update_transforms(self._h, C.cast(d.translations, C.c_void_p), d.translation_count,
                                                         C.cast(d.rotations, C.c_void_p), d.rotation_count, stream))

    def rebuild_tlas(self, builder, stream=None):
        """rt_scene_rebuild_tlas: like update_transforms, but with a NEW top-level tree built on the device (a linear BVH over
        the moved entities), for animations in which the entities travel."""
        d = builder.desc().contents
        self._check(self._lib.rt_scene_rebuild_tlas(self._h, C.cast(d.translations, C.c_void_p), d.translation_count,
                                                    C.cast(d.rotations, C.c_void_p), d.rotation_count, stream))

    def rebuild_blas(self, builder, mesh_index, builder_mesh_index=None, stream=None):
        """rt_scene_rebuild_blas: take the triangles of mesh `builder_mesh_index` (default: `mesh_index`) of `builder` (the same mesh,
        deformed: same triangle count) and build a new tree over them on the device for this scene's mesh `mesh_index`; the boxes of its
        instances and the TLAS above them are refitted."""
        d = builder.desc().contents
        k = mesh_index if builder_mesh_index is None else builder_mesh_index
        if not 0 <= k < d.mesh_count:
            raise RtError(_ffi.RT_E_INVALID, f"the builder has no mesh {k}")
        mesh = d.meshes[k]
        self._check(self._lib.rt_scene_rebuild_blas(self._h, mesh_index, C.cast(mesh.tris, C.c_void_p), mesh.tri_count, stream))

    def info(self):
        out = RtSceneInfo()
        self._check(self._lib.rt_scene_info(self._h, C.byref(out)))
        return out

    def bvh_order(self, mesh_index=-1):
        n = self._check(self._lib.rt_scene_bvh_order(self._h, mesh_index, None, 0))
        out = np.zeros(n, dtype=np.uint32)
        self._check(self._lib.rt_scene_bvh_order(self._h, mesh_index, out.ctypes.data, n))
        return out

    def render_chunk(self, settings, chunk, seed=12345, want_rgb=True, want_rgba=True):
        """The pixel loop of src/render.rs:148-159 for one chunk.
        Returns (rgb float32 (h,w,3) | None, rgba uint8 (h,w,4) | None, RtRayCounters)."""
        h, w = chunk.height, chunk.width
        rgb = np.empty((h, w, 3), dtype=np.float32) if want_rgb else None
        rgba = np.empty((h, w, 4), dtype=np.uint8) if want_rgba else None
        counters = RtRayCounters()
        self._check(self._lib.rt_render_chunk(
            self._h, C.byref(settings.c), C.byref(chunk), seed,
            rgb.ctypes.data if want_rgb else None, rgba.ctypes.data if want_rgba else None, C.byref(counters)))
        return rgb, rgba, counters

    def render_chunk_device(self, settings, chunk, seed, rgb_ptr, rgba_ptr, pitch_pixels, out_x, out_y,
                            counters_ptr=None, stream=None):
        self._check(self._lib.rt_render_chunk_device(self._h, C.byref(settings.c), C.byref(chunk), seed, rgb_ptr,
                                                     rgba_ptr, pitch_pixels, out_x, out_y, counters_ptr, stream))

    def render_chunk_device_noalloc(self, settings, chunk, seed, rgb_ptr, rgba_ptr, pitch_pixels, out_x, out_y,
                                    counters_ptr=None, stream=None):
        """rt_render_chunk_device_noalloc: same pixels, scratch from the library's static device array, nothing allocated."""
        self._check(self._lib.rt_render_chunk_device_noalloc(self._h, C.byref(settings.c), C.byref(chunk), seed, rgb_ptr,
                                                             rgba_ptr, pitch_pixels, out_x, out_y, counters_ptr, stream))

    def render_strips_device(self, settings, strip_height, strip_first, strip_period, seed, rgb_ptr, rgba_ptr,
                             counters_ptr=None, stream=None):
        """Every strip_period-th strip of the frame, one launch, packed output rows."""
        self._check(self._lib.rt_render_strips_device(self._h, C.byref(settings.c), strip_height, strip_first,
                                                      strip_period, seed, rgb_ptr, rgba_ptr, counters_ptr, stream))

    def render_tile_columns_device(self, settings, col_first, col_period, seed, rgb_ptr, rgba_ptr, counters_ptr=None, stream=None):
        """Every col_period-th 8-pixel tile column of the frame, one launch, packed output columns (tile_columns_width pixels per row)."""
        self._check(self._lib.rt_render_tile_columns_device(self._h, C.byref(settings.c), col_first, col_period, seed, rgb_ptr, rgba_ptr,
                                                            counters_ptr, stream))

    def accumulate_chunk_device(self, settings, chunk, seed, sample_first, sample_count, accum_ptr,
                                counters_ptr=None, stream=None):
        self._check(self._lib.rt_accumulate_chunk_device(self._h, C.byref(settings.c), C.byref(chunk), seed,
                                                         sample_first, sample_count, accum_ptr, counters_ptr, stream))

    def cancel(self, cancel=True):
        """rt_scene_cancel: stop (True) every running or enqueued launch on this scene / re-arm it (False).  The reference's
        cancellation token (src/render.rs:131,149), callable from another thread while a render call blocks."""
        self._check(self._lib.rt_scene_cancel(self._h, 1 if cancel else 0))

    def progress(self):
        """rt_scene_progress: (blocks handed out so far - an estimate -, blocks of the call in flight)."""
        done, total = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rt_scene_progress(self._h, C.byref(done), C.byref(total)))
        return done.value, total.value

    def culled_pixels(self, both=False):
        """rt_scene_culled_pixels: pixels of the latest call in empty tiles that were left out (black sky); both=True: also those rendered
        without a walk (a sky that depends on the ray), as a pair."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rt_scene_culled_pixels(self._h, C.byref(a), C.byref(b)))
        return (a.value, b.value) if both else a.value

    def profile_enable(self, enable=True):
        self._check(self._lib.rt_scene_profile_enable(self._h, 1 if enable else 0))

    def profile_read(self):
        """(summed render-kernel milliseconds, launches) since the last read."""
        ms, n = C.c_double(), C.c_uint64()
        self._check(self._lib.rt_scene_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def primary_hits(self, settings, chunk):
        """(t, normal, entity index, triangle index) of the first hit_closest per pixel."""
        h, w = chunk.height, chunk.width
        t = np.empty((h, w), dtype=np.float32)
        n = np.empty((h, w, 3), dtype=np.float32)
        e = np.empty((h, w), dtype=np.int32)
        tri = np.empty((h, w), dtype=np.int32)
        self._check(self._lib.rt_primary_hits(self._h, C.byref(settings.c), C.byref(chunk), t.ctypes.data,
                                              n.ctypes.data, e.ctypes.data, tri.ctypes.data))
        return t, n, e, tri


def render_frame_multi(scenes, settings, mode, seed, rgb_ptr0, rgba_ptr0):
    """rt_render_frame_multi: one frame on len(scenes) replicas (one Scene per GPU), composited in the first one's device
    memory.  mode: _ffi.RT_MULTI_TILES or _ffi.RT_MULTI_SPP.  Returns the summed RtRayCounters."""
    lib = _ffi.load_hip()
    handles = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    counters = RtRayCounters()
    rc = lib.rt_render_frame_multi(handles, len(scenes), C.byref(settings.c), mode, seed, rgb_ptr0, rgba_ptr0, C.byref(counters))
    if rc < 0:
        raise RtError(rc, lib.rt_last_error().decode())
    return counters


def sample_segment():
    """Segment length of the fixed per-pixel sample summation order (rt_sample_segment)."""
    return int(_ffi.load_hip().rt_sample_segment())


def strip_rows(height, strip_height, strip_first, strip_period):
    n = _ffi.load_hip().rt_strip_rows(height, strip_height, strip_first, strip_period)
    if n < 0:
        raise RtError(n, "bad strip arguments")
    return n


def tile_columns_width(width, col_first, col_period):
    """Packed width in pixels of the tile columns col_first, col_first + col_period, ... (rt_tile_columns_width)."""
    n = _ffi.load_hip().rt_tile_columns_width(width, col_first, col_period)
    if n < 0:
        raise RtError(n, "bad tile-column arguments")
    return n


def resolve_device(device, accum_ptr, n_pixels, samples_per_pixel, rgb_ptr, rgba_ptr, stream=None):
    lib = _ffi.load_hip()
    rc = lib.rt_resolve_device(device, accum_ptr, n_pixels, samples_per_pixel, rgb_ptr, rgba_ptr, stream)
    if rc < 0:
        raise RtError(rc, lib.rt_last_error().decode())


def v3_to_rgba(rgb):
    """v3_to_rgba (src/rgba.rs:6-12) over an (..., 3) float array."""
    lib = _ffi.load_hip()
    a = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.empty(a.shape[:-1] + (4,), dtype=np.uint8)
    lib.rt_v3_to_rgba(a.ctypes.data, a.size // 3, out.ctypes.data)
    return out


def decode_image(path):
    """The colour-map decoder on its own: (H, W, 4) uint8, row 0 = top."""
    lib = _ffi.load_samples()
    w, h = C.c_uint32(), C.c_uint32()
    if lib.rt_image_decode(os.fsencode(path), C.byref(w), C.byref(h), None, 0) != 0:
        raise ValueError(lib.rt_samples_last_error().decode())
    out = np.zeros((h.value, w.value, 4), dtype=np.uint8)
    if lib.rt_image_decode(os.fsencode(path), C.byref(w), C.byref(h), out.ctypes.data, out.size) != 0:
        raise ValueError(lib.rt_samples_last_error().decode())
    return out
