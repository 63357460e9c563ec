"""ctypes mirror of include/rt_hip.h and include/rt_samples.h.

Plumbing only: every struct below is a field-for-field copy of the C header,
and the two loaders open the in-tree shared libraries built by
``raytracer_amd/csrc/Makefile``.  There is no CPU fallback: if ``librt_hip.so``
is missing the import of the render entry points fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
ASSET_MESH_DIR = os.path.join(REPO_ROOT, "assets", "meshes")

RT_ABI_VERSION = 1
RT_OK, RT_E_INVALID, RT_E_HIP, RT_E_NO_DEVICE, RT_E_DEPTH, RT_E_IO, RT_E_LIMIT, RT_E_CANCELLED = 0, -1, -2, -3, -4, -5, -6, -7
RT_MULTI_TILES, RT_MULTI_SPP = 0, 1

RT_HITABLE_SPHERE, RT_HITABLE_PLANE, RT_HITABLE_MESH = 0, 1, 2
RT_MAT_LAMBERTIAN, RT_MAT_SPECULAR, RT_MAT_DIELECTRIC = 0, 1, 2
(RT_TEX_COLOR, RT_TEX_CHECKER, RT_TEX_UV_TEST, RT_TEX_XYZ_TEST,
 RT_TEX_COLOR_MAP, RT_TEX_MESH, RT_TEX_MESH_SET) = range(7)
RT_LIGHT_POINT, RT_LIGHT_LAMP, RT_LIGHT_DIRECTIONAL = 0, 1, 2
RT_SKY_DAY, RT_SKY_BLACK = 0, 1
RT_DEVICE_HOST_ONLY = -1
# rt_scene_set_option
RT_OPT_LENS_SAMPLER, RT_OPT_FINE_SPLIT, RT_OPT_ROUND_UNITS, RT_OPT_LENS_COC_PX, RT_OPT_DEBUG_STACK_ENTRIES = 1, 2, 3, 4, 100
RT_OPT_CULL_EMPTY_TILES = 5
RT_OPT_DEBUG_REBUILD_STACK_LIMIT = 101


class RtV3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(x, y, z)

    def tuple(self):
        return (self.x, self.y, self.z)


class RtV2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class RtCamera(C.Structure):
    _fields_ = [("lower_left_corner", RtV3), ("horizontal", RtV3), ("vertical", RtV3),
                ("origin", RtV3), ("u", RtV3), ("v", RtV3), ("lens_radius", C.c_float)]


class RtSphere(C.Structure):
    _fields_ = [("radius", C.c_float), ("mat_id", C.c_uint32), ("tex_id", C.c_uint32)]


class RtPlane(C.Structure):
    _fields_ = [("normal", RtV3), ("u_basis", RtV3), ("v_basis", RtV3), ("radius", C.c_float),
                ("has_radius", C.c_uint32), ("mat_id", C.c_uint32), ("tex_id", C.c_uint32)]


class RtMeshTri(C.Structure):
    _fields_ = [("a", RtV3), ("b", RtV3), ("c", RtV3), ("a_uv", RtV2), ("b_uv", RtV2), ("c_uv", RtV2),
                ("tex_key", C.c_int32)]


class RtMesh(C.Structure):
    _fields_ = [("tris", C.POINTER(RtMeshTri)), ("tri_count", C.c_uint64)]


class RtMeshObject(C.Structure):
    _fields_ = [("mesh_index", C.c_uint32), ("mat_id", C.c_uint32), ("tex_id", C.c_uint32)]


class RtRotation(C.Structure):
    _fields_ = [("axis", RtV3), ("theta", C.c_float)]


class RtEntity(C.Structure):
    _fields_ = [("hitable_kind", C.c_uint32), ("hitable_index", C.c_uint32), ("id", C.c_int32),
                ("translation_first", C.c_uint32), ("translation_count", C.c_uint32),
                ("rotation_first", C.c_uint32), ("rotation_count", C.c_uint32)]


class RtMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("reflectivity", C.c_float), ("fuzz", C.c_float),
                ("opacity", C.c_float), ("ref_index", C.c_float)]


class RtTexture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("color", RtV3), ("color2", RtV3), ("size", C.c_float),
                ("child0", C.c_int32), ("child1", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("texel_first", C.c_uint64)]


class RtLight(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("origin", RtV3), ("direction", RtV3), ("color", RtV3),
                ("intensity", C.c_float), ("angle_deg", C.c_float)]


class RtSceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("camera", RtCamera), ("sky", C.c_uint32),
        ("entities", C.POINTER(RtEntity)), ("entity_count", C.c_uint64),
        ("translations", C.POINTER(RtV3)), ("translation_count", C.c_uint64),
        ("rotations", C.POINTER(RtRotation)), ("rotation_count", C.c_uint64),
        ("spheres", C.POINTER(RtSphere)), ("sphere_count", C.c_uint64),
        ("planes", C.POINTER(RtPlane)), ("plane_count", C.c_uint64),
        ("mesh_objects", C.POINTER(RtMeshObject)), ("mesh_object_count", C.c_uint64),
        ("meshes", C.POINTER(RtMesh)), ("mesh_count", C.c_uint64),
        ("materials", C.POINTER(RtMaterial)), ("material_count", C.c_uint64),
        ("textures", C.POINTER(RtTexture)), ("texture_count", C.c_uint64),
        ("tex_set_entries", C.POINTER(C.c_int32)), ("tex_set_entry_count", C.c_uint64),
        ("texels", C.POINTER(RtV3)), ("texel_count", C.c_uint64),
        ("lights", C.POINTER(RtLight)), ("light_count", C.c_uint64),
    ]


class RtSettings(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_reflections", C.c_uint32),
                ("samples_per_pixel", C.c_uint32)]


class RtRenderChunk(C.Structure):
    _fields_ = [("left", C.c_uint32), ("top", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32)]

    def __repr__(self):
        return f"RenderChunk(left={self.left}, top={self.top}, width={self.width}, height={self.height})"

    def astuple(self):
        return (self.left, self.top, self.width, self.height)


class RtRayCounters(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("secondary", C.c_uint64), ("shadow", C.c_uint64), ("samples", C.c_uint64)]

    @property
    def rays(self):
        return self.primary + self.secondary + self.shadow


class RtSceneInfo(C.Structure):
    _fields_ = [("entity_count", C.c_uint64), ("mesh_count", C.c_uint64), ("triangle_count", C.c_uint64),
                ("instanced_triangle_count", C.c_uint64), ("tlas_node_count", C.c_uint64),
                ("blas_node_count", C.c_uint64), ("tlas_depth", C.c_uint32), ("blas_depth", C.c_uint32),
                ("device_bytes", C.c_uint64)]


class RtCameraConfig(C.Structure):
    _fields_ = [("width", C.c_float), ("height", C.c_float), ("lens_radius", C.c_float), ("fov", C.c_float),
                ("angle_adjust_v", C.c_float), ("angle_adjust_h", C.c_float), ("focus_dist_adjust", C.c_float)]


_P = C.POINTER
_f32p, _u8p, _i32p, _u32p = _P(C.c_float), _P(C.c_uint8), _P(C.c_int32), _P(C.c_uint32)

# name -> (restype, argtypes): exactly the prototypes of include/rt_hip.h
HIP_PROTOTYPES = {
    "rt_device_count": (C.c_int, []),
    "rt_last_error": (C.c_char_p, []),
    "rt_scene_create": (C.c_void_p, [_P(RtSceneDesc), C.c_int]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_render_chunk": (C.c_int, [C.c_void_p, _P(RtSettings), _P(RtRenderChunk), C.c_uint64, C.c_void_p, C.c_void_p,
                                  _P(RtRayCounters)]),
    "rt_render_chunk_device": (C.c_int, [C.c_void_p, _P(RtSettings), _P(RtRenderChunk), C.c_uint64, C.c_void_p,
                                         C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "rt_render_chunk_device_noalloc": (C.c_int, [C.c_void_p, _P(RtSettings), _P(RtRenderChunk), C.c_uint64, C.c_void_p,
                                         C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "rt_noalloc_max_pixels": (C.c_uint64, []),
    "rt_render_strips_device": (C.c_int, [C.c_void_p, _P(RtSettings), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_sample_segment": (C.c_uint32, []),
    "rt_strip_rows": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rt_render_tile_columns_device": (C.c_int, [C.c_void_p, _P(RtSettings), C.c_uint32, C.c_uint32, C.c_uint64,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_tile_columns_width": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "rt_accumulate_chunk_device": (C.c_int, [C.c_void_p, _P(RtSettings), _P(RtRenderChunk), C.c_uint64, C.c_uint32,
                                             C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_resolve_device": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_primary_hits": (C.c_int, [C.c_void_p, _P(RtSettings), _P(RtRenderChunk), C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "rt_create_render_chunks": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P(RtRenderChunk), C.c_uint32]),
    "rt_v3_to_rgba": (None, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "rt_scene_info": (C.c_int, [C.c_void_p, _P(RtSceneInfo)]),
    "rt_scene_bvh_order": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_uint64]),
    "rt_scene_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_scene_set_option": (C.c_int, [C.c_void_p, C.c_uint32, C.c_double]),
    "rt_scene_update_transforms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]),
    "rt_scene_rebuild_tlas": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]),
    "rt_scene_profile_read": (C.c_int, [C.c_void_p, _P(C.c_double), _P(C.c_uint64)]),
    "rt_scene_rebuild_blas": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]),
    "rt_render_frame_multi": (C.c_int, [_P(C.c_void_p), C.c_uint32, _P(RtSettings), C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p,
                                        _P(RtRayCounters)]),
    "rt_render_frame_multi_host": (C.c_int, [_P(C.c_void_p), C.c_uint32, _P(RtSettings), C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p,
                                             _P(RtRayCounters)]),
    "rt_scene_cancel": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_scene_progress": (C.c_int, [C.c_void_p, _P(C.c_uint64), _P(C.c_uint64)]),
    "rt_scene_culled_pixels": (C.c_int, [C.c_void_p, _P(C.c_uint64), _P(C.c_uint64)]),
}

SAMPLES_PROTOTYPES = {
    "rt_samples_last_error": (C.c_char_p, []),
    "rt_make_camera": (C.c_int, [_P(RtCameraConfig), _f32p, _f32p, _P(RtCamera)]),
    "rt_camera_new": (C.c_int, [_f32p, _f32p, C.c_float, C.c_float, C.c_float, C.c_float, _P(RtCamera)]),
    "rt_builder_new": (C.c_void_p, [_P(RtCamera), C.c_uint32]),
    "rt_builder_destroy": (None, [C.c_void_p]),
    "rt_builder_add_material": (C.c_int, [C.c_void_p, _P(RtMaterial)]),
    "rt_builder_add_texture": (C.c_int, [C.c_void_p, _P(RtTexture)]),
    "rt_builder_add_color_map": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "rt_builder_load_color_map": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rt_image_decode": (C.c_int, [C.c_char_p, _P(C.c_uint32), _P(C.c_uint32), C.c_void_p, C.c_uint64]),
    "rt_samples_last_warning": (C.c_char_p, []),
    "rt_builder_add_light": (C.c_int, [C.c_void_p, _P(RtLight)]),
    "rt_builder_add_mesh": (C.c_int, [C.c_void_p, _P(RtMeshTri), C.c_uint64]),
    "rt_builder_load_obj": (C.c_int, [C.c_void_p, C.c_char_p, _P(C.c_int)]),
    "rt_builder_add_entity": (C.c_int, [C.c_void_p, C.c_uint32, _P(RtSphere), _P(RtPlane), _P(RtMeshObject), C.c_int32,
                                        _P(RtV3), C.c_uint32, _P(RtRotation), C.c_uint32]),
    "rt_plane_new": (C.c_int, [_f32p, C.c_uint32, C.c_uint32, C.c_int, C.c_float, _P(RtPlane)]),
    "rt_builder_desc": (_P(RtSceneDesc), [C.c_void_p]),
    "rt_samples_count": (C.c_int, []),
    "rt_samples_name": (C.c_char_p, [C.c_int]),
    "rt_samples_create": (C.c_void_p, [C.c_char_p, _P(RtCameraConfig), C.c_char_p, C.c_char_p]),
}


def _bind(lib, prototypes):
    for name, (res, args) in prototypes.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


_hip_lib = None
_samples_lib = None


def hip_library_path():
    # RT_HIP_LIB: A/B an alternative build of the same library (experiments only)
    return os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")


def samples_library_path():
    return os.path.join(_HERE, "librt_samples.so")


def load_hip():
    """Open librt_hip.so (the HIP render core). Raises if it was not built."""
    global _hip_lib
    if _hip_lib is None:
        path = hip_library_path()
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the render core has no CPU fallback)")
        # One HIP runtime per process: PyTorch ships its own libamdhip64, and if librt_hip.so
        # pulls in /opt/rocm's first, torch afterwards finds no usable GPU ("No HIP GPUs are
        # available").  With torch loaded first both share torch's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _hip_lib = _bind(C.CDLL(path), HIP_PROTOTYPES)
    return _hip_lib


def load_samples():
    global _samples_lib
    if _samples_lib is None:
        path = samples_library_path()
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: run __graft_entry__.build()")
        _samples_lib = _bind(C.CDLL(path), SAMPLES_PROTOTYPES)
    return _samples_lib
