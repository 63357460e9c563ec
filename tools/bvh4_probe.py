"""Primary rays of a bench workload's frame through the two-box walk and through a 4-wide collapse of the same tree
(diagnostics build: make -C raytracer_amd/csrc bvh4).   usage: RT_HIP_LIB=raytracer_amd/librt_hip_bvh4.so python tools/bvh4_probe.py <workload>..."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS
from raytracer_amd import SceneBuilder, CameraConfiguration, Scene, RenderSettings, RenderChunk, _ffi
lib = _ffi.load_hip()
lib.rt_debug_bvh4_primary.restype = C.c_int
lib.rt_debug_bvh4_primary.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
for wl in sys.argv[1:]:
    name, W, H, spp, depth, lens, params = WORKLOADS[wl]
    sc = Scene(SceneBuilder.sample(name, CameraConfiguration(W, H, lens_radius=lens), params=params), 0)
    st, ch = RenderSettings(W, H, depth, 1), RenderChunk(0, 0, W, H)
    a, b, bad = C.c_double(), C.c_double(), C.c_uint64()
    rc = lib.rt_debug_bvh4_primary(sc._h, C.byref(st.c), C.byref(ch), 20, C.byref(a), C.byref(b), C.byref(bad))
    print(f"{wl}: rc {rc}, primary rays of one {W}x{H} frame: two-box walk {a.value:.3f} ms, 4-wide walk {b.value:.3f} ms ({a.value / b.value:.2f}x), "
          f"differing pixels {bad.value}", flush=True)
    sc.close()
