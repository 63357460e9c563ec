"""20 x rt_scene_rebuild_tlas of the 900-instance scene (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import SceneBuilder, CameraConfiguration, Scene
cfg = CameraConfiguration(640, 360, lens_radius=0.1)
a = SceneBuilder.sample("Doot Doot", cfg, params={"Count": 30, "Spacing": 30})
b = SceneBuilder.sample("Doot Doot", cfg, params={"Count": 30, "Spacing": 31})
sc = Scene(a, 0)
for _ in range(10):
    sc.rebuild_tlas(b); sc.rebuild_tlas(a)
sc.close()
