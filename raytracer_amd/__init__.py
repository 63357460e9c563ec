"""raytracer_amd - MI355X-native render core behind deadalusai/raytracer's
scene / settings / render-chunk API."""
from .scene import (CameraConfiguration, RenderChunk, RenderSettings, RtError, Scene, SceneBuilder,  # noqa: F401
                    camera_new, create_render_chunks, render_frame_multi, resolve_device, sample_segment, strip_rows, tile_columns_width, v3_to_rgba, decode_image)
