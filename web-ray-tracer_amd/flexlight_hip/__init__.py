"""flexlight_hip — Python face of libflexlight_hip.so (FlexLight's path-tracing inner loop on MI355X).

`scene_io` is importable anywhere; `capi` needs the built shared library and raises without it."""
from .scene_io import Counters, FrameParams, GBuffers, Scene, SceneView  # noqa: F401
