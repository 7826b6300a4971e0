"""Read .flxs scene files (web-ray-tracer_amd/js/flxs.js) and build the C-ABI structs from them.

A .flxs file is the complete boundary data of one frame — SURVEY.md §8a rows D1–D7 — as the
JavaScript host flattens it: geometry / attribute / id arrays (reference modules/scene.js:190-316),
transform arrays (scene.js:500-521), light array and atlases (modules/pathtracerWGL2.js:85-165),
camera + view matrix (pathtracerWGL2.js:312-318) and the BASELINE frame configuration.
"""
import ctypes as C
import gzip
import json
import os
import struct

import numpy as np

_MAGIC = b"FLXS1\n"
_DTYPES = {"f32": np.float32, "i32": np.int32, "u8": np.uint8}

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")


class SceneView(C.Structure):
    """flx_scene_view (include/flexlight_hip.h)."""
    _fields_ = [
        ("geometry", C.POINTER(C.c_float)),
        ("attributes", C.POINTER(C.c_float)),
        ("n_entries_padded", C.c_uint32),
        ("ids", C.POINTER(C.c_int32)),
        ("n_ids", C.c_uint32),
        ("rotation", C.POINTER(C.c_float)),
        ("shift", C.POINTER(C.c_float)),
        ("n_transforms", C.c_uint32),
        ("lights", C.POINTER(C.c_float)),
        ("n_lights", C.c_uint32),
        ("atlas", C.POINTER(C.c_uint8) * 3),
        ("atlas_w", C.c_uint32 * 3),
        ("atlas_h", C.c_uint32 * 3),
    ]


class FrameParams(C.Structure):
    """flx_frame_params (include/flexlight_hip.h)."""
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("camera", C.c_float * 3),
        ("view_matrix", C.c_float * 9),
        ("samples", C.c_int32), ("max_reflections", C.c_int32),
        ("min_importancy", C.c_float),
        ("use_filter", C.c_int32), ("is_temporal", C.c_int32), ("hdr", C.c_int32),
        ("ambient", C.c_float * 3),
        ("random_seed", C.c_float),
        ("texture_width", C.c_int32),
        ("tile_rows", C.c_uint32), ("tile_index", C.c_uint32), ("tile_count", C.c_uint32),
        ("temporal_samples", C.c_int32),
    ]


class Counters(C.Structure):
    """flx_counters (include/flexlight_hip.h)."""
    _fields_ = [(n, C.c_uint64) for n in (
        "primary_visits", "closest_visits", "shadow_visits", "closest_walks", "shadow_walks",
        "shades", "primary_hits", "atlas_texels")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class GBuffers(C.Structure):
    """flx_gbuffers (include/flexlight_hip.h)."""
    _fields_ = [(n, C.POINTER(C.c_float)) for n in ("color", "color_ip", "original_color", "id", "original_id", "location_id")]


def view_matrix(fx, fy, fov, width, height):
    """modules/pathtracerWGL2.js:312-318, evaluated in float64 like the JavaScript, rounded on upload."""
    inv_fov = 1.0 / fov
    k = height * inv_fov / width
    m = [np.cos(fx) * k, 0.0, np.sin(fx) * k,
         -np.sin(fx) * np.sin(fy) * inv_fov, np.cos(fy) * inv_fov, np.cos(fx) * np.sin(fy) * inv_fov,
         -np.sin(fx) * np.cos(fy), -np.sin(fy), np.cos(fx) * np.cos(fy)]
    return np.asarray(m, dtype=np.float64).astype(np.float32)


class Scene:
    """One loaded .flxs file: `meta` (dict) and `arrays` (name -> contiguous numpy array)."""

    def __init__(self, meta, arrays):
        self.meta = meta
        self.arrays = arrays

    @classmethod
    def load(cls, path):
        opener = gzip.open if path.endswith(".gz") else open
        with opener(path, "rb") as fh:
            blob = fh.read()
        if blob[:len(_MAGIC)] != _MAGIC:
            raise ValueError("not a .flxs file: %s" % path)
        (json_len,) = struct.unpack_from("<I", blob, len(_MAGIC))
        head = len(_MAGIC) + 4
        desc = json.loads(blob[head:head + json_len].decode("utf8"))
        base = (head + json_len + 15) & ~15
        arrays = {}
        for e in desc["arrays"]:
            dt = np.dtype(_DTYPES[e["dtype"]])
            start = base + e["offset"]
            arrays[e["name"]] = np.frombuffer(blob, dtype=dt, count=e["count"], offset=start).copy()
        return cls(desc["meta"], arrays)

    @classmethod
    def golden(cls, name):
        """tests/golden/ref_<name>.flxs.gz — arrays emitted by the reference's own scene.js."""
        return cls.load(os.path.join(GOLDEN_DIR, "ref_%s.flxs.gz" % name))

    # -- C structs ------------------------------------------------------------------------------
    def view(self):
        a = self.arrays
        v = SceneView()
        fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
        v.geometry, v.attributes = fp(a["geometry"]), fp(a["attributes"])
        v.n_entries_padded = a["geometry"].size // 12
        v.ids = a["ids"].ctypes.data_as(C.POINTER(C.c_int32))
        v.n_ids = a["ids"].size
        v.rotation, v.shift = fp(a["rotation"]), fp(a["shift"])
        v.n_transforms = a["shift"].size // 8
        v.lights = fp(a["lights"])
        v.n_lights = a["lights"].size // 6
        for i, key in enumerate(("albedo", "pbr", "tpo")):
            arr = a["atlas" + key.capitalize()]
            w, h = self.meta["atlas"][key]
            assert arr.size == w * h * 4
            v.atlas[i] = arr.ctypes.data_as(C.POINTER(C.c_uint8))
            v.atlas_w[i], v.atlas_h[i] = w, h
        v._keepalive = self
        return v

    def frame_params(self, width=None, height=None, samples=None, max_reflections=None, use_filter=None,
                     min_importancy=0.3, hdr=1, tile=(0, 0, 0)):
        """BASELINE frame of this scene (meta['frame']), any field overridable."""
        fr = self.meta["frame"]
        cam = self.meta["camera"]
        p = FrameParams()
        p.width = int(width if width is not None else fr["width"])
        p.height = int(height if height is not None else fr["height"])
        p.camera[:] = [cam["x"], cam["y"], cam["z"]]
        p.view_matrix[:] = view_matrix(cam["fx"], cam["fy"], cam["fov"], p.width, p.height).tolist()
        p.samples = int(samples if samples is not None else fr["samplesPerRay"])
        p.max_reflections = int(max_reflections if max_reflections is not None else fr["maxReflections"])
        p.min_importancy = min_importancy
        p.use_filter = int(fr["filter"] if use_filter is None else use_filter)
        p.is_temporal = 0
        p.hdr = hdr
        p.ambient[:] = self.meta["ambient"]
        p.random_seed = 0.0
        p.texture_width = int(self.meta["textureWidth"])
        p.tile_rows, p.tile_index, p.tile_count = tile
        p.temporal_samples = 4
        return p
