"""ctypes binding of libflexlight_hip.so (include/flexlight_hip.h).

This is the Python face of the C ABI, used by tests and bench.py; the JavaScript renderer binds the
same entry points through N-API (web-ray-tracer_amd/napi).  There is no fallback: if the shared
library is missing, importing this module raises, and without a GPU Context() raises.
"""
import ctypes as C
import os

import numpy as np

from .scene_io import Counters, FrameParams, GBuffers, SceneView

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLX_LIB") or os.path.join(_HERE, "libflexlight_hip.so")   # FLX_LIB: A/B a variant build

EXPORTS = [
    "flx_context_create", "flx_context_destroy", "flx_last_error", "flx_scene_upload", "flx_transforms_upload",
    "flx_lights_upload", "flx_atlas_upload", "flx_scene_upload_view", "flx_tile_row_count", "flx_tile_row_at",
    "flx_render", "flx_render_device", "flx_sync", "flx_set_stream", "flx_set_counters_enabled", "flx_get_counters",
    "flx_last_frame_ms", "flx_debug_math", "flx_debug_intersect", "flx_debug_walk", "flx_device_info", "flx_version", "flx_set_pipeline", "flx_set_lockstep", "flx_last_pipeline", "flx_get_diag", "flx_set_wavefront_groups", "flx_temporal_reset", "flx_set_walk_scheduler", "flx_render_batch", "flx_render_batch_device", "flx_render_planes_device", "flx_filter_planes_device",
    "flx_mesh_import_obj", "flx_mesh_destroy", "flx_mesh_entry_count", "flx_mesh_triangle_count", "flx_mesh_set_transform", "flx_mesh_move",
    "flx_mesh_scale", "flx_mesh_set_material", "flx_mesh_bounding", "flx_mesh_flatten", "flx_transforms_pack", "flx_fxaa_device", "flx_taa_device", "flx_fxaa", "flx_taa", "flx_taa_reset", "flx_present", "flx_present_device",
    "flx_comm_unique_id", "flx_comm_init_rank", "flx_comm_destroy", "flx_render_gathered_device",
    "flx_group_create", "flx_group_destroy", "flx_group_last_error", "flx_group_size", "flx_group_uses_rccl", "flx_group_context",
    "flx_frame_begin", "flx_frame_end", "flx_frames_in_flight", "flx_set_frame_lanes", "flx_set_server_moving_scenes", "flx_server_moving", "flx_get_tail_diag", "flx_set_frame_chain", "flx_last_chained", "flx_debug_inject_fault", "flx_set_chain_stats", "flx_get_chain_stats", "flx_get_server_stats", "flx_get_server_dump", "flx_set_chain_order", "flx_set_chain_cost", "flx_get_chain_cost",
    "flx_render_gathered_root_device", "flx_comm_count", "flx_frame_begin_gathered", "flx_group_set_gather", "flx_frame_host_slots", "flx_has_experiments", "flx_set_wavefront_organisation", "flx_set_frame_front", "flx_last_organisation",
    "flx_group_scene_upload", "flx_group_transforms_upload", "flx_group_lights_upload", "flx_group_atlas_upload", "flx_group_scene_upload_view", "flx_group_render",
    "flx_group_frame_begin", "flx_group_frame_end", "flx_group_frames_in_flight", "flx_group_set_frame_lanes",
    "flx_frame_server_takes", "flx_frame_target_set", "flx_frame_target_index", "flx_debug_set_server_groups",
    "flx_share_create", "flx_share_join", "flx_share_leave", "flx_frame_begin_shared", "flx_frame_end_shared",
    "flx_render_gathered_rgba8_device", "flx_group_render_rgba8", "flx_debug_set_angle_table", "flx_frame_target_set8", "flx_debug_set_walk_jobs", "flx_debug_set_sample_parallel", "flx_debug_set_tile_order", "flx_debug_tile_cost", "flx_debug_set_adaptive_order", "flx_debug_tile_order_of",
]


EXPERIMENTS_ONLY = ("flx_set_chain_stats", "flx_get_chain_stats", "flx_set_chain_order", "flx_set_chain_cost", "flx_get_chain_cost")      # include/flexlight_hip_debug.h, #ifdef FLX_EXPERIMENTS

SHARE_HANDLE_BYTES = 128      # FLX_SHARE_HANDLE_BYTES
MAX_BATCH_FRAMES = 32          # FLX_MAX_BATCH_FRAMES of include/flexlight_hip.h


class FlexLightHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise FlexLightHipError(
            "libflexlight_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C web-ray-tracer_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    fp, u32, vp = C.POINTER(C.c_float), C.c_uint32, C.c_void_p
    sig = {
        "flx_context_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "flx_context_destroy": (None, [vp]),
        "flx_last_error": (C.c_char_p, [vp]),
        "flx_scene_upload": (C.c_int, [vp, fp, fp, u32, C.POINTER(C.c_int32), u32]),
        "flx_transforms_upload": (C.c_int, [vp, fp, fp, u32]),
        "flx_lights_upload": (C.c_int, [vp, fp, u32]),
        "flx_atlas_upload": (C.c_int, [vp, C.c_int, C.POINTER(C.c_uint8), u32, u32]),
        "flx_scene_upload_view": (C.c_int, [vp, C.POINTER(SceneView)]),
        "flx_tile_row_count": (u32, [C.POINTER(FrameParams)]),
        "flx_tile_row_at": (u32, [C.POINTER(FrameParams), u32]),
        "flx_render": (C.c_int, [vp, C.POINTER(FrameParams), fp, C.POINTER(GBuffers), C.POINTER(Counters)]),
        "flx_render_device": (C.c_int, [vp, C.POINTER(FrameParams), vp]),
        "flx_sync": (C.c_int, [vp]),
        "flx_set_stream": (C.c_int, [vp, vp]),
        "flx_set_counters_enabled": (C.c_int, [vp, C.c_int]),
        "flx_get_counters": (C.c_int, [vp, C.POINTER(Counters)]),
        "flx_last_frame_ms": (C.c_int, [vp, fp, fp]),
        "flx_debug_math": (C.c_int, [vp, C.c_int, fp, fp, fp, u32]),
        "flx_device_info": (C.c_int, [vp, C.c_char_p, u32, C.POINTER(u32)]),
        "flx_version": (C.c_char_p, []),
        "flx_set_pipeline": (C.c_int, [vp, C.c_int]),
        "flx_set_lockstep": (C.c_int, [vp, C.c_int]),
        "flx_get_diag": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "flx_set_wavefront_groups": (C.c_int, [vp, C.c_int]),
        "flx_last_pipeline": (C.c_int, [vp, C.POINTER(C.c_int)]),
        "flx_debug_intersect": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32]),
        "flx_debug_walk": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32]),
        "flx_temporal_reset": (C.c_int, [vp]),
        "flx_set_walk_scheduler": (C.c_int, [vp, C.c_int, C.c_uint32]),
        "flx_render_batch": (C.c_int, [vp, C.POINTER(FrameParams), u32, fp, C.POINTER(Counters)]),
        "flx_render_batch_device": (C.c_int, [vp, C.POINTER(FrameParams), u32, vp]),
        "flx_render_planes_device": (C.c_int, [vp, C.c_void_p, C.c_void_p]),
        "flx_filter_planes_device": (C.c_int, [vp, C.c_void_p, C.c_void_p, C.c_void_p]),
        "flx_mesh_import_obj": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(vp)]),
        "flx_mesh_destroy": (None, [vp]),
        "flx_mesh_entry_count": (C.c_uint32, [vp]),
        "flx_mesh_triangle_count": (C.c_uint32, [vp]),
        "flx_mesh_set_transform": (C.c_int, [vp, C.c_uint32]),
        "flx_mesh_move": (C.c_int, [vp, C.c_double, C.c_double, C.c_double]),
        "flx_mesh_scale": (C.c_int, [vp, C.c_double]),
        "flx_mesh_set_material": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double)]),
        "flx_mesh_bounding": (C.c_int, [vp, C.POINTER(C.c_double)]),
        "flx_mesh_flatten": (C.c_int, [vp, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
        "flx_transforms_pack": (C.c_int, [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "flx_fxaa_device": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
        "flx_taa_device": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
        "flx_fxaa": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
        "flx_taa": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
        "flx_taa_reset": (C.c_int, [vp]),
        "flx_present": (C.c_int, [vp, u32, u32, vp, vp]),
        "flx_present_device": (C.c_int, [vp, u32, u32, vp, vp]),
        "flx_get_tail_diag": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "flx_frame_begin": (C.c_int, [vp, C.POINTER(FrameParams), C.c_int]),
        "flx_frame_end": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_size_t), fp]),
        "flx_frames_in_flight": (C.c_int, [vp]),
        "flx_set_frame_lanes": (C.c_int, [vp, C.c_int]),
        "flx_set_frame_chain": (C.c_int, [vp, C.c_int]),
        "flx_set_server_moving_scenes": (C.c_int, [vp, C.c_int]),
        "flx_server_moving": (C.c_int, [vp]),
        "flx_last_chained": (C.c_int, [vp, C.POINTER(C.c_int)]),
        "flx_debug_inject_fault": (C.c_int, [vp, u32, u32]),
        "flx_set_chain_stats": (C.c_int, [vp, C.c_int]),
        "flx_get_server_stats": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "flx_get_server_dump": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "flx_get_chain_stats": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "flx_set_chain_order": (C.c_int, [vp, C.POINTER(C.c_uint32), u32]),
        "flx_set_chain_cost": (C.c_int, [vp, u32]),
        "flx_get_chain_cost": (C.c_int, [vp, C.POINTER(C.c_uint32)]),
        "flx_comm_unique_id": (C.c_int, [C.c_char_p]),
        "flx_comm_init_rank": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_int]),
        "flx_comm_destroy": (C.c_int, [vp]),
        "flx_render_gathered_device": (C.c_int, [vp, C.POINTER(FrameParams), u32, vp]),
        "flx_render_gathered_root_device": (C.c_int, [vp, C.POINTER(FrameParams), u32, C.c_int, vp]),
        "flx_comm_count": (C.c_int, [vp]),
        "flx_frame_begin_gathered": (C.c_int, [vp, C.POINTER(FrameParams), C.c_int, C.c_int]),
        "flx_group_set_gather": (C.c_int, [vp, C.c_int]),
        "flx_frame_host_slots": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_int)]),
        "flx_has_experiments": (C.c_int, []),
        "flx_set_wavefront_organisation": (C.c_int, [vp, C.c_int]),
        "flx_set_frame_front": (C.c_int, [vp, C.c_int]),
        "flx_last_organisation": (C.c_int, [vp, C.POINTER(C.c_int)]),
        "flx_group_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]),
        "flx_group_destroy": (None, [vp]),
        "flx_group_last_error": (C.c_char_p, [vp]),
        "flx_group_size": (C.c_int, [vp]),
        "flx_group_uses_rccl": (C.c_int, [vp]),
        "flx_group_context": (vp, [vp, C.c_int]),
        "flx_group_scene_upload": (C.c_int, [vp, fp, fp, u32, C.POINTER(C.c_int32), u32]),
        "flx_group_transforms_upload": (C.c_int, [vp, fp, fp, u32]),
        "flx_group_lights_upload": (C.c_int, [vp, fp, u32]),
        "flx_group_atlas_upload": (C.c_int, [vp, C.c_int, C.POINTER(C.c_uint8), u32, u32]),
        "flx_group_scene_upload_view": (C.c_int, [vp, C.POINTER(SceneView)]),
        "flx_group_render": (C.c_int, [vp, C.POINTER(FrameParams), u32, u32, fp, C.POINTER(Counters)]),
        "flx_group_render_rgba8": (C.c_int, [vp, C.POINTER(FrameParams), u32, u32, C.POINTER(C.c_uint8), C.POINTER(Counters)]),
        "flx_render_gathered_rgba8_device": (C.c_int, [vp, C.POINTER(FrameParams), u32, C.c_int, vp]),
        "flx_debug_set_angle_table": (C.c_int, [vp, C.c_int]),
        "flx_frame_target_set8": (C.c_int, [vp, C.POINTER(vp), u32]),
        "flx_group_frame_begin": (C.c_int, [vp, C.POINTER(FrameParams), u32, C.c_int]),
        "flx_group_frame_end": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_float)]),
        "flx_group_frames_in_flight": (C.c_int, [vp]),
        "flx_group_set_frame_lanes": (C.c_int, [vp, C.c_int]),
        "flx_frame_server_takes": (C.c_int, [vp, C.POINTER(FrameParams)]),
        "flx_frame_target_set": (C.c_int, [vp, C.POINTER(vp), u32]),
        "flx_frame_target_index": (C.c_int, [vp]),
        "flx_debug_set_server_groups": (C.c_int, [vp, u32]),
        "flx_share_create": (C.c_int, [vp, u32, u32, u32, C.c_int, C.c_int, C.c_char_p]),
        "flx_share_join": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "flx_share_leave": (C.c_int, [vp]),
        "flx_debug_set_walk_jobs": (C.c_int, [vp, C.c_int]),
        "flx_debug_set_sample_parallel": (C.c_int, [vp, C.c_int]),
        "flx_debug_set_adaptive_order": (C.c_int, [vp, C.c_int]),
        "flx_debug_tile_order_of": (C.c_int, [vp, C.POINTER(C.c_float), u32, C.c_int, C.POINTER(u32)]),
        "flx_debug_set_tile_order": (C.c_int, [vp, C.POINTER(u32), u32]),
        "flx_debug_tile_cost": (C.c_int, [vp, C.POINTER(C.c_uint64), u32]),
        "flx_frame_begin_shared": (C.c_int, [vp, C.POINTER(FrameParams)]),
        "flx_frame_end_shared": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_float)]),
    }
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if name in EXPERIMENTS_ONLY:       # the chain of launches: `make EXPERIMENTS=1` only
                continue
            if os.environ.get("FLX_LIB"):      # an A/B variant (an earlier round's library): what it lacks fails when it is called
                continue
            raise
        fn.restype, fn.argtypes = res, args
    return lib


LIB = _load()


def has_experiments():
    """the loaded library carries the experimental walk schedulers (`make EXPERIMENTS=1`)"""
    return bool(LIB.flx_has_experiments())


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Context:
    """One GPU context (flx_context).  Mirrors the life cycle of the reference renderer object:
    construct -> updateScene()/updatePrimaryLightSources() -> render frames -> halt()."""

    def __init__(self, device=0):
        h = C.c_void_p()
        rc = LIB.flx_context_create(device, C.byref(h))
        if rc != 0:
            raise FlexLightHipError("flx_context_create(%d) failed (%d): %s" % (device, rc, LIB.flx_last_error(None).decode()))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            LIB.flx_context_destroy(self._h)
            self._h = None

    halt = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise FlexLightHipError("%s failed (%d): %s" % (what, rc, LIB.flx_last_error(self._h).decode()))

    # -- uploads ------------------------------------------------------------------------------------
    def update_scene(self, scene):
        """Scene (scene_io.Scene): everything at once, like render()'s first updateScene()."""
        view = scene.view()
        self._check(LIB.flx_scene_upload_view(self._h, C.byref(view)), "flx_scene_upload_view")

    def upload_view(self, view):
        self._check(LIB.flx_scene_upload_view(self._h, C.byref(view)), "flx_scene_upload_view")

    def update_primary_light_sources(self, lights):
        lights = np.ascontiguousarray(lights, np.float32).reshape(-1)
        self._check(LIB.flx_lights_upload(self._h, _fp(lights), lights.size // 6), "flx_lights_upload")

    def update_transforms(self, rotation, shift):
        rotation = np.ascontiguousarray(rotation, np.float32).reshape(-1)
        shift = np.ascontiguousarray(shift, np.float32).reshape(-1)
        self._check(LIB.flx_transforms_upload(self._h, _fp(rotation), _fp(shift), shift.size // 8), "flx_transforms_upload")

    # -- frames ---------------------------------------------------------------------------------------
    @staticmethod
    def tile_row_count(params):
        return int(LIB.flx_tile_row_count(C.byref(params)))

    @staticmethod
    def tile_rows(params):
        return [int(LIB.flx_tile_row_at(C.byref(params), k)) for k in range(Context.tile_row_count(params))]

    def render(self, params, gbuffers=False, counters=False):
        """One frame -> (rgba [rows, W, 4] float32, counters dict or None, gbuffers dict or None)."""
        rows = self.tile_row_count(params)
        out = np.zeros((rows, params.width, 4), np.float32)
        cnt = Counters() if counters else None
        gb, gbs = None, None
        if gbuffers:
            gbs = {n: np.zeros((rows, params.width, 4), np.float32) for n, _ in GBuffers._fields_}
            gb = GBuffers(*[_fp(gbs[n]) for n, _ in GBuffers._fields_])
        rc = LIB.flx_render(self._h, C.byref(params), _fp(out), C.byref(gb) if gb else None, C.byref(cnt) if cnt else None)
        self._check(rc, "flx_render")
        return out, (cnt.as_dict() if cnt else None), gbs

    def render_batch(self, params_list, counters=False):
        """1 .. 32 frames in one pass -> (rgba [n, rows, W, 4] float32, counters dict (summed over the batch) or None)."""
        n = len(params_list)
        arr = (FrameParams * n)(*params_list)
        rows, width = (self.tile_row_count(params_list[0]), params_list[0].width) if n else (0, 0)
        out = np.zeros((n, rows, width, 4), np.float32) if n else np.zeros(4, np.float32)      # an empty batch is the library's to refuse
        cnt = Counters() if counters else None
        self._check(LIB.flx_render_batch(self._h, arr, n, _fp(out), C.byref(cnt) if cnt else None), "flx_render_batch")
        return out, (cnt.as_dict() if cnt else None)

    def render_batch_device(self, params_list, device_ptr):
        n = len(params_list)
        arr = (FrameParams * n)(*params_list)
        self._check(LIB.flx_render_batch_device(self._h, arr, n, C.c_void_p(device_ptr)), "flx_render_batch_device")

    # -- several GPUs, one process per GPU (include/flexlight_hip.h: flx_comm_*) ------------------------
    def comm_init_rank(self, comm_id, n_ranks, rank):
        """join this context to the RCCL communicator of `comm_id` (bytes from comm_unique_id() of rank 0); collective"""
        self._check(LIB.flx_comm_init_rank(self._h, bytes(comm_id), int(n_ranks), int(rank)), "flx_comm_init_rank")

    def comm_destroy(self):
        self._check(LIB.flx_comm_destroy(self._h), "flx_comm_destroy")

    def render_gathered_device(self, params_list, device_ptr):
        """this rank's strips of the frames, all-gathered over RCCL and put in image order: float4[n][H][W] at device_ptr"""
        n = len(params_list)
        arr = (FrameParams * n)(*params_list)
        self._check(LIB.flx_render_gathered_device(self._h, arr, n, C.c_void_p(device_ptr)), "flx_render_gathered_device")

    def render_gathered_root_device(self, params_list, root, device_ptr):
        """the same with one receiver: ncclSend / ncclRecv to rank `root`, which alone gets float4[n][H][W] at device_ptr (0 elsewhere)"""
        n = len(params_list)
        arr = (FrameParams * n)(*params_list)
        self._check(LIB.flx_render_gathered_root_device(self._h, arr, n, int(root), C.c_void_p(device_ptr or 0)), "flx_render_gathered_root_device")

    def render_gathered_rgba8_device(self, params_list, root, device_ptr):
        """the gathered frames as the canvas' RGBA8 (a quarter of the bytes exchanged): uint8[n][H][W][4] at device_ptr; root < 0: all-gather"""
        n = len(params_list)
        arr = (FrameParams * n)(*params_list)
        self._check(LIB.flx_render_gathered_rgba8_device(self._h, arr, n, int(root), C.c_void_p(device_ptr or 0)), "flx_render_gathered_rgba8_device")

    def comm_count(self):
        """ranks of this context's RCCL communicator (ncclCommCount); 0 without one"""
        return int(LIB.flx_comm_count(self._h))

    def frame_begin_gathered(self, params, root=-1):
        """flx_frame_begin over the communicator: the gathered whole frame stays in device memory (frame_end -> its pointer)"""
        self._pending = getattr(self, "_pending", [])
        self._check(LIB.flx_frame_begin_gathered(self._h, C.byref(params), 2, int(root)), "flx_frame_begin_gathered")
        self._pending.append((params.height, params.width, False, True))

    # -- the frame loop: two frames in flight, pixels out of pinned host memory (flx_frame_begin / flx_frame_end) --------
    def set_frame_lanes(self, lanes):
        """2 (default): the frames in flight overlap on the GPU (two streams, two workspaces); 1: one after the other"""
        self._check(LIB.flx_set_frame_lanes(self._h, int(lanes)), "flx_set_frame_lanes")

    def set_frame_chain(self, mode):
        """0 every frame its own launches; 1 a chain of launches (flx_chain.hip); 2 (default) the frame server (flx_server.hip) for thin frames; 3 the server for every frame it takes"""
        self._check(LIB.flx_set_frame_chain(self._h, int(mode)), "flx_set_frame_chain")

    def set_server_moving_scenes(self, on):
        """1 (default): once the lights / transforms have changed, the frame server takes them with every frame; 0: every changed upload ends its launch"""
        self._check(LIB.flx_set_server_moving_scenes(self._h, int(bool(on))), "flx_set_server_moving_scenes")

    def server_moving(self):
        """a launch of the frame server that takes lights and transforms per frame is running"""
        return bool(LIB.flx_server_moving(self._h))

    def frame_server_takes(self, params):
        return bool(LIB.flx_frame_server_takes(self._h, C.byref(params)))

    def frame_target_set(self, images):
        """images: device-visible addresses of float4[H][W] images (2 or 3; [] clears): the frame server resolves this context's strips straight into them"""
        arr = (C.c_void_p * max(1, len(images)))(*[C.c_void_p(int(p)) for p in images])
        self._check(LIB.flx_frame_target_set(self._h, arr if images else None, len(images)), "flx_frame_target_set")

    def frame_target_index(self):
        return int(LIB.flx_frame_target_index(self._h))

    # -- one process per GPU, no collective: the ranks' servers complete one image in the root's memory (flx_share_*) ----
    def share_create(self, width, height, n_images, n_ranks, rank):
        """root: -> the FLX_SHARE_HANDLE_BYTES handle the other ranks join with"""
        buf = C.create_string_buffer(SHARE_HANDLE_BYTES)
        self._check(LIB.flx_share_create(self._h, width, height, n_images, n_ranks, rank, buf), "flx_share_create")
        return buf.raw

    def share_join(self, handle, rank):
        self._check(LIB.flx_share_join(self._h, bytes(handle), rank), "flx_share_join")

    def share_leave(self):
        self._check(LIB.flx_share_leave(self._h), "flx_share_leave")

    def frame_begin_shared(self, params):
        self._check(LIB.flx_frame_begin_shared(self._h, C.byref(params)), "flx_frame_begin_shared")

    def frame_end_shared(self):
        """-> (address of the whole image in the root's device memory (None on the other ranks), ms)"""
        ptr, n, ms = C.c_void_p(), C.c_size_t(), C.c_float()
        self._check(LIB.flx_frame_end_shared(self._h, C.byref(ptr), C.byref(n), C.byref(ms)), "flx_frame_end_shared")
        return ptr.value, ms.value

    def set_angle_table(self, on):
        """the shading's per-triangle table on (default) or off (every shade computes the values itself: the same floats)"""
        self._check(LIB.flx_debug_set_angle_table(self._h, int(bool(on))), "flx_debug_set_angle_table")

    def set_server_groups(self, groups):
        """rehearsal: the frame server's launch takes only `groups` CUs (0: all)"""
        self._check(LIB.flx_debug_set_server_groups(self._h, int(groups)), "flx_debug_set_server_groups")

    def inject_fault(self, watchdog_polls=0, flags=0):
        """tests: the next frames' frame kernels give up after `watchdog_polls` polls; flags 1 = their shade waves drop every batch"""
        self._check(LIB.flx_debug_inject_fault(self._h, int(watchdog_polls), int(flags)), "flx_debug_inject_fault")

    def server_dump(self):
        out = np.zeros((4, 72), np.uint64)
        self._check(LIB.flx_get_server_dump(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))), "flx_get_server_dump")
        return out

    def server_stats(self):
        """-> dict of the frame server's last launch (flx_server.h: SVS_*)"""
        out = np.zeros(16, np.uint64)
        self._check(LIB.flx_get_server_stats(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))), "flx_get_server_stats")
        names = ["start", "end", "frames", "tiles", "batches", "batch_lanes", "rotations", "walk_lane_trips", "walk_trips", "shade_tile_t", "shade_batch_t", "shade_total_t", "post_wait_t"]
        d = {n: int(out[i]) for i, n in enumerate(names)}
        d["host_done"] = (int(out[13]) >> 32, int(out[13]) & 0xffffffff); d["host_posted"] = (int(out[14]) >> 32, int(out[14]) & 0xffffffff); d["next_seq_stop"] = (int(out[15]) >> 32, int(out[15]) & 0xffffffff)
        return d

    def set_sample_parallel(self, on):
        """k_trace_samples (a pixel's samples side by side) instead of k_trace_pixels where the frame allows it (flx_debug_set_sample_parallel)"""
        self._check(LIB.flx_debug_set_sample_parallel(self._h, int(bool(on))), "flx_debug_set_sample_parallel")

    def set_adaptive_order(self, on):
        """the frame kernel's draw order made from the last frame's per-tile cost (flx_debug_set_adaptive_order): on by default"""
        self._check(LIB.flx_debug_set_adaptive_order(self._h, int(bool(on))), "flx_debug_set_adaptive_order")

    def tile_order_of(self, cost, mode=1):
        """the draw order k_tile_order makes of per-tile costs (flx_debug_tile_order_of)"""
        c = np.ascontiguousarray(cost, np.float32)
        out = np.zeros(c.size, np.uint32)
        self._check(LIB.flx_debug_tile_order_of(self._h, c.ctypes.data_as(C.POINTER(C.c_float)), int(c.size), int(mode), out.ctypes.data_as(C.POINTER(C.c_uint32))), "flx_debug_tile_order_of")
        return out

    def set_tile_order(self, order):
        """the frame kernel's draw order over a frame's 8 x 8 screen tiles (flx_debug_set_tile_order): a permutation, or None / empty for the default"""
        o = np.ascontiguousarray(order if order is not None else [], np.uint32)
        self._check(LIB.flx_debug_set_tile_order(self._h, o.ctypes.data_as(C.POINTER(C.c_uint32)), int(o.size)), "flx_debug_set_tile_order")

    def tile_cost(self, n, read=False):
        """counted frames' visits per screen tile (flx_debug_tile_cost): turn on for n tiles (0: off); read=True returns the sums gathered so far first"""
        have = getattr(self, "_tile_cost_n", 0)               # (the library copies min(n or all, what it holds): room for all of it)
        out = np.zeros(max(int(n), have, 1), np.uint64) if read else None
        self._check(LIB.flx_debug_tile_cost(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)) if read else None, int(n)), "flx_debug_tile_cost")
        self._tile_cost_n = int(n)
        return out

    def set_walk_jobs(self, jobs):
        """walk jobs per lane of the frame kernel's walk waves (flx_debug_set_walk_jobs): 1, 2, or 0 = the library's default"""
        self._check(LIB.flx_debug_set_walk_jobs(self._h, int(jobs)), "flx_debug_set_walk_jobs")

    def set_chain_stats(self, on):
        self._check(LIB.flx_set_chain_stats(self._h, int(bool(on))), "flx_set_chain_stats")

    def chain_stats(self):
        """-> uint64 [64 launches, 64 words] (flx_chain.h: CS_*)"""
        out = np.zeros((64, 64), np.uint64)
        self._check(LIB.flx_get_chain_stats(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))), "flx_get_chain_stats")
        return out

    def set_chain_order(self, order):
        """explicit order of a chained frame's screen tiles (a permutation; None: the default)"""
        if order is None:
            self._check(LIB.flx_set_chain_order(self._h, None, 0), "flx_set_chain_order")
            return
        o = np.ascontiguousarray(order, np.uint32)
        self._check(LIB.flx_set_chain_order(self._h, o.ctypes.data_as(C.POINTER(C.c_uint32)), o.size), "flx_set_chain_order")

    def set_chain_cost(self, n):
        self._check(LIB.flx_set_chain_cost(self._h, int(n)), "flx_set_chain_cost")
        self._chain_cost_n = int(n)

    def chain_cost(self):
        out = np.zeros((2, self._chain_cost_n), np.uint32)
        self._check(LIB.flx_get_chain_cost(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32))), "flx_get_chain_cost")
        return out

    def last_chained(self):
        """0 the last frame begun in the loop was not chained, 1 it began a chain, 2 it continued one"""
        v = C.c_int()
        self._check(LIB.flx_last_chained(self._h, C.byref(v)), "flx_last_chained")
        return v.value

    def frame_begin(self, params, rgba8=False, device=False):
        self._pending = getattr(self, "_pending", [])
        self._check(LIB.flx_frame_begin(self._h, C.byref(params), 2 if device else (1 if rgba8 else 0)), "flx_frame_begin")
        self._pending.append((self.tile_row_count(params), params.width, rgba8, device))

    def frame_end(self):
        """-> (pixels [rows, W, 4] float32 or uint8: a COPY of the pinned buffer — or, for a frame begun with device=True, the
        device pointer of float4[rows][W] —, GPU ms of the frame)"""
        ptr, n, ms = C.c_void_p(), C.c_size_t(), C.c_float()
        rc = LIB.flx_frame_end(self._h, C.byref(ptr), C.byref(n), C.byref(ms))
        if rc != 1 or self._pending:           # (FLX_ERR_INVALID with nothing in flight took no frame)
            rows, width, rgba8, device = self._pending.pop(0) if self._pending else (0, 0, False, False)
        self._check(rc, "flx_frame_end")
        if device:
            return ptr.value, ms.value
        dt = np.uint8 if rgba8 else np.float32
        count = n.value // np.dtype(dt).itemsize
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8 if rgba8 else C.c_float)), shape=(count,)).copy() if count else np.zeros(0, dt)
        return a.reshape(rows, width, 4), ms.value

    def frames_in_flight(self):
        return int(LIB.flx_frames_in_flight(self._h))

    def render_device(self, params, device_ptr):
        self._check(LIB.flx_render_device(self._h, C.byref(params), C.c_void_p(device_ptr)), "flx_render_device")

    def sync(self):
        self._check(LIB.flx_sync(self._h), "flx_sync")

    def set_stream(self, stream_ptr):
        self._check(LIB.flx_set_stream(self._h, C.c_void_p(stream_ptr)), "flx_set_stream")

    def set_counters_enabled(self, on):
        self._check(LIB.flx_set_counters_enabled(self._h, int(bool(on))), "flx_set_counters_enabled")

    def set_pipeline(self, pipeline):
        """0 auto, 1 per-pixel kernel, 2 persistent path kernel (same results)."""
        self._check(LIB.flx_set_pipeline(self._h, int(pipeline)), "flx_set_pipeline")

    def set_lockstep(self, on):
        """Small scenes: wave-wide lockstep walk (default) or the lane walk (same results)."""
        self._check(LIB.flx_set_lockstep(self._h, int(bool(on))), "flx_set_lockstep")

    def get_counters(self):
        cnt = Counters()
        self._check(LIB.flx_get_counters(self._h, C.byref(cnt)), "flx_get_counters")
        return cnt.as_dict()

    def temporal_reset(self):
        self._check(LIB.flx_temporal_reset(self._h), "flx_temporal_reset")

    def debug_intersect(self, fn, rows):
        """flx_debug_intersect: rows [n, 16] (triangles: fn 0, 1, 3, 4) or [n, 13] (boxes: fn 2, 5) float32 -> [n, 3] (fn 0, 3) or [n] float32"""
        rows = np.ascontiguousarray(rows, np.float32)
        n = rows.shape[0]
        out = np.zeros((n, 3) if fn in (0, 3) else (n,), np.float32)
        self._check(LIB.flx_debug_intersect(self._h, int(fn), _fp(rows), _fp(out), n), "flx_debug_intersect")
        return out

    def debug_walk(self, variant, rays):
        """flx_debug_walk: rays [n, 7] float32 (origin, direction, l) -> [n, 8] float32 (s, u, v, 2 x transform, entry, entries fetched, shadowed, entries fetched)"""
        rays = np.ascontiguousarray(rays, np.float32)
        out = np.zeros((rays.shape[0], 8), np.float32)
        self._check(LIB.flx_debug_walk(self._h, int(variant), _fp(rays), _fp(out), rays.shape[0]), "flx_debug_walk")
        return out

    def last_pipeline(self):
        v = C.c_int()
        self._check(LIB.flx_last_pipeline(self._h, C.byref(v)), "flx_last_pipeline")
        return v.value

    def set_wavefront_groups(self, groups):
        self._check(LIB.flx_set_wavefront_groups(self._h, int(groups)), "flx_set_wavefront_groups")

    def render_planes_device(self, params, device_ptr):
        """this rank's strips of a filter frame -> uint32[5][rows][width] RGBA8 render targets in device memory"""
        self._check(LIB.flx_render_planes_device(self._h, C.byref(params), C.c_void_p(device_ptr)), "flx_render_planes_device")

    def filter_planes_device(self, params, planes_ptr, out_ptr):
        """uint32[5][height][width] render targets of the whole frame -> float4[height][width] through the denoise chain"""
        self._check(LIB.flx_filter_planes_device(self._h, C.byref(params), C.c_void_p(planes_ptr), C.c_void_p(out_ptr)), "flx_filter_planes_device")

    def fxaa(self, frame):
        """[H, W, 4] float32 frame -> the FXAA pass of the reference over it"""
        a = np.ascontiguousarray(frame, np.float32)
        out = np.empty_like(a)
        self._check(LIB.flx_fxaa(self._h, a.shape[1], a.shape[0], a.ctypes.data, out.ctypes.data), "flx_fxaa")
        return out

    def present(self, frame):
        """[H, W, 4] float32 frame -> the uint8 RGBA of the canvas' drawing buffer"""
        a = np.ascontiguousarray(frame, np.float32)
        out = np.empty(a.shape, np.uint8)
        self._check(LIB.flx_present(self._h, a.shape[1], a.shape[0], a.ctypes.data, out.ctypes.data), "flx_present")
        return out

    def present_device(self, width, height, d_in_rgba, d_out_rgba8):
        """device pointers: float4[H][W] -> the canvas' uint8[H][W][4], enqueued on the context's stream"""
        self._check(LIB.flx_present_device(self._h, width, height, C.c_void_p(d_in_rgba), C.c_void_p(d_out_rgba8)), "flx_present_device")

    def taa(self, frame):
        """the TAA pass: the context keeps the last nine frames"""
        a = np.ascontiguousarray(frame, np.float32)
        out = np.empty_like(a)
        self._check(LIB.flx_taa(self._h, a.shape[1], a.shape[0], a.ctypes.data, out.ctypes.data), "flx_taa")
        return out

    def taa_reset(self):
        self._check(LIB.flx_taa_reset(self._h), "flx_taa_reset")

    def set_wavefront_organisation(self, organisation):
        """0 automatic, 1 rounds (a shade + walk kernel pair per bounce), 2 the frame kernel (all bounces in one persistent launch)"""
        self._check(LIB.flx_set_wavefront_organisation(self._h, int(organisation)), "flx_set_wavefront_organisation")

    def last_organisation(self):
        """what the wavefront pipeline ran for the last frame: 1 rounds, 2 frame kernel, 3 frame kernel with the front inside; 0 another pipeline"""
        v = C.c_int()
        self._check(LIB.flx_last_organisation(self._h, C.byref(v)), "flx_last_organisation")
        return v.value

    def set_frame_front(self, mode):
        """primary rays and bounce-0 shading: 0 k_primary + k_wf_shade0 in front, 3 one kernel in front, 2 inside the frame kernel wherever it runs, 1 (default) automatic"""
        self._check(LIB.flx_set_frame_front(self._h, int(mode)), "flx_set_frame_front")

    def set_walk_scheduler(self, scheduler, suspend_walks=0):
        """0 one walk per lane (default), 1 LDS test queues, 2 lanes + cooperative finisher; identical results"""
        self._check(LIB.flx_set_walk_scheduler(self._h, int(scheduler), int(suspend_walks)), "flx_set_walk_scheduler")

    def get_diag(self):
        out = (C.c_uint64 * 32)()
        self._check(LIB.flx_get_diag(self._h, out), "flx_get_diag")
        return [int(x) for x in out]

    def get_tail_diag(self):
        out = (C.c_uint64 * 40)()
        self._check(LIB.flx_get_tail_diag(self._h, out), "flx_get_tail_diag")
        return [int(x) for x in out]

    def last_frame_ms(self):
        a, b = C.c_float(), C.c_float()
        self._check(LIB.flx_last_frame_ms(self._h, C.byref(a), C.byref(b)), "flx_last_frame_ms")
        return a.value, b.value

    def debug_math(self, fn, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        out = np.empty_like(a)
        bb = np.ascontiguousarray(b, np.float32) if b is not None else None
        self._check(LIB.flx_debug_math(self._h, fn, _fp(a), _fp(bb) if bb is not None else None, _fp(out), a.size), "flx_debug_math")
        return out

    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_uint32()
        self._check(LIB.flx_device_info(self._h, name, 256, C.byref(cus)), "flx_device_info")
        return name.value.decode(), int(cus.value)


def version():
    return LIB.flx_version().decode()


def comm_unique_id():
    """ncclGetUniqueId through the library: FLX_COMM_ID_BYTES bytes rank 0 hands to the other ranks"""
    buf = C.create_string_buffer(128)
    rc = LIB.flx_comm_unique_id(buf)
    if rc != 0:
        raise FlexLightHipError("flx_comm_unique_id failed (%d): %s" % (rc, LIB.flx_last_error(None).decode()))
    return buf.raw


class _Borrowed(Context):
    """a context owned by a Group"""

    def __init__(self, handle):
        self._h = handle

    def close(self):
        self._h = None


class Group:
    """n contexts in one process, one frame split over them in row strips (flx_group_*): what
    `new FlexLight(canvas, {devices: n})` of the JavaScript host sits on."""

    def __init__(self, devices):
        devices = list(devices)
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = LIB.flx_group_create(len(devices), arr, C.byref(h))
        if rc != 0:
            raise FlexLightHipError("flx_group_create(%s) failed (%d): %s" % (devices, rc, LIB.flx_group_last_error(None).decode()))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            LIB.flx_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise FlexLightHipError("%s failed (%d): %s" % (what, rc, LIB.flx_group_last_error(self._h).decode()))

    @property
    def size(self):
        return int(LIB.flx_group_size(self._h))

    @property
    def uses_rccl(self):
        return bool(LIB.flx_group_uses_rccl(self._h))

    def set_gather(self, to_root):
        """True (default): only context 0 (which hands the frame out) receives the strips; False: all-gather"""
        self._check(LIB.flx_group_set_gather(self._h, 1 if to_root else 0), "flx_group_set_gather")

    def context(self, rank):
        h = LIB.flx_group_context(self._h, rank)
        if not h:
            raise FlexLightHipError("flx_group_context: no rank %d" % rank)
        return _Borrowed(C.c_void_p(h))

    def update_scene(self, scene):
        view = scene.view()
        self._check(LIB.flx_group_scene_upload_view(self._h, C.byref(view)), "flx_group_scene_upload_view")

    def update_primary_light_sources(self, lights):
        lights = np.ascontiguousarray(lights, np.float32).reshape(-1)
        self._check(LIB.flx_group_lights_upload(self._h, _fp(lights), lights.size // 6), "flx_group_lights_upload")

    def update_transforms(self, rotation, shift):
        rotation = np.ascontiguousarray(rotation, np.float32).reshape(-1)
        shift = np.ascontiguousarray(shift, np.float32).reshape(-1)
        self._check(LIB.flx_group_transforms_upload(self._h, _fp(rotation), _fp(shift), shift.size // 8), "flx_group_transforms_upload")

    def render(self, params_list, tile_rows=8, counters=False):
        """frames (a list of FrameParams, or one) -> (rgba [n, H, W, 4] float32, counters summed over the contexts or None)"""
        if isinstance(params_list, FrameParams):
            params_list = [params_list]
        n = len(params_list)
        arr = (FrameParams * n)(*params_list)
        h, w = (params_list[0].height, params_list[0].width) if n else (0, 0)
        out = np.zeros((n, h, w, 4), np.float32) if n else np.zeros(4, np.float32)
        cnt = Counters() if counters else None
        self._check(LIB.flx_group_render(self._h, arr, n, tile_rows, _fp(out), C.byref(cnt) if cnt else None), "flx_group_render")
        return out, (cnt.as_dict() if cnt else None)

    def render_rgba8(self, params_list, tile_rows=8):
        """the frames as the canvas' RGBA8, quantised before the exchange -> uint8 [n, H, W, 4]"""
        if isinstance(params_list, FrameParams):
            params_list = [params_list]
        n = len(params_list)
        arr = (FrameParams * n)(*params_list)
        out = np.zeros((n, params_list[0].height, params_list[0].width, 4), np.uint8)
        self._check(LIB.flx_group_render_rgba8(self._h, arr, n, tile_rows, out.ctypes.data_as(C.POINTER(C.c_uint8)), None), "flx_group_render_rgba8")
        return out

    # -- the group's frame loop (flx_group_frame_begin / _end): every context's frame server resolves its strips into ONE image ----
    def set_frame_lanes(self, lanes):
        self._check(LIB.flx_group_set_frame_lanes(self._h, int(lanes)), "flx_group_set_frame_lanes")

    def frames_in_flight(self):
        return int(LIB.flx_group_frames_in_flight(self._h))

    def frame_begin(self, params, tile_rows=8, device=False, rgba8=False):
        self._pending = getattr(self, "_pending", [])
        self._check(LIB.flx_group_frame_begin(self._h, C.byref(params), tile_rows, 2 if device else (1 if rgba8 else 0)), "flx_group_frame_begin")
        self._pending.append((params.height, params.width, device, rgba8))

    def frame_end(self):
        """-> (pixels [H, W, 4] float32: a COPY of the pinned image — or, for a frame begun with device=True, its address in context 0's memory —, ms)"""
        ptr, n, ms = C.c_void_p(), C.c_size_t(), C.c_float()
        rc = LIB.flx_group_frame_end(self._h, C.byref(ptr), C.byref(n), C.byref(ms))
        if rc != 1 or self._pending:
            h, w, device, rgba8 = self._pending.pop(0)
        self._check(rc, "flx_group_frame_end")
        if device:
            return ptr.value, ms.value
        if rgba8:                                   # the canvas' bytes: uint8 [H, W, 4]
            return np.frombuffer((C.c_uint8 * (h * w * 4)).from_address(ptr.value), np.uint8).reshape(h, w, 4).copy(), ms.value
        buf = (C.c_float * (h * w * 4)).from_address(ptr.value)
        return np.frombuffer(buf, np.float32).reshape(h, w, 4).copy(), ms.value


class Mesh:
    """One imported OBJ (+ MTL) in native code: SURVEY 8f N2, include/flexlight_hip.h flx_mesh_*.  Needs no GPU."""
    FIELDS = {"color": 0, "roughness": 1, "metallicity": 2, "emissiveness": 3, "translucency": 4, "ior": 5, "texture_nums": 6}

    def __init__(self, obj_text, mtl_text=None):
        obj = obj_text.encode() if isinstance(obj_text, str) else obj_text
        mtl = (mtl_text.encode() if isinstance(mtl_text, str) else mtl_text) if mtl_text is not None else None
        h = C.c_void_p()
        rc = LIB.flx_mesh_import_obj(obj, len(obj), mtl, len(mtl) if mtl else 0, C.byref(h))
        if rc != 0:
            raise FlexLightHipError("flx_mesh_import_obj failed (%d)" % rc)
        self._h = h

    def close(self):
        if self._h:
            LIB.flx_mesh_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def entries(self):
        return int(LIB.flx_mesh_entry_count(self._h))

    @property
    def triangles(self):
        return int(LIB.flx_mesh_triangle_count(self._h))

    def set_transform(self, number):
        LIB.flx_mesh_set_transform(self._h, int(number))

    def move(self, x, y, z):
        LIB.flx_mesh_move(self._h, float(x), float(y), float(z))

    def scale(self, s):
        LIB.flx_mesh_scale(self._h, float(s))

    def set_material(self, field, values):
        vals = (C.c_double * 3)(*([float(values)] * 3 if np.isscalar(values) else [float(v) for v in values]))
        if LIB.flx_mesh_set_material(self._h, self.FIELDS[field], vals) != 0:
            raise FlexLightHipError("flx_mesh_set_material: unknown field")

    def bounding(self):
        box = (C.c_double * 6)()
        LIB.flx_mesh_bounding(self._h, box)
        return [float(v) for v in box]

    def flatten(self):
        """-> geometry [entries, 12] f32, attributes [entries, 28] f32, ids [triangles] i32, minmax [6] f32"""
        g = np.zeros((self.entries, 12), np.float32)
        a = np.zeros((self.entries, 28), np.float32)
        ids = np.zeros(self.triangles, np.int32)
        box = (C.c_float * 6)()
        if LIB.flx_mesh_flatten(self._h, g.ctypes.data, a.ctypes.data, ids.ctypes.data, box) != 0:
            raise FlexLightHipError("flx_mesh_flatten failed")
        return g, a, ids, np.array(list(box), np.float32)


def transforms_pack(matrices, positions):
    """[T, 3, 3] scale x rotation matrices, [T, 3] positions (float64) -> rotation [T, 24] f32, shift [T, 8] f32"""
    m = np.ascontiguousarray(matrices, np.float64).reshape(-1, 9)
    p = np.ascontiguousarray(positions, np.float64).reshape(-1, 3)
    rot = np.zeros((m.shape[0], 24), np.float32)
    sh = np.zeros((m.shape[0], 8), np.float32)
    if LIB.flx_transforms_pack(m.shape[0], m.ctypes.data, p.ctypes.data, rot.ctypes.data, sh.ctypes.data) != 0:
        raise FlexLightHipError("flx_transforms_pack failed")
    return rot, sh
