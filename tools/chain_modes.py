#!/usr/bin/env python3
"""Loop rate of a rank's share, alternating between the chained loop and the two-lane loop several times in one process (is a mode's rate stable?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
p = sc.frame_params(use_filter=0)
p.tile_rows, p.tile_count, p.tile_index = 8, 8, 0
def rate(frames):
    ctx.frame_begin(p, device=True)
    for _ in range(6):
        ctx.frame_begin(p, device=True); ctx.frame_end()
    t0 = time.perf_counter()
    for _ in range(frames):
        ctx.frame_begin(p, device=True); ctx.frame_end()
    dt = time.perf_counter() - t0
    ctx.frame_end()
    return dt * 1e3 / frames
LANES = 2
def rate(frames):
    for _ in range(LANES - 1):
        ctx.frame_begin(p, device=True)
    for _ in range(6):
        ctx.frame_begin(p, device=True); ctx.frame_end()
    t0 = time.perf_counter()
    for _ in range(frames):
        ctx.frame_begin(p, device=True); ctx.frame_end()
    dt = time.perf_counter() - t0
    while ctx.frames_in_flight():
        ctx.frame_end()
    return dt * 1e3 / frames
for lanes, mode, frames in [(2, 0, 200), (2, 1, 200), (3, 1, 200), (3, 1, 400), (2, 0, 200), (3, 1, 200), (3, 0, 200)]:
    LANES = lanes
    ctx.set_frame_lanes(lanes)
    ctx.set_frame_chain(mode)
    print("%d frames in flight, chain %d, %3d frames: %.3f ms per frame" % (lanes, mode, frames, rate(frames)), flush=True)
