#!/usr/bin/env python3
"""Frame time of ONE rank's share of a tile-sharded frame (what each of N GPUs does), on one GPU.  usage: tile_time.py N"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
if os.environ.get("FLX_SCHED"):                     # "scheduler,suspend_walks"
    a, b = os.environ["FLX_SCHED"].split(",")
    ctx.set_walk_scheduler(int(a), int(b))
if os.environ.get("FLX_GROUPS"):
    ctx.set_wavefront_groups(int(os.environ["FLX_GROUPS"]))
if os.environ.get("FLX_PIPELINE"):
    ctx.set_pipeline(int(os.environ["FLX_PIPELINE"]))
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    p = sc.frame_params(use_filter=0)
    p.tile_rows, p.tile_count, p.tile_index = 8, n, 0
    for _ in range(3): ctx.render(p)
    ms = []
    for _ in range(10):
        ctx.render(p)
        ms.append(ctx.last_frame_ms()[0])
    print("1/%d of the frame: %.3f ms per frame (min %.3f)" % (n, sum(ms) / len(ms), min(ms)))
