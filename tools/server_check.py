#!/usr/bin/env python3
"""The frame server (flx_set_frame_chain(2): one persistent launch renders the loop's frames as they are posted) against single renders: every frame of a
loop with a moving camera must equal its own flx_render bit for bit; then the loop's rate against the two-lane loop.  GPU box.
usage: server_check.py [--workload dragon|dragon_4k] [--count N --index I] [--frames F] [--size WxH] [--lanes 2|3]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="dragon")
ap.add_argument("--count", type=int, default=8)
ap.add_argument("--index", type=int, default=0)
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--size", default="")
ap.add_argument("--time-frames", type=int, default=200)
ap.add_argument("--lanes", type=int, default=3)
ap.add_argument("--no-check", action="store_true")
ap.add_argument("--modes", default="2,0")
ap.add_argument("--stats", action="store_true")
a = ap.parse_args()
sc = Scene.golden("dragon")
size = dict(width=3840, height=2160) if a.workload == "dragon_4k" else {}
if a.size:
    w, h = a.size.split("x"); size = dict(width=int(w), height=int(h))
ctx = capi.Context(0)
ctx.update_scene(sc)

def params(f):
    p = sc.frame_params(use_filter=0, **size)
    if a.count > 1:
        p.tile_rows, p.tile_count, p.tile_index = 8, a.count, a.index
    p.camera[0] += 0.05 * f; p.camera[2] -= 0.03 * f
    p.random_seed = float(f % 4)
    return p

def run(ps, lanes, **kw):
    got, kinds = [], []
    for p in ps:
        if ctx.frames_in_flight() == lanes:
            got.append(ctx.frame_end()[0])
        ctx.frame_begin(p, **kw); kinds.append(ctx.last_chained())
    while ctx.frames_in_flight():
        got.append(ctx.frame_end()[0])
    return got, kinds

if not a.no_check:
    ps = [params(f) for f in range(a.frames)]
    want = [ctx.render(p)[0] for p in ps]
    for lanes in sorted(set([2, a.lanes])):
        ctx.set_frame_lanes(lanes)
        ctx.set_frame_chain(3)
        got, kinds = run(ps, lanes)
        bad = [f for f in range(a.frames) if not np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32))]
        print("server, %d frames in flight: %d frames, kinds %s, frames that differ from their own render: %s" % (lanes, a.frames, kinds, bad or "none"), flush=True)
        for f in bad[:3]:
            d = (got[f].view(np.uint32) != want[f].view(np.uint32)).any(axis=2)
            print("   frame %d: %d pixels differ, rows %s" % (f, int(d.sum()), np.nonzero(d.any(axis=1))[0][:8]))

p = params(0)
for mode in [int(m) for m in a.modes.split(",")]:
    for lanes in ([2, 3] if mode else [2]):
        ctx.set_frame_lanes(lanes)
        ctx.set_frame_chain(3 if mode == 2 else mode)
        best = 1e9
        for rep in range(3):
            for _ in range(lanes - 1):
                ctx.frame_begin(p, device=True)
            for _ in range(6):
                ctx.frame_begin(p, device=True); ctx.frame_end()
            t0 = time.perf_counter()
            for _ in range(a.time_frames):
                ctx.frame_begin(p, device=True); ctx.frame_end()
            dt = time.perf_counter() - t0
            while ctx.frames_in_flight():
                ctx.frame_end()
            best = min(best, dt * 1e3 / a.time_frames)
        print("loop, mode %d (%s), %d frames in flight: %.3f ms per frame" % (mode, {0: "two lanes", 1: "chain of launches", 2: "frame server"}[mode], lanes, best), flush=True)
        if a.stats and mode == 2:
            st = ctx.server_stats()
            dur = (st["end"] - st["start"]) / 100.0
            tot = max(st["shade_total_t"], 1)
            print("   last launch: %.0f us, %d frames (%.1f us per frame), %d tiles, %d batches of %.1f lanes; shade waves: tiles %.0f%% batches %.0f%% of their time; walk waves: %.1f lanes per trip" %
                  (dur, st["frames"], dur / max(st["frames"], 1), st["tiles"], st["batches"], st["batch_lanes"] / max(st["batches"], 1), 100.0 * st["shade_tile_t"] / tot, 100.0 * st["shade_batch_t"] / tot,
                   st["walk_lane_trips"] / max(st["walk_trips"], 1)))
