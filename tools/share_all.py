#!/usr/bin/env python3
"""Every rank's share of a tile-sharded BASELINE frame on ONE GPU: what each of N GPUs renders before the gather (strips of 8 rows dealt round robin),
one frame at a time (HIP events around the share: median of 20, min) and in the frame loop with one frame in flight on one and on two lanes (wall
clock over 60 frames).  GPU box.
usage: share_all.py [--workload dragon|dragon_4k|theater] [--count 8] [--indices 0,1,..] [--front MODE] [--check]
--check: every share's frame is compared with the same rows of the whole frame (bit for bit)."""
import argparse, os, sys, time
import numpy as np
if "--target" in sys.argv:
    import torch                           # (before the library: both must end up on ONE HIP runtime — the first libamdhip64 loaded)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="dragon")
ap.add_argument("--count", type=int, default=8)
ap.add_argument("--indices", default="")
ap.add_argument("--front", type=int, default=-1)
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--check", action="store_true")
ap.add_argument("--no-single", action="store_true")
ap.add_argument("--target", action="store_true", help="also: the share resolved straight into whole images in pinned host memory / device memory (flx_frame_target_set)")
a = ap.parse_args()
work = a.workload
sc = Scene.golden("theater" if work == "theater" else "dragon")
size = dict(width=3840, height=2160) if work == "dragon_4k" else {}
over = dict(samples=16, max_reflections=6) if work == "theater" else {}
ctx = capi.Context(0)
ctx.update_scene(sc)
if a.front >= 0:
    ctx.set_frame_front(a.front)
idx = [int(x) for x in a.indices.split(",")] if a.indices else list(range(a.count))

def params(i, n):
    p = sc.frame_params(use_filter=0, **size, **over)
    if n > 1:
        p.tile_rows, p.tile_count, p.tile_index = 8, n, i
    return p

def single(p):
    for _ in range(3):
        ctx.render(p)
    ms = []
    for _ in range(20):
        ctx.render(p)
        ms.append(ctx.last_frame_ms()[0])
    return float(np.median(ms)), min(ms)

def loop(p, lanes, server=False):
    """ms per frame of the frame loop with `lanes` frames in flight (wall clock over a.frames frames, best of 3); server: through the frame server where it takes the frame"""
    ctx.set_frame_lanes(lanes)
    ctx.set_frame_chain(3 if server else 0)
    best = 1e9
    for rep in range(3):
        for _ in range(max(lanes, 2) - 1):
            ctx.frame_begin(p, device=True)
        for _ in range(4):
            ctx.frame_begin(p, device=True)
            ctx.frame_end()
        t0 = time.perf_counter()
        for _ in range(a.frames):
            ctx.frame_begin(p, device=True)
            ctx.frame_end()
        dt = time.perf_counter() - t0
        while ctx.frames_in_flight():
            ctx.frame_end()
        best = min(best, dt * 1e3 / a.frames)
    served = ctx.last_chained() == 3
    ctx.set_frame_lanes(1)
    ctx.set_frame_chain(2)
    return best if (served or not server) else float("nan")

_images = {}
def loop_target(p, where):
    """the same loop through the frame server with three frames in flight, the share resolved straight into WHOLE images of the caller's (flx_frame_target_set):
    pinned host memory (what flx_group_frame_begin's FLX_FRAME_FLOAT does on every GPU of a group) or device memory"""
    import torch
    if where not in _images:
        _images[where] = torch.zeros((3, p.height, p.width, 4), dtype=torch.float32, device="cuda") if where == "device" else torch.zeros((3, p.height, p.width, 4), dtype=torch.float32).pin_memory()
    img = _images[where]
    ctx.set_frame_lanes(3)
    ctx.set_frame_chain(3)
    ctx.frame_target_set([img[i].data_ptr() for i in range(3)])
    best = 1e9
    try:
        for rep in range(3):
            for _ in range(2):
                ctx.frame_begin(p, device=True)
            for _ in range(4):
                ctx.frame_begin(p, device=True)
                ctx.frame_end()
            t0 = time.perf_counter()
            for _ in range(a.frames):
                ctx.frame_begin(p, device=True)
                ctx.frame_end()
            dt = time.perf_counter() - t0
            while ctx.frames_in_flight():
                ctx.frame_end()
            best = min(best, dt * 1e3 / a.frames)
    finally:
        ctx.frame_target_set([])
        ctx.set_frame_lanes(1)
        ctx.set_frame_chain(2)
    return best

print("workload %s, lib %s" % (work, os.path.basename(capi.LIB_PATH)))
whole = params(0, 1)
wm, wmin = single(whole)
w1, w2 = loop(whole, 1), loop(whole, 2)
org, pipe = ctx.last_organisation(), ctx.last_pipeline()
ws2, ws3 = loop(whole, 2, True), loop(whole, 3, True)
print("columns: one frame at a time (HIP events: median of 20, min) | frame loop, wall clock per frame: 1 lane, 2 lanes, frame server with 2 and with 3 frames in flight (nan: the server does not take the frame)")
print("whole frame      single %7.3f (%7.3f)   loop 1 lane %7.3f   2 lanes %7.3f   server 2: %7.3f  3: %7.3f   organisation %d pipeline %d" % (wm, wmin, w1, w2, ws2, ws3, org, pipe))
full = ctx.render(whole)[0] if a.check else None
rows = []
targets = []
for i in idx:
    p = params(i, a.count)
    m, mn = (0.0, 0.0) if a.no_single else single(p)
    org = ctx.last_organisation()
    l1, l2 = loop(p, 1), loop(p, 2)
    s2, s3 = loop(p, 2, True), loop(p, 3, True)
    ok = ""
    if a.check:
        got = ctx.render(p)[0]
        H = whole.height
        sel = [y for y in range(H) if (y // 8) % a.count == i]
        want = full[sel]
        ok = "  equal" if got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)) else "  DIFFERS"
    rows.append((m, l1, l2, s2, s3))
    tgt = ""
    if a.target:
        th, td = loop_target(p, "host"), loop_target(p, "device")
        targets.append((th, td))
        tgt = "   into a whole image, 3 in flight: pinned host %7.3f  device %7.3f" % (th, td)
        if a.check:
            sel = [y for y in range(whole.height) if (y // 8) % a.count == i]
            ok += "  host image " + ("equal" if np.array_equal(_images["host"][0].numpy()[sel].view(np.uint32), full[sel].view(np.uint32)) else "DIFFERS")
    print("share %d/%d        single %7.3f (%7.3f)   loop 1 lane %7.3f   2 lanes %7.3f   server 2: %7.3f  3: %7.3f   organisation %d%s%s" % (i, a.count, m, mn, l1, l2, s2, s3, org, tgt, ok), flush=True)
r = np.array(rows)
mx = np.nanmax(r, axis=0) if not np.isnan(r[:, 3]).all() else np.concatenate([r[:, :3].max(axis=0), [float("nan")] * 2])
print("max over ranks   single %7.3f             loop 1 lane %7.3f   2 lanes %7.3f   server 2: %7.3f  3: %7.3f" % tuple(mx))
best_whole = min(w1, w2)
print("speed-up of the slowest share over the whole frame, frame after frame (%.3f ms):  single %.2fx   1 lane %.2fx   2 lanes %.2fx   server, 2 in flight %.2fx   server, 3 in flight %.2fx" %
      (w1, w1 / mx[0] if mx[0] else 0, w1 / mx[1], w1 / mx[2], w1 / mx[3], w1 / mx[4]))
print("... and over the whole frame at its best on one GPU (%.3f ms, two lanes):  2 lanes %.2fx   server, 3 in flight %.2fx" % (best_whole, best_whole / mx[2], best_whole / mx[4]))
if targets:
    t = np.array(targets)
    print("max over ranks, share resolved into a whole image with 3 frames in flight:  pinned host memory %.3f ms (%.2fx the whole frame on two lanes)   device memory %.3f ms (%.2fx)" % (t[:, 0].max(), best_whole / t[:, 0].max(), t[:, 1].max(), best_whole / t[:, 1].max()))
