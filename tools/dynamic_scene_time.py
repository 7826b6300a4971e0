#!/usr/bin/env python3
"""A rank's share of the 1080p dragon frame in the frame loop while the SCENE moves: the transforms are uploaded changed before every frame (examples/dragon.js turns its monkey every
tick).  The frame server's launch reads one scene, so every upload ends it (its frames in flight complete first) and the next frame starts another; the two-lane loop keeps a copy
of the arrays per lane.  ms per frame, wall clock.  GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
p = sc.frame_params(use_filter=0)
p.tile_rows, p.tile_count, p.tile_index = 8, 8, 3
rot0 = np.array(sc.arrays["rotation"], np.float32).reshape(-1, 2, 12)
def rot(f):
    r = rot0.copy()
    a = 0.01 * f
    c, s = np.cos(a), np.sin(a)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]) * 2.0
    Ri = np.linalg.inv(R)
    for m, M in ((0, R), (1, Ri)):
        for col in range(3): r[2, m, 4 * col:4 * col + 3] = M[:, col]
    return r.reshape(-1)
N = 100
ctx.set_server_moving_scenes(0)
for label, lanes, chain, moving in (("frame server, 3 in flight, static scene", 3, 3, False), ("frame server, 3 in flight, transforms change every frame: every upload ends the launch (flx_set_server_moving_scenes(0))", 3, 3, True),
                                    ("two lanes (own launches), static scene", 2, 0, False), ("two lanes (own launches), transforms change every frame", 2, 0, True),
                                    ("one lane, transforms change every frame", 1, 0, True),
                                    ("the default (flx_set_frame_chain(2)), 3 in flight, static scene", 3, 2, False),
                                    ("flx_set_server_moving_scenes(0), the default mode, transforms change every frame: frames go to the lanes", 3, 2, True),
                                    ("MOVING SCENES IN THE SERVER (the default): 3 in flight, transforms change every frame", 3, 2, "server"),
                                    ("   ... 2 in flight", 2, 2, "server"),
                                    ("   ... the same launch, the scene standing still again", 3, 2, "still")):
    if moving in ("server", "still"): ctx.set_server_moving_scenes(1)
    ctx.set_frame_lanes(lanes); ctx.set_frame_chain(chain)
    best = 1e9
    for rep in range(3):
        t0 = None
        for f in range(N + 6):
            if f == 6: t0 = time.perf_counter()
            if ctx.frames_in_flight() == max(lanes, 1) or (lanes == 1 and ctx.frames_in_flight() == 1):
                ctx.frame_end()
            if moving and moving != "still": ctx.update_transforms(rot(f), sc.arrays["shift"])
            ctx.frame_begin(p, device=True)
        while ctx.frames_in_flight(): ctx.frame_end()
        best = min(best, (time.perf_counter() - t0) * 1e3 / N)
    print("%-64s %.3f ms per frame%s" % (label, best, "   (launch takes the arrays per frame)" if ctx.server_moving() else ""), flush=True)
ctx.update_transforms(sc.arrays["rotation"], sc.arrays["shift"])
if "--all-ranks" in sys.argv:
    # every rank's eighth, the monkey turning before every frame: the server (3 and 2 in flight) against the two lanes
    print("\nall eight ranks of configs[2], transforms change before every frame, ms per frame (wall clock over %d frames, best of 3):" % N)
    print("rank   server 3 in flight   server 2 in flight   two lanes")
    rows = []
    for i in range(8):
        p.tile_index = i
        row = []
        for lanes, chain, on in ((3, 2, 1), (2, 2, 1), (2, 0, 1)):
            ctx.set_server_moving_scenes(on); ctx.set_frame_lanes(lanes); ctx.set_frame_chain(chain)
            best = 1e9
            for rep in range(3):
                for f in range(N + 6):
                    if f == 6: t0 = time.perf_counter()
                    if ctx.frames_in_flight() == lanes: ctx.frame_end()
                    ctx.update_transforms(rot(f), sc.arrays["shift"])
                    ctx.frame_begin(p, device=True)
                while ctx.frames_in_flight(): ctx.frame_end()
                best = min(best, (time.perf_counter() - t0) * 1e3 / N)
            row.append(best)
        rows.append(row)
        print("%4d   %18.3f   %18.3f   %9.3f" % (i, *row), flush=True)
    a = np.array(rows)
    print(" max   %18.3f   %18.3f   %9.3f     (the slowest rank paces the frame)" % tuple(a.max(axis=0)))
    ctx.update_transforms(sc.arrays["rotation"], sc.arrays["shift"])
