#!/usr/bin/env python3
"""Soak of the frame server: thousands of frames with random numbers of frames in flight, random pauses of the host, shape changes, scene uploads and synchronous
renders in between; every Nth frame compared with its own flx_render.  GPU box.  usage: server_soak.py [frames] [seed]"""
import os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sc = Scene.golden("dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
ctx.set_frame_chain(3)
shapes = [dict(width=480, height=272), dict(width=1920, height=1080, tile=(8, 2, 8)), dict(width=640, height=360, samples=4), dict(width=320, height=200, max_reflections=2)]
def frame(f, shape):
    p = sc.frame_params(use_filter=0, **shape)
    p.camera[0] += 0.01 * (f % 97); p.camera[2] -= 0.007 * (f % 89); p.random_seed = float(f % 4)
    return p
lanes, shape = 3, shapes[0]
ctx.set_frame_lanes(lanes)
inflight, checked, bad, served, served_moving, t0 = [], 0, 0, 0, 0, time.time()
f = 0
ops = []
lights_gen = 0
cur_lights = np.array(sc.arrays["lights"], np.float32).copy()
cur_rot = np.array(sc.arrays["rotation"], np.float32).copy()
moving = False                                               # a stretch of frames with the monkey turning (and the light flickering) before every frame: the launch takes the arrays per frame
def turned(f):
    r = np.array(sc.arrays["rotation"], np.float32).reshape(-1, 2, 12).copy()
    c, s_ = np.cos(0.013 * f), np.sin(0.013 * f)
    R = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]]) * 2.0
    Ri = np.linalg.inv(R)
    for m, M in ((0, R), (1, Ri)):
        for col in range(3): r[2, m, 4 * col:4 * col + 3] = M[:, col]
    return r.reshape(-1)
def fail_dump():
    print('\n'.join(ops[-16:]), flush=True)
import atexit
while f < N or inflight:
    r = rnd.random()
    if f < N and len(inflight) < lanes and (r < 0.6 or not inflight):
        p = frame(f, shape)
        if moving:
            cur_rot = turned(f); ctx.update_transforms(cur_rot, sc.arrays["shift"])
            if rnd.random() < 0.5:
                cur_lights = np.array(sc.arrays["lights"], np.float32).copy(); cur_lights.reshape(-1, 6)[:, 3] *= 1.0 + 0.01 * (f % 50); cur_lights.reshape(-1, 6)[0, 0] += 0.02 * (f % 31)
                ctx.update_primary_light_sources(cur_lights)
        r8 = rnd.random() < 0.3                              # the canvas' bytes: the launch quantises as it resolves (another format: another launch)
        ops.append('begin %d %dx%d rgba8 %d lanes %d' % (f, p.width, p.height, r8, lanes))
        try:
            ctx.frame_begin(p, rgba8=r8)
        except capi.FlexLightHipError:
            fail_dump(); raise
        served += ctx.last_chained() == 3
        served_moving += ctx.last_chained() == 3 and ctx.server_moving()
        inflight.append((f, p, r8, (cur_lights.copy(), cur_rot.copy())))
        f += 1
    elif inflight:
        g, p, r8, gen = inflight.pop(0)
        ops.append('end %d' % g)
        try:
            got = ctx.frame_end()[0]
        except capi.FlexLightHipError:
            fail_dump(); raise
        if (g % 37 == 0 or (moving and g % 7 == 0)) and not inflight:              # (the comparison render would end the launch anyway: only when nothing is in flight)
            ctx.update_primary_light_sources(gen[0]); ctx.update_transforms(gen[1], sc.arrays["shift"])      # the arrays the frame was begun with
            want = ctx.render(p)[0]
            ctx.update_primary_light_sources(cur_lights); ctx.update_transforms(cur_rot, sc.arrays["shift"])
            checked += 1
            ok = np.array_equal(got, ctx.present(want)) if r8 else np.array_equal(got.view(np.uint32), want.view(np.uint32))
            if not ok:
                bad += 1
                d = (got != ctx.present(want)).any(axis=2) if r8 else (got.view(np.uint32) != want.view(np.uint32)).any(axis=2)
                print("frame %d differs: %d pixels, rows %s, rgba8 %d, shape %dx%d tile %d/%d" % (g, int(d.sum()), np.nonzero(d.any(axis=1))[0][:6], r8, p.width, p.height, p.tile_index, p.tile_count)); fail_dump()
    if rnd.random() < 0.01:
        time.sleep(rnd.random() * 0.01)                      # the host pauses: the launch waits for posts
    if rnd.random() < 0.004 and not inflight:
        lanes = rnd.choice([2, 3]); ctx.set_frame_lanes(lanes); ops.append('lanes %d' % lanes)
    if rnd.random() < 0.006:
        ops.append('shape'); shape = rnd.choice(shapes)                           # frames of another shape: the launch ends, another begins
    if rnd.random() < 0.003:
        cur_lights = np.array(sc.arrays["lights"], np.float32).copy(); ctx.update_primary_light_sources(cur_lights)    # (the scene's own lights: the same again — nothing — unless they had been changed)
    if rnd.random() < 0.003:
        cur_lights = np.array(sc.arrays["lights"], np.float32).copy(); cur_lights.reshape(-1, 6)[:, 3] *= rnd.choice([0.5, 1.0, 2.0])
        ops.append('lights changed'); ctx.update_primary_light_sources(cur_lights)                # other lights: the frames after are rendered with them (and compared with them)
    if rnd.random() < 0.004:
        moving = not moving; ops.append('moving %d' % moving)
    if rnd.random() < 0.001 and not inflight:
        ops.append('scene again'); ctx.update_scene(sc); cur_lights = np.array(sc.arrays["lights"], np.float32).copy(); cur_rot = np.array(sc.arrays["rotation"], np.float32).copy()      # the scene uploaded again: it has not moved yet
    if rnd.random() < 0.0005 and inflight:
        ops.append('pause 2.3 s'); time.sleep(2.3)                                      # the host pauses for longer than the launch waits: nothing may be lost
    if rnd.random() < 0.002:
        ops.append('sync render'); ctx.render(frame(f, shapes[3]))                      # a synchronous render while frames are in flight
print("%d frames (%d through the server, %d of them by a launch that takes lights and transforms per frame) in %.1f s, %d compared with their own render, %d differ" % (N, served, served_moving, time.time() - t0, checked, bad))
sys.exit(1 if bad else 0)
