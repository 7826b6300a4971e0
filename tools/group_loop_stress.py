#!/usr/bin/env python3
"""Random walk through the device group's frame loop (flx_group_frame_begin / _end) on ONE GPU: a group of two contexts on device 0, frame shapes that the servers take and
that they do not (ragged last strips, odd widths), float and RGBA8 frames, the transforms changing before random frames, filter frames in between, 2 and 3 frames in flight, random
numbers of frames taken between begins — every frame compared with one context's render of the same arrays (bit for bit; RGBA8: the oracle-identical flx_present of it).  GPU box.
usage: group_loop_stress.py [frames] [seed]"""
import os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sc = Scene.golden("dragon")
one = capi.Context(0)
one.update_scene(sc)
rot0 = np.array(sc.arrays["rotation"], np.float32).reshape(-1, 2, 12)
def arrays(a):
    r = rot0.copy()
    c, s = np.cos(a), np.sin(a)
    R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]) * 2.0
    Ri = np.linalg.inv(R)
    for m, M in ((0, R), (1, Ri)):
        for col in range(3): r[2, m, 4 * col:4 * col + 3] = M[:, col]
    return r.reshape(-1)
shapes = [dict(width=640, height=368), dict(width=480, height=270), dict(width=500, height=264, samples=3), dict(width=320, height=200, max_reflections=2), dict(width=328, height=184)]
done = bad = 0
t0 = time.time()
while done < N:
    ranks = 2                # (on ONE GPU more than two contexts' persistent launches do not get on: the runtime's hardware queues are shared between their streams — profiles/r04_server.txt)
    lanes = rnd.choice([2, 3])
    tile_rows = rnd.choice([8, 8, 16])
    g = capi.Group([0] * ranks)
    g.update_scene(sc)
    g.set_frame_lanes(lanes)
    cur = np.array(sc.arrays["rotation"], np.float32)
    one.update_transforms(cur, sc.arrays["shift"])
    shape = rnd.choice(shapes)
    inflight = []            # (frame params, arrays in force, rgba8)
    ops = ['group of %d, lanes %d, tile_rows %d' % (ranks, lanes, tile_rows)]
    taken = []               # (pixels, frame params, arrays, rgba8): compared when nothing of the group is in flight — on this ONE GPU a render of another context
                             # waits for the CUs the servers' persistent launches hold, and they wait for the host
    def take():
        ops.append('end')
        try:
            tb = time.time(); got = g.frame_end()[0]; ops[-1] += '  (%.1f ms)' % ((time.time() - tb) * 1e3)
        except capi.FlexLightHipError:
            print(ops[0]); print('\n'.join(ops[-14:]))
            for r in range(ranks):
                try:
                    st = g.context(r).server_stats()
                    print(' context', r, 'launch %.3f ms' % ((st['end'] - st['start']) / 1e5), {k: st[k] for k in ('frames', 'tiles', 'rotations', 'host_done', 'host_posted', 'next_seq_stop')})
                except Exception as e:
                    print(' context', r, e)
            raise
        taken.append((got,) + inflight.pop(0))
    def verify():
        global bad, done
        for got, p, arr, rgba8 in taken:
            one.update_transforms(arr, sc.arrays["shift"])
            want = one.render(p)[0]
            ok = np.array_equal(got, one.present(want)) if rgba8 else np.array_equal(got.view(np.uint32), want.view(np.uint32))
            if not ok:
                bad += 1
                print("DIFFERS: ranks %d lanes %d tile_rows %d %dx%d rgba8 %s filter %d" % (ranks, lanes, tile_rows, p.width, p.height, rgba8, p.use_filter), flush=True)
            done += 1
        taken.clear()
    for f in range(rnd.randint(8, 30)):
        if rnd.random() < 0.15: shape = rnd.choice(shapes)
        if rnd.random() < 0.3:
            cur = arrays(rnd.random())
            g.update_transforms(cur, sc.arrays["shift"]); ops.append('transforms changed')
        elif rnd.random() < 0.3:
            g.update_transforms(cur, sc.arrays["shift"])          # the same arrays again: nothing
        if inflight and rnd.random() < 0.02:
            ops.append('pause 2.3 s'); time.sleep(2.3)              # longer than the servers' launches wait for the host: nothing may be lost
        rgba8 = rnd.random() < 0.4
        flt = rnd.random() < 0.08
        p = sc.frame_params(use_filter=1 if flt else 0, **shape)
        p.camera[0] += 0.03 * rnd.randint(0, 20); p.camera[2] -= 0.02 * rnd.randint(0, 20); p.random_seed = float(rnd.randint(0, 3))
        while len(inflight) >= lanes or (inflight and rnd.random() < 0.3):
            take()
        ops.append('begin %dx%d spp %d rgba8 %d filter %d' % (p.width, p.height, p.samples, rgba8, p.use_filter))
        try:
            tb = time.time(); g.frame_begin(p, tile_rows=tile_rows, rgba8=rgba8); ops[-1] += '  (%.1f ms)' % ((time.time() - tb) * 1e3)
        except capi.FlexLightHipError:
            print(ops[0]); print('\n'.join(ops[-14:]))
            for r in range(ranks):
                try:
                    st = g.context(r).server_stats()
                    print(' context', r, 'launch %.3f ms' % ((st['end'] - st['start']) / 1e5), {k: st[k] for k in ('frames', 'tiles', 'rotations', 'host_done', 'host_posted', 'next_seq_stop')})
                except Exception as e:
                    print(' context', r, e)
            raise
        inflight.append((p, cur.copy(), rgba8))
    while inflight:
        take()
    verify()
    g.close()
print("%d frames in %.1f s, %d differ" % (done, time.time() - t0, bad))
sys.exit(1 if bad else 0)
