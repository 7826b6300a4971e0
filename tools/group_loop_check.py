"""Where does a frame of the group's loop differ from one context's render?  (diagnosis; tests/test_group_loop_gpu.py is the test)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "web-ray-tracer_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene


def moving(sc, f, **kw):
    p = sc.frame_params(use_filter=0, **kw)
    p.camera[0] += 0.05 * f
    p.camera[2] -= 0.03 * f
    p.random_seed = float(f % 4)
    return p


def diff_rows(a, b):
    d = (a.view(np.uint32) != b.view(np.uint32)).any(axis=(1, 2))
    return np.nonzero(d)[0]


sc = Scene.golden("dragon")
hip = capi.Context(0)
hip.update_scene(sc)
W, H = 640, 368
for tr in (8, 16):
    hip.set_frame_lanes(3); hip.set_frame_chain(3)
    for r in range(2):
        ps = [moving(sc, f, width=W, height=H, tile=(tr, r, 2)) for f in range(3)]
        want = [hip.render(p)[0] for p in ps]
        for p in ps:
            hip.frame_begin(p)
        got = [hip.frame_end()[0] for _ in ps]
        print("one context, server, tile_rows", tr, "rank", r, "kinds", hip.last_chained(), [len(diff_rows(g, w)) for g, w in zip(got, want)])
    g = capi.Group([0, 0])
    for r in range(2):
        g.context(r).set_server_groups(128)
    g.update_scene(sc)
    ps = [moving(sc, f, width=W, height=H) for f in range(5)]
    want = [hip.render(p)[0] for p in ps]
    for lanes in (2, 3):
        g.set_frame_lanes(lanes)
        got = []
        try:
            for p in ps:
                if g.frames_in_flight() == lanes:
                    got.append(g.frame_end()[0])
                g.frame_begin(p, tile_rows=tr)
            while g.frames_in_flight():
                got.append(g.frame_end()[0])
        except capi.FlexLightHipError as e:
            print("FAILED:", e)
            for r in range(2):
                c = g.context(r)
                print(" context", r, c.server_stats())
                d = c.server_dump()
                for k in range(4):
                    if d[k].any():
                        print("  dump", k, "wg", int(d[k][0]) & 0xffffffff, "wave", int(d[k][1]) & 0xffffffff, "ctl", [int(x) for x in d[k][2:66]], "relay/tileNext", [int(x) for x in d[k][66:72]])
            raise SystemExit(1)
        for f in range(5):
            rows = diff_rows(got[f], want[f])
            others = [k for k in range(5) if len(diff_rows(got[f], want[k])) == 0]
            print("group tile_rows", tr, "lanes", lanes, "frame", f, "rows that differ", len(rows), rows[:12], "equals frame", others)
    g.close()
