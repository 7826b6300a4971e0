#!/usr/bin/env python3
"""Debug aid for the frame kernel (k_wf_frame): render one counted frame and print the control words a wave left behind if its
watchdog tripped (never in a healthy frame).  argv: width height [spp bounces]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
ctx.set_pipeline(3); ctx.set_wavefront_organisation(2)
w, h = int(sys.argv[1]), int(sys.argv[2])
spp, b = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2, 4)
p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=b, use_filter=0)
if os.environ.get("FLX_TILES"):
    p.tile_rows, p.tile_count, p.tile_index = 8, int(os.environ["FLX_TILES"]), 0
t0 = time.time()
_, cnt, _ = ctx.render(p, counters=True)
print("frame %dx%d spp %d bounces %d: %.3f s" % (w, h, spp, b, time.time() - t0), cnt)
t = ctx.get_tail_diag()
names = ["alive", "dry", "sq_tail", "sq_head", "sq_avail", "wq_tail", "wq_head", "wq_avail"]
print("shade waves that gave up:", t[8], {n: t[k] - 1 for k, n in enumerate(names) if t[k]})
print("walk waves that gave up: ", t[18], {n: t[10 + k] - 1 for k, n in enumerate(names) if t[10 + k]})
if t[22]:
    print("item queue found dry: %d workgroups, mean %.0f max %.0f cycles after their start, %.0f paths alive per workgroup then" % (t[22], t[20] / t[22], t[21], t[23] / t[22]))
if t[26]:
    print("walk waves: %d, lifetime mean %.0f max %.0f cycles  -> after the queue ran dry: mean %.0f, max %.0f cycles" % (t[26], t[24] / t[26], t[25], t[24] / t[26] - t[20] / max(1, t[22]), t[25] - t[20] / max(1, t[22])))
if t[39]:
    print("after the queue ran dry, paths alive per workgroup <= 256 / 64 / 16 at (mean over %d workgroups, cycles after the workgroup's start): %.0f / %.0f / %.0f" % (t[39], t[36] / t[39], t[37] / t[39], t[38] / t[39]))
