#!/usr/bin/env python3
"""A/B of the frame kernel's walk waves: one job per lane (k_wf_frame) against two (k_wf_frame2), dragon 1080p 8 spp 4 bounces (or argv: width height spp bounces).
Per variant: ms per frame (flx_last_frame_ms over uncounted frames), then one counted frame's scheduler words (flx_get_tail_diag): when the workgroups found the tile
queue dry, wave lifetimes, and for two jobs the phases' trips and lanes.  profiles/r05_two_walks.txt."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden(os.environ.get("FLX_SCENE", "dragon"))
a = [int(x) for x in sys.argv[1:5]] + [1920, 1080, 8, 4][len(sys.argv[1:5]):]
w, h, spp, b = a
ctx = capi.Context(0)
ctx.update_scene(sc)
p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=b, use_filter=0)
if os.environ.get("FLX_TILES"):
    p.tile_rows, p.tile_count, p.tile_index = 8, int(os.environ["FLX_TILES"]), 0
ref = None
for jobs in [int(x) for x in os.environ.get("FLX_JOBS", "1,2").split(",")]:
    ctx.set_walk_jobs(jobs)
    ms = []
    for i in range(12):
        out = ctx.render(p)[0]
        ms.append(ctx.last_frame_ms()[0])
    if ref is None:
        ref = out
    same = bool(np.array_equal(out.view(np.uint32), ref.view(np.uint32)))
    _, cnt, _ = ctx.render(p, counters=True)
    t = ctx.get_tail_diag()
    print("jobs per lane %d: %.3f ms per frame (median of 12; min %.3f)   equal to the first variant's frame: %s   organisation %d" % (jobs, float(np.median(ms[2:])), min(ms), same, ctx.last_organisation()))
    print("   counted frame:", {k: int(v) for k, v in cnt.items()} if isinstance(cnt, dict) else cnt)
    if t[22]:
        print("   tile queue found dry: %d workgroups, mean %.0f max %.0f cycles after their start, %.0f paths alive per workgroup then" % (t[22], t[20] / t[22], t[21], t[23] / t[22]))
    if t[26]:
        print("   walk waves: %d, lifetime mean %.0f max %.0f cycles" % (t[26], t[24] / t[26], t[25]))
    if t[27]:
        print("   box phases %d x %.1f lanes, triangle phases %d x %.1f lanes  (per walk wave: %.0f box, %.0f triangle phases)" % (t[27], t[28] / t[27], t[29], t[30] / max(1, t[29]), t[27] / t[26], t[29] / t[26]))
