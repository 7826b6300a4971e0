#!/usr/bin/env python3
"""Per-bounce walk-kernel statistics for one rank's share (1/N) of the dragon frame (counted frames; GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = Scene.golden("dragon")
ctx = capi.Context(0)
ctx.update_scene(sc)
prev = None
for bounces in range(1, 5):
    p = sc.frame_params(max_reflections=bounces, use_filter=0)
    p.tile_rows, p.tile_count, p.tile_index = 8, n, 0
    _, cnt, _ = ctx.render(p, counters=True)
    d = ctx.get_diag()
    delta = {k: cnt[k] - (prev[k] if prev else 0) for k in cnt}
    b = bounces - 1
    it, ba = d[2 * b], d[2 * b + 1]
    visits = delta["closest_visits"] + delta["shadow_visits"]
    waves = max(1, d[17 + 3 * b])
    print("bounce %d: paths %d visits %d (%.1f per path) | waves %d, wave-iterations %d (%.0f per wave), lane utilisation %.2f, batches %d | lifetime mean %.0f max %.0f cycles -> %.0f cycles per iteration"
          % (b, delta["shades"], visits, visits / max(1, delta["shades"]), waves, it, it / waves, visits / max(1, it * 64), ba, d[16 + 3 * b] / waves, d[18 + 3 * b], d[16 + 3 * b] / max(1, it)))
    prev = cnt
p = sc.frame_params(max_reflections=1, use_filter=0)
p.tile_rows, p.tile_count, p.tile_index = 8, n, 0
_, cnt, _ = ctx.render(p, counters=True)
d = ctx.get_diag()
fold, refill, inner, life, waves, load, tail = d[8:15]
print("bounce 0 alone: waves %d lifetime %.0f | shares: fold %.3f refill+setup %.3f (loads %.3f) steps %.3f tail rounds %.3f" % (waves, life / max(1, waves), fold / life, refill / life, load / life, inner / life, tail / life))
