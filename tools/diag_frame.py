#!/usr/bin/env python3
"""Per-bounce work counters + scheduler statistics of one dragon frame (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
sc = Scene.golden(name)
ctx = capi.Context(0)
ctx.update_scene(sc)
prev = None
for bounces in range(1, sc.meta["frame"]["maxReflections"] + 1):
    p = sc.frame_params(max_reflections=bounces, use_filter=0)
    _, cnt, _ = ctx.render(p, counters=True)
    d = ctx.get_diag()
    delta = {k: cnt[k] - (prev[k] if prev else 0) for k in cnt}
    b = bounces - 1
    it, ba = d[2 * min(b, 3)], d[2 * min(b, 3) + 1]
    visits = delta["closest_visits"] + delta["shadow_visits"]
    print("bounce %d: shades %d closest walks %d (%.1f visits) shadow walks %d (%.1f visits) | walk-kernel wave-iterations %d -> lane utilisation %.2f, batches %d"
          % (b, delta["shades"], delta["closest_walks"], delta["closest_visits"] / max(1, delta["closest_walks"]), delta["shadow_walks"],
             delta["shadow_visits"] / max(1, delta["shadow_walks"]), it, visits / max(1, it * 64), ba))
    if b == 0:
        fold, refill, inner, life, waves = d[8:13]
        print('   bounce-0 walk waves %d: mean lifetime %.0f cycles; share fold %.3f refill %.3f (record loads %.3f) steps %.3f; cycles per wave-iteration %.0f' % (waves, life / max(1, waves), fold / max(1, life), refill / max(1, life), d[13] / max(1, life), inner / max(1, life), inner / max(1, it)))
    if b < 4 and d[17 + 3 * b]:
        print('   wave lifetimes: %d waves, mean %.0f, max %.0f cycles' % (d[17 + 3 * b], d[16 + 3 * b] / d[17 + 3 * b], d[18 + 3 * b]))
    print('   tail rounds (wave-rounds) %d, consolidating %d, walks moved %d, waves summed %d' % (d[28], d[29], d[30], d[31]))
    prev = cnt
