#!/usr/bin/env python3
"""k_tile_order on its own, for rocprofv3 --kernel-trace --stats: the sort of 4 080 / 8 100 / 16 200 / 32 400 tiles (a rank's eighth, quarter, half, a whole 1080p frame), 20 times each"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "web-ray-tracer_amd"))
from flexlight_hip import capi
ctx = capi.Context(0)
rng = np.random.default_rng(1)
for n in (4080, 8100, 16200, 32400):
    cost = np.exp(rng.normal(9.0, 1.5, n)).astype(np.float32)
    for _ in range(20): ctx.tile_order_of(cost, 1)
