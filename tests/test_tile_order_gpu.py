"""The order in which the frame kernel draws a frame's 8 x 8 screen tiles (include/flexlight_hip_debug.h: flx_debug_set_tile_order, flx_debug_set_adaptive_order)
is a matter of scheduling only: every path writes its radiance to its own slot (flx_wavefront_common.h: finalize_path), so a frame is the same bits whatever
the order — screen order, an explicit permutation, or the adaptive order the library makes from what the tiles cost in the frame before (thin frames: k_resolve sums
the paths' time in the walk lanes per tile, k_tile_order sorts)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("front", [3, 2], ids=["front_in_its_own_kernel", "front_inside"])
def test_frames_do_not_depend_on_the_tile_order(hip, oracle, scenes, front):
    sc = scenes("dragon")
    hip.update_scene(sc)
    p = sc.frame_params(width=320, height=184, samples=2, max_reflections=4, use_filter=0)
    n = (320 // 8) * (184 // 8)
    hip.set_frame_chain(0)
    hip.set_frame_front(front)
    hip.set_adaptive_order(0)
    try:
        base, base_cnt, _ = hip.render(p, counters=True)
        want, want_cnt = oracle.render(sc, p)[:2]
        assert np.array_equal(_bits(base), _bits(want)) and base_cnt == want_cnt
        rng = np.random.default_rng(7)
        for order in (np.arange(n)[::-1], rng.permutation(n), rng.permutation(n)):
            hip.set_tile_order(order.astype(np.uint32))
            got, cnt, _ = hip.render(p, counters=True)
            assert np.array_equal(_bits(got), _bits(base)) and cnt == base_cnt
            got = hip.render(p)[0]                       # (the uncounted kernels)
            assert np.array_equal(_bits(got), _bits(base))
        hip.set_tile_order(None)
        with pytest.raises(Exception):
            hip.set_tile_order(np.zeros(n, np.uint32))   # not a permutation
        with pytest.raises(Exception):
            hip.set_tile_order(np.arange(n, dtype=np.uint32) + 1)
        # an order for another frame shape is not used (and does no harm)
        hip.set_tile_order(np.arange(n // 2, dtype=np.uint32))
        assert np.array_equal(_bits(hip.render(p)[0]), _bits(base))
    finally:
        hip.set_tile_order(None)
        hip.set_adaptive_order(1)
        hip.set_frame_front(1)
        hip.set_frame_chain(2)


def test_the_adaptive_order_leaves_every_frame_as_it_is(hip, scenes):
    """frame after frame with the adaptive order on (the default): the second frame on is drawn in an order made from the frame before; a change of the frame's
    shape, of the camera and of the share starts over; all of them equal the frames rendered in screen order"""
    sc = scenes("dragon")
    hip.update_scene(sc)
    hip.set_frame_chain(0)
    try:
        def frames(adaptive):
            hip.set_adaptive_order(adaptive)
            out = []
            for shape in ((384, 216, None), (384, 216, (8, 1, 2)), (320, 200, None), (384, 216, None)):
                for f in range(3):
                    p = sc.frame_params(width=shape[0], height=shape[1], samples=2, max_reflections=3, use_filter=0)
                    p.camera[0] += 0.05 * f                  # (a moving camera: last frame's costs are only a guess)
                    if shape[2]:
                        p.tile_rows, p.tile_index, p.tile_count = shape[2]
                    out.append(hip.render(p)[0].copy())
            return out
        a, b = frames(1), frames(0)
        assert hip.last_organisation() >= 2                  # the frame kernel ran (what the order is for)
        for x, y in zip(a, b):
            assert np.array_equal(_bits(x), _bits(y))
    finally:
        hip.set_adaptive_order(1)
        hip.set_frame_chain(2)


@pytest.mark.parametrize("front", [3, 2], ids=["front_in_its_own_kernel", "front_inside"])
def test_the_per_tile_visit_counts_add_up_to_the_frames_counters(hip, scenes, front):
    """flx_debug_tile_cost (the measure tools/tile_order_ab.py orders tiles by): a counted frame's visits per 8 x 8 screen tile — first half: the bounce loop's walks, second
    half: the primary rays — sum to the frame's own work counters, and tiles outside the dragon's silhouette cost less than tiles inside"""
    sc = scenes("dragon")
    hip.update_scene(sc)
    w, h = 320, 184
    p = sc.frame_params(width=w, height=h, samples=2, max_reflections=4, use_filter=0)
    n = (w // 8) * (h // 8)
    hip.set_frame_chain(0)
    hip.set_frame_front(front)
    try:
        hip.tile_cost(2 * n)
        _, cnt, _ = hip.render(p, counters=True)
        assert hip.last_organisation() >= 2
        both = hip.tile_cost(2 * n, read=True)
        bounce, primary = both[:n], both[n:]
        assert int(bounce.sum()) == cnt["closest_visits"] + cnt["shadow_visits"]
        assert int(primary.sum()) == cnt["primary_visits"]
        assert bounce.max() > 4 * np.median(bounce)          # (the dragon's tiles against the floor's)
        # a second counted frame starts from zero again (the read above cleared the sums)
        _, cnt2, _ = hip.render(p, counters=True)
        again = hip.tile_cost(0, read=True)
        assert cnt2 == cnt and np.array_equal(again[:2 * n], both)
    finally:
        hip.tile_cost(0)
        hip.set_frame_front(1)
        hip.set_frame_chain(2)


def test_the_sort_makes_a_permutation_heaviest_classes_first(hip):
    """k_tile_order on its own (flx_debug_tile_order_of): whatever the costs — equal, zero, huge, NaN, infinite, any count of tiles — the order is a permutation of the tiles
    (a tile that is missing would never be drawn, one that is there twice drawn twice); classes come heaviest first, screen order inside a class"""
    rng = np.random.default_rng(3)
    for n in (1, 2, 63, 64, 511, 512, 513, 4080, 8100, 32400, 129600):
        cases = [np.zeros(n, np.float32), np.ones(n, np.float32), rng.random(n).astype(np.float32) * 1e4,
                 np.exp(rng.normal(9.0, 1.5, n)).astype(np.float32)]
        odd = np.exp(rng.normal(9.0, 1.5, n)).astype(np.float32)
        odd[rng.integers(0, n, max(1, n // 50))] = np.nan
        odd[rng.integers(0, n, max(1, n // 50))] = np.inf
        odd[rng.integers(0, n, max(1, n // 50))] = 0.0
        odd[rng.integers(0, n, max(1, n // 50))] = 3.0e38
        cases.append(odd)
        for cost in cases:
            for mode in (1, 0):
                order = hip.tile_order_of(cost, mode)
                assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32)), (n, mode)
        # heaviest first: with sixteen classes of (about) equal size the heaviest sixteenth of the tiles comes before the lightest sixteenth, and equal costs keep screen order
        cost = cases[3]
        order = hip.tile_order_of(cost, 1).astype(np.int64)
        if n >= 512:
            k = n // 16
            assert np.median(cost[order[:k]]) > np.median(cost[order[-k:]]) * 4
            pos = np.empty(n, np.int64); pos[order] = np.arange(n)
            heavy = np.argsort(-cost, kind="stable")[: k // 2]
            light = np.argsort(cost, kind="stable")[: k // 2]
            assert pos[heavy].max() < pos[light].min()
        assert np.array_equal(hip.tile_order_of(np.full(n, 7.0, np.float32), 1), np.arange(n, dtype=np.uint32))


def test_the_frame_loop_on_two_lanes_with_the_adaptive_order(hip, scenes):
    """flx_frame_begin / _end with every frame its own launches on two lanes (flx_set_frame_chain 0): each lane keeps its own measure and order; thin frames with a moving
    camera, two in flight, equal the same frames rendered synchronously in screen order"""
    sc = scenes("dragon")
    hip.update_scene(sc)
    hip.set_frame_chain(0)
    hip.set_frame_lanes(2)
    try:
        def params(f):
            p = sc.frame_params(width=640, height=360, samples=2, max_reflections=3, use_filter=0)
            p.camera[0] += 0.03 * f
            p.tile_rows, p.tile_index, p.tile_count = 8, f % 2, 2        # (the lanes see the two shares in turn: the order each made is for the other share's frame next time)
            return p
        hip.set_adaptive_order(1)
        got = []
        n = 12
        for f in range(n):
            if hip.frames_in_flight() == 2:
                got.append(hip.frame_end()[0].copy())
            hip.frame_begin(params(f))
        while hip.frames_in_flight():
            got.append(hip.frame_end()[0].copy())
        hip.set_adaptive_order(0)
        for f in range(n):
            want = hip.render(params(f))[0]
            assert np.array_equal(_bits(got[f]), _bits(want)), f
    finally:
        hip.set_adaptive_order(1)
        hip.set_frame_lanes(2)
        hip.set_frame_chain(2)
