"""The chained frame loop (csrc/flx_chain.hip: consecutive frames overlap inside one persistent launch) and the device error word.

A chained frame is rendered by up to two kernels — the one before it works ahead on it, its own completes it — with walks suspended in flight and taken up
again; per path nothing may change: every frame must equal its own flx_render bit for bit, and through it the oracle (tests/test_parity_gpu.py holds
flx_render against the oracle on the same scenes).  Reference loop: modules/pathtracerWGL2.js:254-303 (frame after frame from one context)."""
import numpy as np
import pytest

from parity_util import bit_mismatches

pytestmark = pytest.mark.gpu


def moving(sc, f, **kw):
    p = sc.frame_params(use_filter=0, **kw)
    p.camera[0] += 0.05 * f
    p.camera[2] -= 0.03 * f
    p.random_seed = float(f % 4)
    return p


def loop(ctx, ps):
    """the frames of ps through flx_frame_begin / _end with two in flight -> (frames, how each was begun: flx_last_chained)"""
    got, kinds = [], []
    ctx.frame_begin(ps[0])
    kinds.append(ctx.last_chained())
    for p in ps[1:]:
        ctx.frame_begin(p)
        kinds.append(ctx.last_chained())
        got.append(ctx.frame_end()[0])
    got.append(ctx.frame_end()[0])
    return got, kinds


chain_of_launches = pytest.mark.experiments      # flx_set_frame_chain(ctx, 1): csrc/flx_chain.hip is in `make EXPERIMENTS=1`'s library only (the frame server replaced it)


@pytest.fixture()
def chained(hip):
    hip.set_frame_lanes(2)
    hip.set_frame_chain(1)
    yield hip
    hip.set_frame_chain(2)


@chain_of_launches
@pytest.mark.parametrize("shape", [dict(width=640, height=360), dict(width=1920, height=1080, tile=(8, 3, 8)), dict(width=500, height=264, samples=3)])
def test_chained_frames_equal_their_own_render(chained, scenes, shape):
    """a camera that moves from frame to frame, a seed that changes: whole frames, a rank's strips of the BASELINE frame, an odd width with three samples"""
    sc = scenes("dragon")
    chained.update_scene(sc)
    ps = [moving(sc, f, **shape) for f in range(7)]
    want = [chained.render(p)[0] for p in ps]
    got, kinds = loop(chained, ps)
    assert kinds == [1] + [2] * 6, kinds                      # the first frame begins the chain, the others continue it
    for f in range(7):
        assert got[f].shape == want[f].shape
        assert bit_mismatches(got[f], want[f]) == 0, "frame %d differs from its own render" % f


@chain_of_launches
def test_a_chain_ends_where_the_frames_change(chained, scenes):
    """another frame shape, a scene upload, a synchronous render in between: each starts a new chain (kind 1) and every frame is still its own render"""
    sc = scenes("dragon")
    chained.update_scene(sc)
    a = [moving(sc, f, width=480, height=272) for f in range(3)]
    b = [moving(sc, f, width=320, height=200) for f in range(3)]
    want = [chained.render(p)[0] for p in a + b]
    got, kinds = loop(chained, a + b)
    assert kinds == [1, 2, 2, 1, 2, 2], kinds
    for f in range(6):
        assert bit_mismatches(got[f], want[f]) == 0, f
    # other lights between two frames: the kernel before must not have worked ahead with the old ones (the SAME lights again are nothing: the chain goes on)
    dim = np.array(sc.arrays["lights"], np.float32).copy()
    dim.reshape(-1, 6)[:, 3] *= 0.5
    chained.update_primary_light_sources(dim)
    want_dim = chained.render(a[1])[0]
    chained.update_primary_light_sources(sc.arrays["lights"])
    chained.frame_begin(a[0])
    chained.update_primary_light_sources(sc.arrays["lights"])
    chained.frame_begin(a[1])
    assert chained.last_chained() == 2
    assert bit_mismatches(chained.frame_end()[0], want[0]) == 0 and bit_mismatches(chained.frame_end()[0], want[1]) == 0
    chained.frame_begin(a[0])
    chained.update_primary_light_sources(dim)
    chained.frame_begin(a[1])
    assert chained.last_chained() == 1
    g0 = chained.frame_end()[0]
    g1 = chained.frame_end()[0]
    chained.update_primary_light_sources(sc.arrays["lights"])
    assert bit_mismatches(g0, want[0]) == 0 and bit_mismatches(g1, want_dim) == 0 and bit_mismatches(want_dim, want[1]) != 0
    # a synchronous render between chained frames uses the same workspace: the chain must not resume state it overwrote
    chained.frame_begin(a[0])
    chained.frame_begin(a[1])
    g0 = chained.frame_end()[0]
    g1 = chained.frame_end()[0]
    mid = chained.render(b[2])[0]
    chained.frame_begin(a[2])
    assert chained.last_chained() == 1
    g2 = chained.frame_end()[0]
    assert bit_mismatches(mid, want[5]) == 0
    assert bit_mismatches(g0, want[0]) == 0 and bit_mismatches(g1, want[1]) == 0 and bit_mismatches(g2, want[2]) == 0


@chain_of_launches
def test_frames_the_chain_does_not_take_run_on_two_lanes(chained, scenes):
    """a scene of fewer than 129 entries (another pipeline), a filter frame, strips that are not a multiple of 8 rows: not chained, still right"""
    sc = scenes("cornell_obj")
    chained.update_scene(sc)
    ps = [moving(sc, f, width=320, height=200, samples=2, max_reflections=3) for f in range(3)]
    want = [chained.render(p)[0] for p in ps]
    got, kinds = loop(chained, ps)
    assert kinds == [0, 0, 0]
    for f in range(3):
        assert bit_mismatches(got[f], want[f]) == 0
    sc = scenes("dragon")
    chained.update_scene(sc)
    ps = [moving(sc, f, width=320, height=204) for f in range(3)]              # 204 rows: the last tile row would straddle the slots
    want = [chained.render(p)[0] for p in ps]
    got, kinds = loop(chained, ps)
    assert kinds == [0, 0, 0]
    for f in range(3):
        assert bit_mismatches(got[f], want[f]) == 0


def test_a_tripped_watchdog_reaches_the_status_code(hip, scenes):
    """fault injection: the frame kernel's shade waves drop every batch and its waves give up after a few hundred polls — the frame is incomplete and
    flx_render says so (FLX_ERR_DEVICE with the bits), the word is cleared, and the context renders on"""
    from flexlight_hip import capi
    sc = scenes("dragon")
    hip.update_scene(sc)
    p = sc.frame_params(width=480, height=272, samples=2, max_reflections=4, use_filter=0)
    want = hip.render(p)[0]
    assert hip.last_organisation() in (2, 3)                  # the frame kernel ran
    hip.inject_fault(400, 1)
    try:
        with pytest.raises(capi.FlexLightHipError) as e:
            hip.render(p)
        assert "device error" in str(e.value) and "watchdog" in str(e.value), str(e.value)
        # through the frame loop too: flx_frame_end reports it
        hip.set_frame_chain(0)
        hip.frame_begin(p)
        with pytest.raises(capi.FlexLightHipError) as e:
            hip.frame_end()
        assert "device error" in str(e.value)
    finally:
        hip.inject_fault(0, 0)
        hip.set_frame_chain(2)
    got = hip.render(p)[0]                                     # the word was cleared, the rings re-initialised
    assert bit_mismatches(got, want) == 0
