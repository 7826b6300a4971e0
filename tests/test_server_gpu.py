"""The frame server (csrc/flx_server.hip): ONE persistent launch renders the frames of the loop as flx_frame_begin posts them — the frames in flight share the
machine inside the launch, every workgroup resolves the screen tiles it made, the host is told through pinned memory.

Per path nothing may change: every frame must equal its own flx_render bit for bit (tests/test_parity_gpu.py holds flx_render against the oracle on the same
scenes).  Reference loop: modules/pathtracerWGL2.js:254-303 (frame after frame from one context, never waiting for the GPU inside a frame)."""
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


def loop(ctx, ps, lanes, **kw):
    got, kinds = [], []
    for p in ps:
        if ctx.frames_in_flight() == lanes:
            got.append(ctx.frame_end()[0])
        ctx.frame_begin(p, **kw)
        kinds.append(ctx.last_chained())
    while ctx.frames_in_flight():
        got.append(ctx.frame_end()[0])
    return got, kinds


@pytest.fixture()
def served(hip):
    hip.set_frame_chain(3)                                    # (every frame the server can take, whatever its size)
    yield hip
    hip.set_frame_lanes(2)
    hip.set_frame_chain(2)


@pytest.mark.parametrize("lanes", [2, 3])
@pytest.mark.parametrize("shape", [dict(width=640, height=360), dict(width=1920, height=1080, tile=(8, 5, 8)), dict(width=500, height=264, samples=3)])
def test_served_frames_equal_their_own_render(served, scenes, shape, lanes):
    """a camera that moves from frame to frame, a seed that changes; two and three frames in flight: whole frames, a rank's strips of the BASELINE frame, an
    odd width with three samples"""
    sc = scenes("dragon")
    served.update_scene(sc)
    served.set_frame_lanes(lanes)
    ps = [moving(sc, f, **shape) for f in range(8)]
    want = [served.render(p)[0] for p in ps]
    got, kinds = loop(served, ps, lanes)
    assert kinds == [3] * 8, kinds                            # every frame went to the server
    for f in range(8):
        assert got[f].shape == want[f].shape
        assert bit_mismatches(got[f], want[f]) == 0, "frame %d differs from its own render" % f


def test_the_server_ends_and_starts_again_where_the_frames_change(served, scenes):
    """another frame shape, a scene upload, a synchronous render between frames in flight, the loop running empty: the launch ends (its frames complete
    first) and another begins; every frame is still its own render"""
    sc = scenes("dragon")
    served.update_scene(sc)
    served.set_frame_lanes(2)
    a = [moving(sc, f, width=480, height=272) for f in range(3)]
    b = [moving(sc, f, width=320, height=200) for f in range(3)]
    want = [served.render(p)[0] for p in a + b]
    got, _ = loop(served, a + b, 2)
    for f in range(6):
        assert bit_mismatches(got[f], want[f]) == 0, f
    served.frame_begin(a[0])
    served.update_primary_light_sources(sc.arrays["lights"])            # (ends the launch: frame a[0] completes first)
    served.frame_begin(a[1])
    g0 = served.frame_end()[0]
    g1 = served.frame_end()[0]
    assert bit_mismatches(g0, want[0]) == 0 and bit_mismatches(g1, want[1]) == 0
    served.frame_begin(a[0])
    served.frame_begin(a[1])
    mid = served.render(b[2])[0]                                        # the workspace goes to this frame: the two in flight are resolved first
    g0 = served.frame_end()[0]
    g1 = served.frame_end()[0]
    served.frame_begin(a[2])
    g2 = served.frame_end()[0]
    assert bit_mismatches(mid, want[5]) == 0
    assert bit_mismatches(g0, want[0]) == 0 and bit_mismatches(g1, want[1]) == 0 and bit_mismatches(g2, want[2]) == 0


def test_an_upload_of_unchanged_lights_and_transforms_is_nothing(served, scenes):
    """the reference refills its transform buffer and its light texture every frame (pathtracerWGL2.js:258-262, 361-365) and so does the JavaScript host here:
    the same arrays again must not end the server's launch (an upload stops it: it reads the scene) — ONE launch serves the whole loop; arrays that did change
    end it, and the frames after show the change"""
    sc = scenes("dragon")
    served.update_scene(sc)
    served.set_frame_lanes(3)
    try:
        ps = [moving(sc, f, width=480, height=272) for f in range(8)]
        want = [served.render(p)[0] for p in ps]
        got = []
        for p in ps:
            if served.frames_in_flight() == 3:
                got.append(served.frame_end()[0])
            served.update_primary_light_sources(sc.arrays["lights"])
            served.update_transforms(sc.arrays["rotation"], sc.arrays["shift"])
            served.frame_begin(p)
            assert served.last_chained() == 3
        while served.frames_in_flight():
            got.append(served.frame_end()[0])
        for f in range(8):
            assert bit_mismatches(got[f], want[f]) == 0, f
        assert served.server_stats()["frames"] == 8                     # one launch rendered all eight
        dim = np.array(sc.arrays["lights"], np.float32).copy()
        dim.reshape(-1, 6)[:, 3] *= 0.5
        served.update_primary_light_sources(dim)
        served.frame_begin(ps[0])
        darker = served.frame_end()[0]
        assert bit_mismatches(darker, want[0]) != 0 and darker[..., :3].sum() < want[0][..., :3].sum()
        served.update_primary_light_sources(sc.arrays["lights"])
        served.frame_begin(ps[0])
        assert bit_mismatches(served.frame_end()[0], want[0]) == 0
    finally:
        served.update_primary_light_sources(sc.arrays["lights"])


def test_the_server_hands_out_the_canvas_bytes(served, oracle, scenes):
    """frames begun as FLX_FRAME_RGBA8 go to the server too: its workgroups quantise the tiles they resolve (ServerArgs::out8) — the bytes are flx_present's of the
    float frame; a change of format ends the launch and starts another"""
    sc = scenes("dragon")
    served.update_scene(sc)
    served.set_frame_lanes(3)
    ps = [moving(sc, f, width=480, height=272) for f in range(6)]
    want = [served.render(p)[0] for p in ps]
    got, kinds = loop(served, ps, 3, rgba8=True)
    assert kinds == [3] * 6, kinds
    for f in range(6):
        assert got[f].dtype == np.uint8 and np.array_equal(got[f], oracle.present(want[f])), f
    mixed, kinds = [], []
    for f, p in enumerate(ps):
        if served.frames_in_flight() == 3:
            mixed.append(served.frame_end()[0])
        served.frame_begin(p, rgba8=(f % 2 == 1))
        kinds.append(served.last_chained())
    while served.frames_in_flight():
        mixed.append(served.frame_end()[0])
    assert kinds == [3] * 6
    for f in range(6):
        if f % 2:
            assert np.array_equal(mixed[f], oracle.present(want[f])), f
        else:
            assert bit_mismatches(mixed[f], want[f]) == 0, f


def test_a_scene_that_moves_goes_to_the_lanes(hip, oracle, scenes):
    """the default mode (flx_set_frame_chain(2)): the server for a rank's share of a frame while the scene stands still; a frame that follows an upload of changed
    transforms goes to the two lanes (the launch would end and start again around it: tools/dynamic_scene_time.py); every frame is the oracle's for its arrays"""
    import copy
    sc = scenes("dragon")
    hip.update_scene(sc)
    hip.set_frame_lanes(3)
    hip.set_frame_chain(2)
    hip.set_server_moving_scenes(0)                           # (the behaviour before the server took moving scenes — and that of scenes whose arrays do not fit a post)
    try:
        p = sc.frame_params(use_filter=0, width=640, height=360, tile=(8, 1, 2))
        rot0 = np.array(sc.arrays["rotation"], np.float32).reshape(-1, 2, 12)

        def arrays(f):
            r = rot0.copy()
            c, s_ = np.cos(0.05 * f), np.sin(0.05 * f)
            R = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]]) * 2.0
            Ri = np.linalg.inv(R)
            for m, M in ((0, R), (1, Ri)):
                for col in range(3):
                    r[2, m, 4 * col:4 * col + 3] = M[:, col]
            return r.reshape(-1)
        hip.frame_begin(p)                                    # (the frame that follows update_scene itself goes to the lanes)
        hip.frame_end()
        kinds, got = [], []
        for f in range(6):
            if hip.frames_in_flight() == 2:
                got.append(hip.frame_end()[0])
            if f >= 3:
                hip.update_transforms(arrays(f), sc.arrays["shift"])
            hip.frame_begin(p)
            kinds.append(hip.last_chained())
        while hip.frames_in_flight():
            got.append(hip.frame_end()[0])
        assert kinds[:3] == [3, 3, 3] and kinds[3:] == [0, 0, 0], kinds
        for f in (0, 4, 5):
            moved = copy.copy(sc)
            moved.arrays = dict(sc.arrays, rotation=arrays(f) if f >= 3 else sc.arrays["rotation"])
            want = oracle.render(moved, p)[0]
            assert bit_mismatches(got[f], want) == 0, f
        hip.frame_begin(p)                                    # the scene stands still again: back to the server
        hip.frame_begin(p)
        assert hip.last_chained() == 3
        while hip.frames_in_flight():
            hip.frame_end()
    finally:
        hip.update_transforms(sc.arrays["rotation"], sc.arrays["shift"])
        hip.set_frame_lanes(2)
        hip.set_server_moving_scenes(1)


def _turned(sc, f):
    """the dragon scene's third transform (its monkey) turned by 0.05 f, as examples/dragon.js does every tick"""
    r = np.array(sc.arrays["rotation"], np.float32).reshape(-1, 2, 12).copy()
    c, s_ = np.cos(0.05 * f), np.sin(0.05 * f)
    R = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]]) * 2.0
    Ri = np.linalg.inv(R)
    for m, M in ((0, R), (1, Ri)):
        for col in range(3):
            r[2, m, 4 * col:4 * col + 3] = M[:, col]
    return r.reshape(-1)


def _lit(sc, f):
    lt = np.array(sc.arrays["lights"], np.float32).reshape(-1, 6).copy()
    lt[0, 0] += 0.4 * f
    lt[0, 3] *= 1.0 + 0.1 * (f % 3)
    return lt.reshape(-1)


@pytest.mark.parametrize("lanes", [2, 3])
def test_a_scene_that_moves_stays_with_the_server(hip, oracle, scenes, lanes):
    """transforms AND lights that change before every frame: after the first changed upload the server's launch takes them with every frame (ServerMail::blob, a
    version per frame in flight) and goes on over the uploads — ONE launch for all the frames that follow; every frame is its own flx_render with its own arrays,
    and the oracle's"""
    import copy
    sc = scenes("dragon")
    hip.update_scene(sc)
    hip.set_frame_lanes(lanes)
    hip.set_frame_chain(2)
    try:
        p = sc.frame_params(use_filter=0, width=640, height=360, tile=(8, 1, 2))
        N = 9
        hip.frame_begin(p)                                    # (the frame that follows update_scene itself goes to the lanes)
        hip.frame_end()
        kinds, got, moving = [], [], []
        for f in range(N):
            if hip.frames_in_flight() == lanes:
                got.append(hip.frame_end()[0].copy())
            if f >= 2:
                hip.update_transforms(_turned(sc, f), sc.arrays["shift"])
            if f >= 4:
                hip.update_primary_light_sources(_lit(sc, f))
            q = copy.copy(p)
            q.random_seed = float(f % 3)
            hip.frame_begin(q)
            kinds.append(hip.last_chained())
            moving.append(hip.server_moving())
        while hip.frames_in_flight():
            got.append(hip.frame_end()[0].copy())
        assert kinds == [3] * N, kinds
        assert moving == [False, False] + [True] * (N - 2), moving
        for f in range(N):
            hip.update_transforms(_turned(sc, f) if f >= 2 else sc.arrays["rotation"], sc.arrays["shift"])
            hip.update_primary_light_sources(_lit(sc, f) if f >= 4 else sc.arrays["lights"])
            q = copy.copy(p)
            q.random_seed = float(f % 3)
            want = hip.render(q)[0]
            assert bit_mismatches(got[f], want) == 0, f
            if f in (3, 8):
                moved = copy.copy(sc)
                moved.arrays = dict(sc.arrays, rotation=_turned(sc, f), lights=_lit(sc, f) if f >= 4 else sc.arrays["lights"])
                assert bit_mismatches(got[f], oracle.render(moved, q)[0]) == 0, f
        # another count of lights is another scene: the launch ends, the frame is right
        two = np.concatenate([_lit(sc, 1), _lit(sc, 2)])
        hip.update_primary_light_sources(two)
        hip.frame_begin(p)
        g = hip.frame_end()[0].copy()
        assert bit_mismatches(g, hip.render(p)[0]) == 0
    finally:
        hip.update_transforms(sc.arrays["rotation"], sc.arrays["shift"])
        hip.update_primary_light_sources(sc.arrays["lights"])
        hip.set_frame_lanes(2)


def test_a_rank_s_eighth_of_the_baseline_frame_with_the_scene_moving(hip, scenes):
    """BASELINE configs[2] at full size, tile 3 of 8 (what one of eight GPUs renders), three frames in flight, the monkey turning and the light moving before every
    frame: one launch takes them all; every frame equals its own flx_render with its own arrays"""
    import copy
    sc = scenes("dragon")
    hip.update_scene(sc)
    hip.set_frame_lanes(3)
    hip.set_frame_chain(2)
    try:
        p = sc.frame_params(use_filter=0, tile=(8, 3, 8))
        assert (p.width, p.height, p.samples, p.max_reflections) == (1920, 1080, 8, 4)
        N = 7
        got = []
        for f in range(N):
            if hip.frames_in_flight() == 3:
                got.append(hip.frame_end()[0].copy())
            hip.update_transforms(_turned(sc, f + 1), sc.arrays["shift"])
            hip.update_primary_light_sources(_lit(sc, f + 1))
            hip.frame_begin(p)
            assert hip.last_chained() == 3 and (f == 0 or hip.server_moving())
        while hip.frames_in_flight():
            got.append(hip.frame_end()[0].copy())
        for f in range(N):
            hip.update_transforms(_turned(sc, f + 1), sc.arrays["shift"])
            hip.update_primary_light_sources(_lit(sc, f + 1))
            assert bit_mismatches(got[f], hip.render(p)[0]) == 0, f
    finally:
        hip.update_transforms(sc.arrays["rotation"], sc.arrays["shift"])
        hip.update_primary_light_sources(sc.arrays["lights"])
        hip.set_frame_lanes(2)


def test_what_else_reads_the_arrays_beside_a_launch_that_takes_them_per_frame(hip, scenes):
    """uploads beside such a launch touch only the host's copies; the device's arrays follow when the launch ends — before a synchronous render with yet other
    transforms, before the per-pixel kernel's table of a filter frame, across flx_sync; and a context is destroyed with such frames in flight"""
    from flexlight_hip import capi
    sc = scenes("dragon")
    hip.update_scene(sc)
    hip.set_frame_lanes(3)
    hip.set_frame_chain(2)
    try:
        p = sc.frame_params(use_filter=0, width=640, height=360, tile=(8, 1, 2))
        for f in range(3):
            hip.update_transforms(_turned(sc, f + 1), sc.arrays["shift"])
            hip.frame_begin(p)
        assert hip.server_moving() and hip.frames_in_flight() == 3
        hip.sync()                                                       # flx_sync in mid-loop: the frames are complete, and handed out as usual
        got = [hip.frame_end()[0].copy() for _ in range(3)]
        for f in range(3):
            hip.update_transforms(_turned(sc, f + 1), sc.arrays["shift"])
            assert bit_mismatches(got[f], hip.render(p)[0]) == 0, f
        for f in range(2):
            hip.update_transforms(_turned(sc, 10 + f), sc.arrays["shift"])
            hip.frame_begin(p)
        hip.update_transforms(_turned(sc, 20), sc.arrays["shift"])       # (beside the launch: the host's copy only)
        want20 = hip.render(p)[0].copy()                                 # ends the launch; renders with the transforms of the LAST upload
        g = [hip.frame_end()[0].copy() for _ in range(2)]
        hip.update_scene(sc)
        hip.update_transforms(_turned(sc, 20), sc.arrays["shift"])
        assert bit_mismatches(want20, hip.render(p)[0]) == 0
        for f in range(2):
            hip.update_transforms(_turned(sc, 10 + f), sc.arrays["shift"])
            assert bit_mismatches(g[f], hip.render(p)[0]) == 0, f
        pf = sc.frame_params(use_filter=1, width=320, height=200)
        for f in range(3):
            hip.update_transforms(_turned(sc, 30 + f), sc.arrays["shift"])
            hip.frame_begin(p)
        a = hip.render(pf)[0].copy()                                     # the per-pixel kernel's table: made from the transforms of the last upload
        while hip.frames_in_flight():
            hip.frame_end()
        hip.set_angle_table(0)
        b = hip.render(pf)[0].copy()
        hip.set_angle_table(1)
        assert bit_mismatches(a, b) == 0
        other = capi.Context(0)
        other.update_scene(sc)
        other.set_frame_lanes(3)
        other.set_server_groups(other.device_info()[1] // 2)
        for f in range(3):
            other.update_transforms(_turned(sc, 40 + f), sc.arrays["shift"])
            other.frame_begin(p)
        assert other.server_moving()
        other.close()                                                    # with three such frames in flight
    finally:
        hip.update_transforms(sc.arrays["rotation"], sc.arrays["shift"])
        hip.set_frame_lanes(2)


def test_a_host_that_pauses_loses_nothing(served, scenes):
    """the launch waits for the host; after two seconds without a word it ends by itself (a safety net: the host normally says when to stop).  An application that
    pauses with frames in flight — all of them complete by then — must find them when it comes back, no error, and the loop goes on with a new launch"""
    import time
    sc = scenes("dragon")
    served.update_scene(sc)
    served.set_frame_lanes(3)
    ps = [moving(sc, f, width=480, height=272) for f in range(6)]
    want = [served.render(p)[0] for p in ps]
    for p in ps[:3]:
        served.frame_begin(p)
    time.sleep(2.6)
    got = [served.frame_end()[0] for _ in range(3)]
    for p in ps[3:]:
        served.frame_begin(p)
        assert served.last_chained() == 3
    got += [served.frame_end()[0] for _ in range(3)]
    for f in range(6):
        assert bit_mismatches(got[f], want[f]) == 0, f


def test_frames_the_server_does_not_take(served, scenes):
    """a scene of fewer than 129 entries: rendered the other ways, and right; RGBA8 frames through the server equal the ones quantised by the kernel of their own"""
    sc = scenes("cornell_obj")
    served.update_scene(sc)
    served.set_frame_lanes(2)
    ps = [moving(sc, f, width=320, height=200, samples=2, max_reflections=3) for f in range(3)]
    want = [served.render(p)[0] for p in ps]
    got, kinds = loop(served, ps, 2)
    assert kinds == [0, 0, 0]
    for f in range(3):
        assert bit_mismatches(got[f], want[f]) == 0
    sc = scenes("dragon")
    served.update_scene(sc)
    ps = [moving(sc, f, width=320, height=200) for f in range(3)]
    want8 = []
    for p in ps:
        served.frame_begin(p, rgba8=True)
        want8.append(served.frame_end()[0])
    served.set_frame_chain(0)
    for f, p in enumerate(ps):
        served.frame_begin(p, rgba8=True)
        assert np.array_equal(served.frame_end()[0], want8[f])
