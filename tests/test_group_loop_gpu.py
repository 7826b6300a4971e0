"""The device group's frame loop (flx_group_frame_begin / _end, csrc/flx_group.hip) and the frame targets under it (flx_frame_target_set).

Every context's frame server resolves its row strips straight into ONE image — pinned host memory or context 0's device memory — so a frame of the group
needs no exchange, no reassembly kernel and no copy, and the host never synchronises a stream (reference loop: modules/pathtracerWGL2.js:254-303).  On this
one-GPU box the group's contexts share the device and each server's launch takes half of the CUs (flx_debug_set_server_groups), so that both run at once as
they do on two GPUs.  Every frame must equal flx_render of ONE context bit for bit (tests/test_parity_gpu.py holds that against the oracle)."""
import ctypes

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


@pytest.fixture()
def pair():
    from flexlight_hip import capi
    g = capi.Group([0, 0])
    cus = g.context(0).device_info()[1]
    for r in range(2):
        g.context(r).set_server_groups(cus // 2)
    yield g
    g.close()


def run_loop(g, ps, lanes, **kw):
    got = []
    for p in ps:
        if g.frames_in_flight() == lanes:
            got.append(g.frame_end()[0])
        g.frame_begin(p, **kw)
    while g.frames_in_flight():
        got.append(g.frame_end()[0])
    return got


@pytest.mark.parametrize("lanes", [2, 3])
@pytest.mark.parametrize("shape", [dict(width=640, height=368), dict(width=500, height=264, samples=3)])
def test_group_loop_frames_equal_one_contexts_render(pair, hip, scenes, shape, lanes):
    """frames into pinned host memory: a camera that moves, a seed that changes; 368 rows = 23 strips of 16 (the last rank's share is shorter), 264 = 16.5"""
    sc = scenes("dragon")
    pair.update_scene(sc)
    hip.update_scene(sc)
    pair.set_frame_lanes(lanes)
    ps = [moving(sc, f, **shape) for f in range(7)]
    want = [hip.render(p)[0] for p in ps]
    got = run_loop(pair, ps, lanes, tile_rows=16)
    for f in range(7):
        assert got[f].shape == want[f].shape
        assert bit_mismatches(got[f], want[f]) == 0, "frame %d differs from one context's render" % f
    assert pair.context(0).last_chained() == 3 and pair.context(1).last_chained() == 3      # both went through their servers


def test_group_loop_into_device_memory_and_across_a_change_of_shape(pair, hip, scenes):
    """FLX_FRAME_DEVICE: the image is in context 0's memory; a change of the frame's size while frames are in flight: they complete in the images they were begun in"""
    hiprt = ctypes.CDLL("libamdhip64.so")
    sc = scenes("dragon")
    pair.update_scene(sc)
    hip.update_scene(sc)
    a = [moving(sc, f, width=480, height=272) for f in range(4)]
    b = [moving(sc, f, width=320, height=208) for f in range(3)]
    want = [hip.render(p)[0] for p in a + b]
    got = []

    def take():
        ptr, _ = pair.frame_end()
        h, w = shapes.pop(0)
        out = np.empty((h, w, 4), np.float32)                # (device -> host: a DMA copy; a copy KERNEL would wait for CUs the two launches leave none of)
        assert hiprt.hipMemcpy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(h * w * 16), 2) == 0
        got.append(out)
    shapes = []
    for p in a + b:
        if pair.frames_in_flight() == 3:
            take()
        pair.frame_begin(p, tile_rows=8, device=True)
        shapes.append((p.height, p.width))
    while pair.frames_in_flight():
        take()
    for f in range(7):
        assert bit_mismatches(got[f], want[f]) == 0, f


def test_frames_the_server_does_not_take_go_through_the_groups_render(pair, hip, scenes):
    """a filter frame and a scene of fewer than 129 entries between served frames: rendered at flx_group_frame_begin, handed out in order"""
    sc = scenes("dragon")
    pair.update_scene(sc)
    hip.update_scene(sc)
    p0, p1 = moving(sc, 0, width=320, height=208), moving(sc, 1, width=320, height=208)
    pf = sc.frame_params(width=320, height=208, use_filter=1)
    want = [hip.render(p)[0] for p in (p0, pf, p1)]
    pair.frame_begin(p0)
    pair.frame_begin(pf)
    pair.frame_begin(p1)
    got = [pair.frame_end()[0] for _ in range(3)]
    for f in range(3):
        assert bit_mismatches(got[f], want[f]) == 0, f
    small = scenes("cornell_obj")
    pair.update_scene(small)
    hip.update_scene(small)
    ps = [moving(small, f, width=256, height=200, samples=2, max_reflections=3) for f in range(3)]
    want = [hip.render(p)[0] for p in ps]
    got = run_loop(pair, ps, 3)
    for f in range(3):
        assert bit_mismatches(got[f], want[f]) == 0, f


def test_a_frame_target_of_the_callers(hip, scenes):
    """flx_frame_target_set on one context: the four ranks' strips of a frame, rendered one rank after the other, complete the caller's images"""
    import torch
    sc = scenes("dragon")
    hip.update_scene(sc)
    hip.set_frame_lanes(3)
    hip.set_frame_chain(3)
    W, H = 448, 264                                           # 33 strips of 8 rows over 4 ranks: 9, 8, 8, 8
    try:
        ps = [moving(sc, f, width=W, height=H) for f in range(3)]
        want = [hip.render(p)[0] for p in ps]
        images = torch.zeros((3, H, W, 4), dtype=torch.float32, device="cuda")
        hip.frame_target_set([images[i].data_ptr() for i in range(3)])
        for r in range(4):
            for f in range(3):
                hip.frame_begin(moving(sc, f, width=W, height=H, tile=(8, r, 4)), device=True)
                assert hip.frame_target_index() == f
            for f in range(3):
                ptr, _ = hip.frame_end()
                assert ptr == images[f].data_ptr()
        torch.cuda.synchronize()
        for f in range(3):
            assert bit_mismatches(images[f].cpu().numpy(), want[f]) == 0, f
        # a frame the server does not take is refused while a target is set
        from flexlight_hip import capi
        with pytest.raises(capi.FlexLightHipError):
            hip.frame_begin(sc.frame_params(width=W, height=H, use_filter=1), device=True)
    finally:
        hip.frame_target_set([])
        hip.set_frame_lanes(2)
        hip.set_frame_chain(2)


def test_group_loop_hands_out_the_canvas_bytes(pair, hip, oracle, scenes):
    """FLX_FRAME_RGBA8: the group's image is the canvas' RGBA8 in pinned host memory, every context's server quantising the tiles it resolves (a quarter of the bytes
    every GPU writes); frames the servers do not take (a filter frame, a small scene) come out as the same bytes the other way"""
    sc = scenes("dragon")
    hip.update_scene(sc)
    pair.update_scene(sc)
    ps = [moving(sc, f, width=640, height=368) for f in range(7)]
    want = [oracle.present(hip.render(p)[0]) for p in ps]
    for lanes in (3, 2):
        pair.set_frame_lanes(lanes)
        got = run_loop(pair, ps, lanes, rgba8=True)
        for f in range(7):
            assert got[f].dtype == np.uint8 and np.array_equal(got[f], want[f]), (lanes, f)
    # float frames after it: the target is made again
    pair.set_frame_lanes(3)
    gotf = run_loop(pair, ps[:3], 3)
    for f in range(3):
        assert bit_mismatches(gotf[f], hip.render(ps[f])[0]) == 0
    # a filter frame between frames of the loop
    flt = sc.frame_params(width=640, height=368, use_filter=1, samples=1, max_reflections=2)
    pair.frame_begin(ps[0], rgba8=True)
    pair.frame_begin(flt, rgba8=True)
    a = pair.frame_end()[0]
    b = pair.frame_end()[0]
    assert np.array_equal(a, want[0]) and np.array_equal(b, oracle.present(hip.render(flt)[0]))


@pytest.mark.parametrize("servers", [True, False])
@pytest.mark.parametrize("rgba8", [False, True])
def test_group_loop_with_a_scene_that_moves(pair, hip, oracle, scenes, rgba8, servers):
    """the transforms change before every frame (examples/dragon.js).  servers: the contexts' launches take the transforms with every frame and go on (flx_server.hip:
    VER) — the frames still complete ONE image with no exchange.  Not servers (flx_set_server_moving_scenes(0): what a scene whose arrays do not fit a post gets): such
    frames do not go to the servers — floats run on the contexts' two lanes and their strips are copied into the frame's image when it is taken, the canvas' bytes go
    through flx_group_render_rgba8 — and the loop goes back to the servers when the scene stands still.  Every frame equals one context's render with the same arrays"""
    sc = scenes("dragon")
    pair.update_scene(sc)
    hip.update_scene(sc)
    for r in range(2):
        pair.context(r).set_server_moving_scenes(servers)
    rot0 = np.array(sc.arrays["rotation"], np.float32).reshape(-1, 2, 12)

    def arrays(f):
        r = rot0.copy()
        c, s_ = np.cos(0.07 * f), np.sin(0.07 * f)
        R = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]]) * 2.0
        Ri = np.linalg.inv(R)
        for m, M in ((0, R), (1, Ri)):
            for col in range(3):
                r[2, m, 4 * col:4 * col + 3] = M[:, col]
        return r.reshape(-1)
    ps = [moving(sc, f, width=640, height=368) for f in range(9)]
    moves = [f for f in range(9) if 2 <= f < 7]                    # frames 2 .. 6 follow an upload of changed transforms
    pair.set_frame_lanes(3)
    got = []
    try:
        for f, p in enumerate(ps):
            if pair.frames_in_flight() == 3:
                got.append(pair.frame_end()[0])
            if f in moves:
                pair.update_transforms(arrays(f), sc.arrays["shift"])
            pair.frame_begin(p, rgba8=rgba8)
            if f == 5:
                assert [pair.context(r).server_moving() for r in range(2)] == [servers, servers]
        while pair.frames_in_flight():
            got.append(pair.frame_end()[0])
        last = None
        for f, p in enumerate(ps):
            if f in moves:
                last = arrays(f)
            hip.update_transforms(last if last is not None else sc.arrays["rotation"], sc.arrays["shift"])
            want = hip.render(p)[0]
            if rgba8:
                assert np.array_equal(got[f], oracle.present(want)), f
            else:
                assert bit_mismatches(got[f], want) == 0, f
    finally:
        hip.update_transforms(sc.arrays["rotation"], sc.arrays["shift"])
        pair.update_transforms(sc.arrays["rotation"], sc.arrays["shift"])
