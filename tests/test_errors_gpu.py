"""Error paths and corner cases of the C ABI on the GPU: everything is reported through the status code + flx_last_error,
nothing throws or aborts, and the context stays usable afterwards."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_empty_tile_share_through_the_device_entry_points(hip, scenes):
    """tile_count greater than the number of strips: a rank without a strip renders nothing — FLX_OK from the *_device entry
    points too (bench.py calls only those), and the frame events are recorded"""
    import torch
    sc = scenes("cornell")
    hip.update_scene(sc)
    out = torch.zeros((64, 4), dtype=torch.float32, device="cuda")
    p = sc.frame_params(width=64, height=16, samples=1, max_reflections=1, use_filter=0, tile=(8, 5, 8))
    assert hip.tile_row_count(p) == 0
    hip.render_device(p, out.data_ptr())
    hip.render_batch_device([p, p], out.data_ptr())
    assert hip.last_frame_ms()[0] >= 0
    f = sc.frame_params(width=64, height=16, samples=1, max_reflections=1, use_filter=1, tile=(8, 5, 8))
    hip.render_planes_device(f, out.data_ptr())
    hip.sync()
    assert float(out.abs().sum()) == 0.0
    got, _, _ = hip.render(p)
    assert got.shape == (0, 64, 4)


def test_two_contexts_on_one_device_render_concurrently(hip, scenes):
    """a second context on the same GPU while the first has frames in flight: separate streams and workspaces"""
    import torch
    from flexlight_hip import capi
    sc, sc2 = scenes("dragon"), scenes("cornell")
    hip.update_scene(sc)
    p = sc.frame_params(width=480, height=270, samples=4, max_reflections=4, use_filter=0)
    p2 = sc2.frame_params(width=128, height=96, samples=2, max_reflections=3, use_filter=0)
    want = hip.render(p)[0]
    other = capi.Context(0)
    try:
        other.update_scene(sc2)
        want2 = other.render(p2)[0]
        a = torch.zeros((3, 270, 480, 4), dtype=torch.float32, device="cuda")
        for i in range(3):
            hip.render_device(p, a[i].data_ptr())          # enqueued, not waited for
        got2 = other.render(p2)[0]                         # the other context renders meanwhile
        hip.sync()
        assert np.array_equal(got2, want2, equal_nan=True)
        for i in range(3):
            assert np.array_equal(a[i].cpu().numpy(), want, equal_nan=True)
    finally:
        other.close()


def test_a_batch_whose_records_do_not_fit_is_refused_with_advice(hip, scenes):
    """32 frames of 8192 x 4320 x 8 samples = 9e9 paths do not even index; 32 x 4K x 8 samples need ~400 GB of path records:
    refused before anything is allocated, the message says what to do, and the context renders on"""
    from flexlight_hip import capi
    sc = scenes("dragon")
    hip.update_scene(sc)
    small = sc.frame_params(width=160, height=90, samples=2, max_reflections=3, use_filter=0)
    want = hip.render(small)[0]
    huge = sc.frame_params(width=8192, height=4320, samples=8, max_reflections=4, use_filter=0)
    with pytest.raises(capi.FlexLightHipError, match="too large"):
        hip.render_batch_device([huge] * 32, 0x1000)
    big = sc.frame_params(width=3840, height=2160, samples=8, max_reflections=4, use_filter=0)
    with pytest.raises(capi.FlexLightHipError, match="fewer frames per batch"):
        hip.render_batch_device([big] * 32, 0x1000)          # (refused before the output pointer is touched)
    assert np.array_equal(hip.render(small)[0], want, equal_nan=True)


def test_set_stream_of_another_device_is_refused(hip):
    import torch
    from flexlight_hip import capi
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: a stream of another device cannot be made on this box")
    with torch.cuda.device(1):
        s = torch.cuda.Stream()
    with pytest.raises(capi.FlexLightHipError, match="belongs to device"):
        hip.set_stream(s.cuda_stream)


def test_dynamic_uploads_reuse_their_buffers(hip, oracle, scenes):
    """transforms and lights re-uploaded every frame (pathtracerWGL2.js:258-262, 361-365) go to persistent device buffers in
    stream order: three frames in flight with three different light arrays each keep their own lights"""
    import torch
    sc = scenes("cornell")
    hip.update_scene(sc)
    p = sc.frame_params(width=96, height=64, samples=2, max_reflections=2, use_filter=0)
    lights = [sc.arrays["lights"].copy() for _ in range(3)]
    for i, l in enumerate(lights):
        l.reshape(-1, 6)[:, 0] += 1.5 * i
        l.reshape(-1, 6)[:, 3] *= (1.0 + 0.5 * i)
    out = torch.zeros((3, 64, 96, 4), dtype=torch.float32, device="cuda")
    for i in range(3):
        hip.update_primary_light_sources(lights[i])
        hip.update_transforms(sc.arrays["rotation"], sc.arrays["shift"])
        hip.render_device(p, out[i].data_ptr())              # no wait between the frames
    hip.sync()
    got = out.cpu().numpy()
    import copy
    for i in range(3):
        sci = copy.copy(sc)
        sci.arrays = dict(sc.arrays, lights=lights[i])
        want, _, _ = oracle.render(sci, p)
        assert np.array_equal(got[i], want, equal_nan=True), "frame %d" % i
    assert not np.array_equal(got[0], got[2])
    hip.update_scene(sc)


def test_frame_loop_two_frames_in_flight(hip, oracle, scenes):
    """flx_frame_begin / flx_frame_end: frames enqueued ahead of the host, taken in order out of pinned memory — float radiance and
    the canvas' RGBA8 —, a third begin refused while two are in flight"""
    from flexlight_hip import capi
    sc = scenes("cornell")
    hip.update_scene(sc)
    p = sc.frame_params(width=160, height=96, samples=2, max_reflections=3, use_filter=0)
    q = type(p).from_buffer_copy(p)
    q.camera[0] += 1.5
    f = sc.frame_params(width=160, height=96, samples=2, max_reflections=3, use_filter=1)
    want_p, want_q, want_f = hip.render(p)[0], hip.render(q)[0], hip.render(f)[0]
    with pytest.raises(capi.FlexLightHipError, match="no frame in flight"):
        hip.frame_end()
    hip.frame_begin(p)
    hip.frame_begin(q, rgba8=True)
    assert hip.frames_in_flight() == 2
    with pytest.raises(capi.FlexLightHipError, match="two frames are in flight"):
        hip.frame_begin(p)
    a, ms = hip.frame_end()
    assert np.array_equal(a, want_p, equal_nan=True) and ms > 0
    hip.frame_begin(f)                                     # the slot of the frame just taken
    b, _ = hip.frame_end()
    assert b.dtype == np.uint8 and np.array_equal(b, oracle.present(want_q))
    c, _ = hip.frame_end()
    assert np.array_equal(c, want_f, equal_nan=True)
    assert hip.frames_in_flight() == 0
    # one lane instead of two: the same frames
    hip.set_frame_lanes(1)
    try:
        hip.frame_begin(p)
        hip.frame_begin(q)
        assert np.array_equal(hip.frame_end()[0], want_p, equal_nan=True) and np.array_equal(hip.frame_end()[0], want_q, equal_nan=True)
    finally:
        hip.set_frame_lanes(2)


def test_frame_loop_lanes_keep_their_own_lights(hip, oracle, scenes):
    """two frames in flight on two lanes (streams + workspaces): lights and transforms uploaded between the begins belong to the
    frame begun after them — each of six frames with a different light array equals the oracle's, whichever lane rendered it —
    and a frame left in device memory (FLX_FRAME_DEVICE) is the same frame"""
    import copy
    import torch
    sc = scenes("cornell")
    hip.update_scene(sc)
    p = sc.frame_params(width=128, height=80, samples=2, max_reflections=3, use_filter=0)
    got, lights = [], []
    for i in range(6):
        l = sc.arrays["lights"].copy()
        l.reshape(-1, 6)[:, 0] += 0.8 * i
        l.reshape(-1, 6)[:, 3] *= (1.0 + 0.3 * i)
        lights.append(l)
        hip.update_primary_light_sources(l)
        hip.update_transforms(sc.arrays["rotation"], sc.arrays["shift"])
        hip.frame_begin(p, device=(i == 5))
        if hip.frames_in_flight() == 2:
            got.append(hip.frame_end()[0])
    while hip.frames_in_flight():
        got.append(hip.frame_end()[0])
    assert len(got) == 6 and isinstance(got[5], int)
    dev = torch.zeros((80, 128, 4), dtype=torch.float32, device="cuda")
    import ctypes
    ctypes.memmove  # (no host copy of a device pointer: read it through torch)
    hip.sync()
    last = torch.empty_like(dev)
    torch.cuda.synchronize()
    # the device pointer stays valid until the second begin after its frame: copy it out with a device-to-device copy
    from flexlight_hip import capi
    import ctypes as C
    hipMemcpy = C.CDLL("libamdhip64.so").hipMemcpy
    hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hipMemcpy(C.c_void_p(last.data_ptr()), C.c_void_p(got[5]), last.numel() * 4, 3) == 0      # hipMemcpyDeviceToDevice
    got[5] = last.cpu().numpy()
    for i in range(6):
        sci = copy.copy(sc)
        sci.arrays = dict(sc.arrays, lights=lights[i])
        want, _, _ = oracle.render(sci, p)
        assert np.array_equal(got[i], want, equal_nan=True), "frame %d" % i
    hip.update_scene(sc)


def test_a_scene_change_between_frames_in_flight(hip, oracle, scenes):
    """flx_scene_upload / flx_atlas_upload while frames are in flight on BOTH lanes of the frame loop: the static scene arrays are
    shared between the lanes, so the upload waits for the second lane's frame (which must finish on the old scene) and the next
    frame of either lane sees the new arrays complete — small scenes, whose uploads go through the asynchronous staging ring, in
    both directions (a larger scene into the smaller one's buffers and back).  Every frame equals the oracle's frame of the scene
    it was begun on."""
    a, b = scenes("cornell"), scenes("theater")                 # 48 and 32 entries: staged copies; different atlases
    pa = a.frame_params(width=160, height=96, samples=2, max_reflections=3, use_filter=0)
    pb = b.frame_params(width=160, height=96, samples=2, max_reflections=3, use_filter=0)
    want = {"a": oracle.render(a, pa)[0], "b": oracle.render(b, pb)[0]}
    hip.update_scene(a)
    hip.frame_begin(pa)
    hip.frame_begin(pa)                                           # second lane
    assert np.array_equal(hip.frame_end()[0], want["a"], equal_nan=True)
    order = []
    for k in range(6):                                            # one frame stays in flight across every scene change
        sc, p, key = (b, pb, "b") if k % 2 == 0 else (a, pa, "a")
        hip.update_scene(sc)                                      # the frame in flight was begun on the other scene
        hip.frame_begin(p)
        order.append(key)
        got = hip.frame_end()[0]
        prev = "a" if k == 0 else order[k - 1]
        assert np.array_equal(got, want[prev], equal_nan=True), "frame before change %d" % k
    assert np.array_equal(hip.frame_end()[0], want[order[-1]], equal_nan=True)
    assert hip.frames_in_flight() == 0
