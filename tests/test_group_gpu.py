"""Several GPUs through the library (include/flexlight_hip.h: flx_group_*, flx_comm_*): the frame split into
row strips over the contexts of a group, gathered and put back in image order inside the library, must equal the frame one
context renders, bit for bit.  The test box has ONE GPU: groups name device 0 several times (strips exchanged by device
copies); the RCCL calls themselves — ncclGetUniqueId, ncclCommInitRank, ncclAllGather on the context's stream — run with a
communicator of one rank."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _moved(sc, p, i):
    from flexlight_hip.scene_io import view_matrix
    cam = sc.meta["camera"]
    q = type(p).from_buffer_copy(p)
    q.camera[:] = [cam["x"] + 0.35 * i, cam["y"] + 0.1 * i, cam["z"] - 0.2 * i]
    q.view_matrix[:] = view_matrix(cam["fx"] + 0.07 * i, cam["fy"] - 0.03 * i, cam["fov"], p.width, p.height).tolist()
    q.random_seed = float(i % 3)
    return q


@pytest.mark.parametrize("name,w,h,spp,bounces,ranks,tile_rows", [
    ("dragon", 320, 180, 2, 4, 2, 8),
    ("dragon", 200, 117, 2, 3, 3, 8),          # ragged: the last strip is cut, the ranks own different numbers of rows
    ("theater", 160, 90, 2, 4, 4, 5),
    ("cornell", 64, 24, 1, 2, 5, 8),           # more ranks than strips: two contexts own nothing
])
def test_group_frame_equals_single_context(hip, scenes, name, w, h, spp, bounces, ranks, tile_rows):
    from flexlight_hip import capi
    sc = scenes(name)
    hip.update_scene(sc)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
    want, want_cnt, _ = hip.render(p, counters=True)
    with capi.Group([0] * ranks) as g:
        assert g.size == ranks and not g.uses_rccl
        g.update_scene(sc)
        got, cnt = g.render(p, tile_rows=tile_rows, counters=True)
        assert np.array_equal(got[0], want, equal_nan=True)
        assert cnt == want_cnt                                     # the strips' work adds up to the frame's
        frames = [_moved(sc, p, i) for i in range(3)]              # a batch of different frames through the same gather
        batch, _ = g.render(frames, tile_rows=tile_rows)
        for i, q in enumerate(frames):
            assert np.array_equal(batch[i], hip.render(q)[0], equal_nan=True), "frame %d" % i


@pytest.mark.parametrize("name,w,h,spp,bounces,ranks", [("cornell_obj", 160, 96, 2, 3, 3), ("dragon", 192, 120, 1, 3, 4)])
def test_group_filter_frame(hip, scenes, name, w, h, spp, bounces, ranks):
    """filter on: the strips' five RGBA8 render targets are gathered, the chain runs on the whole frame on context 0"""
    from flexlight_hip import capi
    sc = scenes(name)
    hip.update_scene(sc)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=1)
    want, _, _ = hip.render(p)
    with capi.Group([0] * ranks) as g:
        g.update_scene(sc)
        got, _ = g.render(p, tile_rows=8)
        assert np.array_equal(got[0], want, equal_nan=True)
        with pytest.raises(capi.FlexLightHipError, match="one by one"):
            g.render([p, p])
        t = type(p).from_buffer_copy(p)
        t.use_filter, t.is_temporal = 0, 1
        with pytest.raises(capi.FlexLightHipError, match="temporal"):
            g.render(t)


@pytest.mark.parametrize("name,w,h,spp,bounces", [("dragon", 3840, 2160, 8, 4), ("theater", 1920, 1080, 16, 6)], ids=["dragon_4k", "theater_1080p"])
def test_eight_rank_strips_of_the_multi_gpu_configs(hip, scenes, name, w, h, spp, bounces):
    """BASELINE.json configs[3] and configs[4] — the 8-GPU workloads — at full size as the strips of eight contexts, gathered by
    the library, against the whole frame on one context (which tests/test_parity_gpu.py holds against the oracle)."""
    from flexlight_hip import capi
    sc = scenes(name)
    hip.update_scene(sc)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
    want, want_cnt, _ = hip.render(p, counters=True)
    with capi.Group([0] * 8) as g:
        g.update_scene(sc)
        got, cnt = g.render(p, tile_rows=8, counters=True)
    assert np.array_equal(got[0], want, equal_nan=True)
    assert cnt == want_cnt


def test_rccl_gather_with_a_communicator_of_one_rank(hip, scenes):
    """the multi-process entry points on the one GPU there is: id, ncclCommInitRank, ncclAllGather + reassembly on the stream"""
    import torch
    from flexlight_hip import capi
    sc = scenes("dragon")
    ctx = capi.Context(0)
    try:
        ctx.update_scene(sc)
        p = sc.frame_params(width=256, height=150, samples=2, max_reflections=3, use_filter=0, tile=(8, 0, 1))
        out = torch.zeros((3, 150, 256, 4), dtype=torch.float32, device="cuda")
        with pytest.raises(capi.FlexLightHipError, match="no communicator"):
            ctx.render_gathered_device([p], out.data_ptr())
        ctx.comm_init_rank(capi.comm_unique_id(), 1, 0)
        with pytest.raises(capi.FlexLightHipError, match="already"):
            ctx.comm_init_rank(capi.comm_unique_id(), 1, 0)
        frames = [_moved(sc, p, i) for i in range(3)]
        ctx.render_gathered_device(frames, out.data_ptr())
        ctx.sync()
        got = out.cpu().numpy()
        for i, q in enumerate(frames):
            q.tile_rows = q.tile_index = q.tile_count = 0
            assert np.array_equal(got[i], hip_render(hip, sc, q), equal_nan=True), "frame %d" % i
        bad = type(p).from_buffer_copy(p)
        bad.tile_index = 1
        with pytest.raises(capi.FlexLightHipError, match="rank"):
            ctx.render_gathered_device([bad], out.data_ptr())
        f = sc.frame_params(width=256, height=150, samples=1, max_reflections=2, use_filter=1, tile=(8, 0, 1))
        ctx.render_gathered_device([f], out.data_ptr())
        ctx.sync()
        f.tile_rows = f.tile_index = f.tile_count = 0
        assert np.array_equal(out[0].cpu().numpy(), hip_render(hip, sc, f), equal_nan=True)
        ctx.comm_destroy()
    finally:
        ctx.close()


def test_gather_to_root_on_repeated_devices_and_with_one_rank(hip, scenes):
    """only the presenting rank receives the strips (flx_group_set_gather / flx_render_gathered_root_device: ncclSend + ncclRecv
    inside one ncclGroupStart / End): the same frame as the all-gather, radiance and filter frames, groups on one device (copies
    into context 0 alone) and a communicator of one rank (the RCCL calls themselves)"""
    import torch
    from flexlight_hip import capi
    sc = scenes("dragon")
    hip.update_scene(sc)
    p = sc.frame_params(width=224, height=126, samples=2, max_reflections=3, use_filter=0)
    f = sc.frame_params(width=224, height=126, samples=1, max_reflections=2, use_filter=1)
    want, want_cnt, _ = hip.render(p, counters=True)
    want_f = hip.render(f)[0]
    with capi.Group([0] * 3) as g:
        g.update_scene(sc)
        for _ in range(2):                                    # the default: only context 0 receives; twice: the send buffers are reused while only context 0 copies
            got, cnt = g.render(p, tile_rows=8, counters=True)
            assert np.array_equal(got[0], want, equal_nan=True) and cnt == want_cnt
        assert np.array_equal(g.render(f, tile_rows=8)[0][0], want_f, equal_nan=True)
        g.set_gather(False)
        assert np.array_equal(g.render(p, tile_rows=8)[0][0], want, equal_nan=True)
    ctx = capi.Context(0)
    try:
        ctx.update_scene(sc)
        assert ctx.comm_count() == 0
        ctx.comm_init_rank(capi.comm_unique_id(), 1, 0)
        assert ctx.comm_count() == 1                          # ncclCommCount of the communicator the frames go through
        out = torch.zeros((126, 224, 4), dtype=torch.float32, device="cuda")
        pt = sc.frame_params(width=224, height=126, samples=2, max_reflections=3, use_filter=0, tile=(8, 0, 1))
        ctx.render_gathered_root_device([pt], 0, out.data_ptr())
        ctx.sync()
        assert np.array_equal(out.cpu().numpy(), want, equal_nan=True)
        ft = sc.frame_params(width=224, height=126, samples=1, max_reflections=2, use_filter=1, tile=(8, 0, 1))
        ctx.render_gathered_root_device([ft], 0, out.data_ptr())
        ctx.sync()
        assert np.array_equal(out.cpu().numpy(), want_f, equal_nan=True)
        with pytest.raises(capi.FlexLightHipError, match="root"):
            ctx.render_gathered_root_device([pt], 1, out.data_ptr())
        with pytest.raises(capi.FlexLightHipError, match="root"):
            ctx.render_gathered_root_device([pt], -1, out.data_ptr())
    finally:
        ctx.close()


def test_frames_gathered_as_the_canvas_rgba8(hip, oracle, scenes):
    """the presenter wants RGBA8 (round-3 review, item 8): every rank quantises its strips where it traced them and a quarter of the bytes is exchanged —
    flx_group_render_rgba8 on repeated devices (ragged strips, more ranks than some frames have strips, both gathers, a batch of moved cameras) and
    flx_render_gathered_rgba8_device with a communicator of one rank (all-gather and to the root); the bytes are flx_present's (= the oracle's) of the
    float frame one context renders"""
    import torch
    from flexlight_hip import capi
    sc = scenes("dragon")
    hip.update_scene(sc)
    p = sc.frame_params(width=224, height=126, samples=2, max_reflections=3, use_filter=0)
    frames = [_moved(sc, p, i) for i in range(3)]
    want = [oracle.present(hip_render(hip, sc, q)) for q in frames]
    assert np.array_equal(want[0], hip.present(hip_render(hip, sc, frames[0])))
    for ranks, tile_rows in ((3, 8), (5, 16), (8, 8)):
        with capi.Group([0] * ranks) as g:
            g.update_scene(sc)
            got = g.render_rgba8(frames, tile_rows=tile_rows)
            for i in range(3):
                assert np.array_equal(got[i], want[i]), (ranks, tile_rows, i)
            g.set_gather(False)
            assert np.array_equal(g.render_rgba8(frames[1], tile_rows=tile_rows)[0], want[1])
            assert np.array_equal(g.render(frames[1], tile_rows=tile_rows)[0][0], hip_render(hip, sc, frames[1]), equal_nan=True)      # (the float path after it: its buffers are its own)
            with pytest.raises(capi.FlexLightHipError, match="filter"):
                g.render_rgba8(sc.frame_params(width=224, height=126, samples=1, max_reflections=2, use_filter=1), tile_rows=tile_rows)
    ctx = capi.Context(0)
    try:
        ctx.update_scene(sc)
        ctx.comm_init_rank(capi.comm_unique_id(), 1, 0)
        out = torch.zeros((3, 126, 224, 4), dtype=torch.uint8, device="cuda")
        tiled = []
        for q in frames:
            t = type(q).from_buffer_copy(q)
            t.tile_rows, t.tile_index, t.tile_count = 8, 0, 1
            tiled.append(t)
        for root in (-1, 0):
            out.zero_()
            ctx.render_gathered_rgba8_device(tiled, root, out.data_ptr())
            ctx.sync()
            got = out.cpu().numpy()
            for i in range(3):
                assert np.array_equal(got[i], want[i]), (root, i)
    finally:
        ctx.close()


def test_frame_loop_over_a_communicator(hip, scenes):
    """flx_frame_begin_gathered: the frames of a camera move alternate between the two lanes, each lane gathering over its own
    communicator (the second is an ncclCommSplit of the first); every frame taken equals its own render — with a communicator of
    one rank, which is what one GPU can run"""
    import torch  # noqa: F401  (device memory is read back through torch below)
    from flexlight_hip import capi
    sc = scenes("dragon")
    ctx = capi.Context(0)
    try:
        ctx.update_scene(sc)
        p = sc.frame_params(width=240, height=136, samples=2, max_reflections=3, use_filter=0, tile=(8, 0, 1))
        with pytest.raises(capi.FlexLightHipError, match="no communicator"):
            ctx.frame_begin_gathered(p)
        ctx._pending.clear()
        ctx.comm_init_rank(capi.comm_unique_id(), 1, 0)
        frames = [_moved(sc, p, i) for i in range(5)]
        got = []

        import ctypes as C
        hipMemcpy = C.CDLL("libamdhip64.so").hipMemcpy
        hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

        def take():
            ptr, ms = ctx.frame_end()                          # the gathered whole frame, in device memory
            host = np.empty((136, 240, 4), np.float32)
            assert hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), host.nbytes, 2) == 0      # hipMemcpyDeviceToHost
            assert ms > 0
            got.append(host)
        for q in frames:
            ctx.frame_begin_gathered(q, root=0)
            if ctx.frames_in_flight() == 2:
                take()
        while ctx.frames_in_flight():
            take()
        for i, q in enumerate(frames):
            q.tile_rows = q.tile_index = q.tile_count = 0
            assert np.array_equal(got[i], hip_render(hip, sc, q), equal_nan=True), "frame %d" % i
        ctx.comm_destroy()
    finally:
        ctx.close()


def hip_render(hip, sc, p):
    hip.update_scene(sc)
    return hip.render(p)[0]


def test_group_arguments():
    from flexlight_hip import capi
    with pytest.raises(capi.FlexLightHipError, match="device"):
        capi.Group([0, 99])
    with pytest.raises(capi.FlexLightHipError):
        capi.Group([])
