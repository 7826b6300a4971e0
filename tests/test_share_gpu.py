"""One process per GPU without a collective (csrc/flx_share.hip): the ranks' frame servers complete ONE image in the root rank's device memory, which the
other ranks map through hipIpc; the ranks tell each other about frames through a page of POSIX shared memory.

Rehearsed here with two PROCESSES on the one GPU of the box, each server's launch taking half of the CUs (flx_debug_set_server_groups) so that both run at
once as they do on two GPUs; the stores of the second process go through its IPC mapping of the first one's allocation.  Every frame the root hands out must
equal flx_render of one context bit for bit (tests/test_parity_gpu.py holds that against the oracle)."""
import ctypes
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, FRAMES, LANES, RANKS = 640, 368, 7, 3, 2


def _moving(sc, f, **kw):
    p = sc.frame_params(use_filter=0, width=W, height=H, **kw)
    p.camera[0] += 0.05 * f
    p.camera[2] -= 0.03 * f
    p.random_seed = float(f % 4)
    return p


def _turned(sc, f):
    """the dragon scene's third transform (its monkey) turned, as examples/dragon.js does every tick"""
    r = np.array(sc.arrays["rotation"], np.float32).reshape(-1, 2, 12).copy()
    c, s_ = np.cos(0.06 * f), np.sin(0.06 * f)
    R = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]]) * 2.0
    Ri = np.linalg.inv(R)
    for m, M in ((0, R), (1, Ri)):
        for col in range(3):
            r[2, m, 4 * col:4 * col + 3] = M[:, col]
    return r.reshape(-1)


def _rank_main(rank, conn, tile_rows, scene_name="dragon", moves=False, in_flight=LANES):
    """one rank: rank 0 creates the share and sends the handle up; the others receive it from the parent"""
    try:
        from flexlight_hip import capi
        from flexlight_hip.scene_io import Scene
        sc = Scene.golden(scene_name)
        ctx = capi.Context(0)
        ctx.update_scene(sc)
        ctx.set_server_groups(ctx.device_info()[1] // RANKS)
        want = None
        if rank == 0:
            want = []
            for f in range(FRAMES):                                           # (before any launch of the other rank holds CUs)
                if moves and f >= 2:
                    ctx.update_transforms(_turned(sc, f), sc.arrays["shift"])
                want.append(ctx.render(_moving(sc, f))[0])
            ctx.update_transforms(sc.arrays["rotation"], sc.arrays["shift"])
            conn.send(ctx.share_create(W, H, LANES, RANKS, 0))
            assert conn.recv() == "go"
        else:
            ctx.share_join(conn.recv(), rank)
            conn.send("joined")
            assert conn.recv() == "go"
        try:
            hiprt = ctypes.CDLL("libamdhip64.so")
        except OSError:
            hiprt = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
        bad, inflight, ptrs = [], [], []

        def take():
            ptr, _ = ctx.frame_end_shared()
            f = inflight.pop(0)
            if rank == 0:
                if in_flight == 1:                                            # the other rank may begin (and finish) the next frame before this one is read: it must not be in this image
                    import time
                    time.sleep(0.03)
                    ptrs.append(ptr)
                out = np.empty((H, W, 4), np.float32)
                assert hiprt.hipMemcpy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(H * W * 16), 2) == 0
                if not np.array_equal(out.view(np.uint32), want[f].view(np.uint32)):
                    bad.append(f)
            else:
                assert ptr is None
        for f in range(FRAMES):
            if len(inflight) == in_flight:
                take()
            if moves and f >= 2:                                              # every rank turns the monkey before the frame: its launch takes the transforms with the frame and goes on
                ctx.update_transforms(_turned(sc, f), sc.arrays["shift"])
            ctx.frame_begin_shared(_moving(sc, f, tile=(tile_rows, rank, RANKS)))
            assert ctx.last_chained() == (3 if scene_name == "dragon" else 0)      # (a scene of <= 128 entries: the lanes, its strips copied into the image)
            assert ctx.server_moving() or not (moves and f >= 2)            # (rank 0's scene has moved before the loop: its first launch is of that kind already)
            inflight.append(f)
        while inflight:
            take()
        if ptrs:                                                              # frame g in image g % n
            assert all(ptrs[g] == ptrs[0] + (g % LANES) * W * H * 16 for g in range(len(ptrs))), ptrs
        # a frame of another size is refused, a second share too
        with pytest.raises(capi.FlexLightHipError):
            ctx.frame_begin_shared(sc.frame_params(use_filter=0, width=W, height=H - 8, tile=(tile_rows, rank, RANKS)))
        ctx.share_leave()
        ctx.close()
        conn.send(("ok", bad))
    except BaseException as e:      # noqa: BLE001 — reported to the parent, which fails the test
        import traceback
        conn.send(("error", traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("tile_rows,scene_name,moves", [(8, "dragon", False), (16, "dragon", False), (8, "theater", False), (8, "dragon", True)])
def test_two_processes_complete_one_image(tile_rows, scene_name, moves):
    """... the theater (23 entries: not a scene the frame server takes): every rank renders on its two lanes and copies its strips into the root's image when the frame is taken;
    moves: every rank uploads changed transforms before every frame from the third on — the ranks' launches take them with the frames"""
    mpc = mp.get_context("spawn")
    pipes = [mpc.Pipe() for _ in range(RANKS)]
    procs = [mpc.Process(target=_rank_main, args=(r, pipes[r][1], tile_rows, scene_name, moves)) for r in range(RANKS)]
    for p in procs:
        p.start()
    try:
        assert pipes[0][0].poll(120), "rank 0 did not create the share"
        handle = pipes[0][0].recv()
        assert isinstance(handle, bytes), handle
        for r in range(1, RANKS):
            pipes[r][0].send(handle)
            assert pipes[r][0].poll(120), "rank %d did not join" % r
            got = pipes[r][0].recv()
            assert got == "joined", got
        for r in range(RANKS):
            pipes[r][0].send("go")
        for r in range(RANKS):
            assert pipes[r][0].poll(120), "rank %d did not finish" % r
            status, detail = pipes[r][0].recv()
            assert status == "ok", detail
            assert detail == [], "frames %s differ from one context's render" % detail
    finally:
        for p in procs:
            p.join(20)
            if p.is_alive():
                p.kill()


def test_frames_begun_and_ended_one_at_a_time_keep_their_images():
    """every rank begins and ends ONE frame at a time, so every frame is a launch of its own that starts on an empty loop (such a launch always began at image 0
    and a non-root rank could write frame g into the image the root was still reading frame g - 1 from): frame g must land in image g % n on every rank — the root copies
    each frame it is handed AFTER the other rank has been allowed to begin the next one, and every one of them must equal one context's render (moving camera)"""
    mpc = mp.get_context("spawn")
    pipes = [mpc.Pipe() for _ in range(RANKS)]
    procs = [mpc.Process(target=_rank_main, args=(r, pipes[r][1], 8, "dragon", False, 1)) for r in range(RANKS)]
    for p in procs:
        p.start()
    try:
        assert pipes[0][0].poll(120), "rank 0 did not create the share"
        handle = pipes[0][0].recv()
        assert isinstance(handle, bytes), handle
        pipes[1][0].send(handle)
        assert pipes[1][0].poll(120) and pipes[1][0].recv() == "joined"
        for r in range(RANKS):
            pipes[r][0].send("go")
        for r in range(RANKS):
            assert pipes[r][0].poll(120), "rank %d did not finish" % r
            status, detail = pipes[r][0].recv()
            assert status == "ok", detail
            assert detail == [], "frames %s differ from one context's render" % detail
    finally:
        for p in procs:
            p.join(20)
            if p.is_alive():
                p.kill()


def _quitter_main(rank, conn):
    """rank 0 creates the share and renders; rank 1 joins and then goes away without rendering a frame (its process ends: flx_share_leave is never called)"""
    try:
        from flexlight_hip import capi
        from flexlight_hip.scene_io import Scene
        sc = Scene.golden("dragon")
        ctx = capi.Context(0)
        ctx.update_scene(sc)
        ctx.set_server_groups(ctx.device_info()[1] // RANKS)
        if rank == 0:
            conn.send(ctx.share_create(W, H, LANES, RANKS, 0))
            assert conn.recv() == "go"
            import time
            t0 = time.time()
            ctx.frame_begin_shared(_moving(sc, 0, tile=(8, 0, RANKS)))
            try:
                ctx.frame_end_shared()
                conn.send(("error", "the root handed out a frame the other rank never rendered"))
                return
            except capi.FlexLightHipError as e:
                waited = time.time() - t0
                msg = str(e)
            # ... and the share stays failed: the next frame is refused at once instead of waiting again
            t1 = time.time()
            try:
                ctx.frame_begin_shared(_moving(sc, 1, tile=(8, 0, RANKS)))
                again = "accepted"
            except capi.FlexLightHipError:
                again = "refused"
            conn.send(("ok", (waited, msg, again, time.time() - t1)))
            ctx.close()
        else:
            ctx.share_join(conn.recv(), rank)
            conn.send("joined")
            os._exit(0)                                   # gone: no flx_share_leave, no error word
    except BaseException as e:      # noqa: BLE001
        import traceback
        conn.send(("error", traceback.format_exc() + repr(e)))


def test_a_rank_that_goes_away_does_not_wedge_the_root():
    """the root waits for every rank's strips of a frame — for at most 5 s: then its call fails with FLX_ERR_DEVICE, and the share stays failed"""
    mpc = mp.get_context("spawn")
    pipes = [mpc.Pipe() for _ in range(RANKS)]
    procs = [mpc.Process(target=_quitter_main, args=(r, pipes[r][1])) for r in range(RANKS)]
    for p in procs:
        p.start()
    try:
        assert pipes[0][0].poll(120)
        handle = pipes[0][0].recv()
        assert isinstance(handle, bytes), handle
        pipes[1][0].send(handle)
        assert pipes[1][0].poll(120) and pipes[1][0].recv() == "joined"
        procs[1].join(30)
        pipes[0][0].send("go")
        assert pipes[0][0].poll(60), "the root is still waiting"
        status, detail = pipes[0][0].recv()
        assert status == "ok", detail
        waited, msg, again, t_again = detail
        assert 4.0 <= waited <= 12.0, waited
        assert "did not complete" in msg or "failed" in msg, msg
        assert again == "refused" and t_again < 1.0
    finally:
        for p in procs:
            p.join(20)
            if p.is_alive():
                p.kill()
