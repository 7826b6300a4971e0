"""GPU parity: libflexlight_hip.so (through the C ABI) against the CPU oracle on the same inputs.

Inputs are the arrays the reference's own scene.js emits for the BASELINE scenes (tests/golden/ref_*).
Bar: per-channel RMS <= 1e-4 (BASELINE.json north_star); because both sides share include/flx_math.h
and are compiled without FMA contraction we additionally expect — and assert — bit equality, and
identical work counters (entries visited, shades, walks)."""
import os

import numpy as np
import pytest

from parity_util import assert_parity

pytestmark = pytest.mark.gpu

# (scene, width, height, samples, max_reflections): sizes the oracle finishes in seconds.
CASES = [
    ("cornell", 256, 256, 1, 1),          # BASELINE config 1, full size
    ("cornell", 37, 21, 2, 0),            # ragged frame (not a multiple of the 8x8 tile), zero bounces
    ("cornell", 128, 96, 3, 4),
    ("cornell_obj", 320, 180, 4, 3),      # config 2 scene, reduced frame, filter off
    ("dragon", 320, 180, 2, 4),           # config 3 scene, reduced frame
    ("theater", 320, 180, 2, 6),          # config 5 scene (9 lights, atlases), reduced frame
]


_ORACLE_CACHE = {}


def _assert_fullsize_hash(key, frame, counters, gbuffers=None):
    """the GPU frame of a BASELINE.json configuration at full size against tests/golden/oracle_fullsize.json (committed hashes of the
    oracle's frames: tests/analysis/make_fullsize_hashes.py) — beside the comparison with the oracle run next to it"""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "analysis"))
    import make_fullsize_hashes as fs
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_fullsize.json")))[key]
    assert counters == want["counters"], key
    assert fs.sha(frame) == want["frame"], key
    if gbuffers is not None:
        assert {k: fs.sha(v) for k, v in gbuffers.items()} == want["gbuffers"], key


@pytest.mark.parametrize("pipeline", [3, 2, 1], ids=["wavefront", "persistent", "per_pixel"])
@pytest.mark.parametrize("name,w,h,spp,bounces", CASES)
def test_radiance_matches_oracle(hip, oracle, scenes, name, w, h, spp, bounces, pipeline):
    sc = scenes(name)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
    hip.update_scene(sc)
    hip.set_pipeline(pipeline)
    try:
        got, got_cnt, _ = hip.render(p, counters=True)
    finally:
        hip.set_pipeline(0)
    key = (name, w, h, spp, bounces)
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = oracle.render(sc, p)[:2]
    want, want_cnt = _ORACLE_CACHE[key]
    rms, mism = assert_parity(got, want, name)
    assert mism == 0, "%s: %d of %d floats differ in bits (rms %s)" % (name, mism, got.size, rms)
    assert got_cnt == want_cnt
    assert (got[..., 3] == 1).sum() == want_cnt["primary_hits"]
    if (name, w, h, spp, bounces) == ("cornell", 256, 256, 1, 1):
        _assert_fullsize_hash("configs[0] cornell 256x256 1spp 1b", got, got_cnt)


@pytest.mark.parametrize("pipeline", [2, 1], ids=["persistent", "per_pixel"])
@pytest.mark.parametrize("name,w,h,spp,bounces", [c for c in CASES if c[0] != "dragon"])
def test_small_scenes_lane_by_lane(hip, oracle, scenes, name, w, h, spp, bounces, pipeline):
    """cornell, cornell.obj and the theater are walked by the wave in lockstep by default (flx_set_lockstep); the lane walk
    over the threaded copy, which larger scenes in one object space take, gives the same frame and the same counters"""
    sc = scenes(name)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
    hip.update_scene(sc)
    hip.set_pipeline(pipeline)
    hip.set_lockstep(False)
    try:
        got, got_cnt, _ = hip.render(p, counters=True)
    finally:
        hip.set_lockstep(True)
        hip.set_pipeline(0)
    key = (name, w, h, spp, bounces)
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = oracle.render(sc, p)[:2]
    want, want_cnt = _ORACLE_CACHE[key]
    assert np.array_equal(got, want, equal_nan=True)
    assert got_cnt == want_cnt


def test_headline_frame_at_full_size(hip, oracle, scenes):
    """The configuration BASELINE.json's target is quoted on (configs[2]) at full size — dragon, 1920x1080, 8 samples,
    4 bounces, filter off, the frame bench.py times —
    against the oracle on all host cores (66 M rays: a few seconds on the GPU box), bit for bit, with the work counters;
    then the same frame as the strips of 8 ranks (what `bench.py --gpus 8` gathers)."""
    sc = scenes("dragon")
    hip.update_scene(sc)
    p = sc.frame_params(use_filter=0)
    assert (p.width, p.height, p.samples, p.max_reflections) == (1920, 1080, 8, 4)
    got, got_cnt, _ = hip.render(p, counters=True)
    want, want_cnt = oracle.render(sc, p, threads=0)[:2]
    assert np.array_equal(got, want, equal_nan=True)
    assert got_cnt == want_cnt
    assert hip.last_pipeline() == 3
    _assert_fullsize_hash("configs[2] dragon 1080p 8spp 4b", got, got_cnt)
    whole = np.full_like(got, np.nan)
    for rank in range(8):
        p.tile_rows, p.tile_index, p.tile_count = 8, rank, 8
        part, _, _ = hip.render(p)
        whole[hip.tile_rows(p)] = part
    assert np.array_equal(whole, want, equal_nan=True)


def test_headline_frame_against_the_bounce_loop_as_written(hip, oracle, scenes):
    """The kernels (and the oracle by default) skip the rayTracer call of fragment:591 whose hit the loop guard discards.  The GPU
    frame of configs[2] at full size equals the oracle's frame with the bounce loop AS THE SHADER IS WRITTEN — every iteration
    ends with that walk — bit for bit; what differs is only the closest-hit work, which the as-written loop does once per bounce
    iteration."""
    sc = scenes("dragon")
    hip.update_scene(sc)
    p = sc.frame_params(use_filter=0)
    got, got_cnt, _ = hip.render(p, counters=True)
    oracle.set_as_written(True)
    try:
        want, want_cnt = oracle.render(sc, p, threads=0)[:2]
    finally:
        oracle.set_as_written(False)
    assert np.array_equal(got, want, equal_nan=True)
    assert want_cnt["closest_walks"] == want_cnt["shades"] > got_cnt["closest_walks"]
    for key in got_cnt:
        if not key.startswith("closest_"):
            assert got_cnt[key] == want_cnt[key], key


def test_synthetic_100k_dragon_at_full_size(hip, oracle, scenes):
    """configs[2] as BASELINE.json words it — "dragon.obj (~100k tris)" — with the deterministic stand-in for the absent
    objects/dragon.obj: dragon_lp.obj split 1 -> 4 (tools/make_dragon_100k.py; 175 598 triangles, 289 189 entries in the scene, a
    13.9 MB threaded tree that no longer fits one XCD's 4 MB L2), arrays emitted by the reference's scene.js
    (tests/golden/ref_dragon_100k.*).  1920x1080, 8 samples, 4 bounces, bit for bit against the oracle with the work counters."""
    sc = scenes("dragon_100k")
    assert sc.meta["textureLength"] == 289189
    hip.update_scene(sc)
    p = sc.frame_params(use_filter=0)
    assert (p.width, p.height, p.samples, p.max_reflections) == (1920, 1080, 8, 4)
    got, got_cnt, _ = hip.render(p, counters=True)
    want, want_cnt = oracle.render(sc, p, threads=0)[:2]
    assert np.array_equal(got, want, equal_nan=True)
    assert got_cnt == want_cnt
    assert hip.last_pipeline() == 3


@pytest.mark.parametrize("name,w,h,spp,bounces", [("dragon", 3840, 2160, 8, 4), ("theater", 1920, 1080, 16, 6)],
                         ids=["dragon_4k", "theater_1080p"])
def test_multi_gpu_configs_at_full_size(hip, oracle, scenes, name, w, h, spp, bounces):
    """BASELINE.json configs[3] and configs[4] (the 8-GPU workloads) at full size on one GPU, bit for bit against the oracle
    on all host cores (265 M and 199 M rays)."""
    sc = scenes(name)
    hip.update_scene(sc)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
    got, got_cnt, _ = hip.render(p, counters=True)
    want, want_cnt = oracle.render(sc, p, threads=0)[:2]
    assert np.array_equal(got, want, equal_nan=True)
    assert got_cnt == want_cnt
    _assert_fullsize_hash({"dragon": "configs[3] dragon 4K 8spp 4b", "theater": "configs[4] theater 1080p 16spp 6b"}[name], got, got_cnt)


def _moved(sc, p, i):
    """frame i of a camera move: another position, view direction, seed and ambient"""
    from flexlight_hip.scene_io import view_matrix
    cam = sc.meta["camera"]
    q = type(p).from_buffer_copy(p)
    q.camera[:] = [cam["x"] + 0.35 * i, cam["y"] + 0.1 * i, cam["z"] - 0.2 * i]
    q.view_matrix[:] = view_matrix(cam["fx"] + 0.07 * i, cam["fy"] - 0.03 * i, cam["fov"], p.width, p.height).tolist()
    q.random_seed = float(i % 3)
    q.ambient[:] = [a * (1.0 + 0.25 * i) for a in sc.meta["ambient"]]
    return q


@pytest.mark.parametrize("name", ["cornell", "cornell_obj", "theater", "dragon"])
def test_random_views_match_oracle(hip, oracle, scenes, name):
    """forty frames per scene from random places: camera position and direction, frame size (ragged against the 8 x 8 tiles),
    samples, bounces, seed, filter on or off, kernel organisation left to the library or forced — against the oracle, bit for bit
    with the counters.  (Looks from inside the geometry, views that miss everything and grazing rays come up this way.)"""
    from flexlight_hip.scene_io import view_matrix
    sc = scenes(name)
    hip.update_scene(sc)
    cam = sc.meta["camera"]
    rng = np.random.default_rng(20261004 + len(name))
    for case in range(40):
        w, h = int(rng.integers(17, 97)), int(rng.integers(9, 61))
        spp, bounces = int(rng.integers(1, 5)), int(rng.integers(0, 6))
        filt = int(rng.integers(0, 2))
        p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=filt)
        reach = 6.0 if name != "dragon" else 25.0
        p.camera[:] = [cam["x"] + rng.uniform(-reach, reach), cam["y"] + rng.uniform(-reach / 2, reach / 2), cam["z"] + rng.uniform(-reach, reach)]
        p.view_matrix[:] = view_matrix(cam["fx"] + rng.uniform(-3.2, 3.2), cam["fy"] + rng.uniform(-1.2, 1.2), cam["fov"] * rng.uniform(0.6, 1.6), w, h).tolist()
        p.random_seed = float(rng.integers(0, 7))
        pipe = 0 if filt else int(rng.integers(0, 4))
        hip.set_pipeline(pipe)
        try:
            got, got_cnt, _ = hip.render(p, counters=True)
        finally:
            hip.set_pipeline(0)
        want, want_cnt = oracle.render(sc, p)[:2]
        tag = "%s case %d (%dx%d %d spp %d bounces filter %d pipeline %d)" % (name, case, w, h, spp, bounces, filt, pipe)
        assert np.array_equal(got, want, equal_nan=True), tag
        assert got_cnt == want_cnt, tag


@pytest.mark.parametrize("pipeline", [3, 2, 1], ids=["wavefront", "persistent", "per_pixel"])
@pytest.mark.parametrize("name,w,h,spp,bounces,tile", [("dragon", 320, 180, 2, 4, (0, 0, 0)), ("dragon", 200, 117, 2, 3, (8, 1, 3)),
                                                       ("theater", 160, 90, 2, 4, (0, 0, 0))])
def test_batch_of_frames_equals_the_frames(hip, scenes, name, w, h, spp, bounces, tile, pipeline):
    """flx_render_batch: five frames of a camera move in one pass (117 rows and strips of 8: frames and tiles that do not
    align) — every frame bit-identical to its own flx_render, the work counters the sum."""
    sc = scenes(name)
    hip.update_scene(sc)
    p0 = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0, tile=tile)
    frames = [_moved(sc, p0, i) for i in range(5)]
    hip.set_pipeline(pipeline)
    try:
        singles = [hip.render(q, counters=True) for q in frames]
        got, cnt = hip.render_batch(frames, counters=True)
        again, _ = hip.render_batch(frames)
    finally:
        hip.set_pipeline(0)
    assert got.shape == (5,) + singles[0][0].shape
    for i, (want, _, _) in enumerate(singles):
        assert np.array_equal(got[i], want, equal_nan=True), "frame %d" % i
        assert np.array_equal(again[i], want, equal_nan=True), "frame %d (uncounted)" % i
    assert not np.array_equal(got[0], got[1])
    assert cnt == {k: sum(c[k] for _, c, _ in singles) for k in cnt}


@pytest.mark.parametrize("name,w,h,spp,bounces", [("cornell_obj", 160, 96, 2, 3), ("dragon", 200, 117, 1, 3)])
def test_batch_of_filter_frames_equals_the_frames(hip, scenes, name, w, h, spp, bounces):
    """filter frames in a batch: one trace pass over the stacked frames, the denoise chain frame by frame — each frame (a
    different camera, seed and ambient) equal to its own flx_render"""
    sc = scenes(name)
    hip.update_scene(sc)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=1)
    frames = [_moved(sc, p, i) for i in range(3)]
    got, _ = hip.render_batch(frames)
    for i, q in enumerate(frames):
        want, _, _ = hip.render(q)
        assert np.array_equal(got[i], want, equal_nan=True), "frame %d" % i
    assert not np.array_equal(got[0], got[1])


def test_batch_arguments(hip, scenes):
    from flexlight_hip import capi
    sc = scenes("cornell")
    hip.update_scene(sc)
    p = sc.frame_params(width=32, height=24, use_filter=0)
    for bad in ([], [p] * 33):
        with pytest.raises(capi.FlexLightHipError, match="1 .. 32"):
            hip.render_batch(bad)
    q = type(p).from_buffer_copy(p)
    q.samples = p.samples + 1
    with pytest.raises(capi.FlexLightHipError, match="differ in camera"):
        hip.render_batch([p, q])
    f = sc.frame_params(width=32, height=24, use_filter=0)
    f.is_temporal = 1
    with pytest.raises(capi.FlexLightHipError, match="cannot be batched"):
        hip.render_batch([f, f])
    f = sc.frame_params(width=32, height=24, use_filter=1, tile=(8, 0, 2))
    with pytest.raises(capi.FlexLightHipError, match="whole frames"):
        hip.render_batch([f, f])
    one, _ = hip.render_batch([p])
    assert np.array_equal(one[0], hip.render(p)[0], equal_nan=True)


def test_tiles_reassemble_full_frame(hip, scenes):
    """Row-strip tile policy (multi-GPU split): strips dealt round-robin reproduce the whole frame."""
    sc = scenes("cornell")
    hip.update_scene(sc)
    full, _, _ = hip.render(sc.frame_params(width=96, height=80, samples=2, max_reflections=3, use_filter=0))
    for tile_rows, count in ((8, 2), (8, 3), (16, 4), (5, 3)):
        frame = np.full_like(full, np.nan)
        seen = 0
        for index in range(count):
            p = sc.frame_params(width=96, height=80, samples=2, max_reflections=3, use_filter=0, tile=(tile_rows, index, count))
            part, _, _ = hip.render(p)
            rows = hip.tile_rows(p)
            assert part.shape[0] == len(rows)
            frame[rows] = part
            seen += len(rows)
        assert seen == 80
        assert np.array_equal(frame, full, equal_nan=True)


def test_counters_off_same_image(hip, scenes):
    sc = scenes("theater")
    hip.update_scene(sc)
    p = sc.frame_params(width=160, height=90, samples=1, max_reflections=3, use_filter=0)
    a, _, _ = hip.render(p, counters=True)
    b, none, _ = hip.render(p)
    assert none is None
    assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("name,w,h,spp,bounces", [("dragon", 320, 180, 2, 4), ("cornell", 96, 80, 3, 3), ("theater", 64, 40, 1, 2)])
def test_wavefront_groups_do_not_change_the_frame(hip, oracle, scenes, name, w, h, spp, bounces):
    """The bounce loop may run as 1..4 independent chains of tile ranges on separate streams; same bits."""
    sc = scenes(name)
    hip.update_scene(sc)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
    want, _, _ = oracle.render(sc, p)
    try:
        for groups in (1, 2, 3, 4):
            hip.set_wavefront_groups(groups)
            for _ in range(2):                       # twice: buffers and streams are reused across frames
                got, _, _ = hip.render(p)
                assert np.array_equal(got, want, equal_nan=True), groups
    finally:
        hip.set_wavefront_groups(1)


@pytest.mark.parametrize("organisation,front", [(1, 0), (2, 0), (2, 2), (1, 3), (2, 3)],
                         ids=["rounds", "frame_kernel", "frame_kernel_with_its_front", "rounds_one_front_kernel", "frame_kernel_one_front_kernel"])
@pytest.mark.parametrize("name,w,h,spp,bounces", [("dragon", 480, 270, 2, 4), ("dragon", 333, 187, 3, 2), ("dragon", 640, 360, 8, 6), ("cornell_obj", 128, 96, 2, 5),
                                                  ("theater", 96, 64, 2, 3), ("cornell", 64, 64, 1, 1), ("dragon", 64, 36, 1, 4)])
def test_both_organisations_of_the_bounce_loop_equal_the_oracle(hip, oracle, scenes, name, w, h, spp, bounces, organisation, front):
    """the wavefront pipeline as rounds (a shade + walk kernel pair per bounce) and as ONE persistent launch (k_wf_frame: walk waves and
    shade waves of a workgroup hand paths to each other through LDS rings) — with k_primary and k_wf_shade0 in front of it, and with the
    primary rays and the bounce-0 shading made by its own shade waves, or by ONE kernel in front (k_wf_front) instead of two (flx_set_frame_front):
    the same frame and the same work counters as the oracle — frames of a few waves (where most workgroups find the item queue dry at once) and frames that fill the machine"""
    sc = scenes(name)
    hip.update_scene(sc)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
    key = (name, w, h, spp, bounces)
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = oracle.render(sc, p)[:2]
    want, want_cnt = _ORACLE_CACHE[key]
    try:
        hip.set_pipeline(3)
        hip.set_wavefront_organisation(organisation)
        hip.set_frame_front(front)
        for _ in range(2):
            got, cnt, _ = hip.render(p, counters=True)
            assert np.array_equal(got, want, equal_nan=True)
            assert cnt == want_cnt
        for _ in range(3):
            got, _, _ = hip.render(p)                               # the build without counters
            assert np.array_equal(got, want, equal_nan=True)
    finally:
        hip.set_pipeline(0)
        hip.set_wavefront_organisation(0)
        hip.set_frame_front(1)


def test_the_front_inside_the_frame_kernel_at_its_edges(hip, oracle, scenes):
    """flx_set_frame_front(2): more samples than a wave has lanes (a tile's units are handed over 64 at a time), so many samples that a tile would not
    fit the workgroup's rings (the front stays outside: refused quietly, same frame), a rank's strips of a tiled frame, a batch of moved cameras —
    against the same frames with k_primary and k_wf_shade0 in front (which the tests above hold against the oracle), bit for bit with equal counters"""
    sc = scenes("dragon")
    hip.update_scene(sc)
    try:
        hip.set_pipeline(3)
        hip.set_wavefront_organisation(2)
        cases = [sc.frame_params(width=160, height=90, samples=70, max_reflections=3, use_filter=0),
                 sc.frame_params(width=96, height=56, samples=130, max_reflections=2, use_filter=0),
                 sc.frame_params(width=333, height=187, samples=5, max_reflections=4, use_filter=0, tile=(8, 1, 3)),
                 sc.frame_params(width=640, height=360, samples=4, max_reflections=4, use_filter=0, tile=(16, 0, 2))]
        # a view that hits nothing (the camera far outside the scene's room, looking away from it): no tile is handed over at all
        from flexlight_hip import scene_io
        cam = sc.meta["camera"]
        away = None
        for dy, fy in ((1.0e3, 1.4), (1.0e3, -1.4), (-1.0e3, 1.4), (-1.0e3, -1.4)):
            q = sc.frame_params(width=256, height=144, samples=4, max_reflections=3, use_filter=0)
            q.camera[1] += dy
            q.view_matrix[:] = scene_io.view_matrix(cam["fx"], fy, cam["fov"], q.width, q.height).tolist()
            hip.set_frame_front(0)
            if hip.render(q, counters=True)[1]["primary_hits"] == 0:
                away = q
                break
        assert away is not None
        cases.append(away)
        for p in cases:
            hip.set_frame_front(0)
            want, want_cnt, _ = hip.render(p, counters=True)
            hip.set_frame_front(2)
            for _ in range(2):
                got, cnt, _ = hip.render(p, counters=True)
                assert np.array_equal(got, want, equal_nan=True) and cnt == want_cnt, (p.width, p.height, p.samples)
            assert np.array_equal(hip.render(p)[0], want, equal_nan=True)
        batch = []
        for i in range(3):
            q = sc.frame_params(width=320, height=180, samples=4, max_reflections=4, use_filter=0)
            q.camera[0] += 0.4 * i
            q.random_seed = float(i)
            batch.append(q)
        hip.set_frame_front(0)
        want, want_cnt = hip.render_batch(batch, counters=True)
        hip.set_frame_front(2)
        got, cnt = hip.render_batch(batch, counters=True)
        assert np.array_equal(got, want, equal_nan=True) and cnt == want_cnt
        assert np.array_equal(hip.render_batch(batch)[0], want, equal_nan=True)
    finally:
        hip.set_pipeline(0)
        hip.set_wavefront_organisation(0)
        hip.set_frame_front(1)


def test_random_frames_agree_between_the_organisations(hip, scenes):
    """a hundred random frames — odd sizes, one to eleven samples, one to seven bounces, whole frames and a rank's strips of every width — rendered as rounds (which the
    tests above hold against the oracle), as the frame kernel, as the frame kernel with the front of the frame inside it, and with one kernel in front of either:
    the same bits and the same work counters"""
    sc = scenes("dragon")
    hip.update_scene(sc)
    rng = np.random.default_rng(20261004)
    try:
        hip.set_pipeline(3)
        for case in range(100):
            w, h = int(rng.integers(1, 700)), int(rng.integers(1, 400))
            spp, bounces = int(rng.integers(1, 12)), int(rng.integers(1, 8))
            tile = (0, 0, 0)
            if rng.random() < 0.5:
                count = int(rng.integers(2, 9))
                tile = (int(rng.choice([1, 3, 8, 16])), int(rng.integers(0, count)), count)
            p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0, tile=tile)
            p.random_seed = float(case)
            hip.set_wavefront_organisation(1)
            hip.set_frame_front(0)
            want, want_cnt, _ = hip.render(p, counters=True)
            for organisation, front in ((2, 0), (2, 2), (2, 3), (1, 3)):
                hip.set_wavefront_organisation(organisation)
                hip.set_frame_front(front)
                got, cnt, _ = hip.render(p, counters=True)
                assert np.array_equal(got, want, equal_nan=True) and cnt == want_cnt, (case, w, h, spp, bounces, tile, front)
                assert np.array_equal(hip.render(p)[0], want, equal_nan=True), (case, w, h, spp, bounces, tile, front)
    finally:
        hip.set_pipeline(0)
        hip.set_wavefront_organisation(0)
        hip.set_frame_front(1)


def test_frame_kernel_hands_paths_over_the_same_way_every_time(hip, oracle, scenes):
    """the frame kernel's walk waves and shade waves pass paths to each other through rings with workgroup-scope release / acquire:
    eighty renders of one frame, alternating with a frame of another size (other buffers, other ring contents), all equal the
    oracle's frame bit for bit (tools/soak_framekernel.py does 645 full-size frames the same way)"""
    sc = scenes("dragon")
    hip.update_scene(sc)
    p = sc.frame_params(width=640, height=360, samples=8, max_reflections=6, use_filter=0)
    q = sc.frame_params(width=333, height=187, samples=3, max_reflections=4, use_filter=0)
    key = ("dragon", 640, 360, 8, 6)
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = oracle.render(sc, p)[:2]
    want, _ = _ORACLE_CACHE[key]
    want_q = None
    try:
        hip.set_pipeline(3)
        hip.set_wavefront_organisation(2)
        for i in range(80):
            hip.set_frame_front(2 if i % 2 else 0)                  # (the front of the frame inside the launch: a third ring, the same scopes)
            assert np.array_equal(hip.render(p)[0], want, equal_nan=True), "frame %d" % i
            if i % 8 == 0:
                got_q = hip.render(q)[0]
                want_q = got_q if want_q is None else want_q
                assert np.array_equal(got_q, want_q, equal_nan=True), "small frame %d" % i
    finally:
        hip.set_pipeline(0)
        hip.set_wavefront_organisation(0)
        hip.set_frame_front(1)


@pytest.mark.experiments
@pytest.mark.parametrize("scheduler,suspend", [(0, 128), (0, 16), (1, 0), (2, 16), (2, 128), (0, 0)])
@pytest.mark.parametrize("name,w,h,spp,bounces", [("dragon", 480, 270, 2, 4), ("cornell_obj", 128, 96, 2, 5), ("theater", 96, 64, 1, 1)])
def test_walk_schedulers_do_not_change_the_frame(hip, oracle, scenes, name, w, h, spp, bounces, scheduler, suspend):
    """Queue scheduler, suspension to the next round, cooperative finisher: other orders of the same work — same bits, same
    work counters (every ray visits the same entries)."""
    sc = scenes(name)
    hip.update_scene(sc)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
    key = (name, w, h, spp, bounces)
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = oracle.render(sc, p)[:2]
    want, want_cnt = _ORACLE_CACHE[key]
    try:
        hip.set_walk_scheduler(scheduler, suspend)
        for _ in range(2):
            got, cnt, _ = hip.render(p, counters=True)
            assert np.array_equal(got, want, equal_nan=True)
            assert cnt == want_cnt
        got, _, _ = hip.render(p)
        assert np.array_equal(got, want, equal_nan=True)
    finally:
        hip.set_walk_scheduler(0, 0)


@pytest.mark.experiments
@pytest.mark.parametrize("w,h,spp,bounces", [(480, 270, 2, 4), (640, 368, 8, 4), (200, 120, 3, 6)])
def test_two_walk_jobs_per_lane_do_not_change_the_frame(hip, oracle, scenes, w, h, spp, bounces):
    """k_wf_frame2 (round 5, measured slower, `make EXPERIMENTS=1` only): the walk waves of the frame kernel hold two independent jobs per lane and step them in a box phase
    and a triangle phase — another order of the same work: same bits, same work counters as the oracle"""
    sc = scenes("dragon")
    hip.update_scene(sc)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=0)
    key = ("dragon", w, h, spp, bounces)
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = oracle.render(sc, p)[:2]
    want, want_cnt = _ORACLE_CACHE[key]
    try:
        hip.set_pipeline(3)
        hip.set_wavefront_organisation(2)
        hip.set_frame_front(2)                                  # the front of the frame inside the launch whatever the frame's size: the kernel the two-job variant replaces
        hip.set_walk_jobs(2)
        for _ in range(2):
            got, cnt, _ = hip.render(p, counters=True)
            assert hip.last_organisation() == 3
            assert np.array_equal(got, want, equal_nan=True)
            assert cnt == want_cnt
        got, _, _ = hip.render(p)
        assert np.array_equal(got, want, equal_nan=True)
    finally:
        hip.set_walk_jobs(1)
        hip.set_pipeline(0)
        hip.set_wavefront_organisation(0)
        hip.set_frame_front(1)


def test_walk_scheduler_arguments(hip):
    from flexlight_hip import capi
    for args in ((3, 0), (-1, 0), (0, 513), (1, 8)):
        with pytest.raises(capi.FlexLightHipError):
            hip.set_walk_scheduler(*args)
    hip.set_walk_scheduler(0, 0)
    if not capi.has_experiments():                                # the shipped library carries the default scheduler only
        for args in ((1, 0), (2, 16), (0, 16)):
            with pytest.raises(capi.FlexLightHipError, match="EXPERIMENTS"):
                hip.set_walk_scheduler(*args)
        with pytest.raises(capi.FlexLightHipError, match="EXPERIMENTS"):
            hip.set_walk_jobs(2)


def test_errors_are_reported_not_thrown(hip, scenes):
    from flexlight_hip import capi
    sc = scenes("cornell")
    hip.update_scene(sc)
    p = sc.frame_params(width=8, height=8)
    p.width = 0
    with pytest.raises(capi.FlexLightHipError, match="width"):
        hip.render(p)
    fresh = capi.Context(0)
    try:
        with pytest.raises(capi.FlexLightHipError, match="before"):
            fresh.render(sc.frame_params(width=8, height=8))
    finally:
        fresh.close()


# ---- filter on: G-buffers (fragment:619-639) and the denoise chain (F0-F3) ------------------------------
FILTER_CASES = [
    ("cornell_obj", 320, 180, 4, 3),      # BASELINE config 2 at reduced size
    ("cornell", 160, 120, 2, 3),          # textured PBR material, translucency flags off
    ("dragon", 240, 136, 2, 4),           # translucent dragon + sphere: glassFilter / vote branch of the first filter
    ("dragon", 130, 75, 8, 5),            # eight samples, a ragged frame, glass paths of several unfiltered bounces
    ("theater", 200, 112, 1, 3),
    ("theater", 96, 54, 16, 6),           # BASELINE config 5's samples and bounces with the filter on
    ("cornell_obj", 100, 60, 3, 3),
    ("cornell", 64, 40, 6, 2),
]


@pytest.mark.parametrize("name,w,h,spp,bounces", FILTER_CASES)
def test_filter_chain_matches_oracle(hip, oracle, scenes, name, w, h, spp, bounces):
    """filter frames: trace with G-buffers + the denoise chain"""
    sc = scenes(name)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=1)
    hip.update_scene(sc)
    got, got_cnt, got_gb = hip.render(p, gbuffers=True, counters=True)
    want, want_cnt, want_gb = oracle.render(sc, p, gbuffers=True)
    for key in want_gb:
        rms, mism = assert_parity(got_gb[key], want_gb[key], "%s G-buffer %s" % (name, key))
        assert mism == 0, "%s G-buffer %s: %d floats differ" % (name, key, mism)
    rms, mism = assert_parity(got, want, name + " filtered")
    assert mism == 0, "%s filtered frame: %d of %d floats differ (rms %s)" % (name, mism, got.size, rms)
    assert got_cnt == want_cnt


def test_filter_frame_at_full_size(hip, oracle, scenes):
    """BASELINE.json configs[1] at full size: cornell_obj, 1920x1080, 4 samples, 3 bounces, filter ON — trace with G-buffers
    and the whole denoise chain against the oracle, bit for bit, with the work counters."""
    sc = scenes("cornell_obj")
    hip.update_scene(sc)
    p = sc.frame_params(use_filter=1)
    assert (p.width, p.height, p.samples, p.max_reflections) == (1920, 1080, 4, 3)
    got, got_cnt, got_gb = hip.render(p, gbuffers=True, counters=True)
    want, want_cnt, want_gb = oracle.render(sc, p, gbuffers=True)
    for key in want_gb:
        assert np.array_equal(got_gb[key], want_gb[key], equal_nan=True), key
    assert np.array_equal(got, want, equal_nan=True)
    assert got_cnt == want_cnt
    assert hip.last_pipeline() == 1
    _assert_fullsize_hash("configs[1] cornell_obj 1080p 4spp 3b filter", got, got_cnt, got_gb)


def test_filter_refuses_tiles(hip, scenes):
    from flexlight_hip import capi
    sc = scenes("cornell")
    hip.update_scene(sc)
    p = sc.frame_params(width=64, height=64, use_filter=1, tile=(8, 0, 2))
    with pytest.raises(capi.FlexLightHipError, match="tiled"):
        hip.render(p)


@pytest.mark.parametrize("name,w,h,spp,bounces,ranks", [("cornell_obj", 160, 96, 2, 3, 3), ("dragon", 192, 120, 1, 3, 4), ("theater", 96, 72, 1, 2, 2)])
def test_filter_frame_from_strips_of_several_ranks(hip, scenes, name, w, h, spp, bounces, ranks):
    """SURVEY 8e with the filter on: every rank traces its strips into the five RGBA8 render targets, the planes are gathered
    into image order, the chain runs on the whole frame — the same bits as the whole frame on one context."""
    import torch
    sc = scenes(name)
    hip.update_scene(sc)
    full = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=1)
    want, _, _ = hip.render(full)
    planes = torch.zeros((5, h, w), dtype=torch.int32, device="cuda")
    seen = 0
    for r in range(ranks):
        p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=1, tile=(8, r, ranks))
        rows = torch.as_tensor(np.asarray(hip.tile_rows(p), dtype=np.int64), device="cuda")
        part = torch.zeros((5, rows.numel(), w), dtype=torch.int32, device="cuda")
        hip.render_planes_device(p, part.data_ptr())
        hip.sync()
        planes[:, rows, :] = part
        seen += rows.numel()
    assert seen == h
    out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    hip.filter_planes_device(full, planes.data_ptr(), out.data_ptr())
    hip.sync()
    got = out.cpu().numpy()
    assert np.array_equal(got, want, equal_nan=True)


def test_render_planes_arguments(hip, scenes):
    from flexlight_hip import capi
    import torch
    sc = scenes("cornell")
    hip.update_scene(sc)
    buf = torch.zeros((5, 64, 64), dtype=torch.int32, device="cuda")
    with pytest.raises(capi.FlexLightHipError, match="use_filter"):
        hip.render_planes_device(sc.frame_params(width=64, height=64, use_filter=0), buf.data_ptr())
    p = sc.frame_params(width=64, height=64, use_filter=1)
    p.is_temporal = 1
    with pytest.raises(capi.FlexLightHipError, match="is_temporal"):
        hip.render_planes_device(p, buf.data_ptr())


# ---- temporal accumulation: history rings in the context, frame f traced with seed f % N ---------
@pytest.mark.parametrize("name,w,h,spp,bounces,filt,frames,n", [
    ("cornell", 96, 72, 1, 3, 0, 6, 4),          # more frames than history slots: the ring wraps
    ("dragon", 160, 90, 1, 3, 0, 5, 4),
    ("cornell_obj", 128, 72, 2, 3, 1, 3, 4),     # temporal + filter: the pass feeds RenderTexture[0] / IpRenderTexture[0]
    ("theater", 96, 54, 1, 2, 0, 4, 3),          # history depth that is not a multiple of four (vec4(0) stand-ins)
])
def test_temporal_sequence_matches_oracle(hip, oracle, scenes, name, w, h, spp, bounces, filt, frames, n):
    sc = scenes(name)
    hip.update_scene(sc)
    p = sc.frame_params(width=w, height=h, samples=spp, max_reflections=bounces, use_filter=filt)
    p.is_temporal, p.temporal_samples = 1, n
    want = oracle.render_sequence(sc, p, frames)
    hip.temporal_reset()
    for f in range(frames):
        p.random_seed = float(f % n)
        got, _, _ = hip.render(p)
        rms, mism = assert_parity(got, want[f], "%s temporal frame %d" % (name, f))
        assert mism == 0, "%s temporal frame %d: %d floats differ (rms %s)" % (name, f, mism, rms)
    # a reset forgets the history: the next frame equals frame 0 of a fresh run
    hip.temporal_reset()
    p.random_seed = 0.0
    again, _, _ = hip.render(p)
    assert np.array_equal(again, want[0], equal_nan=True)
