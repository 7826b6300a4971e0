"""Parity on synthetic scenes that exercise what the BASELINE scenes do not: many rotated / scaled transforms
(more than the walk kernel can pre-transform in LDS -> its on-the-fly variant), zero and several lights,
multi-cell texture atlases, translucent / emissive materials, degenerate triangles, an entry array without
terminator, axis-aligned rays (zero direction components -> the IEEE-division path of the box test)."""
import numpy as np
import pytest

import synth_scene
from parity_util import assert_parity

pytestmark = pytest.mark.gpu

CASES = {
    "many_transforms": dict(seed=1, n_objects=9, tris_per_object=24, n_transforms=9, n_lights=2),
    "four_transforms": dict(seed=2, n_objects=6, tris_per_object=30, n_transforms=4, n_lights=3),
    "no_lights": dict(seed=3, n_objects=3, tris_per_object=40, n_transforms=2, n_lights=0),
    "no_terminator": dict(seed=4, n_objects=3, tris_per_object=50, n_transforms=3, n_lights=1, exact_multiple=True),
    "degenerate_untextured": dict(seed=5, n_objects=2, tris_per_object=30, n_transforms=1, n_lights=1, textured=False, degenerate=6),
    "one_space_axis_aligned": dict(seed=7, n_objects=2, tris_per_object=24, n_transforms=1, n_lights=2, axis_aligned_view=True, width=65, height=33),
    "axis_aligned_view": dict(seed=6, n_objects=3, tris_per_object=30, n_transforms=2, n_lights=2, axis_aligned_view=True, width=65, height=33),
}


@pytest.mark.parametrize("pipeline", [3, 1], ids=["wavefront", "per_pixel"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_synthetic_scene_matches_oracle(hip, oracle, case, pipeline):
    sc = synth_scene.make(**CASES[case])
    p = sc.frame_params(use_filter=0)
    hip.update_scene(sc)
    hip.set_pipeline(pipeline)
    want, want_cnt, _ = oracle.render(sc, p)
    try:
        # the wavefront pipeline as one persistent launch (with the front of the frame inside it, and in front of it) and as rounds
        for organisation, front in (((2, 2), (2, 0), (1, 0), (1, 3)) if pipeline == 3 else ((0, 1),)):
            hip.set_wavefront_organisation(organisation)
            hip.set_frame_front(front)
            got, got_cnt, _ = hip.render(p, counters=True)
            plain, _, _ = hip.render(p)
            rms, mism = assert_parity(got, want, case)
            assert mism == 0, "%s (organisation %d): %d of %d floats differ (rms %s)" % (case, organisation, mism, got.size, rms)
            assert got_cnt == want_cnt, organisation
            assert np.array_equal(plain, got, equal_nan=True)
    finally:
        hip.set_pipeline(0)
        hip.set_wavefront_organisation(0)
        hip.set_frame_front(1)
    assert want_cnt["primary_hits"] > 0.3 * p.width * p.height          # the scene is actually in view


@pytest.mark.parametrize("lockstep", [True, False], ids=["lockstep", "lanes"])
@pytest.mark.parametrize("pipeline", [2, 1], ids=["persistent", "per_pixel"])
@pytest.mark.parametrize("case", ["degenerate_untextured", "one_space_axis_aligned"])
def test_small_one_space_scenes_both_walks(hip, oracle, case, pipeline, lockstep):
    """scenes of <= 128 entries in one object space: the wave-wide lockstep walk and the lane walk, degenerate triangles and
    rays with zero direction components (the box test's IEEE-division path) included"""
    sc = synth_scene.make(**CASES[case])
    assert sc.meta["textureLength"] <= 127
    p = sc.frame_params(use_filter=0)
    hip.update_scene(sc)
    hip.set_pipeline(pipeline)
    hip.set_lockstep(lockstep)
    try:
        got, got_cnt, _ = hip.render(p, counters=True)
        plain, _, _ = hip.render(p)
    finally:
        hip.set_lockstep(True)
        hip.set_pipeline(0)
    want, want_cnt, _ = oracle.render(sc, p)
    assert np.array_equal(got, want, equal_nan=True)
    assert got_cnt == want_cnt
    assert np.array_equal(plain, got, equal_nan=True)


@pytest.mark.parametrize("case", ["four_transforms", "no_lights"])
def test_synthetic_scene_filter_and_temporal(hip, oracle, case):
    sc = synth_scene.make(**CASES[case])
    hip.update_scene(sc)
    p = sc.frame_params(use_filter=1)
    got, _, gb = hip.render(p, gbuffers=True)
    want, _, wgb = oracle.render(sc, p, gbuffers=True)
    for key in wgb:
        _, mism = assert_parity(gb[key], wgb[key], "%s %s" % (case, key))
        assert mism == 0, key
    _, mism = assert_parity(got, want, case + " filtered")
    assert mism == 0
    p = sc.frame_params(use_filter=0)
    p.is_temporal, p.temporal_samples = 1, 4
    seq = oracle.render_sequence(sc, p, 3)
    hip.temporal_reset()
    for f in range(3):
        p.random_seed = float(f)
        frame, _, _ = hip.render(p)
        _, mism = assert_parity(frame, seq[f], "%s temporal %d" % (case, f))
        assert mism == 0


@pytest.mark.parametrize("case", ["many_transforms", "no_lights", "no_terminator"])
def test_synthetic_scene_in_batches(hip, case):
    """flx_render_batch over the walk kernel's on-the-fly variant (nine transforms), a scene without lights and one without
    terminator: three frames with different seeds and ambients, each equal to its own render."""
    sc = synth_scene.make(**CASES[case])
    hip.update_scene(sc)
    p0 = sc.frame_params(use_filter=0)
    frames = []
    for i in range(3):
        q = type(p0).from_buffer_copy(p0)
        q.random_seed = float(i)
        q.ambient[:] = [0.05 * (i + 1)] * 3
        q.camera[0] = p0.camera[0] + 0.2 * i
        frames.append(q)
    singles = [hip.render(q, counters=True) for q in frames]
    got, cnt = hip.render_batch(frames, counters=True)
    for i, (want, _, _) in enumerate(singles):
        assert np.array_equal(got[i], want, equal_nan=True), "frame %d" % i
    assert cnt == {k: sum(c[k] for _, c, _ in singles) for k in cnt}
