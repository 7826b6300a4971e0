"""The shading's per-triangle table (DeviceScene::angle_tan, csrc/flx_kernels.hip: k_angle_tan): clamp(tan(acos(|geometryNormal . n_i|)), 0, 1) of fragment:500-512
depends on the triangle and on its transform only, so it is made once per scene / transform upload by the device function the shading itself would call.  The frames
with the table must equal the frames without it bit for bit, with equal work counters, on every scene and pipeline (the rest of the suite runs WITH the table against
the oracle) — and the table must follow the transforms."""
import copy

import numpy as np
import pytest

from parity_util import bit_mismatches

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,kw", [("dragon", dict(width=480, height=272)), ("theater", dict(width=320, height=180, samples=4, max_reflections=4)),
                                     ("cornell_obj", dict(width=320, height=180)), ("cornell", dict())])
@pytest.mark.parametrize("pipeline", [0, 1, 3])
def test_frames_with_the_table_equal_frames_without(hip, scenes, name, kw, pipeline):
    sc = scenes(name)
    hip.update_scene(sc)
    hip.set_pipeline(pipeline)
    try:
        for use_filter in ((0,) if pipeline == 3 else (0, 1)):      # (the wavefront pipeline renders no filter G-buffers)
            p = sc.frame_params(use_filter=use_filter, **kw)
            hip.set_angle_table(True)
            a, ca, _ = hip.render(p, counters=True)
            hip.set_angle_table(False)
            b, cb, _ = hip.render(p, counters=True)
            assert bit_mismatches(a, b) == 0 and ca == cb, (name, pipeline, use_filter)
    finally:
        hip.set_angle_table(True)
        hip.set_pipeline(0)


def test_the_table_follows_the_transforms(hip, oracle, scenes):
    """the dragon scene's second transform scaled and turned: the table is made again from the new matrices — the frame equals the oracle's for the new arrays, and
    going back gives the first frame again"""
    sc = scenes("dragon")
    hip.update_scene(sc)
    p = sc.frame_params(width=320, height=180, samples=2, max_reflections=3, use_filter=0)
    first = hip.render(p)[0]
    rot = np.array(sc.arrays["rotation"], np.float32).copy().reshape(-1, 2, 12)
    c, s_ = np.cos(0.4), np.sin(0.4)
    R = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]], np.float64) * 0.7
    Ri = np.linalg.inv(R)
    for m, M in ((0, R), (1, Ri)):                       # std140 columns of the matrix and of its inverse
        for col in range(3):
            rot[1, m, 4 * col:4 * col + 3] = M[:, col]
    moved = copy.copy(sc)
    moved.arrays = dict(sc.arrays, rotation=rot.reshape(-1))
    hip.update_transforms(moved.arrays["rotation"], moved.arrays["shift"])
    got = hip.render(p)[0]
    want = oracle.render(moved, p)[0]
    assert bit_mismatches(got, want) == 0
    assert bit_mismatches(got, first) != 0
    hip.update_transforms(sc.arrays["rotation"], sc.arrays["shift"])
    assert bit_mismatches(hip.render(p)[0], first) == 0
