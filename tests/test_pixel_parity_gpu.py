"""GPU frames against literal WHOLE PIXELS (tests/golden/pixel_kat.json.gz, tests/analysis/make_pixel_kat.py: lightTrace with all its bounces and main(),
fragment:464-646, run from the shader text over the reference's arrays; the primary hit is the one input that is not from the text): every pipeline's colour
without the filter, and the five G-buffers the filter frame's trace kernel writes — bit for bit, without the oracle in between."""
import numpy as np
import pytest

from test_oracle_kat import _pixel_kat_cases, assert_pixel_kat, pixel_kat_expectations

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", range(5))
def test_whole_pixels_literal(hip, scenes, k):
    case = _pixel_kat_cases()[k]
    sc, p, rows, want_color, want_gb = pixel_kat_expectations(case, scenes)
    hip.update_scene(sc)
    p.use_filter = 1
    _, _, gbs = hip.render(p, gbuffers=True)                       # the per-pixel kernel with the filter's outputs
    p.use_filter = 0
    try:
        for pipeline, organisation, front in ((1, 0, 1), (2, 0, 1), (3, 1, 0), (3, 2, 0), (3, 2, 2), (3, 1, 3), (0, 0, 1)):
            hip.set_pipeline(pipeline)
            hip.set_wavefront_organisation(organisation)
            hip.set_frame_front(front)
            frame, _, _ = hip.render(p)
            assert_pixel_kat(case, rows, want_color, want_gb if pipeline == 1 else {}, frame, gbs, "GPU pipeline %d organisation %d front %d, case %d" % (pipeline, organisation, front, k))
    finally:
        hip.set_pipeline(0)
        hip.set_wavefront_organisation(0)
        hip.set_frame_front(1)
