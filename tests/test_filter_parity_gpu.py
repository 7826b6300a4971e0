"""The GPU's denoise chain (flx_filter_planes_device: k_filter_first / _second / _final over RGBA8 planes, LDS tiles, batched gathers) against literal answers
computed from the three filter shaders and the host's pass schedule (tests/golden/filter_kat.json.gz, tests/analysis/make_filter_kat.py — SURVEY.md 8a F0 - F3):
bit for bit without the tone mapping, within 2 ulp with it (pow)."""
import numpy as np
import pytest

from test_oracle_kat import _filter_kat_cases, assert_filter_kat, filter_kat_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", range(5))
def test_filter_chain_literal(hip, scenes, k):
    import torch
    case = _filter_kat_cases()[k]
    p, _, want = filter_kat_inputs(case)
    W, H = case["width"], case["height"]
    hip.update_scene(scenes("cornell"))                  # (the chain reads no scene; a context renders only with one)
    planes = np.stack([np.array(pl, np.uint8).reshape(H, W, 4).view(np.uint32)[..., 0] for pl in case["planes"]]).astype(np.uint32)      # RGBA8: R in the low byte
    d_planes = torch.as_tensor(planes.view(np.int32), device="cuda").contiguous()
    out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    hip.filter_planes_device(p, d_planes.data_ptr(), out.data_ptr())
    hip.sync()
    assert_filter_kat(out.cpu().numpy(), want, case["hdr"], "GPU, case %d" % k)
