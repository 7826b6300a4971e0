"""CPU-side checks of the C ABI: the library loads without a GPU, exports every symbol the header
declares, and refuses to create a context without a device (no CPU fallback).  No compute calls."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(headers=("flexlight_hip.h", "flexlight_hip_debug.h"), experiments=False):
    """every function the boundary (flexlight_hip.h) and the instrumentation header (flexlight_hip_debug.h) declare; what the latter keeps under
    #ifdef FLX_EXPERIMENTS exists in `make EXPERIMENTS=1`'s library only"""
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        if not experiments:
            text = re.sub(r"#ifdef FLX_EXPERIMENTS.*?#endif", "", text, flags=re.S)
        names.update(re.findall(r"\b(flx_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_header_symbols_are_exported():
    from flexlight_hip import capi
    names = declared_functions(experiments=capi.has_experiments())
    assert len(names) >= 20
    for name in names:
        assert hasattr(capi.LIB, name), "libflexlight_hip.so does not export %s" % name
    every = declared_functions(experiments=True)
    assert sorted(capi.EXPORTS) == [n for n in every if n in capi.EXPORTS]
    missing = [n for n in every if n not in capi.EXPORTS]
    assert not missing, "capi.EXPORTS lacks %s" % missing
    # the drop-in boundary stays small (SURVEY.md 8b sketches a dozen calls; round 4's header had grown to 110): what a host binds is flexlight_hip.h alone
    boundary = declared_functions(headers=("flexlight_hip.h",))
    assert len(boundary) <= 80, len(boundary)
    assert not [n for n in boundary if n.startswith("flx_debug_") or n.startswith("flx_get_") or "chain_" in n], boundary
    # ... and the shipped library exports nothing that no header declares
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], stdout=subprocess.PIPE, universal_newlines=True).stdout
    exported = sorted(set(re.findall(r" T (flx_[a-z0-9_]+)$", out, flags=re.M)))
    if exported:
        assert not [n for n in exported if n not in names], [n for n in exported if n not in names]


def test_no_cpu_fallback():
    import torch
    from flexlight_hip import capi
    if torch.cuda.is_available():
        return                      # on a GPU box a context is legitimately created
    try:
        capi.Context(0)
    except capi.FlexLightHipError as e:
        assert "no HIP device" in str(e) or "failed" in str(e)
    else:
        raise AssertionError("a context was created without a GPU")


def test_tile_policy_helpers():
    from flexlight_hip import capi
    from flexlight_hip.scene_io import FrameParams
    p = FrameParams()
    p.width, p.height = 64, 37
    assert capi.Context.tile_row_count(p) == 37
    seen = []
    for idx in range(3):
        p.tile_rows, p.tile_index, p.tile_count = 8, idx, 3
        rows = capi.Context.tile_rows(p)
        assert rows == [y for y in range(37) if (y // 8) % 3 == idx]
        seen += rows
    assert sorted(seen) == list(range(37))
    # more ranks than strips (37 rows in strips of 8 = 5 strips, 8 ranks): the ranks past the last strip own nothing
    for idx in range(8):
        p.tile_rows, p.tile_index, p.tile_count = 8, idx, 8
        assert capi.Context.tile_row_count(p) == (8 if idx < 4 else (5 if idx == 4 else 0))


def test_struct_layouts_match_the_c_compiler(tmp_path):
    """ctypes mirrors of the header structs against gcc's own sizeof / offsetof (a drift corrupts every call)."""
    import subprocess
    from flexlight_hip.scene_io import Counters, FrameParams, GBuffers, SceneView
    src = tmp_path / "layout.c"
    src.write_text(r"""
#include <stdio.h>
#include <stddef.h>
#include "flexlight_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu\n", sizeof(flx_frame_params), sizeof(flx_counters), sizeof(flx_gbuffers), sizeof(flx_scene_view));
  printf("%zu %zu %zu %zu %zu\n", offsetof(flx_frame_params, view_matrix), offsetof(flx_frame_params, ambient),
         offsetof(flx_frame_params, tile_rows), offsetof(flx_scene_view, atlas), offsetof(flx_scene_view, atlas_h));
  return 0;
}
""")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes, offs = [list(map(int, l.split())) for l in subprocess.check_output([str(exe)]).decode().splitlines()]
    assert sizes == [C.sizeof(FrameParams), C.sizeof(Counters), C.sizeof(GBuffers), C.sizeof(SceneView)]
    assert offs == [FrameParams.view_matrix.offset, FrameParams.ambient.offset, FrameParams.tile_rows.offset,
                    SceneView.atlas.offset, SceneView.atlas_h.offset]


def test_batch_limit_matches_the_header():
    from flexlight_hip import capi
    text = open(os.path.join(ROOT, "include", "flexlight_hip.h")).read()
    assert int(re.search(r"#define FLX_MAX_BATCH_FRAMES (\d+)", text).group(1)) == capi.MAX_BATCH_FRAMES
    dev = open(os.path.join(ROOT, "web-ray-tracer_amd", "csrc", "flx_device.h")).read()
    assert int(re.search(r"#define FLX_MAX_BATCH (\d+)", dev).group(1)) == capi.MAX_BATCH_FRAMES


def test_the_drivers_build_check_accepts_the_shipped_library():
    """__graft_entry__.build() ends with this check: every function the bindings list is exported, except the chain of launches' five, which are in the experiments library only
    (a check that asked for those too would fail every round's build step)"""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    from flexlight_hip import capi
    entry.check_exports()
    if not capi.has_experiments():
        assert all(not hasattr(capi.LIB, n) for n in capi.EXPERIMENTS_ONLY)
