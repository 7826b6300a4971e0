import sys, os, hashlib
sys.path.insert(0, "web-ray-tracer_amd")
from flexlight_hip import capi
from flexlight_hip.scene_io import Scene
sc = Scene.golden("dragon"); ctx = capi.Context(0); ctx.update_scene(sc)
p = sc.frame_params(use_filter=0)
img, cnt, _ = ctx.render(p, counters=True)
print(os.environ.get("FLX_HOT_ORDER", "-"), hashlib.sha256(img.tobytes()).hexdigest()[:16], cnt["closest_visits"], cnt["shadow_visits"])
