import os, sys
sys.path.insert(0, os.getcwd())
import path_tracing_amd as pt
from path_tracing_amd import workloads
for name in ("mesh1000", "textured"):
    scene, cam, label, sqrt_spp, extra = workloads.load(name)
    W, H = extra.get("width", 1920) // 4, extra.get("height", 1080) // 4
    with pt.DeviceScene(scene) as dev:
        dev.build_bvh_device()
        dev.count_walk(True)
        kw = {}
        if "max_path_segments" in extra: kw["max_path_segments"] = extra["max_path_segments"]
        dev.render(cam, pt.make_params(W, H, 2, seed=1, **kw))
        c = dev.walk_counters()
        print(name, "path rays %d rows/ray %.1f tests/ray %.2f | shadow rays %d rows/ray %.1f tests/ray %.2f" % (
            c["path_rays"], c["box_rows"] / c["path_rays"], c["surface_tests"] / c["path_rays"],
            c["shadow_rays"], c["leaf_rows"] / c["shadow_rays"], c["retry_surface_tests"] / c["shadow_rays"]))
