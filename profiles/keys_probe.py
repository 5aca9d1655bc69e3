"""Per-tile lifetimes and classification words of the demo frame from several camera positions (stamps build):
    RM_LIB_PATH=rusty-marcher_amd/lib/variants/stamps/librusty_marcher_amd.so python3 profiles/keys_probe.py C2 gpurun_out/keys
-> <out>_<k>.bin / .bin.ext per view k (the third frame of the view: its patch order is the view's own), <out>_views.json.
profiles/sim_keys.py replays them: which dispatch keys a launch could have been ordered by, and what each would have cost."""
import json, os, sys
sys.path.insert(0, ".")
import torch
import __graft_entry__ as G, workloads
pkg = G.load_package()
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/keys"
cfg = workloads.CONFIGS[cfgname]
ctx = pkg.backend.Context(0)
scene = workloads.product_scene(pkg, cfg["scene"])
ctx.upload(scene.flatten())
w, h = cfg["width"], cfg["height"]
p = pkg.backend.make_params(workloads.FOV, float(h), float(w), cfg["max_depth"])
f64 = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
u8 = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
stream = torch.cuda.Stream()
walk = [(0., 0., 0.)] + workloads.camera_walk()[:int(os.environ.get("NVIEWS", "12")) - 1]
views = []
for k, cam in enumerate(walk):
    ctx.set_camera(pkg.Vec3f(*cam))
    for f in range(3):
        os.environ["RM_DEBUG_STAMPS"] = "%s_%d.bin" % (out, k) if f == 2 else "/dev/null"
        ctx.render_device_u8(p, f64.data_ptr(), u8.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    views.append({"k": k, "camera": cam, "workgroups": ctx.launch_stats()[0], "tail": ctx.launch_stats()[1]})
json.dump({"config": cfgname, "width": w, "height": h, "views": views}, open(out + "_views.json", "w"))
print("wrote %d views" % len(views))
