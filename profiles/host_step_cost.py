"""Dev helper: CPU cost of the pieces of one distributed bench step (world size 1, RCCL):
the render launch through the C ABI, the asynchronous all-gather, its wait, the
de-interleave at the consumer.  usage (GPU box): python3 profiles/host_step_cost.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import torch, torch.distributed as dist
import __graft_entry__ as G, workloads
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
pkg = G.load_package(); L = pkg.lib()
ctx = pkg.backend.Context(0); ctx.upload(pkg.Scene.create_default().flatten())
w, h = 1920, 1080
p = pkg.backend.make_params(1.5, float(h), float(w), 5, (0, h // 32, 1)); p.flags |= 4
f64 = torch.zeros((h, w, 3), dtype=torch.float64, device=dev)
g = torch.zeros((h // 32 * 32, w, 3), dtype=torch.uint8, device=dev)
disp = torch.zeros_like(g)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
sp = C.c_void_p(s.cuda_stream); fp, gp = C.c_void_p(f64.data_ptr()), C.c_void_p(g.data_ptr()); pr = C.byref(p)
T = {"render": 0., "all_gather": 0., "wait": 0., "deinterleave": 0.}
n = 300
for it in range(n + 20):
    if it == 20:
        torch.cuda.synchronize(); T = dict.fromkeys(T, 0.); t_all = time.perf_counter()
    t0 = time.perf_counter(); L.rm_render_device_u8(ctx.ptr, pr, fp, gp, sp)
    t1 = time.perf_counter(); wk = dist.all_gather_into_tensor(g, g, async_op=True)
    t2 = time.perf_counter(); wk.wait()
    t3 = time.perf_counter(); workloads.deinterleave_rows(g, 3, disp)
    t4 = time.perf_counter()
    T["render"] += t1 - t0; T["all_gather"] += t2 - t1; T["wait"] += t3 - t2; T["deinterleave"] += t4 - t3
torch.cuda.synchronize(); tot = time.perf_counter() - t_all
print("per step: wall %.1f us;  CPU us: %s" % (tot / n * 1e6, "  ".join("%s %.1f" % (k, v / n * 1e6) for k, v in T.items())))
dist.destroy_process_group()
