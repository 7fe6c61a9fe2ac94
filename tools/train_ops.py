"""Where the small launches of a training step come from: torch.profiler over one step (batch 4 x 4096 rays), device time
aggregated by (aten op, input shapes) — the shapes identify the layer for the autograd-side launches, which carry no
Python stack.  Usage: python tools/train_ops.py [--top N] [--all | --fns]   (default: elementwise / copy / reduce ops only)"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--top", type=int, default=60)
ap.add_argument("--all", action="store_true")
ap.add_argument("--kernels", action="store_true", help="device time by kernel name")
ap.add_argument("--fns", action="store_true", help="device time (inclusive) by autograd Function node instead of by aten op")
ap.add_argument("--only", default="", help="comma-separated aten op names (e.g. copy_,add): only those, plus per-op totals")
ap.add_argument("--getz", type=int, default=0, metavar="B", help="profile one inference get_z at batch B instead of a training step")
a = ap.parse_args()
dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(syn.make_full_weights(shapes), strict=True)
model = model.to(dev)
model.eval() if a.getz else model.train()
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
inp = mv(syn.make_inputs(a.getz or 4, 256, 256, 4096, seed=61))
opt = torch.optim.Adam(model.parameters(), lr=1e-5)


def step():
    if a.getz:
        with torch.no_grad():
            model.get_z(inp)
        return
    opt.zero_grad(set_to_none=True)
    out = model(inp, val=False)
    (out["rgb"] - inp["query"]["rgb"]).abs().mean().backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
SMALL = ("copy_", "add", "add_", "mul", "mul_", "fill_", "zero_", "sum", "div", "sub", "neg", "where", "clamp", "cat",
         "threshold_backward", "to", "_to_copy", "sqrt", "rsqrt", "exp", "mean", "index", "masked_fill", "gelu",
         "gelu_backward", "native_layer_norm", "native_layer_norm_backward", "permute", "relu", "sigmoid", "addcmul_",
         "addcdiv_", "lerp_", "_foreach_add_", "_foreach_mul_", "_foreach_addcmul_", "_foreach_addcdiv_", "_foreach_sqrt",
         "_foreach_div_", "_foreach_lerp_", "_foreach_norm")
if a.kernels:
    ker = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            ker[ev.name[:110]][0] += 1
            ker[ev.name[:110]][1] += ev.device_time_total
    print(f"{sum(t for _, t in ker.values()) / 1e3:.2f} ms of kernel time")
    for name, (n, t) in sorted(ker.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"{t / 1e3:8.3f} ms x{n:4d}  {name}")
    sys.exit(0)
if a.fns:
    fn = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.name.startswith("aten::") or ev.device_type != torch.autograd.DeviceType.CPU:
            continue
        if "Backward" in ev.name or ev.name.endswith("Fn") or ev.name.startswith("_"):
            fn[ev.name][0] += 1
            fn[ev.name][1] += getattr(ev, "device_time_total", 0) or 0
    for name, (n, t) in sorted(fn.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"{t / 1e3:8.3f} ms x{n:4d}  {name}")
    sys.exit(0)
agg = collections.defaultdict(lambda: [0, 0.0, ""])
tot = 0.0
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    op = ev.name[6:]
    if a.only:
        if op not in a.only.split(","):
            continue
    elif not a.all and op not in SMALL:
        continue
    dt = getattr(ev, "self_device_time_total", None)
    if dt is None:
        dt = ev.self_cuda_time_total
    if dt <= 0:
        continue
    frame = next((f.strip()[-70:] for f in (ev.stack or []) if "coponerf_amd" in f), "")
    key = (op, str(ev.input_shapes)[:120], frame)
    agg[key][0] += 1
    agg[key][1] += dt
    tot += dt
print(f"{tot / 1e3:.2f} ms of device time in the selected ops")
per = collections.defaultdict(lambda: [0, 0.0])
for (op, _, _), (n, t, _) in agg.items():
    per[op][0] += n
    per[op][1] += t
print("  ".join(f"{op}: {t / 1e3:.2f} ms x{n}" for op, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:12]))
for (op, shp, frame), (n, t, _) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"{t / 1e3:8.3f} ms x{n:4d}  {op:26s} {shp}  {frame}")
