"""Which Python lines of the get_z stack cause the layout copies (aten::copy_ / contiguous / clone) and how much GPU time
they take: torch.profiler with stacks over one forward (+ backward) of get_z at batch B.
Usage: python tools/getz_copies.py [B] [--bwd]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
bwd = "--bwd" in sys.argv
dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(syn.make_full_weights(shapes), strict=True)
model = model.to(dev)
model.train() if bwd else model.eval()
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
inp = mv(syn.make_inputs(B, 256, 256, 64, seed=41))


def step():
    if bwd:
        model.zero_grad(set_to_none=True)
        z, rel, flow = model.get_z(inp)
        (sum(t.float().mean() for t in z) + rel.mean() + sum(f.mean() for f in flow)).backward()
    else:
        with torch.no_grad():
            model.get_z(inp)


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
tot = [0, 0.0]
for ev in prof.events():
    if ev.name not in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::cat", "aten::add", "aten::add_", "aten::mul", "aten::fill_", "aten::zero_"):
        continue
    dt = ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
    if dt <= 0 or ev.name == "aten::contiguous" or ev.name == "aten::clone":       # children carry the device time
        continue
    frame = next((f for f in (ev.stack or []) if "coponerf_amd" in f), "(autograd / library)")
    key = (ev.name, frame.strip()[-110:])
    agg[key][0] += 1
    agg[key][1] += dt
    tot[0] += 1
    tot[1] += dt
print(f"B={B} bwd={bwd}: {tot[0]} copy/elementwise launches, {tot[1] / 1e3:.2f} ms of device time")
for (name, frame), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t / 1e3:8.3f} ms x{n:4d}  {name:14s} {frame}")
