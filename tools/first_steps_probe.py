import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from coponerf_amd import CoPoNeRF, synthetic as syn
dev = torch.device("cuda:0")
H, S = 256, 64
model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
model.load_state_dict(syn.make_render_weights(), strict=False)
model = model.to(dev).eval()
inp = syn.make_inputs(1, H, H, 0, seed=100, full_image=True)
z, rel, flow = syn.make_latents(1, H, H, seed=200)
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else type(o)(mv(v) for v in o))
inp, z, rel, flow = mv(inp), mv(z), rel.to(dev), mv(flow)
model._engine.call_lanes = 1
ts = []
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        model(inp, z=z, rel_pose=rel, val=True, flow=flow)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join("%.1f" % t for t in ts))
