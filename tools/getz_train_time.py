"""get_z forward + backward alone (batch of pairs), for profiling the training path of the UFC / encoder side."""
import argparse, json, time
import torch
from coponerf_amd import CoPoNeRF, synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--warmup", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(syn.make_full_weights(shapes), strict=True)
model = model.to(dev).train()
inp = syn.make_inputs(a.batch, 256, 256, 64, seed=61)
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
inp = mv(inp)
fw, bw = [], []
for it in range(a.warmup + a.steps):
    model.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    z, rel, flow = model.get_z(inp, val=False)
    loss = sum(t.float().mean() for t in z) + rel.mean() + sum(f.mean() for f in flow)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if it >= a.warmup:
        fw.append(t1 - t0); bw.append(t2 - t1)
print(json.dumps({"get_z_fwd_ms": 1e3 * sum(fw) / len(fw), "get_z_bwd_ms": 1e3 * sum(bw) / len(bw), "batch": a.batch}))
