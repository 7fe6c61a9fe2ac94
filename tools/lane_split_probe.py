"""Chunk lanes of ONE render call on disjoint CU shares: does encode_hidden (L1 / write bound) of one chunk run well beside
the hid readers (read bound) of another?   python tools/lane_split_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn              # noqa: E402
from coponerf_amd.streams import CUPartition                     # noqa: E402

dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=64)
model.load_state_dict(syn.make_render_weights(), strict=False)
model = model.to(dev).eval()
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
inp = mv(syn.make_inputs(1, 256, 256, 0, seed=101, full_image=True))
z, rel, flow = syn.make_latents(1, 256, 256, seed=201)
z, rel, flow = mv(list(z)) if isinstance(z, (list, tuple)) else z, rel.to(dev), mv(list(flow))
z = [t.to(dev) for t in z]
flow = [t.to(dev) for t in flow]
eng = model._engine
eng.call_lanes = 1


def run(steps=6):
    with torch.no_grad():
        for _ in range(2):
            out = model(inp, z=z, rel_pose=rel, val=True, flow=flow)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model(inp, z=z, rel_pose=rel, val=True, flow=flow)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out["rgb"].clone()


base, ref = run()
print("one lane, whole chip", round(base, 2), flush=True)
for chunk in (16384, 8192):
    eng.chunk_rays = chunk
    eng.lanes = 1
    t1, _ = run()
    eng.lanes = 2
    eng._lane_streams = []
    t2, r2 = run()
    print(f"chunk {chunk}: one lane {t1:.2f} ms, two ordinary lanes {t2:.2f} ms, equal {torch.equal(r2, ref)}", flush=True)
    for a, b in ((128, 128), (160, 96), (96, 160)):
        part = CUPartition(a, b, dev)
        eng._lane_streams = [part.render, part.getz]
        t3, r3 = run()
        print(f"chunk {chunk}: two lanes on {a}+{b} CUs {t3:.2f} ms, equal {torch.equal(r3, ref)}", flush=True)
        eng._lane_streams = []
        part.close()
