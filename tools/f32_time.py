"""One 256 x 256 x 64 image in the reference-arithmetic mode (RenderEngine.precision = "f32"), timed; under
`rocprofv3 --kernel-trace --stats` the per-kernel sums of the mode.   python tools/f32_time.py [--steps 3] [--layers]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn      # noqa: E402
from tests.helpers import to_device                       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--layers", action="store_true", help="round 5's layer-by-layer form")
ap.add_argument("--chunk", type=int, default=16384)
a = ap.parse_args()
dev = torch.device("cuda:0")
H, S = 256, 64
model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
model.load_state_dict(syn.make_render_weights(), strict=False)
model = model.to(dev).eval()
inp = to_device(syn.make_inputs(1, H, H, H * H, seed=3), dev)
z, rel, flow = syn.make_latents(1, H, H, seed=4)
z, rel, flow = to_device(z, dev), rel.to(dev), to_device(flow, dev)
eng = model._engine
eng.precision, eng.f32_tables, eng.f32_chunk_rays = "f32", not a.layers, a.chunk
eng.call_lanes = 1                      # one stream: per-kernel times are not inflated by a co-running call
with torch.no_grad():
    model(inp, z=z, rel_pose=rel, val=True, flow=flow)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        model(inp, z=z, rel_pose=rel, val=True, flow=flow)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"f32 mode ({'layers' if a.layers else 'tables'}): {dt * 1e3:.1f} ms per image, {H * H / dt / 1e3:.0f} k rays/s")
