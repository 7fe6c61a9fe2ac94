"""Where does the wall time of one render step go?  CPU timers with device syncs around the phases of forward()."""
import time
import torch
from coponerf_amd import CoPoNeRF, synthetic as syn
from coponerf_amd import render as R
from coponerf_amd.aux_outputs import aux_outputs

dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2)
model.load_state_dict(syn.make_render_weights(seed=7), strict=False)
model = model.to(dev).eval()
inp = syn.make_inputs(1, 256, 256, 65536, seed=3, full_image=True)
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
inp = mv(inp)
z, rel, flow = syn.make_latents(1, 256, 256, seed=4)
z = [t.to(dev) for t in z]; rel = rel.to(dev); flow = mv(flow) if isinstance(flow, dict) else [f.to(dev) for f in flow]

marks = []
def tick(name):
    torch.cuda.synchronize(); marks.append((name, time.perf_counter()))

orig_build = R.build_camera_block
def build(*a, **k):
    tick("enter build_camera_block")
    out = orig_build(*a, **k)
    tick("build_camera_block (host 4x4 algebra)")
    return out
R.build_camera_block = build
orig_call = R.call
seen = set()
def call(name, *a):
    if name in ("cpn_sample_geometry", "cpn_mask_rgb") or (name in ("cpn_gather_rows", "cpn_encode_hidden") and "g" not in seen):
        tick("before " + name)
        seen.add("g") if name in ("cpn_gather_rows", "cpn_encode_hidden") else None
    if name == "cpn_linear_f32" and a[10] == 32 and "phi" not in seen:   # K = 32: phi.lin_in
        tick("chunks done"); seen.add("phi")
    return orig_call(name, *a)
R.call = call
import coponerf_amd.CoPoNeRF as CM
orig_aux = CM.aux_outputs
def aux(*a, **k):
    tick("before aux_outputs")
    out = orig_aux(*a, **k)
    tick("aux_outputs")
    return out
CM.aux_outputs = aux
orig_empty = torch.empty
def empty(*a, **k):
    if k.get("pin_memory"):
        tick("before pinned alloc")
        out = orig_empty(*a, **k)
        tick("pinned alloc")
        return out
    return orig_empty(*a, **k)
R.torch.empty = empty

with torch.no_grad():
    for it in range(4):
        marks.clear(); seen.clear()
        tick("start")
        out = model(inp, z=z, rel_pose=rel, val=True, flow=flow)
        tick("end")
t0 = marks[0][1]
prev = t0
for n, t in marks:
    print(f"{(t - t0) * 1e3:8.2f} ms  (+{(t - prev) * 1e3:6.2f})  {n}")
    prev = t
