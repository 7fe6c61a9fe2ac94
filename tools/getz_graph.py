"""Can get_z (B = 1, ~1 100 launches) be captured in a HIP graph, and what does a replay cost?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn      # noqa: E402

dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(syn.make_full_weights(shapes), strict=True)
model = model.to(dev).eval()
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
inp = mv(syn.make_inputs(1, 256, 256, 0, seed=300, full_image=True))
inp2 = mv(syn.make_inputs(1, 256, 256, 0, seed=301, full_image=True))
with torch.no_grad():
    for _ in range(3):
        ref = model.get_z(inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        model.get_z(inp)
    torch.cuda.synchronize()
    print("eager get_z ms", (time.perf_counter() - t0) / 5 * 1e3)
    ref2 = model.get_z(inp2)
    static = mv(syn.make_inputs(1, 256, 256, 0, seed=300, full_image=True))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            model.get_z(static)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model.get_z(static)
    torch.cuda.synchronize()

    def copy_in(dst, src):
        if torch.is_tensor(dst):
            dst.copy_(src)
        elif isinstance(dst, dict):
            for k in dst:
                copy_in(dst[k], src[k])
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print("graph replay ms", (time.perf_counter() - t0) / 5 * 1e3)
    copy_in(static, inp2)
    g.replay()
    torch.cuda.synchronize()
    z, rel, flow = out
    d = max(float((a - b).abs().max()) for a, b in zip(z, ref2[0]))
    print("replay on new inputs vs eager: z max abs diff", d, "rel_pose diff", float((rel - ref2[1]).abs().max()))
