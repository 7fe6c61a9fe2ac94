import os, sys, time, cProfile, pstats, io, torch
sys.path.insert(0, os.getcwd())
from coponerf_amd import CoPoNeRF, synthetic as syn
from tests.helpers import to_device
dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(syn.make_full_weights(shapes), strict=True)
model = model.to(dev).eval()
inp = to_device(syn.make_inputs(1, 256, 256, 64, seed=78), dev)
with torch.no_grad():
    for _ in range(3):
        model.get_z(inp)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    model.get_z(inp); torch.cuda.synchronize()
    pr.disable()
buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(12); print(buf.getvalue()[:3000])
