import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coponerf_amd import CoPoNeRF, synthetic as syn
from tests.helpers import to_device
dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(syn.make_full_weights(shapes), strict=True)
model = model.to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
inp = to_device(syn.make_inputs(B, 256, 256, 64, seed=41), dev)
with torch.no_grad():
    for _ in range(2): model.get_z(inp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): z, rel, flow = model.get_z(inp)
    torch.cuda.synchronize(); print("get_z ms per call (B=%d):" % B, (time.perf_counter() - t0) / 5 * 1e3)
