"""get_z per call under the four (cudnn.benchmark, deterministic get_z) settings."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn
from tests.helpers import to_device
dev = torch.device("cuda:0")
for bench in (False, True):
    for det in (True, False):
        torch.backends.cudnn.benchmark = bench
        torch.backends.cudnn.deterministic = False
        model = CoPoNeRF.CoPoNeRF(n_view=2)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        model.load_state_dict(syn.make_full_weights(shapes), strict=True)
        model = model.to(dev).eval()
        model.deterministic_get_z = det
        from coponerf_amd import getz
        getz._DET_OWNED[0] = False
        inp = to_device(syn.make_inputs(1, 256, 256, 64, seed=41), dev)
        with torch.no_grad():
            for _ in range(3):
                model.get_z(inp)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8):
                model.get_z(inp)
            torch.cuda.synchronize()
        print(f"benchmark={bench} deterministic_get_z={det}: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms")
