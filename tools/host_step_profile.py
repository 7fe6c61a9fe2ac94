"""Round 6: where the HOST spends a training step (cProfile over steps that start with the device idle): is the step
launch-bound, and which Python frames / library calls carry the enqueue time."""
import cProfile
import io
import pstats
import sys
import time

import torch

from coponerf_amd import CoPoNeRF, synthetic as syn
from coponerf_amd.train_step import TrainStep


def main():
    dev = torch.device("cuda:0")
    model = CoPoNeRF.CoPoNeRF(n_view=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes), strict=True)
    model = model.to(dev).train()
    inp = syn.make_inputs(4, 256, 256, 4096, seed=61)
    mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
    inp = mv(inp)
    step = TrainStep(model, lr=1e-5)
    gt = inp["query"]["rgb"]
    for _ in range(3):
        step(inp, gt)
    torch.cuda.synchronize()
    # phases with the device idle at the start of each step
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model(inp, val=False)
        t1 = time.perf_counter()
        loss = (gt - out["rgb"]).abs().mean()
        loss.backward()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        model.zero_grad(set_to_none=True)
        print(f"host: forward enqueue {1e3 * (t1 - t0):.1f} ms, backward enqueue {1e3 * (t2 - t1):.1f} ms, device tail {1e3 * (t3 - t2):.1f} ms")
    if "--ops" in sys.argv:
        # host (CPU) time per operator over three steps, the autograd thread included
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            for _ in range(3):
                torch.cuda.synchronize()
                step(inp, gt)
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=70, max_name_column_width=60))
        return
    pr = cProfile.Profile()
    for _ in range(3):
        torch.cuda.synchronize()
        pr.enable()
        step(inp, gt)
        pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
        print(s.getvalue()[:9000])


if __name__ == "__main__":
    main()
