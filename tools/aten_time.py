"""GPU time of one training step by ATen op and input shape (torch.profiler), for the launches that are NOT this
package's kernels: which copies / elementwise ops / reductions of the autograd graph are worth a kernel of their own.
Forward ops carry the innermost coponerf_amd frame; backward ops the name of their autograd node.
    python tools/aten_time.py [--top 70] [--getz]      (--getz: one inference get_z call instead of a training step)"""
import argparse
import collections
import os
import re
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--getz", action="store_true")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--kernels", default="", help="regex: only events that launched a kernel whose name matches")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = CoPoNeRF.CoPoNeRF(n_view=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes), strict=True)
    model = model.to(dev)
    model = model.eval() if a.getz else model.train()
    inp = syn.make_inputs(1 if a.getz else a.batch, 256, 256, 4096, seed=61)
    mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
    inp = mv(inp)
    if not a.getz:
        from coponerf_amd.train_step import TrainStep
        train = TrainStep(model, lr=1e-5)                         # the product's own step (guard, clip, one-launch Adam)

    def step():
        if a.getz:
            with torch.no_grad():
                model.get_z(inp, val=True)
            return
        train(inp, inp["query"]["rgb"])

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()

    # kernel events -> the CPU op that launched them (linked by correlation through the profiler's event tree)
    agg = collections.defaultdict(lambda: [0.0, 0, 0.0])
    total = 0.0
    for ev in prof.events():
        if not ev.kernels:
            continue
        t = sum(k.duration for k in ev.kernels)                   # us
        names = {k.name for k in ev.kernels}
        ours = all(("at::" not in n and "Cijk" not in n and "rocclr" not in n and "igemm" not in n and "miopen" not in n.lower()
                    and "SubTensor" not in n and "rocblas" not in n) for n in names)
        total += t
        if ours:
            continue
        if a.kernels and not any(re.search(a.kernels, n) for n in names):
            continue
        frame = ""
        for fs in ev.stack or []:
            if "coponerf_amd" in fs:
                frame = fs.split("coponerf_amd/")[-1][:60]
                break
        par = ev.cpu_parent
        while par is not None and not frame:
            if par.name.startswith("autograd::engine::evaluate_function: "):
                frame = "bwd " + par.name.split(": ", 1)[1][:50]
            par = par.cpu_parent
        key = (ev.name, str(ev.input_shapes)[:90], frame)
        agg[key][0] += t
        agg[key][1] += len(ev.kernels)
        agg[key][2] = max(agg[key][2], max(k.duration for k in ev.kernels))
    lib = sum(v[0] for v in agg.values())
    print(f"GPU time of the step {total / 1e3:.2f} ms, of which library / ATen kernels {lib / 1e3:.2f} ms")
    by_op = collections.Counter()
    for (n, _, _), v in agg.items():
        by_op[n] += v[0]
    print("by op:", ", ".join(f"{n} {t / 1e3:.2f}" for n, t in by_op.most_common(25)))
    by_site = collections.defaultdict(lambda: [0.0, 0])
    for (n, _, frame), v in agg.items():
        by_site[(n, frame)][0] += v[0]
        by_site[(n, frame)][1] += v[1]
    print("by op and site:")
    for (n, frame), (t, c) in sorted(by_site.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print(f"{t / 1e3:7.3f} ms x{c:4d}  {n:32s} {frame}")
    print("by op, shape and site:")
    for (n, shp, frame), (t, c, mx) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print(f"{t / 1e3:7.3f} ms x{c:4d} max {mx:7.1f} us  {n:32s} {shp:90s} {frame}")


if __name__ == "__main__":
    main()
