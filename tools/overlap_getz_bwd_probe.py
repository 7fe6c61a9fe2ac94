"""Round 6: can HBM-bound work (the render layers' weight gradients, ~5 ms of streaming kernels) hide under get_z's forward /
backward (thousands of small kernels that leave most of the chip idle)?  Times get_z forward + backward alone, a streaming
load alone (copies of a 2 GB tensor on a side stream), and both at once."""
import json
import time

import torch

from coponerf_amd import CoPoNeRF, synthetic as syn


def main():
    dev = torch.device("cuda:0")
    model = CoPoNeRF.CoPoNeRF(n_view=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes), strict=True)
    model = model.to(dev).train()
    inp = syn.make_inputs(4, 256, 256, 4096, seed=61)
    mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
    inp = mv(inp)
    side = torch.cuda.Stream(device=dev)
    big = torch.empty(1 << 29, dtype=torch.float32, device=dev)          # 2 GB
    dst = torch.empty_like(big)

    def getz_step():
        model.zero_grad(set_to_none=True)
        z, rel, flow = model.get_z(inp, val=False)
        loss = sum(t.float().square().mean() for t in z) + rel.square().mean()
        loss.backward()

    def load(n):
        with torch.cuda.stream(side):
            for _ in range(n):
                dst.copy_(big, non_blocking=True)

    def timed(fn, reps=5):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    res = {"getz_fwd_bwd_ms": timed(getz_step)}
    for n in (4, 8):
        res[f"load{n}_ms"] = timed(lambda: load(n))

        def both():
            side.wait_stream(torch.cuda.current_stream())
            load(n)
            getz_step()
            torch.cuda.current_stream().wait_stream(side)
        res[f"both{n}_ms"] = timed(both)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
