"""Time one training step (BASELINE config 3 per-GPU share: batch=4 pairs, 4096 rays/pair, 64 samples) on one GPU."""
import argparse
import json
import time

import torch

from coponerf_amd import CoPoNeRF, synthetic as syn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--detach-z", action="store_true", help="stop the gradient at z (times the render backward alone)")
    ap.add_argument("--flow-loss", action="store_true", help="add a flow / pose term to the loss (the reference's cycle, "
                                                            "ssim and pose losses send gradients through those heads)")
    ap.add_argument("--sort-rays", choices=("none", "rowmajor", "tiles"), default="none",
                    help="order of the query pixels inside each pair (the dataset draws them at random): locality experiment")
    ap.add_argument("--cudnn-benchmark", action="store_true", help="let the convolution library time its algorithms (MIOpen find)")
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = a.cudnn_benchmark
    dev = torch.device("cuda:0")
    model = CoPoNeRF.CoPoNeRF(n_view=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes), strict=True)
    model = model.to(dev).train()
    inp = syn.make_inputs(a.batch, 256, 256, a.rays, seed=61)
    mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
    if a.sort_rays != "none":
        uv, rgb = inp["query"]["uv"], inp["query"]["rgb"]                      # (B,1,R,2), (B,1,R,3)
        x, y = uv[..., 0].long(), uv[..., 1].long()
        key = y * 256 + x if a.sort_rays == "rowmajor" else ((y // 8) * 32 + x // 8) * 64 + (y % 8) * 8 + x % 8
        order = key.argsort(dim=-1)
        inp["query"]["uv"] = torch.gather(uv, 2, order[..., None].expand_as(uv))
        inp["query"]["rgb"] = torch.gather(rgb, 2, order[..., None].expand_as(rgb))
    inp = mv(inp)
    if not (a.detach_z or a.flow_loss):
        # the product's own step (forward, loss, backward, guard + clip, one-launch Adam): coponerf_amd.train_step.TrainStep
        from coponerf_amd.train_step import TrainStep
        step = TrainStep(model, lr=1e-5)
        step.timing = {}
        gt = inp["query"]["rgb"]
        for it in range(a.warmup + a.steps):
            if it == a.warmup:
                torch.cuda.synchronize()
                step.timing = {}
                t0 = time.perf_counter()
            res = step(inp, gt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        ph = step.timing_summary()
        # what the HOST needs to enqueue one step (device idle at the start, the call returns before the device is done):
        # the step is launch-bound as soon as the device needs less than this
        host = []
        for _ in range(3):
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            res = step(inp, gt)
            host.append((time.perf_counter() - h0) * 1e3)
        torch.cuda.synchronize()
        ph["host_enqueue_ms"] = min(host)
        print(json.dumps({"train_ms_per_step": dt * 1e3, "rays_per_s": a.batch * a.rays / dt, **{k: round(v, 3) for k, v in ph.items()},
                          "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30, "batch": a.batch, "rays_per_pair": a.rays,
                          "stepped": bool(res["stepped"]), "loss": float(res["loss"])}))
        return
    opt = torch.optim.Adam(model.parameters(), lr=1e-5)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    parts = []
    for it in range(a.warmup + a.steps):
        if it == a.warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        ev[0].record()
        z, rel, flow = model.get_z(inp, val=False)
        if a.detach_z:
            z, rel, flow = [t.detach() for t in z], rel.detach(), [f.detach() for f in flow]
        ev[1].record()
        out = model(inp, z=z, rel_pose=rel, val=False, flow=flow)
        loss = (out["rgb"] - inp["query"]["rgb"]).abs().mean()
        if a.flow_loss:
            loss = loss + 0.1 * sum(f.abs().mean() for f in flow) + 0.1 * (rel - out["gt_rel_pose"]).abs().mean()
        ev[2].record()
        loss.backward()
        ev[3].record()
        opt.step()
        if it >= a.warmup:
            torch.cuda.synchronize()
            parts.append([ev[i].elapsed_time(ev[i + 1]) for i in range(3)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    p = torch.tensor(parts).mean(0).tolist()
    print(json.dumps({"train_ms_per_step": dt * 1e3, "rays_per_s": a.batch * a.rays / dt, "get_z_ms": p[0],
                      "render_fwd_ms": p[1], "backward_ms": p[2], "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30,
                      "batch": a.batch, "rays_per_pair": a.rays, "loss": float(loss.detach())}))


if __name__ == "__main__":
    main()
