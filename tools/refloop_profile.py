#!/usr/bin/env python
"""Where the time of the reference callers' 18-call loop goes (bench.py `ref_loop`): free-running vs per-call
synchronised wall time (GPU-bound or host-bound?), and the host's cProfile of one loop.

    python tools/refloop_profile.py [--batch 2] [--chunks 18] [--top 30]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn          # noqa: E402
from coponerf_amd.evalloop import render_in_chunks            # noqa: E402


def _to(o, dev):
    if torch.is_tensor(o):
        return o.to(dev)
    if isinstance(o, dict):
        return {k: _to(v, dev) for k, v in o.items()}
    return type(o)(_to(v, dev) for v in o)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--chunks", type=int, default=18)
    ap.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=64)
    model.load_state_dict(syn.make_render_weights(), strict=False)
    model = model.to(dev).eval()
    inp = _to(syn.make_inputs(a.batch, 256, 256, 0, seed=100, full_image=True), dev)
    lat = tuple(_to(t, dev) for t in syn.make_latents(a.batch, 256, 256, seed=200))
    with torch.no_grad():
        for _ in range(2):
            model(inp, z=lat[0], rel_pose=lat[1], val=True, flow=lat[2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            model(inp, z=lat[0], rel_pose=lat[1], val=True, flow=lat[2])
        torch.cuda.synchronize()
        single = (time.perf_counter() - t0) / 5 * 1e3
    for _ in range(2):
        render_in_chunks(model, inp, a.chunks, latents=lat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        render_in_chunks(model, inp, a.chunks, latents=lat)
    torch.cuda.synchronize()
    loop = (time.perf_counter() - t0) / 5 * 1e3
    # host time alone: the same loop with the GPU idle at the start of every call is NOT measurable directly; the
    # enqueue time is what cProfile sees when the stream never blocks the host -> profile one loop
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    render_in_chunks(model, inp, a.chunks, latents=lat)
    pr.disable()
    torch.cuda.synchronize()
    print(f"batch {a.batch}: single full call {single:.2f} ms, {a.chunks}-call loop {loop:.2f} ms "
          f"(x{loop / single:.2f})")
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(a.top)
    print(buf.getvalue())


if __name__ == "__main__":
    main()
