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
    ap.add_argument("--alt-streams", type=int, default=0)
    ap.add_argument("--trace-only", action="store_true", help="three loops and exit (under rocprofv3 --kernel-trace)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=64)
    model.load_state_dict(syn.make_render_weights(), strict=False)
    model = model.to(dev).eval()
    inp = _to(syn.make_inputs(a.batch, 256, 256, 0, seed=100, full_image=True), dev)
    lat = tuple(_to(t, dev) for t in syn.make_latents(a.batch, 256, 256, seed=200))
    if a.alt_streams:
        # potential of running consecutive forward() calls on alternating HIP streams (the calls are independent)
        qry = inp["query"]
        rgb_full, uv_full = qry["rgb"], qry["uv"]
        streams = [torch.cuda.Stream() for _ in range(a.alt_streams)]
        with torch.no_grad():
            for it in range(8):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                keep = []
                for i, (r_, u_) in enumerate(zip(torch.chunk(rgb_full, a.chunks, dim=2), torch.chunk(uv_full, a.chunks, dim=2))):
                    qry["rgb"], qry["uv"] = r_, u_
                    st = streams[i % len(streams)]
                    with torch.cuda.stream(st):
                        keep.append(model(inp, z=lat[0], rel_pose=lat[1], val=True, flow=lat[2]))
                qry["rgb"], qry["uv"] = rgb_full, uv_full
                torch.cuda.synchronize()
                print(f"{a.alt_streams} alternating streams: loop {1e3 * (time.perf_counter() - t0):.2f} ms")
                del keep
        return
    if a.trace_only:
        for _ in range(3):
            render_in_chunks(model, inp, a.chunks, latents=lat)
            torch.cuda.synchronize()
        return
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
    import gc
    gcs = []
    gc.callbacks.append(lambda ph, info: gcs.append((ph, info.get("generation"), time.perf_counter())))
    per = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r_ = render_in_chunks(model, inp, a.chunks, latents=lat)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        del r_
        t3 = time.perf_counter()
        per.append((round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2), round((t3 - t2) * 1e3, 2)))
    print("per loop (ms): (loop returned, drain, free result):", per)
    dur = [(g, round((t1 - t0) * 1e3, 2)) for (p0, g, t0), (p1, _, t1) in zip(gcs[::2], gcs[1::2])]
    print("gc passes during those loops (generation, ms):", dur)
    gc.callbacks.pop()
    # host time alone: the same loop with the GPU idle at the start of every call is NOT measurable directly; the
    # enqueue time is what cProfile sees when the stream never blocks the host -> profile one loop
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    render_in_chunks(model, inp, a.chunks, latents=lat)
    pr.disable()
    torch.cuda.synchronize()
    # statement-level wall clock of one loop, free-running, and host-only time of forward() with the GPU drained first
    qry = inp["query"]
    rgb_full, uv_full = qry["rgb"], qry["uv"]
    T = {"chunk": 0.0, "forward": 0.0, "del_cpu": 0.0, "forward_gpu_idle": 0.0, "drain_after": 0.0}
    with torch.no_grad():
        for mode in ("free", "drained"):
            t = time.perf_counter()
            rc, uc = torch.chunk(rgb_full, a.chunks, dim=2), torch.chunk(uv_full, a.chunks, dim=2)
            T["chunk"] += time.perf_counter() - t
            keep = []
            for r_, u_ in zip(rc, uc):
                qry["rgb"], qry["uv"] = r_, u_
                if mode == "drained":
                    torch.cuda.synchronize()
                t = time.perf_counter()
                o = model(inp, z=lat[0], rel_pose=lat[1], val=True, flow=lat[2])
                dt = time.perf_counter() - t
                if mode == "drained":
                    T["forward_gpu_idle"] += dt
                    t = time.perf_counter()
                    torch.cuda.synchronize()
                    T["drain_after"] += time.perf_counter() - t
                else:
                    T["forward"] += dt
                    t = time.perf_counter()
                    for k in ("z", "coords", "at_wts"):
                        del o[k]
                    o["pixel_val"] = o["pixel_val"].cpu()
                    T["del_cpu"] += time.perf_counter() - t
                keep.append(o)
            qry["rgb"], qry["uv"] = rgb_full, uv_full
            torch.cuda.synchronize()
            del keep
    print({k: round(v * 1e3, 3) for k, v in T.items()}, "(ms per loop; forward_gpu_idle = host time of the 18 calls "
          "with an empty GPU queue, drain_after = GPU time left after the call returned)")
    print(f"batch {a.batch}: single full call {single:.2f} ms, {a.chunks}-call loop {loop:.2f} ms "
          f"(x{loop / single:.2f})")
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(a.top)
    print(buf.getvalue())


if __name__ == "__main__":
    main()
