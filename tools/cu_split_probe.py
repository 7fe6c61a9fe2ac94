"""How to split the chip between the render pass and get_z (coponerf_amd/streams.py, pipeline.py).
Per split: the render pass alone on its share, get_z alone on its share, and the pipelined image loop; against the
serial loop and the two-ordinary-streams pipeline.  Also checks that the partitioned results equal the serial ones bit
for bit.   python tools/cu_split_probe.py [--splits 224:32,208:48,192:64,176:80,160:96]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn              # noqa: E402
from coponerf_amd.pipeline import render_images                  # noqa: E402
from coponerf_amd.streams import CUPartition, stream_cus         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--splits", default="224:32,208:48,192:64,176:80,160:96")
ap.add_argument("--images", type=int, default=8)
ap.add_argument("--detail", action="store_true", help="per-stream durations of get_z and of the render pass inside the pipelined loop")
ap.add_argument("--graph", action="store_true", help="get_z as a HIP-graph replay")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = CoPoNeRF.CoPoNeRF(n_view=2)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict(syn.make_full_weights(shapes), strict=True)
model = model.to(dev).eval()
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
pairs = [mv(syn.make_inputs(1, 256, 256, 0, seed=300 + i, full_image=True)) for i in range(4)]
seq = (pairs + pairs)[:a.images]
R = 256 * 256


def timed(fn, reps=1):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


res = {}
with torch.no_grad():
    def serial():
        outs = []
        for p in seq:
            z, rp, fl = model.get_z(p)
            outs.append(model(p, z=z, rel_pose=rp, val=True, flow=fl)["rgb"])
        return outs
    ref = [o.clone() for o in serial()]
    res["serial_ms_per_image"] = timed(serial) / len(seq)
    res["two_streams_ms_per_image"] = timed(lambda: [0 for _ in render_images(model, seq)]) / len(seq)
    z, rp, fl = model.get_z(pairs[0])
    res["render_alone_ms"] = timed(lambda: model(pairs[0], z=z, rel_pose=rp, val=True, flow=fl), 5)
    res["getz_alone_ms"] = timed(lambda: model.get_z(pairs[0]), 5)
    for sp in a.splits.split(","):
        rc, gc = (int(x) for x in sp.split(":"))
        part = CUPartition(rc, gc, dev)
        assert stream_cus(part.render) == rc and stream_cus(part.getz) == gc
        lanes, model._engine.call_lanes = model._engine.call_lanes, 1
        with torch.cuda.stream(part.render):
            r_ms = timed(lambda: model(pairs[0], z=z, rel_pose=rp, val=True, flow=fl), 5)
        with torch.cuda.stream(part.getz):
            g_ms = timed(lambda: model.get_z(pairs[0]), 5)
        model._engine.call_lanes = lanes
        part.close()
        if a.detail:
            ev = {"getz": [], "render": []}
            g0, f0 = model.get_z, model.forward

            def wrap(fn, key):
                def inner(*args, **kw):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    r = fn(*args, **kw)
                    e1.record()
                    ev[key].append((e0, e1))
                    return r
                return inner
            model.get_z, model.forward = wrap(g0, "getz"), wrap(f0, "render")
            for _ in render_images(model, seq, cu_split=(rc, gc)):
                pass
            torch.cuda.synchronize()
            del model.get_z, model.forward
            print(sp, {k: [round(e0.elapsed_time(e1), 1) for e0, e1 in v] for k, v in ev.items()}, flush=True)
        outs = [o["rgb"].clone() for _, o in render_images(model, seq, cu_split=(rc, gc), graph=a.graph)]
        torch.cuda.synchronize()
        same = all(torch.equal(x, y) for x, y in zip(outs, ref))
        ms = timed(lambda: [0 for _ in render_images(model, seq, cu_split=(rc, gc), graph=a.graph)]) / len(seq)
        res[sp] = {"render_on_share_ms": r_ms, "getz_on_share_ms": g_ms, "pipelined_ms_per_image": ms,
                   "image_rays_per_s": R / ms * 1e3, "bit_equal_to_serial": same}
        print(sp, res[sp], flush=True)
print(json.dumps(res))
