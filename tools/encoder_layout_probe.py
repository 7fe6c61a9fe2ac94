"""Does the ResNet-34 trunk of get_z run without layout copies in channels_last?  Times the encoder eagerly at B*V = 2 and
counts its kernel launches (torch profiler) in both memory formats; checks that the outputs agree."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import getz        # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = getz.SpatialEncoder().to(dev).eval()
rgb = torch.rand(1, 2, 256, 256, 3, device=dev) * 2 - 1
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rgb = rgb.repeat(nb, 1, 1, 1, 1)


def run(model, x):
    with torch.no_grad():
        return model(x)


def bench(model, x, tag):
    for _ in range(3):
        out = run(model, x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = run(model, x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run(model, x)
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type is not None and "cuda" in str(e.device_type).lower()]
    busy = sum(e.device_time for e in ev) / 1e3 if ev else float("nan")
    print(f"{tag}: {ms:.3f} ms per call, {len(ev)} device events, busy {busy:.3f} ms")
    return out


x_nchw = getz.imagenet_normalise((rgb.flatten(0, 1).permute(0, 3, 1, 2) + 1) / 2.).contiguous()
x_cl = getz.imagenet_normalise((rgb.flatten(0, 1).permute(0, 3, 1, 2) + 1) / 2.)
print("input strides", x_cl.stride(), "channels_last:", x_cl.is_contiguous(memory_format=torch.channels_last))
a = bench(enc, x_nchw, "NCHW weights, NCHW input")
b0 = bench(enc, x_cl, "NCHW weights, NHWC input")
enc_cl = enc.to(memory_format=torch.channels_last)
b = bench(enc_cl, x_cl, "NHWC weights, NHWC input")
for i, (u, v) in enumerate(zip(a, b)):
    print(i, tuple(u.shape), "max diff", float((u - v).abs().max()), "out channels_last:", v.is_contiguous(memory_format=torch.channels_last))
