"""cpn_trunk_conv_bn_act against the library convolution + cpn_bn_act on the layer3 / layer4 shapes of the ResNet-34 trunk at
1, 2 and 4 stereo pairs.   python tools/trunk_conv_bench.py"""
import json
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd.getz import _bn_act, _trunk_conv      # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.deterministic = True


def timed(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    for pairs in (1, 2, 4):
        N = 2 * pairs
        for (H, Cin, Cout, k, s) in ((64, 128, 256, 3, 2), (32, 256, 256, 3, 1), (64, 128, 256, 1, 2), (32, 256, 512, 3, 2),
                                     (16, 512, 512, 3, 1), (32, 256, 512, 1, 2)):
            conv = nn.Conv2d(Cin, Cout, k, stride=s, padding=k // 2, bias=False).to(dev)
            bn = nn.BatchNorm2d(Cout).to(dev).eval()
            x = torch.randn(N, Cin, H, H, device=dev)
            xh = x.permute(0, 2, 3, 1).contiguous()
            t_lib = timed(lambda: _bn_act(conv(x), bn, True))
            t_hip = timed(lambda: _trunk_conv(xh, conv, bn, True))
            flops = 2.0 * N * (H // s) ** 2 * Cout * Cin * k * k
            print(json.dumps({"pairs": pairs, "in": [N, Cin, H, H], "Cout": Cout, "k": k, "stride": s,
                              "library_plus_bn_act_us": round(t_lib, 1), "cpn_trunk_conv_bn_act_us": round(t_hip, 1),
                              "tflops": round(flops / t_hip / 1e6, 1)}), flush=True)
