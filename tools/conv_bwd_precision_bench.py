"""Round 6: what the ResNet-34 trunk's convolution BACKWARD (data + weight gradient, the library's implicit-GEMM kernels) costs
at the training step's shapes (4 pairs = 8 images of 256 x 256) with fp32 / fp16 / bf16 operands, NCHW and channels-last, and
how far the reduced-precision gradients are from the fp32 ones.  One JSON line per (shape, form)."""
import json
import sys
import time

import torch

SHAPES = [  # (Cin, Cout, H, k, stride, count in the trunk)
    (64, 64, 128, 3, 1, 6), (64, 128, 128, 3, 2, 1), (128, 128, 64, 3, 1, 7), (64, 128, 128, 1, 2, 1),
    (128, 256, 64, 3, 2, 1), (256, 256, 32, 3, 1, 11), (128, 256, 64, 1, 2, 1),
    (256, 512, 32, 3, 2, 1), (512, 512, 16, 3, 1, 5), (256, 512, 32, 1, 2, 1),
]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    N = int(args[0]) if args else 8
    if "--find" in sys.argv:                      # let the library time its solvers per shape instead of its heuristic pick
        torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    total = {}
    for (ci, co, H, k, s, cnt) in SHAPES:
        x = torch.randn(N, ci, H, H, device=dev)
        w = torch.randn(co, ci, k, k, device=dev) * (ci * k * k) ** -0.5
        Ho = (H + 2 * (k // 2) - k) // s + 1
        dy = torch.randn(N, co, Ho, Ho, device=dev) * 1e-3
        flops = 2.0 * N * Ho * Ho * co * ci * k * k * 2
        ref = torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [k // 2, k // 2], [1, 1], False, [0, 0], 1, [True, True, False])
        fwd_us = timed(lambda: torch.ops.aten.convolution(x, w, None, [s, s], [k // 2, k // 2], [1, 1], False, [0, 0], 1))
        total["f32_forward"] = total.get("f32_forward", 0.0) + cnt * fwd_us
        print(json.dumps({"shape": [ci, co, H, k, s], "count": cnt, "form": "f32_forward", "us": round(fwd_us, 1),
                          "tflops": round(flops / 2 / fwd_us * 1e-6, 1)}), flush=True)
        for form in (("f32", "f16") if "--find" in sys.argv else ("f32", "f32_cl", "f16", "f16_cl", "bf16_cl")):
            dt = {"f32": torch.float32, "f16": torch.float16, "bf1": torch.bfloat16}[form[:3]]
            mf = torch.channels_last if form.endswith("_cl") else torch.contiguous_format
            scale = 1024.0 if dt == torch.float16 else 1.0
            xx, ww, dd = x.to(dt).contiguous(memory_format=mf), w.to(dt).contiguous(memory_format=mf), (dy * scale).to(dt).contiguous(memory_format=mf)
            fn = lambda: torch.ops.aten.convolution_backward(dd, xx, ww, None, [s, s], [k // 2, k // 2], [1, 1], False, [0, 0], 1, [True, True, False])
            try:
                us = timed(fn)
                dx, dw, _ = fn()
                ex = ((dx.float() / scale - ref[0]).norm() / ref[0].norm()).item()
                ew = ((dw.float() / scale - ref[1]).norm() / ref[1].norm()).item()
            except Exception as e:                     # a form the library has no kernel for
                print(json.dumps({"shape": [ci, co, H, k, s], "form": form, "error": str(e)[:200]}))
                continue
            # the casts a mixed-precision backward pays on top (dy and x to the operand type, dx and dw back)
            cast_us = 0.0
            if dt != torch.float32:
                cast_us = timed(lambda: ((dy * scale).to(dt).contiguous(memory_format=mf), x.to(dt).contiguous(memory_format=mf), dx.float(), dw.float()))
            total[form] = total.get(form, 0.0) + cnt * (us + cast_us)
            print(json.dumps({"shape": [ci, co, H, k, s], "count": cnt, "form": form, "us": round(us, 1), "cast_us": round(cast_us, 1),
                              "tflops": round(flops / us * 1e-6, 1), "rel_dx": ex, "rel_dw": ew}), flush=True)
    print(json.dumps({"trunk_backward_ms_by_form": {k: round(v * 1e-3, 2) for k, v in total.items()}, "images": N}))


if __name__ == "__main__":
    main()
