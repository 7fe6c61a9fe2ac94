"""Microbenchmark of cpn_gather_rows_bwd on training-shaped inputs (B pairs x R random rays x S samples)."""
import os
import sys
import torch
from coponerf_amd import synthetic as syn
from coponerf_amd.render import RenderEngine
from coponerf_amd._hip import call

B, R, S, H = 4, 4096, 64, 256
dev = torch.device("cuda:0")
inp = syn.make_inputs(B, H, H, R, seed=61)
eng = RenderEngine()
c, q = inp["context"], inp["query"]
g = eng._geometry(c["cam2world"].to(dev), c["intrinsics"].to(dev), q["cam2world"].to(dev), q["intrinsics"].to(dev),
                  q["uv"].to(dev), None, False, S, H, H)
rows = B * R * 2 * S * 2
dx = (torch.randn(rows, 896, device=dev) * 0.1).half()
shapes = [(2 * B, 16, 16, 256), (2 * B, 32, 32, 256), (2 * B, 64, 64, 256), (2 * B, 256, 256, 64)]
st = torch.cuda.current_stream().cuda_stream
from coponerf_amd import _hip
boxes = torch.empty(B * 2 * _hip.lib().cpn_gather_bwd_chunks(R, S) * 16, dtype=torch.int32, device=dev)
for cfg in [""]:
    ts = []
    for it in range(4):
        dm = [torch.zeros(s, device=dev) for s in shapes]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("cpn_gather_rows_bwd", dx.data_ptr(), 896, H, H, g["pixel_val"].data_ptr(), g["sec_grid"].data_ptr(), B, 2, R, S,
             0, B * R, dm[0].data_ptr(), dm[1].data_ptr(), dm[2].data_ptr(), dm[3].data_ptr(), boxes.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"cfg={cfg or 'default':24s} ms={min(ts):8.2f}  sums={[float(d.double().sum()) for d in dm]}")
