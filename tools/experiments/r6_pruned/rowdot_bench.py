"""cpn_gemm_f16_rowdot (key_map_2 + the round-1 logit) on one chunk of rows, behind 1 GB of foreign traffic: ms per launch with
Q (= coords_embed) row-major and in fragment order, for the library COPONERF_HIP_LIB points at (tools/_build/libcpn_noq.so =
-DCPN_ROWDOT_NOQ: the Q rows never loaded)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd._hip import call          # noqa: E402

dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
flush = torch.zeros(128 << 20, dtype=torch.float32, device=dev)
for rows in (16384 * 128, 65536 * 128):
    A = (torch.randn(rows, 128, device=dev) * 0.5).half()
    Q = (torch.randn(rows, 128, device=dev) * 0.5).half()
    # the same matrix in fragment order: [16-row group][32-column block][lane = row + 16 * 8-column group][8]
    Qf = Q.view(rows // 16, 16, 4, 4, 8).permute(0, 2, 3, 1, 4).contiguous()
    W = (torch.randn(128, 128, device=dev) * 0.05).half()
    b = torch.randn(128, device=dev)
    outs = {}
    for name, q, ldq in (("row-major", Q, 128), ("fragment order", Qf, 0)):
        lg = torch.empty(rows, device=dev)
        run = lambda: call("cpn_gemm_f16_rowdot", A.data_ptr(), 128, W.data_ptr(), 128, b.data_ptr(), q.data_ptr(), ldq, lg.data_ptr(), rows, 128, 128, s)
        for _ in range(2):
            run()
        tot = 0.0
        for _ in range(5):
            flush.add_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1) / 5
        outs[name] = lg
        print(os.environ.get("COPONERF_HIP_LIB", "product"), rows, "rows, Q", name + ":", round(tot, 3), "ms;",
              round((rows * 512 + rows * 4) / tot / 1e6, 1), "GB/s algorithmic")
    print("   logits bit-identical between the two layouts:", bool(torch.equal(outs["row-major"], outs["fragment order"])))
